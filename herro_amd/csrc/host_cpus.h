// host_cpus.h — CPUs this process may actually use (shared by the job builder, the PAF parser and the FASTQ reader).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <thread>

namespace herro {

// CPUs the process may actually use: the hardware threads, capped by a cgroup CPU quota when there is one (v2 cpu.max,
// v1 cpu.cfs_quota_us).  A container that shows 256 hardware threads under a 16-CPU quota must not get a 64-thread pool:
// every woken thread reserves a bandwidth slice on its CPU, the quota is gone a quarter into each 100 ms period and the
// whole process — GPU completion waits included — stands still for the rest of it (measured: 75 ms stalls, r2n timeline).
inline uint32_t usable_cpus() {
  uint32_t n = std::max(1u, std::thread::hardware_concurrency());
  long long quota = -1, period = 0;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32] = {0};
    if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
    fclose(f);
  } else {
    if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
    if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
  }
  if (quota > 0 && period > 0) n = std::min<uint32_t>(n, (uint32_t)std::max<long long>(1, (quota + period - 1) / period));
  return n;
}


// worker threads for a host-side parallel section: HERRO_HOST_THREADS if set, else the usable CPUs, capped at `cap`
inline uint32_t host_threads(uint32_t cap) {
  if (const char* e = getenv("HERRO_HOST_THREADS")) return std::max(1u, std::min((uint32_t)std::max(1, atoi(e)), cap));
  return std::max(1u, std::min(usable_cpus(), cap));
}

}  // namespace herro
