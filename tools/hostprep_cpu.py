"""Host half of herro_job_create without a device (herro_debug_host_ctx): windows/s vs threads on this machine.
usage: python tools/hostprep_cpu.py [n_targets]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from herro_amd import api, synth

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sb = synth.generate_parallel(nt, 4 * 4096, 32, seed=7, chunk=32)
lens = (sb.off[1:] - sb.off[:-1]).astype(np.uint32)
for th in [int(x) for x in os.environ.get("HOSTPREP_THREADS", "1,2,4,8").split(",")]:
    os.environ["HERRO_HOST_THREADS"] = str(th)   # read when a context starts its thread pool (first herro_job_create)
    c = api.HostContext(lens)
    os.environ["HERRO_HOST_PROFILE"] = "1" if th == 1 else ""
    if not os.environ["HERRO_HOST_PROFILE"]:
        del os.environ["HERRO_HOST_PROFILE"]
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); job = api.job_from_synth(c, sb, 4096); dt = time.perf_counter() - t
        best = min(best, dt); n = job.n_windows; job.close()
    print(f"threads {th}: {n} windows in {best * 1e3:.1f} ms -> {n / best / 1e3:.1f} k windows/s ({best / n * 1e6:.1f} us/window)")
