"""not-gpu: how the host side sizes itself — worker counts follow the CPUs the process may actually use (hardware
threads, affinity mask, cgroup CPU quota), not the hardware thread count (a 16-CPU quota on a 256-thread
box turned 64-thread pools into 75 ms stalls per 100 ms period)."""
import builtins
import io
import os

from herro_amd import synth


def _with_files(monkeypatch, files):
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if isinstance(path, str) and path.startswith("/sys/fs/cgroup/"):
            if path in files:
                return io.StringIO(files[path])
            raise FileNotFoundError(path)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)


def test_usable_cpus_follows_the_cgroup_quota(monkeypatch):
    monkeypatch.setattr(os, "cpu_count", lambda: 256)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)
    _with_files(monkeypatch, {"/sys/fs/cgroup/cpu.max": "1600000 100000\n"})
    assert synth.usable_cpus() == 16
    _with_files(monkeypatch, {"/sys/fs/cgroup/cpu.max": "max 100000\n"})
    assert synth.usable_cpus() == 256
    _with_files(monkeypatch, {"/sys/fs/cgroup/cpu.max": "250000 100000\n"})           # 2.5 CPUs -> 3 workers
    assert synth.usable_cpus() == 3
    _with_files(monkeypatch, {"/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "800000\n", "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"})   # cgroup v1
    assert synth.usable_cpus() == 8
    _with_files(monkeypatch, {"/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "-1\n", "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"})
    assert synth.usable_cpus() == 256
    _with_files(monkeypatch, {})
    assert synth.usable_cpus() == 256
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(12)), raising=False)                                            # taskset
    assert synth.usable_cpus() == 12
