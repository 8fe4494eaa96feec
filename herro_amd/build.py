"""Builds libherro_amd.so (HIP, gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libherro_amd.so")
HIP_SOURCES = ["pileup.hip", "model.hip", "model_h.hip", "cigar_dev.hip", "build_dev.hip", "herro_api.hip", "ingest.cpp", "fastx.cpp", "pool.cpp"]
HEADERS = ["herro_amd.map", "host_cpus.h", "job_dev.h", "model_dev.h", "pileup_core.h", "windowing.hpp", "cigar_dev.h", "build_dev.h", os.path.join("..", "..", "include", "herro_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
         "-fno-gpu-rdc", "-Wl,--version-script=" + os.path.join(CSRC, "herro_amd.map")]   # exports: the C ABI (herro_*) only


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in HIP_SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_hip(force: bool = False, out: str | None = None, defines: tuple = ()) -> str:
    """out / defines: a second library next to the release one (phase timers, A/B variants), selected at run time by HERRO_LIB."""
    prof_env = os.environ.get("HERRO_PROF_BUILD", "0") not in ("", "0")
    if prof_env and out is None:   # a timer build never replaces the release library: it goes beside it (select it with HERRO_LIB)
        out = os.path.join(HERE, "libherro_amd_prof.so")
    if out is not None:
        force = True
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = ["-DHERRO_PROF_BUILD"] if prof_env else []   # kernel phase timers (job_dev.h)
    extra += ["-D" + d for d in defines]
    cmd = [hipcc] + FLAGS + extra + ["-o", out or LIB] + [os.path.join(CSRC, s) for s in HIP_SOURCES]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout)
    return out or LIB


if __name__ == "__main__":
    print(build_hip(force=True))
