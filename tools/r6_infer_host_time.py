"""Host time of herro_job_infer (batch plan, tile packing, descriptor blob, launches) for one job of 2560 / 4096 windows with the device idle: what the GPU waits for
between featurize and the model when nothing else is queued.   usage: python tools/r6_infer_host_time.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from herro_amd import api, model_io, synth
path, _ = model_io.default_model_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "_cache"))
for n_t in (640, 1024):
    sb = synth.generate_parallel(n_t, 4 * 4096, 32, seed=synth.SEED + 5, workers=16)
    c = api.Context(0); c.load_model(path); c.set_reads(sb.seq, sb.qual, sb.off)
    j = api.job_from_synth(c, sb, 4096)
    for _ in range(3):
        j.featurize(); j.infer(128, 1); j.consensus(); c.synchronize()
    ts = []
    for _ in range(10):
        j.featurize(); c.synchronize()
        t0 = time.perf_counter(); j.infer(128, 1); t1 = time.perf_counter(); c.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    h = sorted(t[0] for t in ts)[len(ts) // 2]; g = sorted(t[1] for t in ts)[len(ts) // 2]
    print(f"{j.n_windows} windows: herro_job_infer returns after {h*1e6:.0f} us of host work (median of 10); the model kernels then run {g*1e6:.0f} us")
    j.close(); c.close()
