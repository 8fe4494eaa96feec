"""Per-kernel HIP-event times of one job of 2560 windows at a given SNP rate (the `sensitivity` leg of bench.py in detail).   usage: python tools/r6_sens_kernels.py 0.03"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from herro_amd import api, model_io, synth
p_snp = float(sys.argv[1]) if len(sys.argv) > 1 else 3e-2
path, _ = model_io.default_model_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "_cache"))
sb = synth.generate_parallel(640, 4 * 4096, 32, seed=synth.SEED + 77, workers=16, p_snp=p_snp)
c = api.Context(0); c.load_model(path); c.set_reads(sb.seq, sb.qual, sb.off)
j = api.job_from_synth(c, sb, 4096)
for _ in range(2):
    j.featurize(); j.infer(128, 1); j.consensus()
c.synchronize()
t = time.perf_counter(); j.featurize(); j.infer(128, 1); j.consensus(); c.synchronize(); el = time.perf_counter() - t
ns = np.array([j.info(w).n_supported for w in range(j.n_windows)])
print(f"p_snp {p_snp}: {j.n_windows} windows, informative rows mean {ns.mean():.1f} max {ns.max()} >64: {(ns > 64).sum()} >256: {(ns > 256).sum()} >512: {(ns > 512).sum()}; wall {el*1e3:.2f} ms = {j.n_windows/el:.0f} windows/s, precision {c.precision()}, rf fused {j.rf_fused()}")
c.timing_enable(True); c.timing_reset()
j.featurize(); j.infer(128, 1); j.consensus(); c.synchronize()
tm = c.timing()
tot = sum(v[0] for v in tm.values())
for k, (ms, n) in sorted(tm.items(), key=lambda kv: -kv[1][0]): print(f"  {k:16s} {ms*1e3:9.1f} us  x{n}")
print(f"  sum {tot*1e3:.1f} us")
