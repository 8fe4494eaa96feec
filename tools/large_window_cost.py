"""GPU: what windows with more than 64 informative rows cost.  The fused encoder stack takes tiles of whole windows of at
most 64 rows; a larger window (up to 512 rows) is spread over sibling tiles of the same stack, which run beside the ordinary
tiles (HERRO_FUSED_BIG=0: through the layer-by-layer bf16x3 kernels as in round 3; HERRO_SIB_STREAM=0: sibling tiles in front of
the ordinary ones instead of beside them).  Times the model kernels (HIP events of the context's KernelTimer) of one collated
batch of B windows, all small, against the same batch with 1 % of the windows enlarged to 100 rows.
usage: python tools/large_window_cost.py [B]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from herro_amd import api, model_io  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = 512
rng = np.random.default_rng(3)
c = api.Context(0)
path, _ = model_io.default_model_file(os.path.join(ROOT, "tests", "_cache"))
c.load_model(path)
c.set_precision(api.DEFAULT_PRECISION)
bases = rng.integers(0, 11, (B, L, 31)).astype(np.uint8)
quals = rng.integers(33, 90, (B, L, 31)).astype(np.uint8)


def run(counts):
    idx = [np.sort(rng.choice(L, size=int(k), replace=False)) for k in counts]
    lens = np.array([len(i) for i in idx], np.int32)
    flat = np.concatenate(idx).astype(np.int32)
    c.model_forward(bases, quals, lens, flat)          # warm-up: scratch, opt-ins
    c.timing_enable(True)
    c.timing_reset()
    for _ in range(5):
        c.model_forward(bases, quals, lens, flat)
    t = c.timing()
    c.timing_enable(False)
    per_call = {k: round(v[0] / v[1] * 1e3, 1) for k, v in t.items() if v[1]}
    return per_call, round(sum(v[0] for v in t.values()) / 5 * 1e3, 1), int(lens.sum())   # µs per call; µs of model kernels per forward


small = np.clip(rng.normal(15.2, 4.5, B).round(), 4, 30).astype(int)
mixed = small.copy()
mixed[rng.choice(B, size=max(1, B // 100), replace=False)] = 100
a, ta, na = run(small)
b, tb, nb = run(mixed)
out = {"windows": B, "rows_small": na, "rows_mixed": nb, "large_windows": int((mixed > 64).sum()),
       "us_per_call_small": a, "us_per_call_mixed": b, "model_us_per_forward_small": ta, "model_us_per_forward_mixed": tb}
print(json.dumps(out))
