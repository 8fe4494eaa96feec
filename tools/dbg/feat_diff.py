"""Print where the product's pileup differs from the oracle's (debug helper, GPU box)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
if torch.cuda.is_available(): torch.cuda.init()
import gpu_common as G, oracle_lib as O
from herro_amd import api, synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sb = synth.generate(3, 4 * 4096, 32, seed=synth.SEED + sum(map(ord, "baseline_w4096")))
c = G.ctx(); G.load_synth(c, sb); store = O.store_from_synth(sb)
job = api.job_from_synth(c, sb, W); job.featurize()
w = 0
for t in range(sb.n_targets):
    rid, rows, cigs = O.target_alignments(sb, t)
    res = store.extract_features(rid, rows, cigs, W)
    for wi in range(len(res)):
        ow, gw = res.window(wi), job.window(w)
        if gw.bases.shape != ow.bases.shape: print("win", w, "shape", gw.bases.shape, ow.bases.shape)
        else:
            d = np.argwhere(gw.bases != ow.bases)
            if len(d):
                print("win", w, "ndiff", len(d), "rows", d[:, 0].min(), d[:, 0].max(), "cols", sorted(set(d[:, 1].tolist()))[:40])
                for r, cc in d[:12]: print("   row", r, "col", cc, "got", chr(gw.bases[r, cc]), "want", chr(ow.bases[r, cc]), "tgt", chr(ow.bases[r, 0]))
            dq = np.argwhere(gw.quals != ow.quals)
            if len(dq): print("win", w, "qual ndiff", len(dq))
            if gw.sup_pos.tolist() != ow.sup_pos.tolist(): print("win", w, "sup differs", len(gw.sup_pos), len(ow.sup_pos))
        w += 1
print("done", w)
