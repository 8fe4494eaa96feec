"""smoke(): one small invocation of the hot path on cuda:0, checked against the oracle (oracle/ and tests/ helpers are
checker code: this file lives next to __graft_entry__.py, outside the herro_amd package, which imports none of it)."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))


def run() -> None:
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from herro_amd import api, model_io, synth
    import oracle_lib as O       # checker only
    import model_ref as MR       # checker only

    W = 1024
    sb = synth.generate(2, 3 * W + 300, 24, seed=1234, flank_min=120, flank_max=200, p_partial=0.2)
    ctx = api.Context(0)
    path, raw = model_io.default_model_file(os.path.join(ROOT, "tests", "_cache"))
    ctx.load_model(path)
    ctx.set_precision(api.DEFAULT_PRECISION)
    ctx.set_reads(sb.seq, sb.qual, sb.off)
    job = api.job_from_synth(ctx, sb, W)
    job.featurize()
    job.infer(4, 0)
    store = O.store_from_synth(sb)
    twin = MR.build(raw, model_io.Hyper())
    w = 0
    n_tok = 0
    fastas = []
    for t in range(sb.n_targets):
        rid, rows, cigs = O.target_alignments(sb, t)
        res = store.extract_features(rid, rows, cigs, W)
        logits = []
        for wi in range(len(res)):
            ow, gw = res.window(wi), job.window(w + wi)
            assert np.array_equal(gw.bases, ow.bases) and np.array_equal(gw.quals, ow.quals), "pileup mismatch"
            assert gw.sup_pos.tolist() == ow.sup_pos.tolist() and gw.sup_ins.tolist() == ow.sup_ins.tolist()
            assert gw.qids.tolist() == ow.qids.tolist()
            if len(ow.sup_pos):
                logits.append(job.logits(w + wi)[1])
        # model: reference grouping, batches of 4 windows of this read
        for g0 in range(0, len(res), 4):
            nb, bt = res.collate(1 << 20, 0)
            break
        sel = [wi for wi in range(len(res)) if len(res.window(wi).sup_pos)]
        for g0 in range(0, len(res), 4):
            grp = [wi for wi in sel if g0 <= wi < g0 + 4]
            if not grp:
                continue
            gws = [job.window(w + wi, encoded=True) for wi in grp]
            Lmax = max(g.info.length for g in gws)
            bases = np.full((len(grp), Lmax, 31), 11, np.uint8)
            quals = np.full((len(grp), Lmax, 31), 126, np.uint8)
            lens, flat = [], []
            for k, g in enumerate(gws):
                bases[k, :g.info.length], quals[k, :g.info.length] = g.bases, g.quals
                tidx = np.flatnonzero(g.bases[:, 0] != 4)
                lens.append(len(g.sup_pos))
                flat.extend((tidx[g.sup_pos] + g.sup_ins).tolist())
            ti, tb = MR.run_batch(twin, bases, quals, np.array(lens, np.int32), np.array(flat, np.int32))
            o = 0
            for k, wi in enumerate(grp):
                gb = job.logits(w + wi)[1]
                assert np.abs(gb - tb[o:o + lens[k]]).max() <= 1e-3, "logits outside 1e-3"
                o += lens[k]
                n_tok += lens[k]
        lg = np.concatenate(logits) if logits else np.zeros((0, 5), np.float32)
        want = res.consensus_fasta(lg)
        assert job.consensus_fasta(t, sb.read_name(rid)) == want, "FASTA mismatch (host decode)"
        fastas.append((t, rid, want))
        w += len(res)
    job.consensus()
    for t, rid, want in fastas:
        assert job.consensus_fasta(t, sb.read_name(rid)) == want, "FASTA mismatch (device consensus)"
    job.close()
    ctx.close()
    print(f"smoke ok: {w} windows, {n_tok} informative positions, pileup bit-exact, logits within 1e-3, FASTA identical")
