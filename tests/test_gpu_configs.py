"""-m gpu: BASELINE.json configs[1] at its stated size — 10 000 synthetic windows (4096 bp, 32 overlaps), batch 64, one GPU —
through the size-independent properties the domain offers (the oracle takes minutes at this size; the parity tests proper run it
on smaller sets): one corrected record per target, independently computed copies of a target agree, the output alphabet,
idempotence of a whole pass, the lean and the planes feature paths agree, and SURVEY §8(a)'s invariants of the pileup matrix on a
sample of windows."""
import numpy as np
import pytest

import gpu_common as G
from herro_amd import api, synth

pytestmark = pytest.mark.gpu

W, N_OVL, WPT = 4096, 32, 4
N_BASE, COPIES = 625, 4            # 2500 targets = 10 000 windows; 625 are generated, the rest are copies with read ids of their own


def _bodies(text: bytes):
    recs = text.split(b">")[1:]
    return {r.split(b" ", 1)[0]: r.split(b"\n", 1)[1] for r in recs}


def test_config1_10k_windows_batch64_properties():
    sb = synth.replicate_targets(synth.generate_parallel(N_BASE, WPT * W, N_OVL, seed=synth.SEED + 1), COPIES)
    assert sb.n_targets * WPT == 10000
    c = G.ctx()
    G.load_synth(c, sb)
    group = 250                                              # targets per job: 1000 windows
    fasta, n_windows, n_sup = {}, 0, 0
    first = None
    for t0 in range(0, sb.n_targets, group):
        ts = list(range(t0, min(t0 + group, sb.n_targets)))
        job = api.job_from_synth(c, sb, W, targets=ts)
        job.featurize()
        job.infer(64, 1)                                     # configs[1]: batch 64, windows batched across reads
        job.consensus()
        ids = [f"read{int(sb.tgt_rid[t])}" for t in ts]
        text = job.fasta(ids)
        fasta.update(_bodies(text))
        n_windows += job.n_windows
        n_sup += job.stats()["sum_supported"]
        if first is None:
            first = (job, ts, ids, text)
        else:
            job.close()
    assert n_windows == 10000 and n_sup > 10000
    # one record per target, nothing but bases in it
    assert len(fasta) == sb.n_targets
    assert all(set(b) <= set(b"ACGT\n") for b in list(fasta.values())[::37])
    # every copy of a target was corrected on its own (another job, another batch composition) to the same bases
    for t in range(0, N_BASE, 5):
        b0 = fasta[f"read{int(sb.tgt_rid[t])}".encode()]
        for k in range(1, COPIES):
            assert fasta[f"read{int(sb.tgt_rid[k * N_BASE + t])}".encode()] == b0, (t, k)
    # a corrected read is close to its target's length (consensus removes noise, not sequence)
    lens = np.array([len(b) - 1 for b in fasta.values()])
    assert np.all(np.abs(lens - WPT * W) < 0.02 * WPT * W)
    # idempotence of a whole pass, and the planes path gives the same records as the lean one
    job, ts, ids, text = first
    job.featurize(); job.infer(64, 1); job.consensus()
    assert job.fasta(ids) == text
    try:
        c.featurize_planes(True)
        job.featurize(); job.infer(64, 1); job.consensus()
        assert job.fasta(ids) == text
    finally:
        c.featurize_planes(False)
    # SURVEY §8(a) invariants on a sample of windows (planes built on request)
    job.featurize()
    strands = {}
    for a in range(int(sb.tgt_aln_off[ts[0]]), int(sb.tgt_aln_off[ts[-1] + 1])):
        strands[int(sb.aln[a, 0])] = int(sb.aln[a, 4])
    for w in range(0, job.n_windows, 97):
        gw = job.window(w, encoded=True)
        b = gw.bases
        assert gw.info.length == b.shape[0] >= gw.info.win_len
        assert not np.any(b[:, 0] == 10)                                        # the target column never shows '.'
        assert np.count_nonzero(b[:, 0] != 4) == gw.info.win_len                # one base row per target position
        informative = (b != 10) & (b != 4) & (b != 9)
        assert np.all(informative[:, : gw.info.n_alns + 1].any(axis=1))         # no all-gap rows survive (features.rs:531-545)
        assert np.all(b[:, gw.info.n_alns + 1:] == 10)                          # padding columns
        key = gw.sup_pos.astype(np.int64) * 256 + gw.sup_ins
        assert np.all(np.diff(key) > 0) and (len(gw.sup_ins) == 0 or gw.sup_ins.max() <= 50)
        for k in range(gw.info.n_alns):
            col = b[:, 1 + k]
            ok = {5, 6, 7, 8, 9, 10} if strands[int(gw.qids[k])] else {0, 1, 2, 3, 4, 10}
            assert set(np.unique(col).tolist()) <= ok, (w, k)
    job.close()
