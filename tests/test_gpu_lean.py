"""-m gpu: the two feature-generation paths are two independent derivations of the same results.

The LEAN path (round 5, the default: k_rows) works in position space — bit-sliced symbol counters over the selected
columns' planes, insertion rows from per-row accumulators — and never writes the [31][L'] token planes
(features.rs:547-556); the PLANES path (rounds 3-4: k_tokens) counts symbols while it writes every cell.  Here both run
on the same jobs and must agree on everything downstream (informative rows, logits bit for bit, corrected FASTA), and
the receptive-field records the model reads on the lean path must be the planes' cells (which test_gpu_features pins
to the oracle)."""
import numpy as np
import pytest

import gpu_common as G
import oracle_lib as O
from herro_amd import api, synth

pytestmark = pytest.mark.gpu

CASES = {
    "baseline_w4096": dict(W=4096, n=3, tl=4 * 4096, ov=32, kw={}),
    "low_coverage": dict(W=512, n=4, tl=2048, ov=3, kw=dict(flank_min=60, flank_max=90, p_partial=0.5)),
    "noisy_w256": dict(W=256, n=4, tl=1500, ov=16, kw=dict(flank_min=30, flank_max=60, p_sub=0.05, p_ins=0.05, p_del=0.05, p_partial=0.2)),
    "many_overlaps": dict(W=256, n=2, tl=1024, ov=70, kw=dict(flank_min=30, flank_max=60)),
    "w8192": dict(W=8192, n=2, tl=3 * 8192 + 100, ov=12, kw=dict(p_partial=0.3)),
    "w1000": dict(W=1000, n=3, tl=3500, ov=10, kw=dict(flank_min=100, flank_max=200, p_partial=0.3)),
    "diverged_haplotypes": dict(W=4096, n=2, tl=2 * 4096, ov=32, kw=dict(p_snp=0.03)),
    "very_diverged": dict(W=4096, n=2, tl=2 * 4096, ov=32, kw=dict(p_snp=0.09)),       # windows above 256 informative rows: k_rows leaves the receptive fields to k_rfq
}


def _run(job, ids, batch):
    job.featurize()
    job.infer(batch, 1)
    job.consensus()
    wins = []
    for w in range(job.n_windows):
        wi = job.info(w)
        sp = np.zeros(wi.n_supported, np.uint16)
        si = np.zeros(wi.n_supported, np.uint8)
        job.ctx._chk(job._l.herro_job_window_copy(job.h, w, 1, None, None, sp.ctypes.data, si.ctypes.data, None))
        info, base = job.logits(w)
        wins.append((wi.length, wi.n_supported, wi.n_alns, sp.copy(), si.copy(), info.copy(), base.copy()))
    return wins, job.fasta(ids)


@pytest.mark.parametrize("name", list(CASES))
def test_lean_path_equals_planes_path(name):
    cs = CASES[name]
    sb = synth.generate(cs["n"], cs["tl"], cs["ov"], seed=synth.SEED + 17 + sum(map(ord, name)), **cs["kw"])
    c = G.ctx()
    G.load_synth(c, sb)
    ids = [f"read{t}" for t in range(sb.n_targets)]
    job = api.job_from_synth(c, sb, cs["W"])
    other = None
    try:
        c.featurize_planes(False)
        lean, fa_lean = _run(job, ids, 64)                # a job on its own: receptive fields by k_rfq behind the counts
        assert not job.rf_fused()
        other = api.job_from_synth(c, sb, cs["W"], targets=[0])
        other.featurize()                                 # a pending job: the caller pipelines -> k_rows gathers the receptive fields itself
        lean2, fa_lean2 = _run(job, ids, 64)
        fused = job.rf_fused()
        other.close(); other = None
        c.featurize_planes(True)
        planes, fa_planes = _run(job, ids, 64)
    finally:
        c.featurize_planes(False)
        if other is not None:
            other.close()
    assert fused == (sum(a[1] for a in lean) > 0) or name == "low_coverage"   # (a window above the 256 rows k_rows stages no longer takes the job off the fused gather: k_rfq fills that window)
    assert fa_lean2 == fa_lean
    for a, b in zip(lean, lean2):
        assert a[:3] == b[:3] and np.array_equal(a[5], b[5]) and np.array_equal(a[6], b[6])   # both gathers feed the model the same records
    assert len(lean) == len(planes) > 0
    n_sup = 0
    for w, (a, b) in enumerate(zip(lean, planes)):
        assert a[:3] == b[:3], (w, a[:3], b[:3])
        assert a[3].tolist() == b[3].tolist() and a[4].tolist() == b[4].tolist(), w
        assert np.array_equal(a[5], b[5]) and np.array_equal(a[6], b[6]), (w, "logits differ between the two paths")
        n_sup += a[1]
    assert fa_lean == fa_planes
    if name != "low_coverage":
        assert n_sup > 0
    job.close()


@pytest.mark.parametrize("name", ["baseline_w4096", "noisy_w256", "w1000", "very_diverged"])
def test_receptive_field_records_are_the_planes_cells(name):
    cs = CASES[name]
    sb = synth.generate(cs["n"], cs["tl"], cs["ov"], seed=synth.SEED + 29 + sum(map(ord, name)), **cs["kw"])
    c = G.ctx()
    G.load_synth(c, sb)
    job = api.job_from_synth(c, sb, cs["W"])
    job.featurize()
    job.infer(64, 1)
    assert not job.rf_fused()                  # a job on its own: k_rfq gathers behind the copy of the counts (the host plans meanwhile)
    other = api.job_from_synth(c, sb, cs["W"], targets=[0])
    other.featurize()                          # another job of the context is featurized and waits for its infer: the caller pipelines ...
    job.featurize()
    job.infer(64, 1)
    big = sum(1 for w in range(job.n_windows) if job.info(w).n_supported > 256)
    assert job.rf_fused() and job.rf_left() == big   # ... and k_rows gathers the receptive fields itself; a window with more rows than it stages is filled by k_rfq, alone (round 6)
    assert (big > 0) == (name == "very_diverged")
    other.close()
    n_rec = 0
    for w in range(job.n_windows):
        rf = job.rf_records(w)                 # read by the model on the lean path
        gw = job.window(w, encoded=True)       # planes built on request; test_gpu_features pins them to the oracle
        L = gw.info.length
        rows = np.flatnonzero(gw.bases[:, 0] != 4)           # row of every target position (inference.rs:255-268)
        for k in range(gw.info.n_supported):
            r0 = int(rows[gw.sup_pos[k]]) + int(gw.sup_ins[k])
            for d in range(5):
                r = r0 - 2 + d
                if 0 <= r < L:
                    assert rf[k, :, d].tolist() == gw.bases[r].tolist(), (w, k, d, "tokens")
                    assert rf[k, :, 8 + d].tolist() == gw.quals[r].tolist(), (w, k, d, "qualities")
                    n_rec += 1
    assert n_rec > 0
    job.close()


TOKMAP = np.full(256, 255, np.uint8)
for _i, _ch in enumerate("ACGT*acgt#."):     # inference.rs BASES_MAP: the token of every cell of the oracle's matrix
    TOKMAP[ord(_ch)] = _i


@pytest.mark.parametrize("name", ["baseline_w4096", "low_coverage", "noisy_w256", "diverged_haplotypes"])
def test_receptive_field_records_are_the_oracles_cells(name):
    """the records the model reads on the lean path (written by k_rfq behind a job on its own, by k_rows itself for a
    pipelining caller) against cells cut from the ORACLE's [L', 31] bases / quals (features.rs:110-266; rows from get_target_indices,
    inference.rs:255-268) — byte for byte, with no token plane of the product involved: the records are read out BEFORE anything asks for
    the planes (the planes path is not the reference here, the oracle is)."""
    cs = CASES[name]
    n = 6 if name == "baseline_w4096" else cs["n"]          # 24 windows of the BASELINE workload (4096 bp x 32 overlaps)
    sb = synth.generate(n, cs["tl"], cs["ov"], seed=synth.SEED + 41 + sum(map(ord, name)), **cs["kw"])
    store = O.store_from_synth(sb)
    c = G.ctx()
    G.load_synth(c, sb)
    c.featurize_planes(False)
    job = api.job_from_synth(c, sb, cs["W"])
    job.featurize()
    job.infer(64, 1)
    assert not job.rf_fused()
    recs = {"k_rfq": [job.rf_records(w) for w in range(job.n_windows)]}
    other = api.job_from_synth(c, sb, cs["W"], targets=[0])
    other.featurize()                          # a pending job: k_rows gathers the receptive fields itself
    job.featurize()
    job.infer(64, 1)
    if job.rf_fused():
        recs["k_rows"] = [job.rf_records(w) for w in range(job.n_windows)]
    other.close()
    assert name == "low_coverage" or "k_rows" in recs
    n_cells = w = 0
    for t in range(sb.n_targets):
        rid, rows, cigs = O.target_alignments(sb, t)
        res = store.extract_features(rid, rows, cigs, cs["W"])
        for wi in range(len(res)):
            ow = res.window(wi)
            enc = TOKMAP[ow.bases]
            L = enc.shape[0]
            tidx = np.flatnonzero(enc[:, 0] != 4)
            r0 = (tidx[ow.sup_pos.astype(np.int64)] + ow.sup_ins).astype(np.int64)
            for who, rr in recs.items():
                rf = rr[w]
                assert rf.shape[0] == len(r0), (who, t, wi)
                for d in range(5):
                    r = r0 - 2 + d
                    ok = (r >= 0) & (r < L)
                    assert np.array_equal(rf[ok][:, :, d], enc[r[ok]]), (who, t, wi, d, "tokens")
                    assert np.array_equal(rf[ok][:, :, 8 + d], ow.quals[r[ok]]), (who, t, wi, d, "qualities")
                    n_cells += int(ok.sum()) * 31
            w += 1
    assert w == job.n_windows and (n_cells > 0 or name == "low_coverage")
    job.close()


@pytest.mark.parametrize("trial", G.sweep_trials(10))
def test_lean_equals_planes_on_random_configurations(trial):
    """A seeded sweep (round 6: k_rows evaluates the insertion rows one per thread from positions the run walkers leave and cover counts in LDS planes; k_rfq's records come in a
    5-row instantiation): informative rows, logits bit for bit and FASTA of the lean path against the planes path on window sizes, depths and — above all — insertion
    rates the named cases do not reach."""
    g = np.random.default_rng(0xbb67ae85 + trial)
    W = int(g.choice([100, 256, 512, 1000, 2048, 4096]))
    tl = int(g.integers(1, 4)) * W + int(g.integers(0, W))
    ov = int(g.integers(2, 45))
    fl = max(8, min(W // 4, 300))
    kw = dict(flank_min=fl // 2 + 1, flank_max=fl + 2, p_sub=float(g.choice([0.004, 0.02])), p_ins=float(g.choice([0.004, 0.03, 0.08])),
              p_del=float(g.choice([0.004, 0.03])), p_partial=float(g.choice([0.0, 0.4])), p_snp=float(g.choice([0.0, 0.01, 0.05])))
    sb = synth.generate(int(g.integers(1, 4)), tl, ov, seed=int(g.integers(1, 1 << 30)), **kw)
    c = G.ctx()
    G.load_synth(c, sb)
    ids = [f"read{t}" for t in range(sb.n_targets)]
    job = api.job_from_synth(c, sb, W)
    other = None
    try:
        c.featurize_planes(False)
        lean, fa_lean = _run(job, ids, 64)
        other = api.job_from_synth(c, sb, W, targets=[0])
        other.featurize()                                 # the caller pipelines: the fused gather
        lean2, fa_lean2 = _run(job, ids, 64)
        other.close(); other = None
        c.featurize_planes(True)
        planes, fa_planes = _run(job, ids, 64)
    finally:
        c.featurize_planes(False)
        if other is not None:
            other.close()
    assert fa_lean == fa_planes == fa_lean2, (W, tl, ov, kw)
    assert len(lean) == len(planes) > 0
    for w, (a, b, a2) in enumerate(zip(lean, planes, lean2)):
        assert a[:3] == b[:3] == a2[:3], (w, a[:3], b[:3])
        assert a[3].tolist() == b[3].tolist() and a[4].tolist() == b[4].tolist(), (w, W, kw)
        assert np.array_equal(a[5], b[5]) and np.array_equal(a[6], b[6]) and np.array_equal(a[5], a2[5]) and np.array_equal(a[6], a2[6]), (w, "logits differ")
    job.close()
