"""-m gpu: herro_job_create builds a job's windows and descriptors ON THE DEVICE behind the CIGAR scan (csrc/build_dev.hip, round 6;
extract_windows windowing.rs:44-273 + the per-overlap bookkeeping of extract_features, features.rs:337-361) — the arrays must be the ones
the host build of rounds 3-5 writes (which tests/test_host_job_layout.py holds to the oracle's extract_windows without a device), field
for field, and whatever the device does not settle itself must fall back to the host build and fail there with the reference's message."""
import dataclasses
import re

import numpy as np
import pytest

import gpu_common as G
from herro_amd import api, synth

pytestmark = pytest.mark.gpu

CASES = {
    "baseline_w4096": dict(W=4096, n=6, tl=4 * 4096, ov=32, kw={}),
    "ragged_tail_w4096": dict(W=4096, n=3, tl=3 * 4096 + 1717, ov=20, kw=dict(p_partial=0.3)),
    "noisy_w256": dict(W=256, n=5, tl=1500, ov=16, kw=dict(flank_min=30, flank_max=60, p_sub=0.05, p_ins=0.05, p_del=0.05, p_partial=0.2)),
    "w16": dict(W=16, n=3, tl=700, ov=8, kw=dict(flank_min=8, flank_max=20, p_partial=0.3)),
    "w1000": dict(W=1000, n=3, tl=3500, ov=10, kw=dict(flank_min=100, flank_max=200, p_partial=0.3)),
    "w8192": dict(W=8192, n=2, tl=3 * 8192 + 100, ov=12, kw=dict(p_partial=0.3)),
    "many_overlaps": dict(W=256, n=2, tl=1024, ov=70, kw=dict(flank_min=30, flank_max=60)),
    "long_indels": dict(W=512, n=3, tl=2500, ov=12, kw=dict(flank_min=60, flank_max=90, p_long_indel=0.002)),
}


def _arrays(c, sb, W, host, targets=None):
    c.host_build(host)
    try:
        job = api.job_from_synth(c, sb, W, targets)
    finally:
        c.host_build(False)
    built = c._l.herro_debug_job_dev_built(job.h)
    arr = c.job_arrays(job)
    sk = job.skipped()
    return job, arr, built, sk


def _same(a, b, tag):
    assert set(a) == set(b)
    for k in a:
        if k == "ops":
            continue          # both read the op array the scan kernel wrote
        assert a[k].dtype == b[k].dtype and len(a[k]) == len(b[k]), (tag, k, len(a[k]), len(b[k]))
        if a[k].dtype.names:
            for f in a[k].dtype.names:
                assert np.array_equal(a[k][f], b[k][f]), (tag, k, f, np.flatnonzero(a[k][f] != b[k][f])[:5])
        else:
            assert np.array_equal(a[k], b[k]), (tag, k)


@pytest.mark.parametrize("name", list(CASES))
def test_device_built_descriptors_equal_the_host_built_ones(name):
    cs = CASES[name]
    sb = synth.generate(cs["n"], cs["tl"], cs["ov"], seed=synth.SEED + 61 + sum(map(ord, name)), **cs["kw"])
    c = G.ctx()
    G.load_synth(c, sb)
    jd, ad, built_d, sk_d = _arrays(c, sb, cs["W"], host=False)
    jh, ah, built_h, sk_h = _arrays(c, sb, cs["W"], host=True)
    try:
        assert built_d == 1 and built_h == 0
        assert len(ad["ow"]) > 0 and sk_d == sk_h
        _same(ad, ah, name)
        # ... and the pileup computed from them is the same (informative rows, L', kept overlaps of every window)
        jd.featurize(); jh.featurize()
        for w in range(jd.n_windows):
            a, b = jd.info(w), jh.info(w)
            assert (a.rid, a.wid, a.n_total_wins, a.length, a.n_supported, a.n_overlaps, a.n_alns) == (b.rid, b.wid, b.n_total_wins, b.length, b.n_supported, b.n_overlaps, b.n_alns), w
    finally:
        jd.close(); jh.close()


def test_skipped_alignments_and_name_classes():
    """parse_paf's rules (self overlaps, a second alignment of a (query, target) pair: overlaps.rs:175-185) and the ratio classes by read NAME
    (features.rs:494) are settled by the host's pre-pass for the device build: same descriptors, same counts, same first message."""
    sb = synth.generate(4, 1200, 10, seed=synth.SEED + 63, flank_min=30, flank_max=60)
    aln = sb.aln.copy()
    t0 = int(sb.tgt_aln_off[1])
    aln[t0 + 2, 0] = aln[t0, 0]                 # the third alignment of target 1 repeats the query of its first
    aln[t0 + 2, 1] = aln[t0, 1]
    t2 = int(sb.tgt_aln_off[2])
    aln[t2 + 1, 0] = sb.tgt_rid[2]              # a self overlap
    aln[t2 + 1, 1] = aln[t2 + 1, 6]
    sb2 = dataclasses.replace(sb, aln=aln)
    cls = np.arange(sb.n_reads, dtype=np.uint32)
    cls[7] = 3                                  # two reads with one name
    c = G.ctx()
    c.set_reads(sb2.seq, sb2.qual, sb2.off, cls)
    jd, ad, built_d, sk_d = _arrays(c, sb2, 256, host=False)
    jh, ah, built_h, sk_h = _arrays(c, sb2, 256, host=True)
    try:
        assert built_d == 1 and built_h == 0 and sk_d == sk_h and sk_d[0] == 2
        _same(ad, ah, "skips")
    finally:
        jd.close(); jh.close()
        G.load_synth(c, sb)


def test_unusual_inputs_fall_back_to_the_host_build():
    """What the device does not settle itself raises a flag and the host builds the job as before — same results for legal input (lengths padded
    to 11+ digits: the scan kernel does not read them), the reference's message for input it panics on."""
    sb = synth.generate(2, 1100, 6, seed=synth.SEED + 65, flank_min=30, flank_max=60)
    c = G.ctx()
    G.load_synth(c, sb)
    a = int(sb.tgt_aln_off[0])
    cg = sb.cigar(a).decode()
    m = re.match(r"(\d+)M", cg)
    padded = (f"{int(m.group(1)):012d}M" + cg[m.end():]).encode()
    cigs = [padded if i == a else sb.cigar(i) for i in range(len(sb.aln))]
    off = np.asarray(sb.tgt_aln_off, np.uint64)
    j1 = c.create_job(sb.tgt_rid, sb.aln, off, cigs, 256)
    j2 = api.job_from_synth(c, sb, 256)
    try:
        assert c._l.herro_debug_job_dev_built(j1.h) == 0 and c._l.herro_debug_job_dev_built(j2.h) == 1
        a1, a2 = c.job_arrays(j1), c.job_arrays(j2)
        for k in ("win", "tile_win", "tile_r0", "tgt_win_off"):
            assert np.array_equal(a1[k], a2[k]), k
        for f in a1["ow"].dtype.names:
            if f != "op_begin":     # (the padded text has room for fewer ops in front of it: the op array is laid out by text length)
                assert np.array_equal(a1["ow"][f], a2["ow"][f]), f
    finally:
        j1.close(); j2.close()
    rows = sb.aln[:1].copy()
    for bad, code in ((b"100M5X919M", -3), (b"5000M", -3)):     # a letter CigarIter panics on; a slice that overruns the target window
        with pytest.raises(api.HerroError) as e:
            c.create_job(sb.tgt_rid[:1], rows, np.array([0, 1], np.uint64), [bad], 256)
        assert e.value.code == code, str(e.value)
