"""-m gpu: pileup feature generation (HIP, through the C-ABI) is bit-exact with the oracle's
restatement of features.rs:326-583 on seeded synthetic overlap batches, including the edge
cases the reference's code paths distinguish."""
import numpy as np
import pytest

import gpu_common as G
import oracle_lib as O
from herro_amd import api, synth

pytestmark = pytest.mark.gpu

CASES = {
    "baseline_w4096": dict(W=4096, n=3, tl=4 * 4096, ov=32, kw={}),
    "ragged_tail_w4096": dict(W=4096, n=2, tl=2 * 4096 + 1234, ov=20, kw=dict(p_partial=0.3)),
    "few_overlaps": dict(W=512, n=4, tl=2048, ov=3, kw=dict(flank_min=60, flank_max=90, p_partial=0.5)),
    "many_overlaps": dict(W=256, n=2, tl=1024, ov=70, kw=dict(flank_min=30, flank_max=60)),
    "long_indels_filtered": dict(W=512, n=3, tl=2048, ov=24, kw=dict(flank_min=60, flank_max=90, p_long_indel=0.05)),
    "noisy": dict(W=256, n=4, tl=1500, ov=16, kw=dict(flank_min=30, flank_max=60, p_sub=0.05, p_ins=0.05, p_del=0.05, p_partial=0.2)),
    "n_bases_quirk": dict(W=256, n=3, tl=1024, ov=12, kw=dict(flank_min=30, flank_max=60, p_n_base=0.01)),
    "tiny_window": dict(W=16, n=3, tl=200, ov=8, kw=dict(flank_min=2, flank_max=5)),
    "no_overlaps": dict(W=256, n=2, tl=700, ov=0, kw=dict(flank_min=30, flank_max=60)),
    "largest_window_w8192": dict(W=8192, n=2, tl=3 * 8192 + 100, ov=12, kw=dict(p_partial=0.3)),
    "window_not_a_multiple_of_32": dict(W=1000, n=3, tl=3500, ov=10, kw=dict(flank_min=100, flank_max=200, p_partial=0.3)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_features_bit_exact(name):
    cs = CASES[name]
    sb = synth.generate(cs["n"], cs["tl"], cs["ov"], seed=synth.SEED + sum(map(ord, name)), **cs["kw"])
    c = G.ctx()
    G.load_synth(c, sb)
    store = O.store_from_synth(sb)
    job = api.job_from_synth(c, sb, cs["W"])
    job.featurize()
    n = G.compare_features(job, sb, store, cs["W"])
    assert n > 0
    job.close()


def _with_cigars(sb, new):
    """SynthBatch with some alignments replaced: new = {alignment index: (row of 9 or None, cigar bytes)}."""
    import dataclasses
    cigs = [sb.cigar(a) for a in range(len(sb.aln))]
    aln = sb.aln.copy()
    for a, (row, cg) in new.items():
        cigs[a] = cg
        if row is not None:
            aln[a, :9] = row
    off = np.zeros(len(cigs), np.uint64)
    o = 0
    for a, cg in enumerate(cigs):
        off[a] = o
        aln[a, 9] = len(cg)
        o += len(cg)
    return dataclasses.replace(sb, aln=aln, cig=np.frombuffer(b"".join(cigs), np.uint8).copy(), cig_off=off)


def test_consecutive_and_leading_insertions():
    """CIGARs minimap2 never writes but the reference processes (features.rs:213-229): consecutive insertion ops (the later
    one overwrites the earlier one's rows from the first) and an alignment that starts inside a window with an insertion
    (its bases land behind the position in front of the overlap)."""
    import re
    W = 256
    sb = synth.generate(3, 1100, 14, seed=77, flank_min=30, flank_max=60, p_ins=0.04, p_del=0.02)
    new, n_pairs, n_lead = {}, 0, 0
    for a in range(len(sb.aln)):
        cg = sb.cigar(a).decode()
        row = [int(x) for x in sb.aln[a, :9]]
        kind = a % 4
        if kind in (0, 1, 2):   # split insertions of length >= 2 (>= 3) into two (three) consecutive ops
            def split(m, kind=kind):
                n = int(m.group(1))
                if kind == 0 and n >= 2: return f"1I{n - 1}I"          # the second, longer one hides the first completely
                if kind == 1 and n >= 2: return f"{n - 1}I1I"          # the second hides only the first row
                if kind == 2 and n >= 3: return f"1I1I{n - 2}I"
                return m.group(0)
            cg2 = re.sub(r"(\d+)I", split, cg)
            if cg2 != cg:
                new[a] = (None, cg2.encode()); n_pairs += 1
        else:                   # start 5 bases into the first window, with a 2-base insertion in front
            m = re.match(r"(\d+)M", cg)
            if m and int(m.group(1)) >= 8 and row[7] % W == 0:
                lead = int(m.group(1))
                row2 = list(row)
                row2[7] += 5                                   # tstart
                if row[4] == 0: row2[2] += 3                   # forward: 5 bases cut, 2 inserted bases kept -> qstart + 3
                else: row2[3] -= 3                             # reverse: the alignment's start is the query's end
                new[a] = (row2, (f"2I{lead - 5}M" + cg[m.end():]).encode()); n_lead += 1
    # an insertion split at a window boundary leaves its second half at the head of the next window's slice, where the
    # reference panics (max_ins[tpos - 1], tpos == 0): such candidates are dropped here (the oracle says which) ...
    tgt_of = np.searchsorted(sb.tgt_aln_off, np.arange(len(sb.aln)), side="right") - 1
    kept, panicking = {}, []
    for a, m in new.items():
        trial = _with_cigars(sb, {**kept, a: m})
        try:
            rid, rows, cigs = O.target_alignments(trial, int(tgt_of[a]))
            O.store_from_synth(trial).extract_features(rid, rows, cigs, W)
            kept[a] = m
        except O.OracleError:
            panicking.append(a)
    n_pairs = sum(1 for a in kept if kept[a][0] is None)
    n_lead = len(kept) - n_pairs
    assert n_pairs >= 10 and n_lead >= 1, (n_pairs, n_lead, len(panicking))
    sb2 = _with_cigars(sb, kept)
    c = G.ctx()
    G.load_synth(c, sb2)
    store = O.store_from_synth(sb2)
    job = api.job_from_synth(c, sb2, W)
    assert job.skipped() == (0, 0)
    job.featurize()
    assert G.compare_features(job, sb2, store, W) > 0
    job.close()
    for a in panicking[:3]:   # ... and refused by the library like every input the reference panics on
        bad = _with_cigars(sb, {a: new[a]})
        with pytest.raises(api.HerroError) as e:
            api.job_from_synth(c, bad, W, targets=[int(tgt_of[a])])
        assert e.value.code == -3, str(e.value)


def test_thousands_of_overlaps_in_one_window():
    """features.rs:376-409 ranks any number of overlaps: 4200 in one window (threshold of pass 1 = 420 columns)."""
    sb = synth.generate(1, 128, 4200, seed=31, flank_min=4, flank_max=10)
    c = G.ctx()
    G.load_synth(c, sb)
    store = O.store_from_synth(sb)
    job = api.job_from_synth(c, sb, 64)
    job.featurize()
    assert job.info(0).n_overlaps > 4000
    G.compare_features(job, sb, store, 64)
    job.close()


def test_duplicate_read_names_share_ratio():
    # the reference keys haplotype ratios by read *name* (features.rs:494): two reads with the same id
    sb = synth.generate(1, 1024, 10, seed=99, flank_min=30, flank_max=60)
    names = [sb.read_name(i) for i in range(sb.n_reads)]
    names[3] = names[5]
    cls = np.arange(sb.n_reads, dtype=np.uint32)
    cls[5] = 3
    c = G.ctx()
    c.set_reads(sb.seq, sb.qual, sb.off, cls)
    store = O.Store(sb.seq, sb.qual, sb.off, names)
    job = api.job_from_synth(c, sb, 256)
    job.featurize()
    G.compare_features(job, sb, store, 256)
    job.close()


def test_reference_panics_become_errors():
    sb = synth.generate(1, 1024, 4, seed=5, flank_min=30, flank_max=60)
    c = G.ctx()
    G.load_synth(c, sb)
    rows = sb.aln[:1].copy()
    with pytest.raises(api.HerroError) as e:
        c.create_job(sb.tgt_rid[:1], rows, np.array([0, 1], np.uint64), [b"100M5X919M"], 256)
    assert e.value.code == -3
    with pytest.raises(api.HerroError):
        c.create_job(sb.tgt_rid[:1], rows, np.array([0, 1], np.uint64), [b"5000M"], 256)  # overruns the target


def test_properties_full_size():
    """Size-independent invariants (SURVEY.md §8 a) on a larger full-size batch."""
    sb = synth.generate(8, 4 * 4096, 32, seed=4242)
    c = G.ctx()
    G.load_synth(c, sb)
    job = api.job_from_synth(c, sb, 4096)
    job.featurize()
    for w in range(job.n_windows):
        g = job.window(w, encoded=True)
        b = g.bases
        assert (b[:, 0] <= 4).all()                               # target column: ACGT or '*', never '.'
        isbase = (b < 4) | ((b >= 5) & (b <= 8))
        assert isbase.any(axis=1).all()                           # no all-gap rows survive
        assert (b[:, 0] != 4).sum() == g.info.win_len             # every target base has its row
        assert (b[:, g.info.n_alns + 1:] == 10).all()             # unused rows are '.'
        key = g.sup_pos.astype(np.int64) * 256 + g.sup_ins
        assert (np.diff(key) > 0).all()                           # informative positions strictly increasing
        assert g.quals.min() >= 33
        assert ((g.quals == 33) | isbase).all()                   # non-base cells carry the default quality
    job.close()


def test_job_from_paf_text_equals_job_from_arrays():
    """PAF text -> herro_paf_parse -> herro_job_create gives the same windows as the array entry (f2: ingest)."""
    sb = synth.generate(4, 1500, 12, seed=9, flank_min=40, flank_max=80, p_partial=0.3)
    c = G.ctx()
    G.load_synth(c, sb)
    names = [sb.read_name(i).encode() for i in range(sb.n_reads)]
    lines = []
    for t in range(sb.n_targets):
        for a in range(int(sb.tgt_aln_off[t]), int(sb.tgt_aln_off[t + 1])):
            r = sb.aln[a]
            lines.append(b"\t".join([names[r[0]], str(r[1]).encode(), str(r[2]).encode(), str(r[3]).encode(), b"-" if r[4] else b"+",
                                      names[r[5]], str(r[6]).encode(), str(r[7]).encode(), str(r[8]).encode(), b"60", b"60", b"255",
                                      b"cg:Z:" + sb.cigar(a)]))
    paf = api.Paf(names, text=b"\n".join(lines) + b"\n")
    assert paf.targets.tolist() == sb.tgt_rid.tolist()
    ja = api.job_from_synth(c, sb, 512)
    jp = c.create_job_from_paf(paf, 512)
    ja.featurize()
    jp.featurize()
    assert ja.n_windows == jp.n_windows
    for w in range(ja.n_windows):
        a, b = ja.window(w, encoded=True), jp.window(w, encoded=True)
        assert a.info.length == b.info.length and a.info.n_supported == b.info.n_supported
        assert np.array_equal(a.bases, b.bases) and np.array_equal(a.quals, b.quals)
    ja.close(); jp.close(); paf.close()


def test_features_directory_equals_oracle(tmp_path):
    """`herro features` sink (features.rs:724-764): files written by the library (herro_job_write_features) == files written
    from the oracle's windows by an independent writer (numpy)."""
    import os
    from herro_amd import io as hio
    sb = synth.generate(3, 900, 10, seed=21, flank_min=30, flank_max=60, p_partial=0.2)
    c = G.ctx()
    G.load_synth(c, sb)
    store = O.store_from_synth(sb)
    job = api.job_from_synth(c, sb, 256)
    job.featurize()
    w0 = 0
    for t in range(sb.n_targets):
        rid, rows, cigs = O.target_alignments(sb, t)
        res = store.extract_features(rid, rows, cigs, 256)
        d = tmp_path / "oracle" / sb.read_name(rid)
        os.makedirs(d, exist_ok=True)
        for w in range(len(res)):
            ow = res.window(w)
            np.save(d / f"{w}.features.npy", np.ascontiguousarray(np.stack([ow.bases, ow.quals], axis=0)))
            sup = np.zeros(len(ow.sup_pos), hio.SUPPORTED_DTYPE)
            sup["pos"], sup["ins"] = ow.sup_pos, ow.sup_ins
            np.save(d / f"{w}.supported.npy", sup)
            (d / f"{w}.ids.txt").write_text("".join(sb.read_name(int(q)) + "\n" for q in ow.qids))
        w0 += len(res)
    n = hio.write_job_features(job, str(tmp_path / "product"), [sb.read_name(i) for i in range(sb.n_reads)])
    assert n == w0 == job.n_windows
    n_files = 0
    for root, _, files in os.walk(tmp_path / "oracle"):
        for f in files:
            a = open(os.path.join(root, f), "rb").read()
            b = open(os.path.join(str(root).replace("/oracle/", "/product/"), f), "rb").read()
            assert a == b, (root, f)
            n_files += 1
    assert n_files == 3 * w0
    job.close()


def test_zero_copy_job_creation_from_a_registered_blob():
    """herro_host_register (round 5): a job whose CIGAR texts lie in a registered buffer is created without the staging pass — the
    range goes up as it is, every text keeps its host alignment modulo 16 and the scan kernel skips the bytes in front of it.
    The blob is put at an odd address so that no text is 16-byte aligned; features must equal the oracle's as on the staged path."""
    W = 4096
    sb = synth.generate(3, 2 * 4096 + 777, 20, seed=91, p_partial=0.3)
    c = G.ctx()
    G.load_synth(c, sb)
    store = O.store_from_synth(sb)
    blob = np.concatenate([np.full(5, ord("M"), np.uint8), sb.cig, np.full(40, ord("I"), np.uint8)])   # letters in front and behind: they must not leak into a text
    view = blob[5:5 + len(sb.cig)]
    before = api.lib().herro_debug_zero_copy_jobs()
    c.register_host(blob)
    try:
        job = c.create_job(sb.tgt_rid, sb.aln, sb.tgt_aln_off, None, W, cig_blob=view, cig_off=sb.cig_off)
        assert api.lib().herro_debug_zero_copy_jobs() == before + 1
        job.featurize()
        assert G.compare_features(job, sb, store, W) > 0
        job.close()
        # single targets (texts in the middle of the range), and a job after unregistering takes the staged path again
        job = c.create_job(sb.tgt_rid[1:2], sb.aln[int(sb.tgt_aln_off[1]):int(sb.tgt_aln_off[2])], sb.tgt_aln_off[1:3] - sb.tgt_aln_off[1], None, W,
                           cig_blob=view, cig_off=sb.cig_off[int(sb.tgt_aln_off[1]):int(sb.tgt_aln_off[2])])
        job.featurize()
        assert G.compare_features(job, sb, store, W, targets=[1]) > 0
        job.close()
    finally:
        c.unregister_host(blob)
    n = api.lib().herro_debug_zero_copy_jobs()
    job = c.create_job(sb.tgt_rid, sb.aln, sb.tgt_aln_off, None, W, cig_blob=view, cig_off=sb.cig_off)
    assert api.lib().herro_debug_zero_copy_jobs() == n
    job.featurize()
    assert G.compare_features(job, sb, store, W) > 0
    job.close()


def test_window_slice_with_a_query_span_above_the_staged_planes():
    """k_cols stages a slice's query planes in LDS — cols_qcap(nw) = nw + 40 words; a window whose slice consumes more query than that (thirty insertions of 50 bases,
    the longest the long-indel filter keeps, inside one 256-base window: 1650 query bases = 53 words against 48) reads the read store directly instead.  Both
    strands; the deletions that keep the alignment's spans are spread over the other windows."""
    W, n_ins, ins = 256, 30, 50
    sb = synth.generate(2, 2300, 12, seed=4242, flank_min=30, flank_max=60)
    tgt_of = np.searchsorted(sb.tgt_aln_off, np.arange(len(sb.aln)), side="right") - 1
    kept, strands = {}, set()
    for a in range(len(sb.aln)):
        row = [int(x) for x in sb.aln[a, :9]]
        qspan, tspan = row[3] - row[2], row[8] - row[7]
        head = (-row[7]) % W + 8                                  # the insertions start 8 positions into a window
        used_t, used_q = head + 5 * n_ins, head + (ins + 5) * n_ins
        rest_t, rest_q = tspan - used_t, qspan - used_q
        d_total = rest_t - rest_q
        if rest_q < 64 or d_total < 0 or a % 3 == 2:
            continue
        dels = [40] * (d_total // 40) + ([d_total % 40] if d_total % 40 else [])
        if rest_q < len(dels) + 1:
            continue
        ms = [rest_q // (len(dels) + 1)] * (len(dels) + 1)
        ms[-1] += rest_q - sum(ms)
        tail = "".join(f"{m}M{d}D" for m, d in zip(ms, dels)) + f"{ms[-1]}M"
        cg = f"{head}M" + "".join(f"{ins}I5M" for _ in range(n_ins)) + tail
        trial = _with_cigars(sb, {**kept, a: (None, cg.encode())})
        try:
            rid, rows, cigs = O.target_alignments(trial, int(tgt_of[a]))
            O.store_from_synth(trial).extract_features(rid, rows, cigs, W)
            kept[a] = (None, cg.encode())
            strands.add(row[4])
        except O.OracleError:
            pass
    assert len(kept) >= 4 and strands == {0, 1}, (len(kept), strands)
    sb2 = _with_cigars(sb, kept)
    c = G.ctx()
    G.load_synth(c, sb2)
    store = O.store_from_synth(sb2)
    job = api.job_from_synth(c, sb2, W)
    assert job.skipped() == (0, 0)
    job.featurize()
    assert G.compare_features(job, sb2, store, W) > 0
    job.close()


@pytest.mark.parametrize("trial", G.sweep_trials(16))
def test_features_bit_exact_on_random_configurations(trial):
    """A seeded sweep over window sizes, depths, error rates and partial overlaps (round 6: k_cols' plane build was re-formulated — op lanes write the word they start in, one op
    per word left to the walk — and every such combination runs the multi-batch, clipped and reverse-strand branches in different proportions): oracle vs HIP, cell by cell."""
    g = np.random.default_rng(0x6a09e667 + trial)
    W = int(g.choice([16, 48, 100, 256, 512, 1000, 2048, 4096, 8192]))
    n_win = int(g.integers(1, 4))
    tl = n_win * W + int(g.integers(0, W))
    ov = int(g.integers(1, 40))
    fl = max(2, min(W // 4, 400))
    kw = dict(flank_min=fl // 2 + 1, flank_max=fl + 2, p_sub=float(g.choice([0.002, 0.01, 0.05])), p_ins=float(g.choice([0.002, 0.01, 0.06])),
              p_del=float(g.choice([0.002, 0.01, 0.06])), p_partial=float(g.choice([0.0, 0.3, 0.7])), p_long_indel=float(g.choice([0.0, 0.0, 0.02])),
              p_snp=float(g.choice([0.0, 0.01])))
    sb = synth.generate(int(g.integers(1, 4)), tl, ov, seed=int(g.integers(1, 1 << 30)), **kw)
    c = G.ctx()
    G.load_synth(c, sb)
    store = O.store_from_synth(sb)
    job = api.job_from_synth(c, sb, W)
    job.featurize()
    assert G.compare_features(job, sb, store, W) > 0, (W, tl, ov, kw)
    job.close()
