"""Hand-traced known-answer example, re-derived from the reference
code: windowing.rs:44-273, features.rs:44-95,110-266,585-679.  Window size 5."""
import numpy as np
import pytest

import oracle_lib as O

T = b"ACGTACGTACGT"
Q = b"ACGTGACGCGT"
CIG = b"4M1I3M2D3M"
ROW = (1, 11, 0, 11, 0, 0, 12, 0, 12)  # qid,qlen,qs,qe,strand,tid,tlen,ts,te


def test_extract_windows_kat():
    w = O.extract_windows(ROW, CIG, 3, 5)
    # window, tstart, qstart, qend, cs_idx, cs_off, ce_idx, ce_off
    assert w.tolist() == [[0, 0, 0, 6, 0, 0, 6, 1], [1, 5, 6, 9, 4, 1, 10, 1]]


def _store(q=Q, t=T):
    seq = np.frombuffer(t + q, np.uint8)
    qual = np.frombuffer(bytes([40] * len(t)) + bytes(range(50, 50 + len(q))), np.uint8)
    off = np.array([0, len(t), len(t) + len(q)], np.uint64)
    return O.Store(seq, qual, off, ["t", "q desc"])


def test_extract_features_kat():
    st = _store()
    res = st.extract_features(0, [ROW], [CIG], 5)
    assert len(res) == 3
    w0, w1, w2 = (res.window(i) for i in range(3))
    assert w0.p1_acc.tolist() == [np.float32(5) / np.float32(6)]
    assert w1.p1_acc.tolist() == [np.float32(3) / np.float32(5)]
    assert w0.max_ins.tolist() == [0, 0, 0, 1, 0] and w0.p1_L == 6
    assert w1.max_ins.tolist() == [0, 0, 0, 0, 0] and w1.p1_L == 5
    assert w0.bases[:, 0].tobytes() == b"ACGT*A" and w0.bases[:, 1].tobytes() == b"ACGTGA"
    assert w1.bases[:, 0].tobytes() == b"CGTAC" and w1.bases[:, 1].tobytes() == b"CG**C"
    assert (w0.bases[:, 2:] == ord(".")).all() and (w1.bases[:, 2:] == ord(".")).all()
    # quals: target 40 everywhere there is a base, '!' elsewhere; query quals follow the bases
    assert w0.quals[:, 0].tolist() == [40, 40, 40, 40, 33, 40]
    assert w0.quals[:, 1].tolist() == [50, 51, 52, 53, 54, 55]
    assert w1.quals[:, 1].tolist() == [56, 57, 33, 33, 58]
    # tail window (2 target bases) has no overlap with W=5: single target column
    assert w2.bases.shape == (2, 31) and w2.bases[:, 0].tobytes() == b"GT" and len(w2.qids) == 0
    assert w0.n_alns == 1 and len(w0.sup_pos) == 0
    # n_alns == 1 everywhere -> consensus returns None -> read not written (consensus.rs:90-98)
    assert res.consensus_fasta(np.zeros((0, 5), np.float32)) == ""


def test_reverse_strand_kat():
    # same alignment, query stored reverse-complemented: lower-case bases, '#' gaps, reversed quals
    rc = bytes({65: 84, 67: 71, 71: 67, 84: 65}[c] for c in reversed(Q))
    st = _store(q=rc)
    row = (1, 11, 0, 11, 1, 0, 12, 0, 12)
    res = st.extract_features(0, [row], [CIG], 5)
    w0, w1 = res.window(0), res.window(1)
    assert w0.bases[:, 1].tobytes() == b"acgtga"
    assert w1.bases[:, 1].tobytes() == b"cg##c"
    assert w0.quals[:, 1].tolist() == [60, 59, 58, 57, 56, 55]
    assert w1.quals[:, 1].tolist() == [54, 53, 33, 33, 52]


def test_bad_cigar_panics():
    st = _store()
    with pytest.raises(O.OracleError):
        st.extract_features(0, [ROW], [b"4M1X7M"], 5)  # aligners.rs:281-286


def test_merged_synthetic_batches_keep_every_target_intact():
    """synth.generate_parallel / merge (what bench.py feeds the GPU): a target of the merged batch gives the
    oracle exactly the windows it gave inside its own chunk."""
    import oracle_lib as O
    from herro_amd import synth
    parts = [synth.generate(2, 600, 6, seed=50 + i, flank_min=20, flank_max=40) for i in range(3)]
    m = synth.merge(parts)
    assert m.n_targets == 6 and m.n_reads == sum(p.n_reads for p in parts)
    sm = O.store_from_synth(m)
    t = 0
    for p in parts:
        sp = O.store_from_synth(p)
        for tl in range(p.n_targets):
            rid_p, rows_p, cigs_p = O.target_alignments(p, tl)
            rid_m, rows_m, cigs_m = O.target_alignments(m, t)
            assert cigs_p == cigs_m and m.read_seq(rid_m) == p.read_seq(rid_p)
            rp, rm_ = sp.extract_features(rid_p, rows_p, cigs_p, 256), sm.extract_features(rid_m, rows_m, cigs_m, 256)
            assert len(rp) == len(rm_)
            for w in range(len(rp)):
                a, b = rp.window(w), rm_.window(w)
                assert (a.bases == b.bases).all() and (a.quals == b.quals).all() and list(a.sup_pos) == list(b.sup_pos)
            t += 1
    g = synth.generate_parallel(5, 600, 6, seed=9, chunk=2, flank_min=20, flank_max=40)
    assert g.n_targets == 5 and int(g.tgt_aln_off[-1]) == len(g.aln) and int(g.off[-1]) == len(g.seq)
