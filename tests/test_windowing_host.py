"""Product host windowing (herro_amd/csrc/windowing.hpp, op-index slices) against the oracle's
restatement of windowing.rs:44-273 (byte-index slices) — no GPU needed."""
import re

import numpy as np
import pytest

import oracle_lib as O
from herro_amd import api, synth


def op_byte_ranges(cigar: bytes):
    out, pos = [], 0
    for m in re.finditer(rb"\d+[MID]", cigar):
        out.append((m.start(), m.end()))
        pos = m.end()
    assert pos == len(cigar)
    return out


def to_byte_rows(rows, cigar):
    rng = op_byte_ranges(cigar)
    conv = []
    for w, ts, qs, qe, lo, hi, so, eo in rows.tolist():
        b0 = rng[lo][0] if lo < len(rng) else len(cigar)
        conv.append([w, ts, qs, qe, b0, so, rng[hi - 1][1], eo])
    return conv


def test_appendix_a():
    rows = api.debug_extract_windows((1, 11, 0, 11, 0, 0, 12, 0, 12), b"4M1I3M2D3M", 3, 5)
    assert rows.tolist() == [[0, 0, 0, 6, 0, 3, 0, 1], [1, 5, 6, 9, 2, 5, 1, 1]]


@pytest.mark.parametrize("W,tl,kw", [(64, 1000, dict(p_partial=0.4, flank_min=10, flank_max=40)),
                                     (100, 1037, dict(p_partial=0.3, flank_min=0, flank_max=30, p_long_indel=0.02)),
                                     (4096, 3 * 4096 + 777, dict(p_partial=0.3)),
                                     (512, 2048, dict(p_ins=0.05, p_del=0.05, p_partial=0.2, flank_min=60, flank_max=90))])
def test_against_oracle(W, tl, kw):
    sb = synth.generate(6, tl, 12, seed=W * 7 + tl, **kw)
    n_emitted = 0
    for a in range(len(sb.aln)):
        row = tuple(int(x) for x in sb.aln[a, :9])
        cig = sb.cigar(a)
        nwin = (row[6] + W - 1) // W
        try:
            want = O.extract_windows(row, cig, nwin, W).tolist()
        except O.OracleError:
            with pytest.raises(api.HerroError):
                api.debug_extract_windows(row, cig, nwin, W)
            continue
        got = to_byte_rows(api.debug_extract_windows(row, cig, nwin, W), cig)
        assert got == want, (row, cig[:200])
        n_emitted += len(got)
    assert n_emitted > 0


def test_hand_cases():
    # op ending exactly on a boundary followed by an insertion: insertion stays with the earlier window
    row = (1, 40, 0, 23, 0, 0, 20, 0, 20)
    cig = b"10M3I10M"
    want = O.extract_windows(row, cig, 2, 10).tolist()
    assert to_byte_rows(api.debug_extract_windows(row, cig, 2, 10), cig) == want
    assert want[0][6] == len(b"10M3I") and want[0][7] == 3
    # one op spanning several windows, deletion across a boundary
    for cig, q in ((b"35M", 35), (b"8M7D20M", 28), (b"12M1I3M4D16M", 32)):
        row = (1, 64, 0, q, 0, 0, 35, 0, 35)
        want = O.extract_windows(row, cig, 4, 10).tolist()
        assert to_byte_rows(api.debug_extract_windows(row, cig, 4, 10), cig) == want
    # bad op
    with pytest.raises(api.HerroError):
        api.debug_extract_windows((1, 64, 0, 35, 0, 0, 35, 0, 35), b"30M5X", 4, 10)
    with pytest.raises(api.HerroError):
        api.debug_extract_windows((1, 64, 0, 35, 0, 0, 35, 0, 35), b"0M35M", 4, 10)


def test_reference_windowing_vectors():
    """The target-side cases of the reference's own windowing tests (windowing.rs:309-606; the oracle is held to their expected
    values in test_oracle_ref_windowing.py) through the PRODUCT's windowing: identical rows."""
    import test_oracle_ref_windowing as R
    q1, t1 = R.Q1, R.T1
    cases = [(R.EDIT, len(q1), len(t1), 0), (R.GAP, len(q1), len(t1), 0), (R.GAP, len(q1), len(t1) + 5, 5),
             ("2=2D20=3X2=", 27, 29, 0), ("4=20I16=", 40, 20, 0)]
    for cig, ql, tl, ts in cases:
        text = R._as_mid(cig)
        row = (0, ql, 0, ql, 0, 1, tl, ts, tl)
        nwin = (tl + R.W - 1) // R.W
        want = O.extract_windows(row, text, nwin, R.W).tolist()
        assert want and to_byte_rows(api.debug_extract_windows(row, text, nwin, R.W), text) == want
