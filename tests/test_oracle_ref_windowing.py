"""not-gpu: the oracle's `extract_windows` against the expected values of the reference's OWN windowing tests
(windowing.rs:309-606, `test_extract_windows1..7`, window size 5).

Those tests no longer compile: they call a WFA aligner module that is gone and an `OverlapWindow::new` with six
arguments (overlap, start of the window on the OTHER read, cigar_start_idx, cigar_start_offset, cigar_end_idx,
cigar_end_offset).  What they hold is still data: two sequences and, per window, the five numbers above.  The CIGAR the
aligner produced is not in the file; it is reconstructed here from the sequences (column by column: '=' where the bases
agree, 'X' where they differ, the inserted / deleted bases where the lengths demand) and is the only alignment of those
sequences consistent with the expected query starts.  The old op alphabet had separate Match / Mismatch ops — the current
enum still has both and `extract_windows` treats them alike (windowing.rs:125) — so '=' and 'X' runs go in as SEPARATE 'M'
ops (CigarIter does not merge neighbours, aligners.rs:264-293), which keeps the op indices the tests expect.

One thing did change since: the tail window.  Today `extract_windows` emits the window that ends with the read only if the
walk got past `len - (0.1 * window_size) as u32` (windowing.rs:66-71, 263), and at window size 5 that threshold is `len - 0`:
never.  The expected tail windows of tests 1, 2, 5 and 7 (the reads are 29, 31 and 36 bases long) therefore have no
counterpart; the oracle follows today's code and must NOT produce them (`_but_tail`).  Tests 3, 4 and 6 walk reads of 20 and
35 bases and match in full.

Tests 1, 2, 3, 5, 7 walk the target (is_target = true; the strand of test 5 plays no role on that side).  Tests 4 and 6 walk
the QUERY of the same alignment (is_target = false): the iterator they built swapped insertions and deletions (and, on the
reverse strand, the op order); the swapped CIGAR is written out below."""
import re

import numpy as np
import pytest

import oracle_lib as O

W = 5
Q1, T1 = b"AACTAAGTGTCGGTGGCTACTATATATCAGGTCCT", b"AGCTAGTGTCAATGGCTACTTTTCAGGTCCT"
EDIT = "1=1X3=1I5=2X8=1I1=1I1=1I9="        # test 1: edit-distance alignment (7 edits)
GAP = "1=1X3=1I5=2X8=3I1=1X9="             # tests 4-7: the three inserted bases as one gap


def _check_alignment(cig: str, q: bytes, t: bytes, tstart: int = 0):
    """the reconstructed CIGAR really aligns q to t[tstart:]: '=' on equal bases, 'X' on different ones, all bases used"""
    qi, ti = 0, tstart
    for n, op in re.findall(r"(\d+)([=XID])", cig):
        n = int(n)
        if op in "=X":
            for k in range(n):
                assert (q[qi + k] == t[ti + k]) == (op == "=")
            qi, ti = qi + n, ti + n
        elif op == "I":
            qi += n
        else:
            ti += n
    assert (qi, ti) == (len(q), len(t))


def _as_mid(cig: str) -> bytes:
    return cig.replace("=", "M").replace("X", "M").encode()


def _op_index(text: bytes):
    starts = [m.start() for m in re.finditer(rb"\d+[MID]", text)] + [len(text)]
    return {b: k for k, b in enumerate(starts)}


def _run(cig: str, q_len: int, t_len: int, tstart: int, is_target: bool, strand: int = 0):
    text = _as_mid(cig)
    row = (0, q_len, 0, q_len, strand, 1, t_len, tstart, t_len)   # Overlap::new(0, q_len, 0, q_len, strand, 1, t_len, tstart, t_len)
    walked = t_len if is_target else q_len
    rows = O.extract_windows(row, text, (walked + W - 1) // W, W, is_target)
    idx = _op_index(text)
    lens = [int(m.group(1)) for m in re.finditer(rb"(\d+)[MID]", text)]

    def old_end(b1, eo):
        """The END of a window changed representation since those tests were written: today it is (byte end of the last op the
        window touches, bases of that op inside the window) (windowing.rs:203-224, the old assignments are still there as
        comments `cigar_idx + 1` / `+ 2`); then it was (index of the first op NOT entirely inside, bases of it inside).  Same
        slice of the CIGAR: a partly used op k is (k + 1, off) today and (k, off) then; a fully used one (k + 1, len) and (k + 1, 0)."""
        k = idx[b1] - 1
        return (k, eo) if eo < lens[k] else (k + 1, 0)
    # -> window: (start on the other read, cigar_start_idx, cigar_start_offset, cigar_end_idx, cigar_end_offset)
    return {int(w): (int(qs), idx[int(b0)], int(so)) + old_end(int(b1), int(eo)) for w, ts, qs, qe, b0, so, b1, eo in rows}


def _but_tail(expected: dict) -> dict:
    return {w: v for w, v in expected.items() if w != max(expected)}


def test_reference_test1_edit_distance_alignment():
    _check_alignment(EDIT, Q1, T1)
    assert _run(EDIT, len(Q1), len(T1), 0, True) == _but_tail({  # windowing.rs:335-341
        0: (0, 0, 0, 4, 0), 1: (6, 4, 0, 5, 0), 2: (11, 5, 0, 6, 3), 3: (16, 6, 3, 8, 0),
        4: (22, 8, 0, 12, 3), 5: (29, 12, 3, 12, 8), 6: (34, 12, 8, 13, 0)})


def test_reference_test2_deletion_then_long_run():
    q, t = b"AATTTTTTTTTTTTTTTTTTTTGCACC", b"AAGCTTTTTTTTTTTTTTTTTTTTCGTCC"
    cig = "2=2D20=3X2="
    _check_alignment(cig, q, t)
    assert _run(cig, len(q), len(t), 0, True) == _but_tail({     # windowing.rs:371-376
        0: (0, 0, 0, 2, 1), 1: (3, 2, 1, 2, 6), 2: (8, 2, 6, 2, 11), 3: (13, 2, 11, 2, 16), 4: (18, 2, 16, 3, 1), 5: (23, 3, 1, 5, 0)})


def test_reference_test3_twenty_base_insertion_stays_with_its_window():
    q, t = b"ATCGTTTTTTTTTTTTTTTTTTTTATCGAAAAAAAAAAAA", b"ATCGATCGAAAAAAAAAAAA"
    cig = "4=20I16="
    _check_alignment(cig, q, t)
    assert _run(cig, len(q), len(t), 0, True) == {               # windowing.rs:408-411
        0: (0, 0, 0, 2, 1), 1: (25, 2, 1, 2, 6), 2: (30, 2, 6, 2, 11), 3: (35, 2, 11, 3, 0)}


EXPECT_GAP = {0: (0, 0, 0, 4, 0), 1: (6, 4, 0, 5, 0), 2: (11, 5, 0, 6, 3), 3: (16, 6, 3, 8, 0),
              4: (24, 8, 0, 10, 3), 5: (29, 10, 3, 10, 8), 6: (34, 10, 8, 11, 0)}


def test_reference_test5_reverse_strand_target_side():
    _check_alignment(GAP, Q1, T1)
    assert _run(GAP, len(Q1), len(T1), 0, True, strand=1) == _but_tail(EXPECT_GAP)          # windowing.rs:503-509


def test_reference_test7_overlap_starting_at_a_window_boundary():
    t = b"TTTTT" + T1
    _check_alignment(GAP, Q1, t, tstart=5)
    assert _run(GAP, len(Q1), len(t), 5, True) == _but_tail({w + 1: v for w, v in EXPECT_GAP.items()})   # windowing.rs:591-597: window 0 stays empty


def _swap(cig: str, reverse: bool) -> str:
    ops = re.findall(r"\d+[=XID]", cig)
    ops = [o[:-1] + {"I": "D", "D": "I"}.get(o[-1], o[-1]) for o in ops]
    return "".join(reversed(ops) if reverse else ops)


def test_reference_test6_query_side():
    assert _run(_swap(GAP, False), len(Q1), len(T1), 0, False) == {              # windowing.rs:547-553
        0: (0, 0, 0, 3, 0), 1: (5, 3, 0, 4, 4), 2: (9, 4, 4, 6, 2), 3: (14, 6, 2, 6, 7),
        4: (19, 6, 7, 9, 0), 5: (21, 9, 0, 10, 4), 6: (26, 10, 4, 11, 0)}


def test_reference_test4_query_side_reverse_strand():
    assert _run(_swap(GAP, True), len(Q1), len(T1), 0, False, strand=1) == {     # windowing.rs:459-465
        0: (0, 0, 0, 0, 5), 1: (5, 0, 5, 2, 0), 2: (10, 2, 0, 4, 1), 3: (12, 4, 1, 4, 6),
        4: (17, 4, 6, 6, 1), 5: (22, 6, 1, 8, 0), 6: (26, 8, 0, 11, 0)}
