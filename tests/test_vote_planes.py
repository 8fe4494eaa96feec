"""not-gpu: the rules k_rows applies in position space (csrc/pileup_core.h, compiled for host and device from the same source) against the
reference's own rules on EVERY count vector.

k_rows (the lean feature path, round 5) never builds the [L',31] matrix: a base row's symbol counts live in saturating 2-bit bit-sliced
counters (0, 1, 2, ">= 3").  That is enough for the informative-row rule (features.rs:681-722 with thresh = (31 * 0.1) as usize = 3) and —
the part that needs an argument — for the decoder's majority vote (consensus.rs:178-200), because the vote is only consulted on rows that are
NOT informative, where at most one symbol reaches 3.  Here the claim is checked exhaustively instead of argued."""
import itertools

import numpy as np

from herro_amd import api


def ref_supported(c):                       # features.rs:694-718 on the final matrix
    return sum(1 for x in c if x >= 3) >= 2


def ref_vote(c, tb):                        # consensus.rs:186-200
    order = sorted(range(5), key=lambda q: -c[q])      # stable: ties keep A C G T * order
    m0, m1, b0, b1 = c[order[0]], c[order[1]], order[0], order[1]
    return tb if (m0 < 2 or (m0 == m1 and (b0 == tb or b1 == tb))) else b0


def test_base_row_rules_on_every_count_vector():
    L = api.lib()
    cases = [(c, t) for c in itertools.product(range(6), repeat=5) for t in range(4) if c[t] >= 1]   # the target column shows its own base
    counts = np.array([c for c, _ in cases], np.uint8)
    target = np.array([t for _, t in cases], np.uint8)
    rng = np.random.default_rng(1)
    for split in (None, (rng.integers(0, 6, counts.shape)).astype(np.uint8)):      # one counter set / two sets merged (the halves of the workgroup)
        sup = np.zeros(len(cases), np.uint8)
        vote = np.full(len(cases), 9, np.uint8)
        rc = L.herro_debug_base_row_votes(counts.ctypes.data, None if split is None else split.ctypes.data, target.ctypes.data, len(cases),
                                          sup.ctypes.data, vote.ctypes.data)
        assert rc == 0
        n_vote = 0
        for i, (c, t) in enumerate(cases):
            assert bool(sup[i]) == ref_supported(c), (c, t)
            assert vote[i] <= 4, (c, t, vote[i])                          # a defined code on every row, informative or not
            if not sup[i]:
                assert vote[i] == ref_vote(c, t), (c, t, int(vote[i]), ref_vote(c, t))
                n_vote += 1
        assert n_vote > 4000
    # counts far above the saturation point behave like 3
    big = np.array([[31, 0, 0, 0, 0], [20, 11, 0, 0, 0], [1, 0, 0, 0, 30], [2, 2, 0, 0, 27], [1, 2, 2, 0, 0]], np.uint8)
    tg = np.array([0, 1, 0, 1, 0], np.uint8)
    sup = np.zeros(5, np.uint8); vote = np.zeros(5, np.uint8)
    assert L.herro_debug_base_row_votes(big.ctypes.data, None, tg.ctypes.data, 5, sup.ctypes.data, vote.ctypes.data) == 0
    assert sup.tolist() == [0, 1, 0, 0, 0]
    assert [int(vote[i]) for i in (0, 2, 3, 4)] == [ref_vote(big[i].tolist(), int(tg[i])) for i in (0, 2, 3, 4)]


def test_vote_on_exact_counts_equals_the_reference_rule():
    """insertion rows: exact counts, the target shows '*' (tb = 4); and the general form for any target"""
    L = api.lib()
    for c in itertools.product(range(5), repeat=5):
        arr = np.array(c, np.uint32)
        for tb in range(5):
            assert L.herro_debug_vote5(arr.ctypes.data, tb) == ref_vote(c, tb), (c, tb)
