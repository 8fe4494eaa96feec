"""Pin the oracle's 2-bit codec against the reference's own live unit tests
(haec_io.rs:191-299) — the only golden vectors the reference holds for this path."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "codec_vectors.json")))


@pytest.mark.parametrize("v", G["encode"], ids=lambda v: v["test"])
def test_encode(v):
    w = O.encode(v["seq"].encode())
    assert w.tolist() == v["words"]


@pytest.mark.parametrize("v", G["decode"], ids=lambda v: v["test"])
def test_decode(v):
    out = O.decode(np.array(v["words"], np.uint64), v["length"], v["start"], v["end"], v["rc"])
    assert out == v["out"].encode()


@pytest.mark.parametrize("v", G["subseq"], ids=lambda v: v["test"])
def test_subseq(v):
    w = O.encode(v["seq"].encode())
    out = O.decode(w, len(v["seq"]), v["start"], v["end"], v["rc"])
    assert out == v["out"].encode()


def test_case_folding_and_n_quirk():
    # haec_io.rs:7-15: lower case folds; haec_io.rs:126-128: 255 is OR-ed unmasked, so a
    # non-ACGT byte turns bases i..i+3 of the same 32-base word into T.
    assert O.encode(b"acgt").tolist() == O.encode(b"ACGT").tolist()
    w = O.encode(b"ACNACGTAC")
    assert O.decode(w, 9, 0, 9, False) == b"ACTTTTTAC"
    # at the end of a word the overflow bits are dropped
    s = b"A" * 31 + b"N" + b"CC"
    w = O.encode(s)
    assert O.decode(w, len(s), 0, len(s), False) == b"A" * 31 + b"T" + b"CC"


def test_round_trip_random():
    rng = np.random.default_rng(1)
    for n in (1, 31, 32, 33, 64, 1000):
        s = bytes(rng.choice(list(b"ACGT"), n).astype(np.uint8))
        w = O.encode(s)
        assert len(w) == (n + 31) // 32
        assert O.decode(w, n, 0, n, False) == s
        rc = bytes({65: 84, 67: 71, 71: 67, 84: 65}[c] for c in reversed(s))
        assert O.decode(w, n, 0, n, True) == rc


def test_decode_out_of_bounds_panics():
    w = O.encode(b"ACGT")
    with pytest.raises(O.OracleError):
        O.decode(w, 4, 0, 5, False)
