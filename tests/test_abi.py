"""not-gpu: the C-ABI library loads, exports every symbol include/herro_amd.h declares, the host
utilities work without a device, and device calls fail loudly (no CPU fallback)."""
import json
import os
import re

import numpy as np
import pytest

from herro_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "herro_amd.h")).read()
    declared = set(re.findall(r"\b(herro_[a-z0-9_]+)\s*\(", hdr))
    L = api.lib()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(api.EXPORTS)


def test_codec_matches_reference_vectors():
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "codec_vectors.json")))
    for v in G["encode"]:
        assert api.encode_2bit(v["seq"].encode()).tolist() == v["words"]
    for v in G["decode"]:
        assert api.decode_2bit(np.array(v["words"], np.uint64), v["length"], v["start"], v["end"], v["rc"]) == v["out"].encode()
    for v in G["subseq"]:
        w = api.encode_2bit(v["seq"].encode())
        assert api.decode_2bit(w, len(v["seq"]), v["start"], v["end"], v["rc"]) == v["out"].encode()
    assert api.decode_2bit(api.encode_2bit(b"ACNACGTAC"), 9, 0, 9, False) == b"ACTTTTTAC"  # haec_io.rs:126-128 quirk
    with pytest.raises(api.HerroError):
        api.decode_2bit(api.encode_2bit(b"ACGT"), 4, 0, 5, False)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.HerroError) as e:
        api.Context(0)
    assert e.value.code == -2
