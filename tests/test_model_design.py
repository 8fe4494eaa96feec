"""CPU check that the product's model formulation (folded BN/embedding, receptive-field
evaluation at informative rows only, ragged windows with reconstructed batch padding) is
numerically the dense PyTorch twin — i.e. what a TorchScript executor would compute."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from herro_amd import model_io as mio
import model_numpy as MN


def _rand_batch(rng, B, L, win_len):
    bases = rng.integers(0, 11, (B, L, 31)).astype(np.uint8)
    quals = rng.integers(33, 90, (B, L, 31)).astype(np.uint8)
    for b in range(B):
        bases[b, win_len[b]:] = 11   # collate padding (inference.rs:86-97)
        quals[b, win_len[b]:] = 126
    return bases, quals


@pytest.mark.parametrize("kw", [3, 5])
def test_receptive_field_equals_dense_twin(kw):
    import model_ref as MR
    hp = mio.Hyper(kw=kw, c1=32, c2=32, d_model=64, n_heads=2, d_ff=128, n_layers=2)
    raw = mio.random_raw_params(hp, seed=7)
    F = mio.fold(raw, hp)
    rng = np.random.default_rng(3)
    B, L = 3, 40
    win_len = np.array([40, 33, 37])
    bases, quals = _rand_batch(rng, B, L, win_len)
    # informative rows incl. both edges of every window and rows next to the padding
    idx = [np.array([0, 1, 5, 20, 38, 39]), np.array([0, 2, 31, 32]), np.array([36, 35, 3])]
    idx = [np.sort(i) for i in idx]
    lens = np.array([len(i) for i in idx], np.int32)
    flat = np.concatenate(idx).astype(np.int32)
    twin = MR.build(raw, hp)
    ti, tb = MR.run_batch(twin, bases, quals, lens, flat)
    ni, nb = MN.forward(F, hp, bases, quals, lens, flat, win_len=win_len)
    assert np.abs(ti - ni).max() < 2e-5 and np.abs(tb - nb).max() < 2e-5


def test_gemm_formulation_of_the_twin_equals_the_module():
    """forward_gemm (what the GPU end-to-end test runs on the device) == the nn.Module forward, incl. collate padding,
    windows without informative rows and rows next to both window edges."""
    import model_ref as MR
    hp = mio.Hyper()
    raw = mio.random_raw_params(hp, seed=9)
    rng = np.random.default_rng(6)
    B, L = 5, 48
    win_len = np.array([48, 40, 48, 31, 45])
    bases, quals = _rand_batch(rng, B, L, win_len)
    idx = [np.array([0, 1, 2, 30, 46, 47]), np.array([39, 38, 5]), np.zeros(0, np.int64), np.arange(31), np.array([44])]
    idx = [np.sort(i) for i in idx]
    lens = np.array([len(i) for i in idx], np.int32)
    flat = np.concatenate(idx).astype(np.int32)
    twin = MR.build(raw, hp)
    ti, tb = MR.run_batch(twin, bases, quals, lens, flat)
    gi, gb = MR.run_batch(twin, bases, quals, lens, flat, gemm=True)
    assert ti.shape == gi.shape and tb.shape == gb.shape
    assert np.abs(ti - gi).max() < 2e-5 and np.abs(tb - gb).max() < 2e-5


def test_window_without_informative_rows_is_skipped():
    import model_ref as MR
    hp = mio.Hyper(c1=32, c2=32, d_model=64, n_heads=2, d_ff=128, n_layers=1)
    raw = mio.random_raw_params(hp, seed=8)
    F = mio.fold(raw, hp)
    rng = np.random.default_rng(4)
    win_len = np.array([20, 20])
    bases, quals = _rand_batch(rng, 2, 20, win_len)
    lens = np.array([0, 3], np.int32)
    flat = np.array([1, 7, 19], np.int32)
    twin = MR.build(raw, hp)
    ti, tb = MR.run_batch(twin, bases, quals, lens, flat)
    ni, nb = MN.forward(F, hp, bases, quals, lens, flat, win_len=win_len)
    assert ti.shape == (3,) and np.abs(tb - nb).max() < 2e-5 and np.abs(ti - ni).max() < 2e-5


def test_flat_file_round_trip(tmp_path):
    import struct
    hp = mio.Hyper()
    raw = mio.random_raw_params(hp, seed=1)
    p = str(tmp_path / "m.hrro")
    mio.export(raw, hp, p)
    blob = open(p, "rb").read()
    hdr = struct.unpack("<11If", blob[:48])
    assert hdr[0] == mio.MAGIC and hdr[2] == 31 and hdr[3] == hp.kw and hdr[10] == len(mio.fold(raw, hp))
