"""tools/export_weights.py on a TorchScript archive (the stand-in for model_R10_v0.1.pt, reference
inference.rs:185-186): the recovered hyper-parameters and the flat file must equal what model_io writes from the
raw parameters, the conversion check against the archive must pass, and archives that are not the architecture the
kernels implement must be refused with the reason."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from herro_amd import model_io as mio
import export_weights as EW
import scripted_twin as ST


@pytest.mark.parametrize("hp", [mio.Hyper(), mio.Hyper(kw=5, c1=32, c2=32, d_model=64, n_heads=2, d_ff=128, n_layers=2)])
def test_archive_to_flat_file(tmp_path, hp):
    raw = mio.random_raw_params(hp, seed=21)
    pt, out, ref = str(tmp_path / "m.pt"), str(tmp_path / "m.hrro"), str(tmp_path / "ref.hrro")
    ST.save_archive(pt, raw, hp)
    got_hp, got_raw, err = EW.convert(pt, out, dump=str(tmp_path / "dump.txt"), do_verify=True, quiet=True)
    assert got_hp == hp
    assert set(got_raw) == set(raw) and all(np.array_equal(got_raw[k], raw[k]) for k in raw)
    mio.export(raw, hp, ref)
    assert open(out, "rb").read() == open(ref, "rb").read()
    assert err is not None and err <= 2e-5
    dump = open(tmp_path / "dump.txt").read()
    assert "forward code" in dump and "self_attn.in_proj_weight" in dump and "aten::" in dump
    assert "GFLOP" in EW.flop_report(hp)


VARIANTS = {   # members of the family an archive may hold: recognised, converted and checked instead of refused
    "gelu": dict(act=1),
    "post_ln": dict(norm_first=0),
    "post_ln_no_final_norm_gelu": dict(norm_first=0, final_norm=0, act=1),
    "learned_position": dict(pe=1, pe_rows=96),
    "no_position_no_batchnorm": dict(pe=2, bn=0),
    "head_dim_64": dict(d_model=128, n_heads=2, d_ff=256),
    "kw1_two_layers": dict(kw=1, c1=32, c2=32, n_layers=2),
    "kw7_eight_layers_pre_ln_no_final_norm": dict(kw=7, c1=32, c2=32, n_layers=8, final_norm=0),
    "d_model_320_gelu_learned": dict(d_model=320, n_heads=10, d_ff=640, act=1, pe=1, pe_rows=64),
}


@pytest.mark.parametrize("name", list(VARIANTS))
def test_variant_archives_convert(tmp_path, name):
    base = dict(c1=32, c2=32, d_model=64, n_heads=2, d_ff=128, n_layers=1)
    hp = mio.Hyper(**{**base, **VARIANTS[name]})
    raw = mio.random_raw_params(hp, seed=23)
    pt, out, ref = str(tmp_path / "m.pt"), str(tmp_path / "m.hrro"), str(tmp_path / "ref.hrro")
    ST.save_archive(pt, raw, hp)
    got_hp, got_raw, err = EW.convert(pt, out, do_verify=True, quiet=True)
    assert got_hp == hp, (got_hp, hp)
    assert set(got_raw) == set(raw) and all(np.array_equal(got_raw[k], raw[k]) for k in raw)
    mio.export(raw, hp, ref)
    assert open(out, "rb").read() == open(ref, "rb").read()
    assert err is not None and err <= 2e-5


@pytest.mark.parametrize("kw,needle", [
    (dict(conv_act="tanh"), "tanh"),               # an operator outside the family
    (dict(extra_param=True), "no rule claims"),    # a parameter nothing accounts for
])
def test_foreign_architectures_are_refused(tmp_path, kw, needle):
    hp = mio.Hyper(c1=32, c2=32, d_model=64, n_heads=2, d_ff=128, n_layers=1)
    pt = str(tmp_path / "m.pt")
    ST.save_archive(pt, mio.random_raw_params(hp, seed=22), hp, **kw)
    with pytest.raises(EW.Unsupported) as e:
        EW.convert(pt, str(tmp_path / "m.hrro"), quiet=True)
    assert needle.lower() in str(e.value).lower()
    assert not os.path.exists(tmp_path / "m.hrro")
