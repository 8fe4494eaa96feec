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


@pytest.mark.parametrize("kw,needle", [
    (dict(activation="gelu"), "gelu"),
    (dict(norm_first=False), "post-LN"),
    (dict(extra_param=True), "no rule claims"),
])
def test_foreign_architectures_are_refused(tmp_path, kw, needle):
    hp = mio.Hyper(c1=32, c2=32, d_model=64, n_heads=2, d_ff=128, n_layers=1)
    pt = str(tmp_path / "m.pt")
    ST.save_archive(pt, mio.random_raw_params(hp, seed=22), hp, **kw)
    with pytest.raises(EW.Unsupported) as e:
        EW.convert(pt, str(tmp_path / "m.hrro"), quiet=True)
    assert needle.lower() in str(e.value).lower()
    assert not os.path.exists(tmp_path / "m.hrro")
