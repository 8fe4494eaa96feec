"""Shared helpers for the -m gpu parity tests (HIP path through the C-ABI vs the oracle)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import oracle_lib as O
from herro_amd import api, model_io, synth

CACHE = os.path.join(ROOT, "tests", "_cache")
_CTX = {}



def sweep_trials(n):
    """Trial numbers of a seeded sweep: 0 .. n - 1 in the suite; HERRO_SWEEP_FIRST / HERRO_SWEEP_TRIALS move and widen the range for a soak run on the GPU box
    (tools/r6_sweep_soak.sh; every trial is its own seed, so a failing one is reproduced by its number)."""
    first = int(os.environ.get("HERRO_SWEEP_FIRST", "0") or 0)
    return range(first, first + int(os.environ.get("HERRO_SWEEP_TRIALS", str(n)) or n))

def ctx():
    if "c" not in _CTX:
        c = api.Context(0)
        path, raw = model_io.default_model_file(CACHE)
        c.load_model(path)
        _CTX["c"], _CTX["raw"] = c, raw
    _CTX["c"].set_precision(api.DEFAULT_PRECISION)   # every test starts from the shipped default, whatever the previous one selected
    return _CTX["c"]


LOOSE = 4e-3   # bound of an f16 mode that this model's calibration REFUSES and the test hook selects anyway (measured, not a product mode)


def select_precision(c, mode):
    """herro_set_precision(mode) -> True; a mode the model's load-time calibration refuses (> 5e-4 on the calibration batch) is selected
    through the test hook and reported as not selectable -> False: whatever a caller CAN select is held to the 1e-3 contract."""
    try:
        c.set_precision(mode)
        return True
    except api.HerroError as e:
        if "refused" not in str(e):
            raise
    c.force_precision(True)
    try:
        c.set_precision(mode)
    finally:
        c.force_precision(False)
    return False


def raw_params():
    ctx()
    return _CTX["raw"]


def twin():
    if "twin" not in _CTX:
        import model_ref as MR
        _CTX["twin"] = MR.build(raw_params(), model_io.Hyper())
    return _CTX["twin"]


def load_synth(c, sb):
    c.set_reads(sb.seq, sb.qual, sb.off)


def compare_features(job, sb, store, W, targets=None):
    """Bit-exact comparison of every window of the job with the oracle.  Returns #windows."""
    ts = list(range(sb.n_targets)) if targets is None else list(targets)
    w = 0
    for t in ts:
        rid, rows, cigs = O.target_alignments(sb, t)
        res = store.extract_features(rid, rows, cigs, W)
        for wi in range(len(res)):
            ow = res.window(wi)
            gw = job.window(w)
            tag = f"target {t} window {wi}"
            assert (gw.info.rid, gw.info.wid, gw.info.n_total_wins) == (rid, wi, len(res)), tag
            assert gw.info.n_overlaps == len(ow.qids), tag
            assert gw.qids.tolist() == ow.qids.tolist(), tag
            assert gw.info.n_alns == ow.n_alns, tag
            assert gw.info.length == ow.bases.shape[0], (tag, gw.info.length, ow.bases.shape)
            assert np.array_equal(gw.bases, ow.bases), tag
            assert np.array_equal(gw.quals, ow.quals), tag
            assert gw.sup_pos.tolist() == ow.sup_pos.tolist() and gw.sup_ins.tolist() == ow.sup_ins.tolist(), tag
            w += 1
    assert w == job.n_windows
    return w
