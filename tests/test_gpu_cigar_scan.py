"""-m gpu: the CIGAR text scan on the device (csrc/cigar_dev.hip) against the host decoder (windowing.hpp scan_cigar),
through herro_job_create: a job created on a device context (text -> ops + cut records on the GPU, windows cut from the
records) must carry the same descriptors as the job the device-free host context builds from the same alignments, and
its op array must hold the same ops in every overlap's slice.  The host decoder is itself pinned to the oracle's
extract_windows in tests/test_host_job_layout.py."""
import numpy as np
import pytest

import gpu_common as G
from herro_amd import api, synth
from test_host_job_layout import _hand_alignment

pytestmark = pytest.mark.gpu


def _same_jobs(hc, jd, jh):
    ad, ah = hc.job_arrays(jd), hc.job_arrays(jh)
    for k in ("win", "tile_win", "tile_r0", "tgt_win_off"):
        assert np.array_equal(ad[k], ah[k]), k
    assert len(ad["ow"]) == len(ah["ow"])
    for f in api.OW_DTYPE.names:
        if f != "op_begin":     # device jobs index the scan's (gapped) op array, host jobs a compact one
            assert np.array_equal(ad["ow"][f], ah["ow"][f]), f
    od, oh = ad["ops"], ah["ops"]
    for d, h in zip(ad["ow"], ah["ow"]):
        n = int(d["op_cnt"])
        assert np.array_equal(od[int(d["op_begin"]):int(d["op_begin"]) + n], oh[int(h["op_begin"]):int(h["op_begin"]) + n])
    return len(ad["ow"])


@pytest.mark.parametrize("W,tl,nt,kw", [(64, 700, 5, dict(p_partial=0.4, flank_min=10, flank_max=40)),
                                        (256, 1500, 7, dict(p_partial=0.3, flank_min=20, flank_max=60, p_long_indel=0.02)),
                                        (4096, 3 * 4096 + 333, 12, dict(p_partial=0.2)),
                                        (1000, 30000, 3, dict())])
def test_device_scan_gives_the_host_jobs(W, tl, nt, kw):
    sb = synth.generate(nt, tl, 12, seed=3, **kw)
    c = G.ctx()
    G.load_synth(c, sb)
    lens = (sb.off[1:] - sb.off[:-1]).astype(np.uint32)
    hc = api.HostContext(lens)
    jd = api.job_from_synth(c, sb, W)
    jh = api.job_from_synth(hc, sb, W)
    assert _same_jobs(hc, jd, jh) > 0
    assert jd.skipped() == jh.skipped()
    jd.close(); jh.close(); hc.close()


def test_device_scan_text_corner_cases():
    """multi-chunk texts (> 4096 bytes), zero padding, 5..10 digit lengths, digits across chunk borders; malformed text
    fails with the message of the byte-wise reader"""
    rng = np.random.default_rng(5)
    lens = np.array([40000, 40000], np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(off[-1]))]
    qual = rng.integers(33, 80, int(off[-1])).astype(np.uint8)
    c = G.ctx()
    c.set_reads(seq, qual, off)
    hc = api.HostContext(lens.astype(np.uint32))
    rid = np.array([0], np.uint32)
    aoff = np.array([0, 1], np.uint64)
    for W in (64, 1000, 4096):
        for widths in ([1], [1, 3], [1, 5, 7], [1, 8, 9], [4], [7], [10], [1, 12], [17]):
            row, text, canon = _hand_alignment(rng, 40000, widths)
            jd = c.create_job(rid, row, aoff, [text], W)
            jh = hc.create_job(rid, row, aoff, [canon], W)
            assert _same_jobs(hc, jd, jh) > 0, (W, widths)
            jd.close(); jh.close()
    row, text, canon = _hand_alignment(rng, 40000, [1])
    for bad, what in [(canon + b"12", "ends inside an op"), (b"M" + canon, "longer than 0"), (canon.replace(b"M", b"M0I", 1), "longer than 0"),
                      (canon.replace(b"M", b"m", 1), "Unexpected cigar operation"), (canon.replace(b"M", b":", 1), "Unexpected cigar operation"),
                      (canon.replace(b"M", b"\xc8", 1), "Unexpected cigar operation"), (b"99999999999M" + canon, "overflows 30 bits"),
                      (canon[:5000] + b"X" + canon[5000:], ""),
                      # round 6 (the scan keeps a lane's ops in eight registers and decodes in 32-bit arithmetic): nine letters inside sixteen bytes with every one of the
                      # first eight well-formed; lengths at and around the 30-bit limit, with and without leading zeros
                      (canon[:4096] + b"1M1I1M1I1M1I1M1IM" + canon[4096:], "longer than 0"), (canon[:1600] + b"MMMMMMMMMMMMMMMM" + canon[1600:], "longer than 0"),
                      (b"1073741824M" + canon, "overflows 30 bits"), (b"4294967297M" + canon, "overflows 30 bits"), (b"9999999999M" + canon, "overflows 30 bits"),
                      (b"01073741824M" + canon, "")]:
        with pytest.raises(api.HerroError) as e:
            c.create_job(rid, row, aoff, [bad], 1000)
        assert what in str(e.value), (bad[:30], str(e.value))
        assert "internal" not in str(e.value)
    # consecutive insertions: processed like the reference does (the later op overwrites the earlier one's rows); nothing is left out
    pair = canon.replace(b"M", b"M2I3I", 1)
    q_extra = 5
    row2 = row.copy(); row2[0, 3] += q_extra
    jd = c.create_job(rid, row2, aoff, [pair], 1000)
    jh = hc.create_job(rid, row2, aoff, [pair], 1000)
    assert jd.skipped() == jh.skipped() == (0, 0)
    _same_jobs(hc, jd, jh)
    jd.close(); jh.close(); hc.close()
