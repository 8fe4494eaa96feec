"""not-gpu: the host encoder of the precision-6 weight copies (f32 -> OCP e4m3, round to nearest even, saturating) against a brute-force
nearest-code search over the 127 non-negative finite codes, and against the bytes v_cvt_pk_fp8_f32 produced on an MI355X for the same inputs
(tools/probes/mx_probe.hip — the device converts the activations, the host the weights: both must be the format the MX MFMA reads)."""
import math

import numpy as np

from herro_amd import api


def decode(code: int) -> float:
    e, m = (code >> 3) & 15, code & 7
    v = m * 2.0 ** -9 if e == 0 else (1 + m / 8) * 2.0 ** (e - 7)
    return -v if code & 0x80 else v


VALS = [decode(c) for c in range(0x7f)]          # 0 .. 448, strictly increasing (0x7f is NaN)


def nearest(a: float) -> int:
    if a >= 448:
        return 0x7e
    hi = next(i for i, v in enumerate(VALS) if v >= a)
    if VALS[hi] == a or hi == 0:
        return hi
    lo = hi - 1
    dl, dh = a - VALS[lo], VALS[hi] - a
    if dl != dh:
        return lo if dl < dh else hi
    return lo if lo % 2 == 0 else hi               # tie: the even code (codes are consecutive integers, so code parity = mantissa parity)


def test_e4m3_encoder():
    L = api.lib()
    enc = lambda x: int(L.herro_debug_e4m3(float(x)))
    assert VALS[0x38] == 1.0 and VALS[0x7e] == 448.0 and VALS[1] == 2.0 ** -9
    for c in range(0x7f):                          # every finite code is a fixed point, both signs
        assert enc(VALS[c]) == c and enc(-VALS[c]) == (c | 0x80 if c else 0x80), c
    for c in range(0x7e):                          # midpoints go to the even neighbour; just off the midpoint to the nearer one
        mid = (VALS[c] + VALS[c + 1]) / 2
        assert enc(np.float32(mid)) == (c if c % 2 == 0 else c + 1), (c, mid)
        for x in (np.nextafter(np.float32(mid), np.float32(0)), np.nextafter(np.float32(mid), np.float32(1e9))):
            assert enc(x) == nearest(float(x)), (c, float(x))
    rng = np.random.default_rng(8)
    xs = np.concatenate([rng.uniform(-460, 460, 4000), rng.normal(0, 1, 4000) * 2.0 ** rng.integers(-14, 8, 4000)]).astype(np.float32)
    for x in xs:
        want = nearest(abs(float(x))) | (0x80 if x < 0 else 0)
        assert enc(x) == want, (float(x), enc(x), want)
    assert enc(500.0) == 0x7e and enc(-1e30) == 0xfe and enc(math.inf) == 0x7e and enc(math.nan) & 0x7f == 0x7f
    # what the device's v_cvt_pk_fp8_f32 returned (tools/probes/mx_probe.hip on gfx950) for values below its NaN threshold
    for x, code in ((0.0, 0x00), (1.0, 0x38), (0.5, 0x30), (1.75, 0x3e), (448.0, 0x7e), (449.0, 0x7e), (2.0 ** -9, 0x01), (2.0 ** -10, 0x00),
                    (0.0176, 0x09), (3.3, 0x45), (0.1, 0x1d), (240.0, 0x77)):
        assert enc(x) == code, (x, enc(x), code)
