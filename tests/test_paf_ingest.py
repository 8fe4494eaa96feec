"""not-gpu: herro_paf_parse / herro_oec_read (host C++, herro_amd/csrc/ingest.cpp) against the restatement of
overlaps.rs:117-202 / 292-323 in oracle/paf_ref.py: skips, duplicates, self overlaps, `core`, repeated read names,
the dropped last byte, wrapping numbers, every panic; a zstd .oec file written with pyarrow's codec."""
import os
import sys

import numpy as np
import pytest

from herro_amd import api

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import paf_ref as R  # noqa: E402


def _line(q, ql, qs, qe, st, t, tl, ts, te, cig, extra=(b"60", b"100", b"255")):
    return b"\t".join([q, str(ql).encode(), str(qs).encode(), str(qe).encode(), st, t, str(tl).encode(),
                       str(ts).encode(), str(te).encode(), *extra, b"cg:Z:" + cig])


def _check(text, names, core_names=None, threads=0):
    core = None if core_names is None else np.array([1 if n in core_names else 0 for n in names], np.uint8)
    want_order, want = R.parse_paf(text, names, None if core_names is None else set(core_names))
    ix = api.NameIndex(names)           # the same parse through an index built once, and without copying the text
    for who, view in ((names, False), (ix, False), (ix, True)):
        p = api.Paf(who, text=text, core=core, threads=threads, view=view)
        assert p.targets.tolist() == want_order
        rows = p.rows()
        k = 0
        for ti, t in enumerate(want_order):
            assert int(p.aln_off[ti + 1] - p.aln_off[ti]) == len(want[t])
            for w in want[t]:
                assert rows[k] == w
                k += 1
        assert k == p.n_alns
        p.close()
    ix.close()


NAMES = [b"r0", b"r1", b"r2", b"r3", b"dup", b"r5", b"dup"]  # "dup" -> index 6 (last one wins)


def test_rules_and_quirks():
    L = [
        _line(b"r1", 100, 0, 90, b"+", b"r0", 120, 5, 95, b"90M"),
        _line(b"r2", 100, 0, 90, b"-", b"r0", 120, 5, 95, b"40M2I48M"),
        _line(b"r1", 100, 1, 91, b"+", b"r0", 120, 6, 96, b"90M"),          # same pair again: dropped
        _line(b"zz", 100, 0, 90, b"+", b"r0", 120, 5, 95, b"90M"),          # unknown query
        _line(b"r1", 100, 0, 90, b"+", b"zz", 120, 5, 95, b"90M"),          # unknown target
        _line(b"r3", 100, 0, 90, b"+", b"r3", 100, 0, 90, b"90M"),          # self overlap
        _line(b"r0", 120, 5, 95, b"+", b"r1", 100, 0, 90, b"90M"),          # new target, (r0, r1) is a different pair
        _line(b"dup", 50, 0, 50, b"-", b"r1", 100, 10, 60, b"50M"),         # repeated name: last index
        _line(b"r5", 4294967295 + 7, 0, 5, b"+", b"r2", 99999999999, 1, 6, b"5M"),  # u32 wrap
        b"",                                                                 # empty line: skipped
        _line(b"r3", 10, 0, 9, b"+", b"r2", 100, 0, 9, b"9M", extra=()),    # no optional columns: cg is field 10
    ]
    text = b"\n".join(L) + b"\n"
    _check(text, NAMES)
    _check(text, NAMES, threads=3)
    # final line without newline loses its last character (here: the CIGAR's 'M')
    _check(b"\n".join(L), NAMES)
    # core: only listed targets are kept
    _check(text, NAMES, core_names={b"r0"})
    _check(text, NAMES, core_names=set())
    _check(b"", NAMES)


def test_many_lines_parallel_matches_serial():
    rng = np.random.default_rng(4)
    names = [f"read{i}".encode() for i in range(300)]
    L = []
    for _ in range(20000):
        q, t = rng.integers(0, 320, 2)
        qn = names[q] if q < 300 else b"unknown%d" % q
        tn = names[t] if t < 300 else b"unknown%d" % t
        n = int(rng.integers(1, 5000))
        L.append(_line(qn, n + 10, 0, n, b"+" if rng.integers(0, 2) else b"-", tn, n + 20, 3, n + 3, b"%dM" % n))
    text = b"\n".join(L) + b"\n"
    _check(text, names, threads=1)
    _check(text, names, threads=8)


def test_text_of_several_megabytes_crosses_the_byte_ranges_of_the_threads():
    """The copy and the search for line ends run over byte ranges of >= 1 MiB per thread: long lines (CIGARs of kilobytes) that
    straddle the range boundaries, with and without a final newline, against the restatement."""
    rng = np.random.default_rng(8)
    names = [f"read{i}".encode() for i in range(64)]
    L = []
    for _ in range(2500):
        q, t = rng.integers(0, 70, 2)
        qn = names[q] if q < 64 else b"unknown%d" % q
        tn = names[t] if t < 64 else b"unknown%d" % t
        cig = b"".join(b"%d%s" % (int(rng.integers(1, 60)), [b"M", b"I", b"M", b"D"][k % 4]) for k in range(int(rng.integers(1, 1500)))) + b"7M"
        L.append(_line(qn, 50000, 0, 40000, b"+" if rng.integers(0, 2) else b"-", tn, 50000, 3, 40003, cig))
    text = b"\n".join(L)
    assert len(text) > 5 << 20
    for tail in (b"\n", b""):
        for th in (1, 3, 8):
            _check(text + tail, names, threads=th)


@pytest.mark.parametrize("bad,msg", [
    (_line(b"r1", "1x0", 0, 90, b"+", b"r0", 120, 5, 95, b"90M"), "valid digit"),
    (_line(b"r1", 100, 0, 90, b"*", b"r0", 120, 5, 95, b"90M"), "Invalid strand"),
    (_line(b"r1", 100, 0, 90, b"", b"r0", 120, 5, 95, b"90M"), "index out of bounds"),
    (_line(b"r1", 100, 0, 90, b"+", b"r0", 120, 5, "9 5", b"90M"), "valid digit"),
    (b"r1\t100\t0\t90\t+\tr0\t120\t5\t95\n", "unwrap"),                       # nothing after tend
    (b"r1\t100\t0\t90\t+\tr0\t120\t5\t95\tcg\n", "out of range"),             # last column shorter than 5 bytes
    (b"r1\t100\t0\n", "unwrap"),
])
def test_panics(bad, msg):
    good = _line(b"r2", 100, 0, 90, b"-", b"r0", 120, 5, 95, b"90M")
    text = good + b"\n" + (bad if bad.endswith(b"\n") else bad + b"\n")
    with pytest.raises(R.ReferencePanic):
        R.parse_paf(text, NAMES)
    with pytest.raises(api.HerroError) as e:
        api.Paf(NAMES, text=text)
    assert e.value.code == -3 and msg in str(e.value) and "line 2" in str(e.value)
    # a panicking line behind a skip condition is never reached: unknown query name first
    if bad.startswith(b"r1"):
        _check(good + b"\n" + b"zz" + bad[2:] + (b"" if bad.endswith(b"\n") else b"\n"), NAMES)


def test_oec_zst_file(tmp_path):
    pa = pytest.importorskip("pyarrow")
    if not pa.Codec.is_available("zstd"):
        pytest.skip("no zstd codec to write the fixture with")
    L = [
        _line(b"r1", 100, 0, 90, b"+", b"r0", 120, 5, 95, b"90M"),
        _line(b"r2", 100, 0, 90, b"-", b"r0", 120, 5, 95, b"40M2I48M"),
        _line(b"r0", 120, 5, 95, b"+", b"r1", 100, 0, 90, b"44M1D45M"),
    ]
    body = b"2\nr0\nr1\n" + b"\n".join(L) + b"\n"      # header: n_targets + ids (scripts/batch.py, overlaps.rs:274-279)
    comp = pa.compress(body, codec="zstd", asbytes=True)
    f = tmp_path / "0.oec.zst"
    f.write_bytes(comp)
    want_order, want = R.read_batch(body, NAMES)
    ix = api.NameIndex(NAMES)
    for who in (NAMES, ix):
        p = api.Paf(who, path=str(f))
        assert p.targets.tolist() == want_order
        assert p.rows() == [w for t in want_order for w in want[t]]
        p.close()
    with pytest.raises(api.HerroError):
        api.Paf(NAMES, path=str(tmp_path / "missing.oec.zst"))


def test_randomised_differential():
    """Random PAF-like text (valid lines, unknown names, duplicates, truncated / corrupted fields, missing final
    newline): the product and the restatement agree on the result, or both report the reference's panic at the same line."""
    rng = np.random.default_rng(12345)
    names = [b"q%d" % i for i in range(12)] + [b"q3"]            # one repeated name
    pool = names + [b"nope", b""]
    for trial in range(300):
        L = []
        for _ in range(int(rng.integers(0, 12))):
            q, t = pool[int(rng.integers(0, len(pool)))], pool[int(rng.integers(0, len(pool)))]
            f = [q, b"%d" % rng.integers(0, 5000), b"%d" % rng.integers(0, 100), b"%d" % rng.integers(100, 5000),
                 [b"+", b"-", b"*", b""][int(rng.choice(4, p=[0.45, 0.45, 0.05, 0.05]))], t,
                 b"%d" % rng.integers(0, 5000), b"%d" % rng.integers(0, 100), b"%d" % rng.integers(100, 5000),
                 b"60", b"cg:Z:%dM" % rng.integers(1, 900)]
            r = rng.random()
            if r < 0.06:
                f[int(rng.integers(1, 9))] = b"12a"                # bad digit somewhere
            elif r < 0.10:
                f = f[:int(rng.integers(1, 10))]                   # truncated line
            elif r < 0.13:
                f[-1] = b"cg"                                      # short last column
            elif r < 0.16:
                f = f[:9] + [f[-1]]                                # no optional columns
            L.append(b"\t".join(f))
        text = b"\n".join(L) + (b"\n" if rng.random() < 0.8 and L else b"")
        core = None if rng.random() < 0.7 else {names[int(i)] for i in rng.integers(0, len(names), 4)}
        flags = None if core is None else np.array([1 if n in core else 0 for n in names], np.uint8)
        try:
            want_order, want = R.parse_paf(text, names, core)
        except R.ReferencePanic:
            # which line: re-run line by line prefixes until the panic appears
            k = 1
            lines = text.split(b"\n")
            while True:
                try:
                    R.parse_paf(b"\n".join(lines[:k]) + b"\n", names, core)
                except R.ReferencePanic:
                    break
                k += 1
                assert k <= len(lines) + 1
            with pytest.raises(api.HerroError) as e:
                api.Paf(names, text=text, core=flags)
            # the prefix test re-adds a newline the last line may not have had: only check the line when it is not the last
            if k < len(lines):
                assert f"line {k})" in str(e.value), (trial, k, str(e.value))
            continue
        p = api.Paf(names, text=text, core=flags, threads=int(rng.integers(1, 4)))
        assert p.targets.tolist() == want_order, trial
        assert p.rows() == [w for t in want_order for w in want[t]], trial
        p.close()


def test_job_from_paf_equals_job_from_arrays_host_half():
    """PAF text -> herro_paf_parse -> herro_job_create: the same descriptors as the array entry (device-free context)."""
    from herro_amd import synth
    sb = synth.generate(4, 1500, 12, seed=9, flank_min=40, flank_max=80, p_partial=0.3)
    lens = (sb.off[1:] - sb.off[:-1]).astype(np.uint32)
    c = api.HostContext(lens)
    names = [sb.read_name(i).encode() for i in range(sb.n_reads)]
    lines = []
    for a in range(len(sb.aln)):
        r = sb.aln[a]
        lines.append(b"\t".join([names[r[0]], b"%d" % r[1], b"%d" % r[2], b"%d" % r[3], b"-" if r[4] else b"+", names[r[5]],
                                 b"%d" % r[6], b"%d" % r[7], b"%d" % r[8], b"60", b"60", b"255", b"cg:Z:" + sb.cigar(a)]))
    ix = api.NameIndex(names)
    for who in (names, ix):
        paf = api.Paf(who, text=b"\n".join(lines) + b"\n")
        assert paf.targets.tolist() == sb.tgt_rid.tolist()
        ja, jp = api.job_from_synth(c, sb, 512), c.create_job_from_paf(paf, 512)
        A, B = c.job_arrays(ja), c.job_arrays(jp)
        for k in A:
            assert np.array_equal(A[k], B[k]), k
        ja.close(); jp.close(); paf.close()
    ix.close()
    c.close()
