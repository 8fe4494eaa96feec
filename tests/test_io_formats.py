"""not-gpu: the reads reader (get_reads rules, haec_io.rs:37-75, needletail's record rules) and the `herro features` file
layout (features.rs:724-764), both behind the C ABI (csrc/fastx.cpp), against independent checkers: a plain Python parser
and numpy's own .npy writer."""
import gzip
import os

import numpy as np
import pytest

import oracle_lib as O
from herro_amd import io as hio, synth


def _py_fastq(text: bytes, min_length=0, keep=None):
    """Checker: 4-line FASTQ only, get_reads' rules restated in Python."""
    out = []
    lines = text.split(b"\n")
    i = 0
    while i + 3 < len(lines) + 1 and i < len(lines) and lines[i]:
        h, s, q = lines[i].rstrip(b"\r")[1:], lines[i + 1].rstrip(b"\r"), lines[i + 3].rstrip(b"\r")
        i += 4
        if len(s) < min_length:
            continue
        cut = min((k for k in (h.find(b" "), h.find(b"\t")) if k >= 0), default=-1)
        rid, desc = (h, None) if cut < 0 else (h[:cut], h[cut + 1:])
        if keep is not None and rid.decode() not in keep:
            continue
        out.append((rid, desc, s, q))
    return out


@pytest.mark.parametrize("chunk", [None, "1", "3", "7", "64", "4096"])
def test_read_fastx_rules(tmp_path, monkeypatch, chunk):
    # the reader streams the file in chunks (32 MiB) and re-parses a record that straddles a chunk end with more text behind
    # it; tiny chunks put a chunk end inside every header, sequence, '+' line, quality line, CRLF pair and gzip block
    if chunk is None:
        monkeypatch.delenv("HERRO_FASTX_CHUNK", raising=False)
    else:
        monkeypatch.setenv("HERRO_FASTX_CHUNK", chunk)
    recs = [(b"r0 some description\twith tab", b"ACGTACGT", b"IIIIIIII"), (b"r1", b"AC", b"II"),
            (b"r2\tdesc", b"ACGTN", b"!!!!!"), (b"r3", b"ACGTAAAA", b"56789:;<")]
    txt = b"".join(b"@" + h + b"\n" + s + b"\n+\n" + q + b"\n" for h, s, q in recs)
    p = tmp_path / "x.fastq"
    p.write_bytes(txt)
    pg = tmp_path / "x.fastq.gz"
    with gzip.open(pg, "wb") as f:
        f.write(txt)
    pc = tmp_path / "crlf.fastq"
    pc.write_bytes(txt.replace(b"\n", b"\r\n"))
    want = _py_fastq(txt, 3)
    for path in (str(p), str(pg), str(pc)):
        r = hio.read_fastx(path, min_length=3)
        assert r.ids == [w[0] for w in want] == [b"r0", b"r2", b"r3"]
        assert r.descriptions == [w[1] for w in want] == [b"some description\twith tab", b"desc", None]
        assert r.off.tolist() == [0, 8, 13, 21]
        assert bytes(r.seq) == b"".join(w[2] for w in want) and bytes(r.qual) == b"".join(w[3] for w in want)
    r = hio.read_fastx(str(p), core={"r0"}, neighbour={"r3"})
    assert r.ids == [b"r0", b"r3"]
    r = hio.read_fastx(str(p), core={"r0"})          # filter needs both sets (haec_io.rs:63)
    assert len(r.ids) == 4
    # multi-line FASTQ (needletail accepts it): sequence and qualities wrapped at 3
    ml = tmp_path / "ml.fastq"
    ml.write_bytes(b"@m0 d\nACG\nTAC\nGT\n+m0\nIII\nIII\nII\n@m1\nAC\n+\nII\n")
    r = hio.read_fastx(str(ml))
    assert r.ids == [b"m0", b"m1"] and bytes(r.seq) == b"ACGTACGTAC" and bytes(r.qual) == b"IIIIIIIIII" and r.off.tolist() == [0, 8, 10]
    # sequence on one line, qualities wrapped (and the reverse); an empty record; no newline at the end of the file
    mx = tmp_path / "mixed.fastq"
    mx.write_bytes(b"@a\nACGTAC\n+\nIII\nJJJ\n@b\nAC\nGT\n+\nKKKK\n@e\n\n+\n\n@z\nTT\n+\nLL")
    r = hio.read_fastx(str(mx))
    assert r.ids == [b"a", b"b", b"e", b"z"] and bytes(r.seq) == b"ACGTACACGTTT" and bytes(r.qual) == b"IIIJJJKKKKLL"
    assert r.off.tolist() == [0, 6, 10, 10, 12]
    # what the reference panics on
    fa = tmp_path / "x.fasta"
    fa.write_bytes(b">r0\nACGT\nACGT\n")
    with pytest.raises(ValueError, match="Qualities should be present"):
        hio.read_fastx(str(fa))
    assert hio.read_fastx(str(fa), min_length=100).ids == []          # ... unless the length filter drops the record first
    bad = tmp_path / "bad.fastq"
    bad.write_bytes(b"@r0\nACGT\n+\nII\n")
    with pytest.raises(ValueError, match="Error parsing fastx file"):
        hio.read_fastx(str(bad))
    with pytest.raises(ValueError, match="Cannot open"):
        hio.read_fastx(str(tmp_path / "missing.fastq"))
    # a gzip stream cut short exactly at a record boundary is an error, not a shorter read set (gzread returns 0 there as at a clean end)
    import zlib
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    head = co.compress(b"".join(b"@" + h + b"\n" + sq + b"\n+\n" + q + b"\n" for h, sq, q in recs[:2])) + co.flush(zlib.Z_FULL_FLUSH)
    cut = tmp_path / "cut.fastq.gz"
    cut.write_bytes(head)                       # two whole records, no end of stream, no trailer
    with pytest.raises(ValueError, match="truncated or damaged gzip"):
        hio.read_fastx(str(cut))
    whole = tmp_path / "whole.fastq.gz"
    whole.write_bytes(head + co.compress(b"@r9\nACGT\n+\nIIII\n") + co.flush())
    assert hio.read_fastx(str(whole)).ids == [b"r0", b"r1", b"r9"]


def _same_reads(a, b):
    assert a.ids == b.ids and a.descriptions == b.descriptions
    assert np.array_equal(a.off, b.off) and np.array_equal(a.seq, b.seq) and np.array_equal(a.qual, b.qual)


@pytest.mark.parametrize("case", ["four_line", "at_qualities", "blank_lines_crlf", "multi_line", "one_huge_read", "no_final_newline", "tiny_records", "fasta_inside"])
def test_parallel_byte_ranges_equal_one_sequential_pass(tmp_path, monkeypatch, capfd, case):
    """A plain file of some size is read by several threads over byte ranges (two passes: boundaries + counts, then every record to its
    final place).  Whatever the ranges cut through — quality lines that start with '@', blank lines, a record longer than a range,
    multi-line records (which the boundary guess cannot handle: the reader notices and reads sequentially) — the result is the
    sequential reader's, errors included."""
    rng = np.random.default_rng(hash(case) % 1000)
    recs = []
    n = {"one_huge_read": 12, "tiny_records": 3000}.get(case, 400)
    for i in range(n):
        ln = int(rng.integers(1, 40)) if case == "tiny_records" else int(rng.integers(30, 900))
        if case == "one_huge_read" and i == 5:
            ln = 60000
        s_ = bytes(rng.choice(list(b"ACGT"), ln).tolist())
        q = rng.integers(33, 90, ln).astype(np.uint8)
        if case == "at_qualities":
            q[0] = ord("@")                                   # every quality line looks like a header
            if i % 3 == 0 and ln > 1:
                s_ = b"A" + s_[1:]
        head = b"r%d" % i + (b" desc %d\tx" % i if i % 4 == 0 else b"")
        recs.append((head, s_, q.tobytes()))
    nl = b"\r\n" if case == "blank_lines_crlf" else b"\n"
    out = []
    for i, (h, s_, q) in enumerate(recs):
        if case == "multi_line" and i % 5 == 0 and len(s_) > 10:
            k = len(s_) // 3
            out.append(b"@" + h + nl + s_[:k] + nl + s_[k:] + nl + b"+" + nl + q[:k] + nl + q[k:] + nl)
        else:
            out.append(b"@" + h + nl + s_ + nl + b"+" + (h if i % 7 == 0 else b"") + nl + q + nl)
        if case == "blank_lines_crlf" and i % 6 == 0:
            out.append(nl + nl)
    text = b"".join(out)
    if case == "no_final_newline":
        text = text.rstrip(b"\n")
    if case == "fasta_inside":
        text = text[: len(text) // 2].rsplit(b"@r", 1)[0] + b">fa\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n" + b"@r" + text[len(text) // 2:].split(b"@r", 1)[1]
    path = str(tmp_path / f"{case}.fastq")
    open(path, "wb").write(text)
    monkeypatch.setenv("HERRO_FASTX_THREADS", "1")

    def read(**kw):
        try:
            return hio.read_fastx(path, **kw)
        except ValueError as e:
            return str(e)
    want = {k: read(**kw) for k, kw in (("all", {}), ("min", dict(min_length=200)), ("keep", dict(core={"r3", "r11", "r%d" % (n - 1)}, neighbour={"r4"})))}
    if case == "fasta_inside":
        assert want["all"] == "Qualities should be present."
    else:
        assert len(want["all"].ids) == n
    monkeypatch.setenv("HERRO_FASTX_TRACE", "1")
    capfd.readouterr()
    n_par = n_seq = 0
    for threads, range_min in (("2", "64"), ("3", "1000"), ("7", "1"), ("8", "5000"), ("16", "300")):
        monkeypatch.setenv("HERRO_FASTX_THREADS", threads)
        monkeypatch.setenv("HERRO_FASTX_RANGE_MIN", range_min)
        for k, kw in (("all", {}), ("min", dict(min_length=200)), ("keep", dict(core={"r3", "r11", "r%d" % (n - 1)}, neighbour={"r4"}))):
            got = read(**kw)
            if isinstance(want[k], str):
                assert got == want[k], (threads, range_min, k)
            else:
                _same_reads(got, want[k])
            trace = capfd.readouterr().err
            n_par += "read in parallel" in trace
            n_seq += "sequential pass" in trace
    assert n_par + n_seq == 15
    if case in ("four_line", "at_qualities", "blank_lines_crlf", "one_huge_read", "no_final_newline", "tiny_records"):
        assert n_par == 15, (n_par, n_seq)          # plain four-line records: the ranges always verify
    if case == "fasta_inside":
        assert n_seq == 15                          # an error anywhere belongs to the sequential reader


def test_features_files_equal_numpy_written(tmp_path):
    sb = synth.generate(2, 700, 8, seed=3, flank_min=30, flank_max=50)
    store = O.store_from_synth(sb)
    rid, rows, cigs = O.target_alignments(sb, 0)
    res = store.extract_features(rid, rows, cigs, 256)
    d = str(tmp_path / sb.read_name(rid))
    for w in range(len(res)):
        ow = res.window(w)
        hio.write_window_features(d, w, [sb.read_name(int(q)) for q in ow.qids], ow.bases, ow.quals, ow.sup_pos, ow.sup_ins)
    for w in range(len(res)):
        ow = res.window(w)
        # independent writer: numpy
        ref = tmp_path / "np"
        os.makedirs(ref, exist_ok=True)
        np.save(ref / f"{w}.features.npy", np.ascontiguousarray(np.stack([ow.bases, ow.quals], axis=0)))
        sup = np.zeros(len(ow.sup_pos), hio.SUPPORTED_DTYPE)
        sup["pos"], sup["ins"] = ow.sup_pos, ow.sup_ins
        np.save(ref / f"{w}.supported.npy", sup)
        for name in (f"{w}.features.npy", f"{w}.supported.npy"):
            assert open(os.path.join(d, name), "rb").read() == open(ref / name, "rb").read(), name
        f = np.load(os.path.join(d, f"{w}.features.npy"))
        assert f.dtype == np.uint8 and f.shape == (2, ow.bases.shape[0], 31) and not np.isfortran(f)
        assert np.array_equal(f[0], ow.bases) and np.array_equal(f[1], ow.quals)
        s = np.load(os.path.join(d, f"{w}.supported.npy"))
        assert s.dtype == hio.SUPPORTED_DTYPE and s["pos"].tolist() == list(ow.sup_pos) and s["ins"].tolist() == list(ow.sup_ins)
        assert open(os.path.join(d, f"{w}.ids.txt")).read().split("\n")[:-1] == [sb.read_name(int(q)) for q in ow.qids]
