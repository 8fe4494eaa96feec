"""not-gpu: the FASTQ reader (get_reads rules, haec_io.rs:37-75) and the `herro features` file layout
(features.rs:724-764) written from the oracle's windows."""
import gzip
import os

import numpy as np

import oracle_lib as O
from herro_amd import io as hio, synth


def test_read_fastq_rules(tmp_path):
    recs = [(b"r0 some description\twith tab", b"ACGTACGT", b"IIIIIIII"), (b"r1", b"AC", b"II"),
            (b"r2\tdesc", b"ACGTN", b"!!!!!"), (b"r3", b"ACGTAAAA", b"56789:;<")]
    txt = b"".join(b"@" + h + b"\n" + s + b"\n+\n" + q + b"\n" for h, s, q in recs)
    p = tmp_path / "x.fastq"
    p.write_bytes(txt)
    pg = tmp_path / "x.fastq.gz"
    with gzip.open(pg, "wb") as f:
        f.write(txt)
    for path in (str(p), str(pg)):
        r = hio.read_fastq(path, min_length=3)
        assert r.ids == [b"r0", b"r2", b"r3"]
        assert r.descriptions == [b"some description\twith tab", b"desc", None]
        assert r.off.tolist() == [0, 8, 13, 21]
        assert bytes(r.seq[8:13]) == b"ACGTN" and bytes(r.qual[13:21]) == b"56789:;<"
    r = hio.read_fastq(str(p), core={"r0"}, neighbour={"r3"})
    assert r.ids == [b"r0", b"r3"]
    r = hio.read_fastq(str(p), core={"r0"})          # filter needs both sets (haec_io.rs:63)
    assert len(r.ids) == 4


def test_features_layout_from_oracle_windows(tmp_path):
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
        f = np.load(os.path.join(d, f"{w}.features.npy"))
        assert f.dtype == np.uint8 and f.shape == (2, ow.bases.shape[0], 31) and not np.isfortran(f)
        assert np.array_equal(f[0], ow.bases) and np.array_equal(f[1], ow.quals)
        s = np.load(os.path.join(d, f"{w}.supported.npy"))
        assert s.dtype == hio.SUPPORTED_DTYPE and s["pos"].tolist() == list(ow.sup_pos) and s["ins"].tolist() == list(ow.sup_ins)
        assert open(os.path.join(d, f"{w}.ids.txt")).read().split("\n")[:-1] == [sb.read_name(int(q)) for q in ow.qids]
        hdr = open(os.path.join(d, f"{w}.features.npy"), "rb").read(128)
        assert hdr[:6] == b"\x93NUMPY" and b"'descr': '|u1'" in hdr and b"'fortran_order': False" in hdr
