"""The third-party pin of the pileup oracle (tests/golden/features_w256, written by tools/make_golden_features.py): input files a
reference `herro` binary can read and the oracle's `herro features` output for them.  Here: the committed files are what the
oracle produces today (not-gpu), the product's readers turn the input FILES back into the generator's arrays (not-gpu), and the HIP
path run from those files writes the same bytes (gpu)."""
import hashlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLD = os.path.join(ROOT, "tests", "golden", "features_w256")


def _manifest():
    return dict((ln.split("  ", 1)[1].rstrip("\n"), ln.split("  ", 1)[0]) for ln in open(os.path.join(GOLD, "MANIFEST.sha256")))


def test_golden_files_match_manifest_and_oracle(tmp_path):
    import make_golden_features as mg
    man = _manifest()
    assert len(man) == 2 + 3 * 15
    for rel, h in man.items():
        assert hashlib.sha256(open(os.path.join(GOLD, rel), "rb").read()).hexdigest() == h, rel
    assert mg.write(str(tmp_path / "again")) == man     # the oracle (and the generator) still produce exactly these bytes
    f = np.load(os.path.join(GOLD, "features", "read0", "0.features.npy"))
    assert f.dtype == np.uint8 and f.ndim == 3 and f.shape[0] == 2 and f.shape[2] == 31


def test_golden_inputs_read_back_by_the_product_readers():
    """reads.fastq / overlaps.paf through herro_fastx_read / herro_paf_parse == the arrays they were written from."""
    import make_golden_features as mg
    from herro_amd import api, io as hio
    sb = mg.case()
    r = hio.read_fastx(os.path.join(GOLD, "reads.fastq"))
    assert r.ids == [sb.read_name(i).encode() for i in range(sb.n_reads)]
    assert bytes(r.seq) == bytes(sb.seq) and bytes(r.qual) == bytes(sb.qual) and r.off.tolist() == sb.off.tolist()
    paf = api.Paf(r.ids, text=open(os.path.join(GOLD, "overlaps.paf"), "rb").read())
    assert paf.targets.tolist() == sb.tgt_rid.tolist() and paf.aln_off.tolist() == sb.tgt_aln_off.tolist()
    rows = paf.rows()
    assert len(rows) == len(sb.aln)
    for a in (0, len(rows) // 2, len(rows) - 1):
        assert list(rows[a][:9]) == [int(x) for x in sb.aln[a][:9]] and rows[a][9] == sb.cigar(a)
    paf.close()


@pytest.mark.gpu
def test_hip_path_from_the_golden_inputs_writes_the_golden_files(tmp_path):
    from herro_amd import api, io as hio
    import gpu_common as G
    r = hio.read_fastx(os.path.join(GOLD, "reads.fastq"))
    c = G.ctx()
    c.set_reads(np.frombuffer(bytes(r.seq), np.uint8), np.frombuffer(bytes(r.qual), np.uint8), np.asarray(r.off, np.uint64))
    paf = api.Paf(r.ids, text=open(os.path.join(GOLD, "overlaps.paf"), "rb").read())
    job = c.create_job_from_paf(paf, 256)
    assert job.skipped() == (0, 0)
    job.featurize()
    n = hio.write_job_features(job, str(tmp_path / "out"), [i.decode() for i in r.ids])
    man = _manifest()
    assert n == sum(1 for k in man if k.endswith(".features.npy"))
    for rel, h in man.items():
        if rel.startswith("features/"):
            assert hashlib.sha256(open(os.path.join(str(tmp_path / "out"), rel[len("features/"):]), "rb").read()).hexdigest() == h, rel
    job.close(); paf.close()
