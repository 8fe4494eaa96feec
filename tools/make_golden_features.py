#!/usr/bin/env python
"""Writes the third-party pin of the pileup oracle: a fixed seeded case as INPUT files a real `herro` binary can
read (reads.fastq, overlaps.paf) and the oracle's output for it in the reference's own exchange format — `herro features`
(features.rs:724-764): <read>/<wid>.features.npy u8 [2, L', 31] (ASCII bases; qualities), <wid>.supported.npy ({pos:<u2, ins:u1}),
<wid>.ids.txt — with the SHA-256 of every file in MANIFEST.sha256.

    python tools/make_golden_features.py [out_dir]          (default tests/golden/features_w256; deterministic: byte-identical reruns)

A holder of a reference build pins the oracle with
    herro features -w 256 --read-alns <dir with overlaps.paf> reads.fastq ref_out/      (or: --write-alns / minimap2-free: the PAF carries cg:Z:)
    diff -r ref_out/ tests/golden/features_w256/features/
The oracle is test infrastructure (oracle/herro_oracle.hpp); nothing in the product reads these files except tests/test_golden_features.py,
which checks that the oracle still reproduces them and (on a GPU) that the HIP path writes the same bytes.
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SEED, N_TARGETS, TARGET_LEN, N_OVL, W = 20240917, 3, 1100, 10, 256


def case():
    from herro_amd import synth
    return synth.generate(N_TARGETS, TARGET_LEN, N_OVL, seed=SEED, flank_min=40, flank_max=90, p_partial=0.25)


def paf_text(sb) -> bytes:
    names = [sb.read_name(i).encode() for i in range(sb.n_reads)]
    lines = []
    for a in range(len(sb.aln)):
        r = sb.aln[a]
        lines.append(b"\t".join([names[r[0]], b"%d" % r[1], b"%d" % r[2], b"%d" % r[3], b"-" if r[4] else b"+", names[r[5]],
                                 b"%d" % r[6], b"%d" % r[7], b"%d" % r[8], b"60", b"60", b"255", b"cg:Z:" + sb.cigar(a)]))
    return b"\n".join(lines) + b"\n"


def fastq_text(sb) -> bytes:
    out = []
    for i in range(sb.n_reads):
        o0, o1 = int(sb.off[i]), int(sb.off[i + 1])
        out.append(b"@" + sb.read_name(i).encode() + b"\n" + bytes(sb.seq[o0:o1]) + b"\n+\n" + bytes(sb.qual[o0:o1]) + b"\n")
    return b"".join(out)


def write(out_dir: str) -> dict[str, str]:
    import oracle_lib as O
    from herro_amd import io as hio
    sb = case()
    os.makedirs(out_dir, exist_ok=True)
    files = {"reads.fastq": fastq_text(sb), "overlaps.paf": paf_text(sb)}
    store = O.store_from_synth(sb)
    for t in range(sb.n_targets):
        rid, rows, cigs = O.target_alignments(sb, t)
        res = store.extract_features(rid, rows, cigs, W)
        for w in range(len(res)):
            ow = res.window(w)
            import io
            b = io.BytesIO(); np.save(b, np.ascontiguousarray(np.stack([ow.bases, ow.quals], axis=0)))
            sup = np.zeros(len(ow.sup_pos), hio.SUPPORTED_DTYPE)
            sup["pos"], sup["ins"] = ow.sup_pos, ow.sup_ins
            s = io.BytesIO(); np.save(s, sup)
            d = f"features/{sb.read_name(rid)}"
            files[f"{d}/{w}.features.npy"] = b.getvalue()
            files[f"{d}/{w}.supported.npy"] = s.getvalue()
            files[f"{d}/{w}.ids.txt"] = "".join(sb.read_name(int(q)) + "\n" for q in ow.qids).encode()
    digest = {}
    for rel, data in sorted(files.items()):
        p = os.path.join(out_dir, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "wb") as f:
            f.write(data)
        digest[rel] = hashlib.sha256(data).hexdigest()
    with open(os.path.join(out_dir, "MANIFEST.sha256"), "w") as f:
        for rel, h in sorted(digest.items()):
            f.write(f"{h}  {rel}\n")
    return digest


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "features_w256")
    d = write(out)
    print(f"{len(d)} files under {out}; {sum(1 for k in d if k.endswith('.features.npy'))} windows")
