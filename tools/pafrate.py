"""Host ingest rate: herro_paf_parse on PAF text made from a synthetic batch (no device needed).
usage: python tools/pafrate.py [n_targets]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from herro_amd import api, synth

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sb = synth.generate_parallel(nt, 4 * 4096, 32, seed=5, chunk=32)
names = [sb.read_name(i).encode() for i in range(sb.n_reads)]
lines = []
for a in range(len(sb.aln)):
    r = sb.aln[a]
    lines.append(b"\t".join([names[r[0]], b"%d" % r[1], b"%d" % r[2], b"%d" % r[3], b"-" if r[4] else b"+", names[r[5]],
                             b"%d" % r[6], b"%d" % r[7], b"%d" % r[8], b"60", b"60", b"255", b"cg:Z:" + sb.cigar(a)]))
text = b"\n".join(lines) + b"\n"
windows = nt * 4
print(f"{len(lines)} overlaps, {len(text) / 1e6:.1f} MB of PAF, {windows} windows of 4096 bp")
ix = api.NameIndex(names)          # built once per read set (the reference's name_to_id)
for th in (1, 2, 4, 8, 16, 32):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        p = api.Paf(ix, text=text, threads=th, view=True)        # herro_paf_parse_view: the text stays where it is
        best = min(best, time.perf_counter() - t0)
        p.close()
    print(f"threads {th:2d}: {best * 1e3:7.1f} ms  {len(text) / best / 1e6:8.1f} MB/s  {windows / best / 1e3:8.1f} k windows/s")
