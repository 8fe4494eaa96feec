#!/usr/bin/env python
"""From a rocprofv3 kernel-trace CSV: per-kernel average duration and the idle gap that precedes each
kernel on the device (start[i] - max(end[<i])).  usage: trace_gaps.py kernel_trace.csv"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
last_end = None
for s, e, n in ks:
    n = re.sub(r"\(.*", "", n); n = re.sub(r"^void ", "", n)
    dur[n].append(e - s)
    if last_end is not None: gap[n].append(max(0, s - last_end))
    last_end = e if last_end is None else max(last_end, e)
busy = sum(sum(v) for v in dur.values()); span = ks[-1][1] - ks[0][0]
print(f"kernels {len(ks)}  span {span/1e3:.0f} us  busy {busy/1e3:.0f} us ({100*busy/span:.1f} %)")
print(f"{'kernel':60s} {'calls':>6s} {'avg us':>8s} {'gap before us (median)':>22s}")
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    g = sorted(gap[n]); med = g[len(g)//2]/1e3 if g else 0
    print(f"{n[:60]:60s} {len(v):6d} {sum(v)/len(v)/1e3:8.1f} {med:22.1f}")
