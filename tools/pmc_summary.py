#!/usr/bin/env python
"""Average rocprofv3 --pmc counters per kernel from *_counter_collection.csv."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("herro::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in agg.items():
    if flt in k:
        print(k, {c: round(sum(x) / len(x)) for c, x in sorted(v.items())})
