#!/bin/bash
# round 6: what herro_job_create costs the GPU — rocprofv3 kernel stats of an end_to_end leg with ONE feeder (no other job's kernels beside them), 4096-window jobs
# usage: gpurun --timeout 600 -- bash tools/r6_create_kernels.sh tag
tag=${1:-r6ck}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
q="--no-cpu-baseline --self-check 0 --strong-windows 0 --repeats 0 --steps 64 --warmup 32 --sustained 0 --sensitivity 0 --long-run-steps 0 --settle 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o ck -- python bench.py $q --e2e-feeders 1 --e2e-jobs 10 > $out/line.json 2> $out/err.txt < /dev/null
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/kernel_stats.csv; rm -rf $out/prof
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
for r in rows[:40]:
    n=r["Name"].replace("herro::(anonymous namespace)::","").replace("herro::","").replace("void ","").split("(")[0]
    print(f'{n:40s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.1f} min_us {float(r["MinNs"])/1e3:9.1f} pct {r["Percentage"]}')
PY
