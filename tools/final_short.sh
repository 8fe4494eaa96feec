#!/bin/bash
# the round's closing record on a short GPU budget: rocprofv3 kernel stats of the device-resident leg (one stream), then the full
# default bench line, then (if time is left) the line with the driver's arguments.  usage: gpurun -- bash tools/final_short.sh r3k
tag=${1:-rX}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o $tag -- \
  python bench.py --no-cpu-baseline --self-check 0 --streams 1 --e2e-jobs 0 --strong-windows 0 --repeats 0 > $out/bench_streams1.json 2> $out/prof.err < /dev/null
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $out/kernel_stats.csv; fi
rm -rf $out/prof
timeout 170 python bench.py > $out/bench.json 2> $out/bench.err < /dev/null
timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_args.json 2>> $out/bench.err < /dev/null
head -8 $out/kernel_stats.csv 2>/dev/null | cut -c1-150
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], {k:round(v["avg_us"]) for k,v in d.get("kernels",{}).items()}, (d.get("end_to_end") or {}).get("windows_per_s"), (d.get("strong") or {}).get("windows_per_s"), (d.get("self_check") or {}).get("ok"), d["roofline"]["frac"])
    except Exception as e: print(f, "unreadable", e)
PY
