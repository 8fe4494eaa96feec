#!/bin/bash
# The round's closing record from ONE GPU box: (1) rocprofv3 per-kernel stats of the device-resident leg (one stream) + its bench
# line, (2) HBM traffic per kernel (two separate --pmc passes), (3) MFMA-pipe / wave-cycle counters of the model kernels,
# (4) the full line with the driver's arguments (end_to_end, strong, cpu_baseline, long_run = the default size).  usage: gpurun --timeout 1500 -- bash tools/round_record.sh r4z
tag=${1:-rX}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -v Warning | tail -12 ) > $out/gpu_tests.log 2>&1; tail -6 $out/gpu_tests.log
q="--no-cpu-baseline --self-check 0 --e2e-jobs 0 --strong-windows 0 --long-run-steps 0 --sustained 0 --sensitivity 0"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o $tag -- python bench.py $q --streams 1 --repeats 0 > $out/bench_streams1.json 2> $out/prof.err < /dev/null
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/kernel_stats.csv; rm -rf $out/prof
cmd="python bench.py $q --streams 1 --repeats 0 --settle 0 --steps 64 --warmup 32"
timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -o f -- $cmd > /dev/null 2> $out/f.err < /dev/null
timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -o w -- $cmd > /dev/null 2> $out/w.err < /dev/null
ff=$(find $out/f -name "*counter_collection.csv" | head -1); fw=$(find $out/w -name "*counter_collection.csv" | head -1)
prec=$(python -c "import json; print(json.loads([x for x in open('$out/bench_streams1.json') if x.startswith('{')][-1])['config']['precision'])" 2>/dev/null || echo 4)   # the tier the load-time calibration chose (round 6)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_traffic.py "$ff" "$fw" 32 $out/traffic.json $prec > $out/traffic_summary.txt 2>&1
rm -rf $out/f $out/w
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $out/m -o m -- $cmd > /dev/null 2> $out/m.err < /dev/null
fm=$(find $out/m -name "*counter_collection.csv" | head -1); [ -n "$fm" ] && python tools/pmc_summary.py "$fm" "k_" > $out/pmc_sq_mfma_counters.txt 2>&1; rm -rf $out/m
# (the full default-size line is left out since round 5: the driver-arguments line carries the default-size figure as long_run; GPU budget)
: > $out/bench.err
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_args.json 2>> $out/bench.err < /dev/null
head -14 $out/kernel_stats.csv 2>/dev/null | cut -c1-140; cat $out/traffic_summary.txt; grep -E "k_layers|k_conv|k_fc|k_cols|k_rows|k_win|k_rfq|k_layout" $out/pmc_sq_mfma_counters.txt | cut -c1-330
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], {k:round(v["avg_us"]) for k,v in d.get("kernels",{}).items()}, (d.get("end_to_end") or {}).get("windows_per_s"), (d.get("strong") or {}).get("windows_per_s"), (d.get("self_check") or {}).get("ok"), d["roofline"]["frac"], (d.get("long_run") or {}).get("value"))
    except Exception as e: print(f, "unreadable", e)
PY
