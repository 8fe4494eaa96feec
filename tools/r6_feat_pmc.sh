#!/bin/bash
# round 6: the three figures VERDICT r5 item 2 names — featurize ms per step with the driver's arguments, SQ_INSTS_VALU of the featurize kernels, their HBM traffic (separate --pmc passes).
# usage: gpurun --timeout 900 -- bash tools/r6_feat_pmc.sh tag
tag=${1:-r6pmc}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
q="--no-cpu-baseline --self-check 0 --e2e-jobs 0 --strong-windows 0 --long-run-steps 0 --sustained 0 --sensitivity 0"
timeout 200 python bench.py $q --steps 20 --warmup 5 > $out/driver.json 2> $out/bench.err < /dev/null
cmd="python bench.py $q --streams 1 --repeats 0 --settle 0 --steps 64 --warmup 32"
timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -o f -- $cmd > /dev/null 2> $out/f.err < /dev/null
timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -o w -- $cmd > /dev/null 2> $out/w.err < /dev/null
ff=$(find $out/f -name "*counter_collection.csv" | head -1); fw=$(find $out/w -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_traffic.py "$ff" "$fw" 32 $out/traffic.json 5 > $out/traffic_summary.txt 2>&1
rm -rf $out/f $out/w
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $out/m -o m -- $cmd > /dev/null 2> $out/m.err < /dev/null
fm=$(find $out/m -name "*counter_collection.csv" | head -1); [ -n "$fm" ] && python tools/pmc_summary.py "$fm" "k_" > $out/pmc_sq_counters.txt 2>&1; rm -rf $out/m
python - <<PY
import json
d=json.loads([x for x in open("$out/driver.json") if x.startswith("{")][-1])
print("driver args: value", round(d["value"]), "stage_ms_per_step", d.get("stage_ms_per_step"), "frac", d["roofline"]["frac"])
PY
cat $out/traffic_summary.txt; grep -E "k_cols|k_rows|k_win|k_rfq|k_layout" $out/pmc_sq_counters.txt | cut -c1-300
