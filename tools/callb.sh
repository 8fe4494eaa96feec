#!/bin/bash
mkdir -p gpurun_out/r3e
timeout 150 python tools/large_window_cost.py 1024 > gpurun_out/r3e/large_window_cost.json 2> gpurun_out/r3e/large.err < /dev/null
q="--no-cpu-baseline --self-check 0 --e2e-jobs 0 --strong-windows 0 --repeats 2"
timeout 100 python bench.py $q --steps 20 --warmup 5 --min-jobs 2 > gpurun_out/r3e/minjobs2.json 2> /dev/null < /dev/null
timeout 100 python bench.py $q --steps 20 --warmup 5 > gpurun_out/r3e/minjobs1.json 2> /dev/null < /dev/null
timeout 100 python bench.py $q --precision 5 --streams 1 > gpurun_out/r3e/precision5_streams1.json 2> /dev/null < /dev/null
cat gpurun_out/r3e/large_window_cost.json; tail -3 gpurun_out/r3e/large.err
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r3e/*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1])
        if "value" in d: print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], d.get("repeat_ms_per_step"), {k:round(v["avg_us"]) for k,v in d.get("kernels",{}).items()})
    except Exception as e: print(f, e)
PY
