#!/bin/bash
# every -m gpu test, then the driver's command line in full (device-resident leg, long_run, end_to_end, strong at 100 000 windows, cpu_baseline)
# usage: gpurun --timeout 1500 -- bash tools/r5_full.sh tag
tag=${1:-r5f}
out=gpurun_out/$tag; mkdir -p $out
( time timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -v Warning | tail -30 ) > $out/tests.log 2>&1; tail -12 $out/tests.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench.err < /dev/null ) 2> $out/bench_time.txt; cat $out/bench_time.txt
python - <<PY
import json
try:
    d=json.loads([x for x in open("$out/bench_driver_args.json") if x.startswith("{")][-1])
    print("value", round(d["value"]), "ms/step", d["ms_per_step"], "stage", d.get("stage_ms_per_step"))
    print("kernels", {k:round(v["avg_us"]) for k,v in d.get("kernels",{}).items()})
    print("e2e", {k:v for k,v in (d.get("end_to_end") or {}).items() if k!="note"})
    print("strong", {k:v for k,v in (d.get("strong") or {}).items() if k not in ("timed",)})
    print("long_run", d.get("long_run")); print("self_check", d.get("self_check")); print("roofline", d["roofline"]); print("feat", d["roofline_featurize_group"])
    print("cpu", d.get("cpu_baseline"))
except Exception as e: print("unreadable", e)
PY
grep -v amdgpu.ids $out/bench.err | tail -8
