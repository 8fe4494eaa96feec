#!/bin/bash
# round 6: the driver's 20-step region as ONE launch group on one stream (the default) against two groups of 10 steps on two streams (--min-jobs 2); one box.
# usage: gpurun --timeout 900 -- bash tools/r6_value_streams.sh tag
tag=${1:-r6vs}; out=gpurun_out/$tag; mkdir -p $out
( time timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | grep -v Warning | tail -6 ) > $out/gpu_tests.log 2>&1; tail -4 $out/gpu_tests.log
q="--no-cpu-baseline --self-check 0 --e2e-jobs 0 --strong-windows 0 --long-run-steps 0 --sustained 0 --sensitivity 0 --steps 20 --warmup 5"
for rep in 1 2; do
  for mj in 1 2; do
    timeout 200 python bench.py $q --min-jobs $mj > $out/mj${mj}_$rep.log 2>&1
    python - <<PY
import json
j=[l for l in open("$out/mj${mj}_$rep.log") if l.startswith("{")]
if j:
    d=json.loads(j[-1]); print("min-jobs $mj rep$rep value", round(d["value"]), "repeats", [round(20*128/(x*20)*1e3) for x in d.get("repeat_ms_per_step",[])], "streams", d["config"]["streams_per_gpu"], {k:round(v["avg_us"]) for k,v in d["kernels"].items()})
else: print("min-jobs $mj rep$rep: no line")
PY
  done
done 2>&1 | tee $out/summary.txt
