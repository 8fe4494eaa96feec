#!/bin/bash
# round 6: featurize kernels before / after on one box — libraries given by name (release + timer builds), one stream; kernel times from HIP events, phase cycles from the timer builds.
# usage: gpurun --timeout 900 -- bash tools/r6_ab_feat.sh tag libA.so libB.so [timerA.so timerB.so]
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
q="--no-cpu-baseline --self-check 0 --e2e-jobs 0 --strong-windows 0 --repeats 0 --settle 0 --long-run-steps 0 --sustained 0 --sensitivity 0"
for rep in 1 2; do
for lib in "$@"; do
  for st in 1 0; do
    sa=""; [ $st = 1 ] && sa="--streams 1"
    HERRO_LIB=$PWD/herro_amd/$lib HERRO_PROF=1 timeout 200 python bench.py $q $sa > $out/${lib}_s${st}_$rep.log 2>&1
    python - <<PY
import json
L=[l for l in open("$out/${lib}_s${st}_$rep.log")]
j=[l for l in L if l.startswith("{")]
if j:
    d=json.loads(j[-1]); print("$lib streams=$st rep$rep value", round(d["value"]), "featurize ms/step", d.get("stage_ms_per_step",{}).get("featurize"), {k:round(v["avg_us"]) for k,v in d["kernels"].items() if k in ("cols","win","layout","rows","rf_quals")})
for l in L:
    if l.startswith("PROF kernel 0") or l.startswith("PROF kernel 6"): print("   ", l.strip())
PY
  done
done
done 2>&1 | tee $out/summary.txt
