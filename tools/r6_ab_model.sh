#!/bin/bash
# round 6: model kernels before / after on one box — libraries given by name; the driver's 20-step size and the default size on one stream; kernel times from HIP events.
# usage: gpurun --timeout 900 -- bash tools/r6_ab_model.sh tag libA.so libB.so
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
q="--no-cpu-baseline --self-check 0 --e2e-jobs 0 --strong-windows 0 --repeats 2 --long-run-steps 0 --sustained 0 --sensitivity 0"
for rep in 1 2; do
for lib in "$@"; do
  for sz in 20 256; do
    sa="--streams 1"; [ $sz = 20 ] && sa="--steps 20 --warmup 5"
    HERRO_LIB=$PWD/herro_amd/$lib timeout 200 python bench.py $q $sa > $out/${lib}_${sz}_$rep.log 2>&1
    python - <<PY
import json
j=[l for l in open("$out/${lib}_${sz}_$rep.log") if l.startswith("{")]
if j:
    d=json.loads(j[-1]); print("$lib steps=$sz rep$rep value", round(d["value"]), "repeats", [round(128/x*1e3) for x in d.get("repeat_ms_per_step",[])], {k:round(v["avg_us"]) for k,v in d["kernels"].items() if k in ("conv_fused","fc_gemm","layers_fused","consensus","build_tokens")})
else: print("$lib steps=$sz rep$rep: no line")
PY
  done
done
done 2>&1 | tee $out/summary.txt
