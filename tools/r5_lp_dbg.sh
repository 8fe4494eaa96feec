#!/bin/bash
# round 5: timing experiments on k_layers_p's GEMM calls (libraries built with -DHERRO_PROF_BUILD -DHERRO_LP_DBG=v: 1 half of the LDS fragment reads, 2 half of the
# MFMAs, 4 no weight loads inside a call — WRONG results, timing only): which resource bounds the stack?   usage: gpurun --timeout 400 -- bash tools/r5_lp_dbg.sh r5q "0 1 2 4"
tag=$1; out=gpurun_out/$tag; mkdir -p $out
q="--no-cpu-baseline --self-check 0 --e2e-jobs 0 --strong-windows 0 --repeats 0 --settle 0 --long-run-steps 0 --streams 1"
for v in $2; do
  HERRO_LIB=$PWD/herro_amd/libherro_lp$v.so HERRO_PROF=1 timeout 100 python bench.py $q > $out/lp$v.json 2> $out/lp$v.err < /dev/null
  grep -E "^PROF k_layers_p" $out/lp$v.err | tail -1 | cut -c1-400 > $out/lp${v}_phases.txt
  python - <<PY
import json
try:
    d=json.loads([x for x in open("$out/lp$v.json") if x.startswith("{")][-1])
    print("LP_DBG=$v", round(d["value"]), {k:round(x["avg_us"]) for k,x in d["kernels"].items() if k in ("layers_fused","fc_gemm","conv_fused")})
except Exception as e: print("lp$v", e)
PY
  cat $out/lp${v}_phases.txt
done
