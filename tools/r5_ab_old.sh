#!/bin/bash
# same-box A/B of two builds of the library (release defaults): herro_amd/libherro_old.so (the tree before the precision-6 work: r5y) against the current one
# usage: gpurun --timeout 500 -- bash tools/r5_ab_old.sh r5z2
tag=$1; out=gpurun_out/$tag; mkdir -p $out
q="--no-cpu-baseline --self-check 2 --e2e-jobs 0 --strong-windows 0 --repeats 1 --long-run-steps 0"
for rep in 1 2; do for v in old new; do
  lib=$PWD/herro_amd/libherro_amd.so; [ $v = old ] && lib=$PWD/herro_amd/libherro_old.so
  HERRO_LIB=$lib timeout 100 python bench.py $q --steps 20 --warmup 5 > $out/${v}_driver_$rep.json 2>> $out/err < /dev/null
  HERRO_LIB=$lib timeout 100 python bench.py $q --streams 1 > $out/${v}_s1_$rep.json 2>> $out/err < /dev/null
done; done
for v in old new; do
  lib=$PWD/herro_amd/libherro_amd.so; [ $v = old ] && lib=$PWD/herro_amd/libherro_old.so
  HERRO_LIB=$lib timeout 100 python bench.py $q > $out/${v}_default.json 2>> $out/err < /dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], {k:round(v["avg_us"]) for k,v in d.get("kernels",{}).items() if k in ("layers_fused","fc_gemm","conv_fused")}, round(d["roofline"]["frac"],4), (d.get("self_check") or {}).get("ok"))
    except Exception as e: print(f, e)
PY
grep -v amdgpu.ids $out/err | tail -3
