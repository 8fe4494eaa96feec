#!/bin/bash
# round 5: where does precision 6's time go?  kernel times of precision 4 / 5 / 6 and of precision 6 without its K = 128 instructions (8), without the e4m3 conversions (16), without both (24)
# (HERRO_LP_DBG libraries: wrong results, timing only)   usage: gpurun --timeout 400 -- bash tools/r5_p6_dbg.sh r5v
tag=$1; out=gpurun_out/$tag; mkdir -p $out
q="--no-cpu-baseline --self-check 0 --e2e-jobs 0 --strong-windows 0 --repeats 0 --settle 0 --long-run-steps 0 --streams 1"
for p in 4 5 6; do timeout 100 python bench.py $q --precision $p > $out/p$p.json 2>> $out/err < /dev/null; done
for v in 8 16 24; do HERRO_FORCE_PRECISION=1 HERRO_LIB=$PWD/herro_amd/libherro_lp$v.so timeout 100 python bench.py $q --precision 6 > $out/p6_dbg$v.json 2>> $out/err < /dev/null; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]), {k:round(v["avg_us"]) for k,v in d.get("kernels",{}).items() if k in ("layers_fused",)})
    except Exception as e: print(f, e)
PY
grep -v amdgpu.ids $out/err | tail -3
