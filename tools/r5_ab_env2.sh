#!/bin/bash
# NOTE (round 5): measurement switches (HERRO_RF_FUSED, HERRO_LAYERS_Q, HERRO_FC_G, HERRO_TILE_PACK, ...) exist only in libraries built with
# -DHERRO_PROF_BUILD: export HERRO_LIB=$PWD/herro_amd/libherro_amd_prof.so for them (tools/prof.sh says how it is built).
# parity tests, then a same-box A/B of one environment switch on the device-resident leg (one stream and the driver's size), twice each
# usage: gpurun --timeout 700 -- bash tools/r5_ab_env2.sh tag VAR "v0 v1" [pytest targets]
tag=$1; var=$2; vals=$3; shift 3
out=gpurun_out/$tag; mkdir -p $out
if [ -n "$*" ]; then timeout 500 python -m pytest "$@" -x -q -m gpu 2>&1 | grep -v Warn | tail -6 | tee $out/tests.log; fi
q="--no-cpu-baseline --self-check 2 --e2e-jobs 0 --strong-windows 0 --repeats 1 --long-run-steps 0"
for rep in 1 2; do for v in $vals; do
  env $var=$v timeout 120 python bench.py $q --streams 1 > $out/${var}${v}_s1_$rep.json 2>> $out/bench.err < /dev/null
done; done
for v in $vals; do
  env $var=$v timeout 120 python bench.py $q --steps 20 --warmup 5 > $out/${var}${v}_driver.json 2>> $out/bench.err < /dev/null
  env $var=$v timeout 120 python bench.py $q > $out/${var}${v}_default.json 2>> $out/bench.err < /dev/null
done
if [ -f herro_amd/libherro_amd_prof.so ]; then
  HERRO_LIB=$PWD/herro_amd/libherro_amd_prof.so HERRO_PROF=1 timeout 150 python bench.py $q --streams 1 --repeats 0 --settle 0 2>&1 | grep -E "^PROF" > $out/prof.txt
fi
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], d.get("stage_ms_per_step"), {k:round(v["avg_us"]) for k,v in d.get("kernels",{}).items()}, (d.get("self_check") or {}).get("ok"))
    except Exception as e: print(f, e)
PY
grep -v amdgpu.ids $out/bench.err | tail -3; cat $out/prof.txt 2>/dev/null | cut -c1-300
