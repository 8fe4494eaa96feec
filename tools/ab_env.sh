#!/bin/bash
# NOTE (round 5): measurement switches (HERRO_RF_FUSED, HERRO_LAYERS_Q, HERRO_FC_G, HERRO_TILE_PACK, ...) exist only in libraries built with
# -DHERRO_PROF_BUILD: export HERRO_LIB=$PWD/herro_amd/libherro_amd_prof.so for them (tools/prof.sh says how it is built).
# A/B of one environment switch on the same box: device-resident bench leg at the default launch size and at the driver's.
# usage: gpurun -- bash tools/ab_env.sh tag VAR "values" [pytest targets]
tag=$1; var=$2; vals=$3; shift 3
mkdir -p gpurun_out/$tag
if [ -n "$*" ]; then timeout 600 python -m pytest "$@" -x -q -m gpu 2>&1 | grep -v Warn | tail -4 | tee gpurun_out/$tag/tests.log; fi
q="--no-cpu-baseline --self-check 2 --e2e-jobs 0 --strong-windows 0 --repeats 1"
for v in $vals; do
  env $var=$v timeout 100 python bench.py $q > gpurun_out/$tag/${var}${v}.json 2>> gpurun_out/$tag/bench.err < /dev/null
  env $var=$v timeout 100 python bench.py $q --steps 20 --warmup 5 > gpurun_out/$tag/${var}${v}_driver.json 2>> gpurun_out/$tag/bench.err < /dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/$tag/*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1])
        if "value" in d: print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], d.get("repeat_ms_per_step"), {k:round(v["avg_us"]) for k,v in d.get("kernels",{}).items()}, (d.get("self_check") or {}).get("ok"))
    except Exception as e: print(f, e)
PY
grep -v amdgpu.ids gpurun_out/$tag/bench.err | tail -5
