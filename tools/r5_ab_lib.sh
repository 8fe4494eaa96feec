#!/bin/bash
# A/B of two builds of the library on one box: the release one and $2 (HERRO_LIB), device-resident leg on one stream + at the driver's size.
# usage: gpurun --timeout 600 -- bash tools/r5_ab_lib.sh tag herro_amd/libherro_amd_X.so [pytest targets]
tag=$1; alt=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
if [ -n "$*" ]; then timeout 400 python -m pytest "$@" -x -q -m gpu 2>&1 | grep -v Warn | tail -4 | tee $out/tests.log; fi
q="--no-cpu-baseline --self-check 2 --e2e-jobs 0 --strong-windows 0 --repeats 1 --long-run-steps 0"
for rep in 1 2; do
  timeout 120 python bench.py $q --streams 1 > $out/rel_s1_$rep.json 2>> $out/bench.err < /dev/null
  HERRO_LIB=$PWD/$alt timeout 120 python bench.py $q --streams 1 > $out/alt_s1_$rep.json 2>> $out/bench.err < /dev/null
done
timeout 120 python bench.py $q --steps 20 --warmup 5 > $out/rel_driver.json 2>> $out/bench.err < /dev/null
HERRO_LIB=$PWD/$alt timeout 120 python bench.py $q --steps 20 --warmup 5 > $out/alt_driver.json 2>> $out/bench.err < /dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], {k:round(v["avg_us"]) for k,v in d.get("kernels",{}).items()}, (d.get("self_check") or {}).get("ok"))
    except Exception as e: print(f, e)
PY
grep -v amdgpu.ids $out/bench.err | tail -3
