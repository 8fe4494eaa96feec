#!/bin/bash
# A/B of the release library against $2 on one box (one stream, twice each) + the phase timers of the HERRO_PROF_BUILD library
# usage: gpurun --timeout 600 -- bash tools/r5_ab_lib2.sh tag herro_amd/libherro_amd_alt.so [pytest targets]
tag=$1; alt=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
if [ -n "$*" ]; then timeout 300 python -m pytest "$@" -x -q -m gpu 2>&1 | grep -v Warn | tail -4 | tee $out/tests.log; fi
q="--no-cpu-baseline --self-check 2 --e2e-jobs 0 --strong-windows 0 --repeats 1 --long-run-steps 0"
for rep in 1 2; do
  timeout 120 python bench.py $q --streams 1 > $out/rel_s1_$rep.json 2>> $out/bench.err < /dev/null
  HERRO_LIB=$PWD/$alt timeout 120 python bench.py $q --streams 1 > $out/alt_s1_$rep.json 2>> $out/bench.err < /dev/null
done
HERRO_LIB=$PWD/herro_amd/libherro_amd_prof.so HERRO_PROF=1 timeout 150 python bench.py $q --streams 1 --repeats 0 --settle 0 2>&1 | grep -E "^PROF" > $out/prof.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], {k:round(v["avg_us"]) for k,v in d.get("kernels",{}).items()}, (d.get("self_check") or {}).get("ok"))
    except Exception as e: print(f, e)
PY
grep -v amdgpu.ids $out/bench.err | tail -3; cat $out/prof.txt | cut -c1-300
