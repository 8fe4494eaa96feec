#!/bin/bash
# round 6 iteration check: parity tests, bench at both sizes (one stream and default), phase cycles of the timer build.
# usage: gpurun --timeout 1200 -- bash tools/r6_quick.sh tag ["pytest targets"]
tag=${1:-r6q}
out=gpurun_out/$tag; mkdir -p $out
timeout 800 python -m pytest ${2:-tests/test_gpu_model.py tests/test_gpu_e2e.py tests/test_gpu_lean.py tests/test_gpu_sib_retry.py} -x -q -m gpu -s 2>&1 | grep -v Warning | tail -60 > $out/tests.log; tail -5 $out/tests.log
cp gpurun_out/e2e_errors.json $out/e2e_logit_errors.json 2>/dev/null
q="--no-cpu-baseline --self-check 2 --e2e-jobs 0 --strong-windows 0 --repeats 1 --long-run-steps 0"
timeout 150 python bench.py $q --streams 1 > $out/s1.json 2>> $out/bench.err < /dev/null
timeout 150 python bench.py $q --streams 1 --precision 4 > $out/s1_p4.json 2>> $out/bench.err < /dev/null
timeout 150 python bench.py $q > $out/default.json 2>> $out/bench.err < /dev/null
timeout 150 python bench.py $q --steps 20 --warmup 5 > $out/driver.json 2>> $out/bench.err < /dev/null
if [ -f herro_amd/libherro_amd_prof.so ]; then
  HERRO_LIB=$PWD/herro_amd/libherro_amd_prof.so HERRO_PROF=1 timeout 150 python bench.py $q --streams 1 --repeats 0 --settle 0 --precision 4 2>&1 | grep -E "^PROF" > $out/prof.txt
fi
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1])
        if "value" in d: print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], d["config"].get("precision"), d.get("stage_ms_per_step"), {k:round(v["avg_us"]) for k,v in d.get("kernels",{}).items()}, (d.get("self_check") or {}).get("ok"))
    except Exception as e: print(f, e)
PY
grep -v amdgpu.ids $out/bench.err | tail -5; cat $out/prof.txt 2>/dev/null | cut -c1-400
