#!/bin/bash
# round 5, call p: (1) the sibling-tile recovery test + the model tests (k_conv_m took a parameter), (2) same-box A/B of conv / FC chunk by chunk
# (HERRO_CONV_CHUNK tokens per chunk, HERRO_PROF_BUILD library) on the device-resident leg: one stream at the default size twice, the driver's size once
# usage: gpurun --timeout 600 -- bash tools/r5_ab_chunk.sh r5p
tag=$1; out=gpurun_out/$tag; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_sib_retry.py tests/test_gpu_model.py -x -q -m gpu -s 2>&1 | grep -v Warn | grep -E "passed|failed|Error|error|sibling tiles vs|assert" | tail -12 | tee $out/tests.log
export HERRO_LIB=$PWD/herro_amd/libherro_amd_prof.so
q="--no-cpu-baseline --self-check 2 --e2e-jobs 0 --strong-windows 0 --repeats 1 --long-run-steps 0"
for rep in 1 2; do for v in 0 16384 32768; do
  HERRO_CONV_CHUNK=$v timeout 100 python bench.py $q --streams 1 > $out/chunk${v}_s1_$rep.json 2>> $out/bench.err < /dev/null
done; done
for v in 0 16384; do
  HERRO_CONV_CHUNK=$v timeout 100 python bench.py $q --steps 20 --warmup 5 > $out/chunk${v}_driver.json 2>> $out/bench.err < /dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], d.get("stage_ms_per_step"), {k:round(v["avg_us"]) for k,v in d.get("kernels",{}).items()}, (d.get("self_check") or {}).get("ok"))
    except Exception as e: print(f, e)
PY
grep -v amdgpu.ids $out/bench.err | tail -3
