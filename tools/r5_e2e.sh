#!/bin/bash
# lean-path parity + quick bench, then the end_to_end leg with 3 / 4 / 6 / 8 feeders (zero-copy text) on one box
# usage: gpurun --timeout 1200 -- bash tools/r5_e2e.sh tag
tag=${1:-r5e2e}
out=gpurun_out/$tag; mkdir -p $out
timeout 500 python -m pytest tests/test_gpu_lean.py tests/test_gpu_features.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | grep -v Warning | tail -25 > $out/tests.log; tail -5 $out/tests.log
q="--no-cpu-baseline --self-check 2 --strong-windows 0 --repeats 1 --long-run-steps 0"
timeout 150 python bench.py $q --e2e-jobs 0 --streams 1 > $out/s1.json 2>> $out/bench.err < /dev/null
timeout 150 python bench.py $q --e2e-jobs 0 --steps 20 --warmup 5 > $out/driver.json 2>> $out/bench.err < /dev/null
for nf in 3 4 6 8; do
  timeout 200 python bench.py $q --e2e-feeders $nf > $out/e2e_f$nf.json 2>> $out/bench.err < /dev/null
done
HERRO_ZERO_COPY=0 timeout 200 python bench.py $q --e2e-feeders 6 > $out/e2e_f6_staged.json 2>> $out/bench.err < /dev/null
if [ -f herro_amd/libherro_amd_prof.so ]; then
  HERRO_LIB=$PWD/herro_amd/libherro_amd_prof.so HERRO_PROF=1 timeout 150 python bench.py $q --e2e-jobs 0 --streams 1 --repeats 0 --settle 0 2>&1 | grep -E "^PROF" > $out/prof.txt
fi
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1])
        e=d.get("end_to_end") or {}
        print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], d.get("stage_ms_per_step"), {k:round(v["avg_us"]) for k,v in d.get("kernels",{}).items()}, "e2e", e.get("windows_per_s"), e.get("host_prepare_windows_per_s_per_feeder"), (d.get("self_check") or {}).get("ok"))
    except Exception as e: print(f, e)
PY
grep -v amdgpu.ids $out/bench.err | tail -5; cat $out/prof.txt 2>/dev/null | cut -c1-300
