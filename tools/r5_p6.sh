#!/bin/bash
# round 5: precision 6 (f16 + e4m3 remainder on the MX MFMA) — model tests, then precision 4 against 6 on one box (one stream default size twice, the driver's size, two streams)
# usage: gpurun --timeout 600 -- bash tools/r5_p6.sh r5r [pytest targets]
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
t=${*:-tests/test_gpu_model.py}
timeout 400 python -m pytest $t -x -q -m gpu -s 2>&1 | grep -v Warn | grep -E "passed|failed|Error|error|precision [0-9]|assert|p6|calibration" | tail -30 | tee $out/tests.log
q="--no-cpu-baseline --self-check 2 --e2e-jobs 0 --strong-windows 0 --repeats 1 --long-run-steps 0"
for rep in 1 2; do for p in 4 6; do
  timeout 100 python bench.py $q --streams 1 --precision $p > $out/p${p}_s1_$rep.json 2>> $out/bench.err < /dev/null
done; done
for p in 4 6; do
  timeout 100 python bench.py $q --steps 20 --warmup 5 --precision $p > $out/p${p}_driver.json 2>> $out/bench.err < /dev/null
  timeout 100 python bench.py $q --precision $p > $out/p${p}_default.json 2>> $out/bench.err < /dev/null
done
if [ -f herro_amd/libherro_amd_prof.so ]; then for p in 4 6; do
  HERRO_LIB=$PWD/herro_amd/libherro_amd_prof.so HERRO_PROF=1 timeout 100 python bench.py $q --streams 1 --repeats 0 --settle 0 --self-check 0 --precision $p 2>&1 | grep -E "^PROF k_layers" | tail -1 | cut -c1-400 > $out/phases_p$p.txt
  echo "p$p $(cat $out/phases_p$p.txt)"
done; fi
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], d.get("stage_ms_per_step"), {k:round(v["avg_us"]) for k,v in d.get("kernels",{}).items() if k in ("layers_fused","fc_gemm","conv_fused")}, d["roofline"]["frac"], (d.get("self_check") or {}))
    except Exception as e: print(f, e)
PY
grep -v amdgpu.ids $out/bench.err | tail -3
