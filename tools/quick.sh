#!/bin/bash
# short GPU check while iterating on a kernel: the parity tests named in $TESTS (default: the model + end-to-end ones), then the
# device-resident bench leg at the default launch size and at the driver's.  usage: gpurun -- bash tools/quick.sh tag
tag=${1:-q}
mkdir -p gpurun_out/$tag
timeout 400 python -m pytest ${TESTS:-tests/test_gpu_model.py tests/test_gpu_e2e.py} -x -q -m gpu 2>&1 | grep -v Warning | tail -6 > gpurun_out/$tag/tests.log; tail -3 gpurun_out/$tag/tests.log
q="--no-cpu-baseline --self-check 2 --e2e-jobs 0 --strong-windows 0 --repeats 1"
timeout 100 python bench.py $q ${BENCH_ARGS} > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err < /dev/null
timeout 100 python bench.py $q --steps 20 --warmup 5 ${BENCH_ARGS} > gpurun_out/$tag/bench_driver.json 2>> gpurun_out/$tag/bench.err < /dev/null
cat gpurun_out/e2e_errors.json 2>/dev/null | head -c 600; echo
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/$tag/*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1])
        if "value" in d: print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], d.get("repeat_ms_per_step"), {k:round(v["avg_us"]) for k,v in d.get("kernels",{}).items()}, (d.get("self_check") or {}).get("ok"))
    except Exception as e: print(f, e)
PY
