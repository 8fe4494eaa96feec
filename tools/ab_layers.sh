#!/bin/bash
# A/B of the encoder stack: HERRO_LAYERS_Q=1 (32-token tiles for the short last round; during round 4 also k_layers_q) against 64-token tiles only
# (HERRO_LAYERS_Q=0), device-resident bench leg, default launch size and the driver's.  usage: gpurun -- bash tools/ab_layers.sh tag
tag=${1:-ab}
mkdir -p gpurun_out/$tag
timeout 500 python -m pytest ${TESTS:-tests/test_gpu_model.py tests/test_gpu_e2e.py} -x -q -m gpu 2>&1 | grep -v Warning | tail -6 > gpurun_out/$tag/tests.log; tail -3 gpurun_out/$tag/tests.log
q="--no-cpu-baseline --self-check 2 --e2e-jobs 0 --strong-windows 0 --repeats 1"
for v in 1 0; do
HERRO_LAYERS_Q=$v timeout 100 python bench.py $q --streams 1 > gpurun_out/$tag/q${v}_s1.json 2>> gpurun_out/$tag/bench.err < /dev/null
HERRO_LAYERS_Q=$v timeout 100 python bench.py $q > gpurun_out/$tag/q${v}_s2.json 2>> gpurun_out/$tag/bench.err < /dev/null
HERRO_LAYERS_Q=$v timeout 100 python bench.py $q --steps 20 --warmup 5 > gpurun_out/$tag/q${v}_driver.json 2>> gpurun_out/$tag/bench.err < /dev/null
done
cat gpurun_out/e2e_errors.json 2>/dev/null | head -c 600; echo
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/$tag/*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1])
        if "value" in d: print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], d.get("repeat_ms_per_step"), {k:round(v["avg_us"]) for k,v in d.get("kernels",{}).items()}, (d.get("self_check") or {}).get("ok"), d["roofline"]["frac"])
    except Exception as e: print(f, e)
PY
tail -5 gpurun_out/$tag/bench.err
