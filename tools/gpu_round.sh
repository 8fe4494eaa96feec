#!/bin/bash
# one GPU round: parity tests, then the driver's bench line; summaries into gpurun_out/
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -${TAILN:-25} > gpurun_out/tests.log; cat gpurun_out/tests.log
timeout 400 python bench.py --steps ${STEPS:-20} --warmup 5 ${BENCH_ARGS} > gpurun_out/bench.log 2>&1
python - <<PY
import json
l=[x for x in open("gpurun_out/bench.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print(d["value"], d["ms_per_step"], d["stage_ms_per_step"]); print({k:round(v["avg_us"],1) for k,v in d["kernels"].items()}); print(d["self_check"]); print("e2e", d["end_to_end"]["windows_per_s"])
else: print(open("gpurun_out/bench.log").read()[-2000:])
PY
