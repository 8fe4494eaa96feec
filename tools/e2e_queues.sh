#!/bin/bash
# end_to_end leg against the number of hardware queues the HIP runtime multiplexes its streams onto (GPU_MAX_HW_QUEUES, default 4):
# 6 contexts x 2 streams share them, and a job's text copy can sit behind another context's kernels.  usage: gpurun -- bash tools/e2e_queues.sh tag
tag=${1:-q}
mkdir -p gpurun_out/$tag
for q in ${QUEUES:-16 8}; do
  GPU_MAX_HW_QUEUES=$q timeout 120 python bench.py --no-cpu-baseline --self-check 0 --strong-windows 0 --repeats 0 --steps 64 --warmup 32 > gpurun_out/$tag/q$q.json 2> gpurun_out/$tag/q$q.err < /dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/$tag/q*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1]); e=d["end_to_end"]
        print(f.split("/")[-1], round(d["value"]), round(e["windows_per_s"]), round(e["host_prepare_windows_per_s_per_feeder"]))
    except Exception as ex: print(f, ex)
PY
