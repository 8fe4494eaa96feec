#!/bin/bash
# host timeline of the end_to_end leg: HERRO_HOST_PROFILE=1 prints, per herro_job_create, the staging / copy / scan / windowing
# split.  usage: gpurun -- bash tools/e2e_hostprof.sh tag
tag=${1:-q}
mkdir -p gpurun_out/$tag
HERRO_HOST_PROFILE=1 timeout 150 python bench.py --no-cpu-baseline --self-check 0 --strong-windows 0 --repeats 0 --steps 64 --warmup 32 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/hostprof.txt < /dev/null
grep -c "cigar scan" gpurun_out/$tag/hostprof.txt
tail -60 gpurun_out/$tag/hostprof.txt | cut -c1-260
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/$tag/bench.json") if x.startswith("{")][-1])
e=d["end_to_end"]; print(d["value"], {k:v for k,v in e.items() if k!="note"})
PY
