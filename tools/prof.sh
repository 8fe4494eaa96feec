#!/bin/bash
# in-kernel phase cycles of the device-resident bench leg (pileup kernels: job_dev.h PROF_MARK; encoder stack: model_h.hip LP_MARK).
# The library with the phase timers is built HERE (no compile on the GPU box):
#   python -c "from herro_amd import build; build.build_hip(out='herro_amd/libherro_amd_prof.so', defines=('HERRO_PROF_BUILD',))"
# and selected by HERRO_LIB.  usage: gpurun -- bash tools/prof.sh tag [bench args]
tag=${1:-prof}; shift
mkdir -p gpurun_out/$tag
for size in "" "--steps 20 --warmup 5"; do
HERRO_LIB=$PWD/herro_amd/libherro_amd_prof.so HERRO_PROF=1 timeout 200 python bench.py --no-cpu-baseline --self-check 0 --streams 1 --e2e-jobs 0 --strong-windows 0 --repeats 0 --settle 0 $size "$@" 2>&1 | grep -E "^PROF|^\{" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
    else: print(l.strip())" | tee -a gpurun_out/$tag/prof.txt
done
