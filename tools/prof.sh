#!/bin/bash
# in-kernel phase cycles of the device-resident bench leg (pileup kernels: job_dev.h PROF_MARK; encoder stack: model_h.hip LP_MARK):
# rebuilds the library with the phase timers (HERRO_PROF_BUILD),
# runs with HERRO_PROF=1, and puts the release build back
HERRO_PROF_BUILD=1 python -c "from herro_amd import build; build.build_hip(force=True)"
trap 'python -c "from herro_amd import build; build.build_hip(force=True)"' EXIT
HERRO_PROF=1 timeout 200 python bench.py --no-cpu-baseline --self-check 0 --streams 1 --e2e-jobs 0 --repeats 0 --settle 0 --steps 20 --warmup 5 2>&1 | grep -E "^PROF|^\{" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
    else: print(l.strip())"
