#!/bin/bash
# in-kernel phase cycles (HERRO_PROF=1) of the device-resident bench leg
HERRO_PROF=1 timeout 200 python bench.py --no-cpu-baseline --self-check 0 --streams 1 --e2e-jobs 0 --repeats 0 --settle 0 --steps 20 --warmup 5 2>&1 | grep -E "^PROF|^\{" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
    else: print(l.strip())"
