#!/bin/bash
for d in 0 1 2 3 4 8 16; do
  HERRO_QDBG=$d timeout 200 python bench.py --no-cpu-baseline --self-check 0 --streams 1 --e2e-jobs 0 --repeats 0 --settle 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('dbg=$d', {k:round(v['avg_us'],1) for k,v in d['kernels'].items() if k in ('rf_quals','tokens','cols')})"
done
