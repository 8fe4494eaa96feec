#!/bin/bash
# round 6: end_to_end leg with the CIGAR text copied up (HERRO_ZERO_COPY=1, default) against the scan kernel reading the registered host range itself (=2), one box
tag=${1:-r6zc}; out=gpurun_out/$tag; mkdir -p $out
HERRO_ZERO_COPY=2 timeout 300 python -m pytest tests/test_gpu_features.py tests/test_gpu_build_dev.py -x -q -m gpu 2>&1 | tail -2
q="--no-cpu-baseline --self-check 2 --strong-windows 0 --repeats 0 --steps 64 --warmup 32 --sustained 0 --sensitivity 0 --long-run-steps 0"
for rep in 1 2 3; do
  for zc in 1 2; do
    for nf in 4 6; do
      HERRO_ZERO_COPY=$zc timeout 200 python bench.py $q --e2e-feeders $nf 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); e=d['end_to_end']; print('zero_copy $zc feeders', e['feeders_per_gpu'], 'value', round(d['value']), 'e2e', round(e['windows_per_s']), round(e['windows_per_s']/d['value'],3), 'check', (d.get('self_check') or {}).get('ok'))" | tee -a $out/summary.txt
    done
  done
done
