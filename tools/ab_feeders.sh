#!/bin/bash
# end_to_end leg of bench.py against the number of feeder threads (one context each), same box.  usage: gpurun -- bash tools/ab_feeders.sh "4 6 8"
q="${Q:---steps 8 --warmup 4} --no-cpu-baseline --self-check 0 --strong-windows 0 --repeats 0 --long-run-steps 0"
for f in ${1:-4 6 8}; do
  timeout 150 python bench.py $q --e2e-feeders $f 2>/dev/null | python -c "
import sys,json
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); e=d['end_to_end']; print('feeders $f', round(e['windows_per_s']), 'prepare/feeder', round(e['host_prepare_windows_per_s_per_feeder']), 'windows', e['windows'])"
done
