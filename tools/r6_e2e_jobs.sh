#!/bin/bash
# round 6: the end_to_end leg with 6 / 12 jobs per feeder (driver's arguments otherwise): how much of the figure is pipeline fill and drain, and what the longer leg costs in run time
q="--no-cpu-baseline --self-check 0 --strong-windows 0 --repeats 0 --steps 20 --warmup 5 --sustained 0 --sensitivity 0 --long-run-steps 0"
mkdir -p gpurun_out/r6e2ej
for rep in 1 2; do for nj in 8 32; do
  t0=$(date +%s.%N)
  timeout 700 python bench.py $q --e2e-jobs $nj > gpurun_out/r6e2ej/j${nj}_$rep.json 2> gpurun_out/r6e2ej/j${nj}_$rep.err < /dev/null
  t1=$(date +%s.%N)
  python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r6e2ej/j${nj}_$rep.json") if x.startswith("{")][-1]); e=d["end_to_end"]
print("jobs/feeder $nj rep$rep value", round(d["value"]), "e2e", round(e["windows_per_s"]), "ratio", round(e["windows_per_s"]/d["value"],3), "bench wall %.0f s" % ($t1 - $t0))
PY
done; done 2>&1 | tee gpurun_out/r6e2ej/summary.txt
