#!/bin/bash
# round 6: the end_to_end leg of two libraries on one box, alternating, three repeats (4096-window jobs, six feeders)
# usage: gpurun --timeout 1500 -- bash tools/r6_e2e_libs.sh tag libA.so libB.so
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
q="--no-cpu-baseline --self-check 0 --strong-windows 0 --repeats 0 --steps ${STEPS:-64} --warmup ${WARM:-32} --sustained 0 --sensitivity 0 --long-run-steps 0"
for rep in 1 2 3 4; do
  for lib in "$@"; do
    HERRO_LIB=$PWD/herro_amd/$lib timeout 250 python bench.py $q > $out/${lib}_$rep.json 2>> $out/err.txt < /dev/null
    python - <<PY
import json
try:
    d=json.loads([x for x in open("$out/${lib}_$rep.json") if x.startswith("{")][-1]); e=d["end_to_end"]
    print("$lib rep$rep value", round(d["value"]), "e2e", round(e["windows_per_s"]), "ratio", round(e["windows_per_s"]/d["value"],3), "prep/feeder", round(e["host_prepare_windows_per_s_per_feeder"]))
except Exception as ex: print("$lib rep$rep", ex)
PY
  done
done 2>&1 | tee $out/summary.txt
