#!/bin/bash
# round 6: end_to_end leg, feeder modes and counts with the device build (one box)   usage: gpurun --timeout 1200 -- bash tools/r6_e2e_modes.sh tag
tag=${1:-r6m}; out=gpurun_out/$tag; mkdir -p $out
q="--no-cpu-baseline --self-check 0 --strong-windows 0 --repeats 0 --steps 64 --warmup 32 --sustained 0 --sensitivity 0 --long-run-steps 0"
for rep in 1 2; do
  for mode in serial producer; do
    for nf in 2 3 4 6; do
      timeout 200 python bench.py $q --e2e-feeders $nf --e2e-mode $mode > $out/${mode}_f${nf}_$rep.json 2>> $out/err.txt < /dev/null
    done
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1]); e=d["end_to_end"]
        print(f.split("/")[-1], "value", round(d["value"]), "e2e", round(e["windows_per_s"]), "ratio", round(e["windows_per_s"]/d["value"],3), "prep/feeder", round(e["host_prepare_windows_per_s_per_feeder"]), "feeders", e["feeders_per_gpu"])
    except Exception as ex: print(f, ex)
PY
