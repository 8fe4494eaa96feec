#!/bin/bash
# round 6: the end_to_end leg with the job's windows / descriptors built on the device (default) against the host build of rounds 3-5, one box; + the host timeline of both
# usage: gpurun --timeout 1200 -- bash tools/r6_e2e_ab.sh tag
tag=${1:-r6e2e}; out=gpurun_out/$tag; mkdir -p $out
q="--no-cpu-baseline --self-check 0 --strong-windows 0 --repeats 0 --steps 64 --warmup 32 --sustained 0 --sensitivity 0 --long-run-steps 0"
for rep in 1 2; do
  for hb in 0 1; do
    for nf in 4 6; do
      HERRO_HOST_BUILD=$hb timeout 200 python bench.py $q --e2e-feeders $nf > $out/hb${hb}_f${nf}_$rep.json 2>> $out/err.txt < /dev/null
    done
  done
done
for hb in 0 1; do
  HERRO_HOST_BUILD=$hb HERRO_HOST_PROFILE=1 timeout 200 python bench.py $q --e2e-feeders 1 --e2e-jobs 8 > $out/prof_hb$hb.json 2> $out/prof_hb$hb.txt < /dev/null
  grep "herro_job_create" $out/prof_hb$hb.txt | tail -6 | cut -c1-200
  grep "cigar scan" $out/prof_hb$hb.txt | tail -3 | cut -c1-200
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads([x for x in open(f) if x.startswith("{")][-1]); e=d["end_to_end"]
        print(f.split("/")[-1], "value", round(d["value"]), "e2e", round(e["windows_per_s"]), "ratio", round(e["windows_per_s"]/d["value"],3), "prep/feeder", round(e["host_prepare_windows_per_s_per_feeder"]), "feeders", e["feeders_per_gpu"])
    except Exception as ex: print(f, ex)
PY
