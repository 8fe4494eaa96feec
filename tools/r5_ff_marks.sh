#!/bin/bash
# round 5: finer phase marks inside the feed-forward loop of k_layers_p (HERRO_PROF_BUILD library), precision 4, 5 and 6   usage: gpurun --timeout 300 -- bash tools/r5_ff_marks.sh r5z3
tag=$1; out=gpurun_out/$tag; mkdir -p $out
q="--no-cpu-baseline --self-check 0 --e2e-jobs 0 --strong-windows 0 --repeats 0 --settle 0 --long-run-steps 0 --streams 1"
for p in 4 5 6; do
  HERRO_LIB=$PWD/herro_amd/libherro_amd_prof.so HERRO_PROF=1 timeout 100 python bench.py $q --precision $p 2>&1 | grep -E "^PROF k_layers" | tail -1 | cut -c1-600 > $out/phases_p$p.txt
  echo "p$p $(cat $out/phases_p$p.txt)"
done
