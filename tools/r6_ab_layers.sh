#!/bin/bash
# round 6: k_layers_p before / after on one box — the round-5 library against this tree's (both timer builds), precisions 4 and 5, one stream; kernel time from HIP events + phase cycles.
# usage: gpurun --timeout 900 -- bash tools/r6_ab_layers.sh tag [more libs]
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
q="--no-cpu-baseline --self-check 0 --e2e-jobs 0 --strong-windows 0 --repeats 0 --settle 0 --long-run-steps 0 --streams 1"
for rep in 1 2; do
for lib in ${@:-libherro_amd_r5prof.so libherro_amd_prof.so}; do
  for p in 4 5; do
    HERRO_FORCE_PRECISION=1 HERRO_LIB=$PWD/herro_amd/$lib HERRO_PROF=1 timeout 120 python bench.py $q --precision $p > $out/${lib}_p${p}_$rep.log 2>&1
    python - <<PY
import json
L=[l for l in open("$out/${lib}_p${p}_$rep.log")]
j=[l for l in L if l.startswith("{")]
ph=[l for l in L if l.startswith("PROF k_layers")]
if j:
    d=json.loads(j[-1]); print("$lib p$p rep$rep", round(d["value"]), {k:round(v["avg_us"]) for k,v in d["kernels"].items() if k in ("layers_fused","conv_fused","fc_gemm")})
if ph: print("   ", ph[-1].strip()[60:460])
PY
  done
done
done 2>&1 | tee $out/summary.txt
