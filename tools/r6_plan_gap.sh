#!/bin/bash
# round 6: what the single job of the driver's 20-step run pays for the batch plan — host time of herro_job_infer with the device idle, and the fused gather forced on / off
# (A/B switches exist only in the timer build; HERRO_PROF unset: no phase timers run).   usage: gpurun --timeout 600 -- bash tools/r6_plan_gap.sh tag
tag=${1:-r6pg}; out=gpurun_out/$tag; mkdir -p $out
timeout 200 python tools/r6_infer_host_time.py 2>&1 | grep -v Warn | tee $out/infer_host_time.txt
q="--no-cpu-baseline --self-check 0 --e2e-jobs 0 --strong-windows 0 --long-run-steps 0 --sustained 0 --sensitivity 0 --steps 20 --warmup 5"
for rep in 1 2; do
  for f in -1 1 0; do
    HERRO_LIB=$PWD/herro_amd/libherro_amd_prof.so HERRO_RF_FUSED=$f timeout 200 python bench.py $q > $out/fused${f}_$rep.log 2>&1
    python - <<PY
import json
j=[l for l in open("$out/fused${f}_$rep.log") if l.startswith("{")]
if j:
    d=json.loads(j[-1]); print("HERRO_RF_FUSED=$f rep$rep value", round(d["value"]), "repeats", [round(128/x*1e3) for x in d.get("repeat_ms_per_step",[])], {k:round(v["avg_us"]) for k,v in d["kernels"].items()})
else: print("HERRO_RF_FUSED=$f rep$rep: no line")
PY
  done
done 2>&1 | tee $out/summary.txt
