#!/bin/bash
# Everything the round's record needs from one GPU box, most important first: parity tests, the bench lines + rocprofv3 kernel
# stats (tools/profile_round.sh), the PMC traffic passes (tools/pmc_round.sh), then the A/B of the tile packing.
# usage: gpurun --timeout 1000 -- bash tools/final_round.sh r3d
tag=${1:-rX}
mkdir -p gpurun_out/$tag
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -15 > gpurun_out/$tag/tests.log
cat gpurun_out/$tag/tests.log | tail -5
bash tools/profile_round.sh $tag > gpurun_out/$tag/profile_round.log 2>&1
bash tools/pmc_round.sh $tag > gpurun_out/$tag/pmc_round.log 2>&1
q="--no-cpu-baseline --self-check 0 --e2e-jobs 0 --strong-windows 0 --repeats 1"
HERRO_TILE_PACK=0 timeout 120 python bench.py $q > gpurun_out/$tag/ab_pack0.json 2> /dev/null < /dev/null
HERRO_TILE_PACK=0 timeout 120 python bench.py $q --steps 20 --warmup 5 > gpurun_out/$tag/ab_pack0_driver.json 2> /dev/null < /dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/$tag/*.json")):
    try:
        l=[x for x in open(f) if x.startswith("{")]
        d=json.loads(l[-1])
        if "value" in d:
            print(f.split("/")[-1], round(d["value"]), d["ms_per_step"], {k:round(v["avg_us"]) for k,v in d.get("kernels",{}).items()}, (d.get("end_to_end") or {}).get("windows_per_s"), (d.get("strong") or {}).get("windows_per_s"), (d.get("self_check") or {}).get("ok"))
    except Exception as e:
        print(f, "unreadable", e)
PY
