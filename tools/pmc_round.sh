#!/bin/bash
# HBM traffic per launch for every kernel of the bench command (device-resident leg, one stream): two separate counter
# passes (FETCH_SIZE, WRITE_SIZE), summarised into profiles-style traffic.json.  usage: gpurun -- bash tools/pmc_round.sh r2z
tag=${1:-rX}
out=gpurun_out/${tag}_pmc
mkdir -p $out
export TMPDIR=/tmp
cmd="python bench.py --no-cpu-baseline --self-check 0 --streams 1 --e2e-jobs 0 --repeats 0 --settle 0 --steps 96 --warmup 32"
timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -o f -- $cmd > $out/f.json 2> $out/f.err < /dev/null
timeout 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -o w -- $cmd > $out/w.json 2> $out/w.err < /dev/null
ff=$(find $out/f -name "*counter_collection.csv" | head -1)
fw=$(find $out/w -name "*counter_collection.csv" | head -1)
if [ -n "$ff" ] && [ -n "$fw" ]; then python tools/pmc_traffic.py "$ff" "$fw" 32 $out/traffic.json 4 > $out/summary.txt 2>&1 < /dev/null; fi
rm -rf $out/f $out/w
cat $out/summary.txt 2>/dev/null | head -20
