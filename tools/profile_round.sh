#!/bin/bash
# Round profile on the GPU box: rocprofv3 per-kernel stats of the bench command (one stream, so kernel durations are not
# inflated by a second stream) + the bench lines to be committed under profiles/.  usage: gpurun -- bash tools/profile_round.sh r2z
tag=${1:-rX}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
( cd /tmp && true )
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o $tag -- \
  python bench.py --no-cpu-baseline --self-check 0 --streams 1 --e2e-jobs 2 --e2e-feeders 2 --repeats 0 > $out/bench_streams1.json 2> $out/prof.err < /dev/null
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $out/kernel_stats.csv; fi
rm -rf $out/prof
timeout 400 python bench.py > $out/bench.json 2> $out/bench.err < /dev/null
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_args.json 2>> $out/bench.err < /dev/null
head -20 $out/kernel_stats.csv 2>/dev/null | cut -c1-140
