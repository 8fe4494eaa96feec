#!/bin/bash
# rocprofv3 per-kernel stats of the device-resident bench leg alone (one stream, no end_to_end leg): the averages the
# bench line's HIP-event kernel times must agree with.  usage: gpurun -- bash tools/profile_kernels.sh r2z
tag=${1:-rX}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o $tag -- \
  python bench.py --no-cpu-baseline --self-check 0 --streams 1 --e2e-jobs 0 --repeats 0 --settle 0 > $out/bench_streams1.json 2> $out/prof.err < /dev/null
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $out/kernel_stats.csv; fi
rm -rf $out/prof
head -8 $out/kernel_stats.csv 2>/dev/null | cut -c1-140
