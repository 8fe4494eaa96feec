#!/bin/bash
# SQ counters per kernel for the device-resident bench leg (one stream), two counter passes.  usage: gpurun -- bash tools/pmc_sq.sh r3a [filter]
tag=${1:-rX}
out=gpurun_out/${tag}_sq
mkdir -p $out
export TMPDIR=/tmp
cmd="python bench.py --no-cpu-baseline --self-check 0 --streams 1 --e2e-jobs 0 --repeats 0 --settle 0 --steps 40 --warmup 20"
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/a -o a -- $cmd > $out/a.json 2> $out/a.err < /dev/null
timeout 240 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --output-format csv -d $out/b -o b -- $cmd > $out/b.json 2> $out/b.err < /dev/null
for d in a b; do
  f=$(find $out/$d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" "${2:-k_}" > $out/$d.txt 2>&1; fi
  rm -rf $out/$d
done
cat $out/a.txt $out/b.txt | cut -c1-400
