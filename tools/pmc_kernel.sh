#!/bin/bash
# SQ / LDS / MFMA counters of the kernels matching a name filter, device-resident bench leg (one stream).
# usage: gpurun -- bash tools/pmc_kernel.sh tag filter [env assignments]
tag=$1; flt=$2; shift 2
out=gpurun_out/${tag}_pmc
mkdir -p $out
export TMPDIR=/tmp
cmd="python bench.py --no-cpu-baseline --self-check 0 --streams 1 --e2e-jobs 0 --strong-windows 0 --repeats 0 --settle 0 --steps 64 --warmup 32"
env "$@" timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $out/a -o a -- $cmd > $out/a.json 2> $out/a.err < /dev/null
env "$@" timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $out/b -o b -- $cmd > $out/b.json 2> $out/b.err < /dev/null
for d in a b; do
  f=$(find $out/$d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" "$flt" > $out/$d.txt 2>&1; else tail -3 $out/$d.err; fi
  rm -rf $out/$d
done
cat $out/a.txt $out/b.txt | cut -c1-700
