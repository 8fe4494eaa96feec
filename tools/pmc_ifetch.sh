#!/bin/bash
# instruction-fetch side of the model kernels (k_layers_p is 64 KB of code, the instruction cache 64 KB per CU pair):
# SQC instruction-cache requests / misses and the SQ's instruction-fetch waits.  usage: gpurun -- bash tools/pmc_ifetch.sh r3g
tag=${1:-rX}
out=gpurun_out/${tag}_if
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o -i "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAVE_CYCLES\|SQ_BUSY_CU_CYCLES\|SQ_VALU_MFMA_BUSY_CYCLES\|SQ_INSTS_MFMA\|SQ_INSTS_VALU_MFMA[A-Z_0-9]*" | sort -u > $out/avail.txt
cat $out/avail.txt | tr '\n' ' '
cmd="python bench.py --no-cpu-baseline --self-check 0 --streams 1 --e2e-jobs 0 --strong-windows 0 --repeats 0 --settle 0 --steps 64 --warmup 32"
timeout 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/a -o a -- $cmd > $out/a.json 2> $out/a.err < /dev/null
f=$(find $out/a -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then python tools/pmc_summary.py "$f" "k_" > $out/a.txt 2>&1; fi
rm -rf $out/a
cat $out/a.txt | cut -c1-400; tail -3 $out/a.err
