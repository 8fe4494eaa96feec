#!/bin/bash
# encoder stack under the counters: MFMA pipe busy, wave-cycle split, waves per CU, L2 hit rate — for k_layers_q (HERRO_LAYERS_Q=1) and
# k_layers_p (=0).  usage: gpurun -- bash tools/pmc_layers.sh tag ["0 1"]
tag=${1:-rX}
out=gpurun_out/${tag}_pmc
mkdir -p $out
export TMPDIR=/tmp
cmd="python bench.py --no-cpu-baseline --self-check 0 --streams 1 --e2e-jobs 0 --strong-windows 0 --repeats 0 --settle 0 --steps 64 --warmup 32"
for v in ${2:-1 0}; do
HERRO_LAYERS_Q=$v timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $out/a$v -o a -- $cmd > $out/a$v.json 2> $out/a$v.err < /dev/null
HERRO_LAYERS_Q=$v timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $out/b$v -o b -- $cmd > $out/b$v.json 2> $out/b$v.err < /dev/null
for d in a$v b$v; do
  f=$(find $out/$d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_summary.py "$f" "${3:-k_layers}" > $out/$d.txt 2>&1; else tail -3 $out/$d.err; fi
  rm -rf $out/$d
done
echo "Q=$v"; cat $out/a$v.txt $out/b$v.txt | cut -c1-600
done
