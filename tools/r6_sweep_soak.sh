#!/bin/bash
# round 6: the two seeded sweeps (featurizer vs oracle cell by cell; lean path vs planes path incl. logits bit for bit and FASTA) over trial numbers the suite does not run.
# usage: gpurun --timeout 1200 -- bash tools/r6_sweep_soak.sh tag first count
tag=${1:-r6soak}; first=${2:-100}; count=${3:-200}; out=gpurun_out/$tag; mkdir -p $out
export HERRO_SWEEP_FIRST=$first HERRO_SWEEP_TRIALS=$count
( time timeout 1100 python -m pytest tests/test_gpu_features.py tests/test_gpu_lean.py -q -m gpu -k "random_configurations" -p no:cacheprovider 2>&1 | grep -v Warning | tail -25 ) > $out/soak.log 2>&1
tail -8 $out/soak.log
