#!/bin/bash
mkdir -p gpurun_out/r3f
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -6 > gpurun_out/r3f/tests.log; tail -3 gpurun_out/r3f/tests.log
timeout 150 python tools/large_window_cost.py 1024 > gpurun_out/r3f/large_window_cost.json 2> gpurun_out/r3f/large.err < /dev/null
cat gpurun_out/r3f/large_window_cost.json; tail -3 gpurun_out/r3f/large.err
