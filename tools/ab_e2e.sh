# A/B of the end_to_end leg on the GPU box: feeder modes x host pool sizes, with the cgroup throttling counters around each run
# (gpurun -- bash tools/ab_e2e.sh; results under gpurun_out/r2o).
mkdir -p gpurun_out/r2o
thr() { grep -E "nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; }
HOSTPREP_THREADS=1,4,8,16,32,64 python tools/hostprep_cpu.py 1024 > gpurun_out/r2o/hostprep.log 2>&1
run() { name=$1; shift; echo "== $name before: $(thr)" >> gpurun_out/r2o/thr.log; env "$@" python bench.py --no-cpu-baseline --self-check 0 --e2e-jobs 8 --e2e-mode $MODE > gpurun_out/r2o/$name.json 2> gpurun_out/r2o/$name.err; echo "== $name after: $(thr)" >> gpurun_out/r2o/thr.log; }
MODE=serial run serial_def A=1
MODE=producer run producer_def A=1
MODE=serial run serial_t8 HERRO_HOST_THREADS=8
MODE=producer run producer_t8 HERRO_HOST_THREADS=8
MODE=serial run serial_t64 HERRO_HOST_THREADS=64
cat gpurun_out/r2o/thr.log; tail -8 gpurun_out/r2o/hostprep.log
