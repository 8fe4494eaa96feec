import time, sys, os
sys.path.insert(0, os.getcwd())
from herro_amd import api, synth, model_io
sb = synth.generate(256, 4*4096, 32, seed=7)
ctx = api.Context(0)
ctx.set_reads(sb.seq, sb.qual, sb.off)
for rep in range(3):
    t = time.perf_counter(); job = api.job_from_synth(ctx, sb, 4096); dt = time.perf_counter() - t
    print(f"job_create: {job.n_windows} windows in {dt*1e3:.1f} ms -> {job.n_windows/dt:.0f} windows/s host ({dt/job.n_windows*1e6:.1f} us/window)")
    job.close()
