"""herro_job_create throughput (host CIGAR parse + windowing + descriptor upload) vs HERRO_HOST_THREADS.
Needs a device (the job's buffers are allocated and uploaded).  usage: python tools/hostrate.py [n_targets]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
from herro_amd import api, synth

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sb = synth.generate(nt, 4 * 4096, 32, seed=7)
ctx = api.Context(0)
ctx.set_reads(sb.seq, sb.qual, sb.off)
for th in (1, 8, 32, 64, 128):
    os.environ["HERRO_HOST_THREADS"] = str(th)
    best = 1e9
    for rep in range(3):
        t = time.perf_counter(); job = api.job_from_synth(ctx, sb, 4096); dt = time.perf_counter() - t
        best = min(best, dt); n = job.n_windows; job.close()
    print(f"threads {th:3d}: {n} windows in {best*1e3:7.1f} ms -> {n/best/1e3:7.1f} k windows/s ({best/n*1e6:5.1f} us/window)")
