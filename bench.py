#!/usr/bin/env python
"""bench.py — throughput of the HERRO hot path (pileup featurisation + correction-model forward + consensus)
on synthetic overlap batches, one process per GPU.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of `--batch` (128) 4096-bp windows with 32 overlaps each
(BASELINE.json configs[2]): featurise the windows on the GPU, batch the windows with >= 1 informative position across
reads, run the model, decode the corrected bases on the device.  `value` counts exactly `--steps` steps with the inputs
(2-bit read store, window descriptors) resident in HBM before the timed region starts.  The same JSON line also carries
  end_to_end    the same work with herro_job_create (CIGAR text staged and scanned on the GPU, windowing, descriptor
                upload) and the D2H of the corrected bases INSIDE the timed region, fresh inputs every job, six feeder
                threads (contexts) per GPU; N = 1 only by default;
  roofline      the dominant kernel against its roof, durations from HIP events on the launch stream
                (roofline_next_kernels: the two behind it; roofline_featurize_group: the featurize kernels together);
  repeat_ms_per_step   three further passes of the same K steps (spread; `value` is always the first timed pass);
  cpu_baseline  the reference algorithm on the host cores (oracle feature generation + PyTorch-CPU twin), N = 1 only;
  self_check    windows of a timed job compared with the oracle after the timing (features bit-exact, FASTA identical).
  strong        after the weak measurement, ONE fixed job of --strong-windows windows through the sharded data path (rank 0
                ingests, the window work is scattered, the corrected reads are gathered: herro_amd/shard.py), with ranks_seen.
Multi-GPU: the path shards by target read.  `value` is the weak figure (every rank its own batches; the process group only
carries the barrier and the max-over-ranks time); the `strong` object in the same line times the scatter / gather path of
north_star on a fixed job; `--scaling strong` prints that path as a line of its own.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TF = 2500.0     # dense bf16 / f16 MFMA
W, N_OVL, WINS_PER_TARGET = 4096, 32, 4
DTYPE = {0: "f32", 1: "bf16x3", 2: "f32-valu", 3: "bf16x3", 4: "f16 (encoder proj / FF GEMMs: activation hi+lo)", 5: "f16", 6: "f16 (encoder proj / FF GEMMs: activation f16 + an e4m3 remainder term on the MX MFMA)",
         7: "f16 (encoder proj GEMM: activation hi+lo; FF single)", 8: "f16 (encoder FF GEMMs: activation hi+lo; proj single)"}
# MFMA products issued per algorithmic product in the encoder GEMMs (4: QKV one, proj / FF1 / FF2 two; 6: their second term on the MX MFMA at half an f16 product's pipe time;
# 7 / 8: the mixed tiers of round 6 — QKV 3 + proj 1 + FF 8 flop units per layer)
MFMA_TERMS = {1: 3, 3: 3, 4: 1.75, 5: 1, 6: 1.375, 7: 13 / 12, 8: 20 / 12}


def cpu_baseline(seed: int) -> dict:
    """Reference algorithm on the host: oracle restatement (features) + PyTorch-CPU twin (model) on a bounded sample of
    the same workload.  The reference runs the two stages on different threads (lib.rs:154-200), so the pipeline rate is
    the slower stage's.  Checker code is used ONLY here (and in self_check)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import concurrent.futures as cf
    import oracle_lib as O
    import model_ref as MR
    import torch
    from herro_amd import model_io, synth
    cores = synth.usable_cpus()   # = worker threads used below
    n_tgt = max(256, min(64 * cores, 1024))   # 1024 targets = 4096 windows on 16+ usable CPUs: ~0.5 s per pass, ~10 CPU-seconds
    sb = synth.generate_parallel(n_tgt, WINS_PER_TARGET * W, N_OVL, seed=seed, chunk=32)
    store = O.store_from_synth(sb)
    tasks = [O.target_alignments(sb, t) for t in range(sb.n_targets)]

    def feat(task):
        rid, rows, cigs = task
        return len(store.extract_features(rid, rows, cigs, W))   # released at once: the next task recycles the memory

    def feat_rate(workers, sample):
        with cf.ThreadPoolExecutor(workers) as ex:   # the oracle is a C++ library behind ctypes: the GIL is released
            list(ex.map(feat, sample[:2 * workers]))  # warm-up: threads started, store paged in, allocator arenas grown
            t0 = time.perf_counter()
            n = sum(ex.map(feat, sample))
            return n / (time.perf_counter() - t0)
    workers = max(1, min(cores, len(tasks)))
    feat_all = feat_rate(workers, tasks)
    feat_t4 = feat_rate(min(4, cores), tasks[:24])           # the reference's `-t 4` configuration
    # model: dense fp32 twin — what libtorch executes for the TorchScript graph — on ONE collated batch of `mb`
    # windows of one read group (a batch of 128 such windows takes minutes on the CPU; the batch size is stated)
    torch.set_num_threads(cores)
    _, raw = model_io.default_model_file(os.path.join(ROOT, "tests", "_cache"))
    twin = MR.build(raw, model_io.Hyper())
    # ONE collated batch of 128 windows across reads — the grouping of the GPU run (identical batch grouping):
    # the windows with informative rows of as many reads as it takes, padded to the batch's longest window with token 11 / quality 126 as
    # collate does (inference.rs:86-97).  The dense twin holds ~120 MB of activations per window, so the batch is EXECUTED in slices of 8
    # windows, each already padded to the batch's length: the windows of a batch do not interact (attention is per window), the numbers
    # and the arithmetic are those of one forward over the 128.
    MB, SLICE = 128, 8
    parts, k = [], 0
    while sum(len(p_["lens"]) for p_ in parts) < MB + WINS_PER_TARGET and k < len(tasks):
        r_ = store.extract_features(*tasks[k], W)
        nb_, bt_ = r_.collate(WINS_PER_TARGET, 0)
        if bt_ is not None and len(bt_["lens"]):
            parts.append(bt_)
        k += 1
    warm = parts[0]
    MR.run_batch(twin, warm["bases"], warm["quals"], warm["lens"], warm["indices"])   # warm-up: threads, allocator, oneDNN primitives
    parts = parts[1:]
    lmax = max(p_["bases"].shape[1] for p_ in parts)
    wins = []   # (bases [lmax,31], quals, n informative, indices)
    for p_ in parts:
        o = 0
        for i in range(len(p_["lens"])):
            n_i = int(p_["lens"][i])
            b_ = np.full((lmax, 31), 11, np.uint8); q_ = np.full((lmax, 31), 126, np.uint8)
            b_[:p_["bases"].shape[1]] = p_["bases"][i]; q_[:p_["quals"].shape[1]] = p_["quals"][i]
            wins.append((b_, q_, n_i, p_["indices"][o:o + n_i]))
            o += n_i
    wins = wins[:MB]
    mb = len(wins)
    t0 = time.perf_counter()
    for s0 in range(0, mb, SLICE):
        sl = wins[s0:s0 + SLICE]
        MR.run_batch(twin, np.stack([x[0] for x in sl]), np.stack([x[1] for x in sl]), np.array([x[2] for x in sl], np.int32),
                     np.concatenate([x[3] for x in sl]).astype(np.int32))
    model_s = time.perf_counter() - t0
    model_rate = mb / model_s
    n_model = mb
    return {"value": min(feat_all, model_rate), "unit": "windows/s", "cores": cores, "kind": "port",
            "feature_windows_per_s": feat_all, "feature_windows_per_s_4_threads": feat_t4, "model_windows_per_s": model_rate,
            "model_batch": mb, "model_batch_seconds": model_s, "model_batch_padded_length": int(lmax), "model_batch_executed_in_slices_of": SLICE,
            "sample": f"pipelined stages, rate of the slower one: oracle extract_features on {sb.n_targets * WINS_PER_TARGET} windows "
                      f"({workers} threads: {feat_all:.0f} win/s; 4 threads, the reference's -t 4: {feat_t4:.0f} win/s) | dense PyTorch-CPU fp32 "
                      f"twin of the assumed architecture, warmed, ONE cross-read batch of {mb} windows padded to {lmax} rows as collate pads it (the GPU run's grouping), executed in slices "
                      f"of {SLICE} windows to bound memory ({cores} threads: {model_s:.1f} s = {model_rate:.2f} win/s). "
                      "The reference itself runs the model on a GPU through libtorch; this is the same algorithm on the host cores "
                      f"(usable CPUs {cores} of {os.cpu_count()} hardware threads: affinity / cgroup quota)"}


def self_check(job, sb, targets, n_check: int, seed: int) -> dict:
    """After the timing: windows of a job that ran in the timed region against the oracle — pileup bit-exact, device FASTA
    identical to the oracle's consensus.rs restatement decoding the job's logits."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    rng = np.random.default_rng(seed)
    store = O.store_from_synth(sb)
    picks = sorted(rng.choice(len(targets), size=min(n_check, len(targets)), replace=False).tolist())
    n_win = 0
    for k in picks:
        t = targets[k]
        rid, rows, cigs = O.target_alignments(sb, t)
        res = store.extract_features(rid, rows, cigs, W)
        lg = []
        for wi in range(len(res)):
            ow, gw = res.window(wi), job.window(k * WINS_PER_TARGET + wi)
            ok = (np.array_equal(gw.bases, ow.bases) and np.array_equal(gw.quals, ow.quals) and gw.qids.tolist() == ow.qids.tolist()
                  and gw.sup_pos.tolist() == ow.sup_pos.tolist() and gw.sup_ins.tolist() == ow.sup_ins.tolist())
            if not ok:
                return {"ok": False, "failed": f"pileup of target {t} window {wi}"}
            if len(ow.sup_pos):
                lg.append(job.logits(k * WINS_PER_TARGET + wi)[1])
            n_win += 1
        want = res.consensus_fasta(np.concatenate(lg) if lg else np.zeros((0, 5), np.float32))
        if job.consensus_fasta(k, sb.read_name(rid)) != want:
            return {"ok": False, "failed": f"FASTA of target {t}"}
    return {"ok": True, "windows": n_win, "targets": len(picks), "what": "pileup bit-exact vs oracle, device FASTA == oracle consensus of the job's logits"}


T_PROCESS_START = time.time()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--pool", type=int, default=2, help="distinct synthetic jobs per stream cycled through (one job touches ~0.9 GB of read "
                                                        "store + ~1.2 GB of planes at the default group size: nothing of it survives in the 256 MB "
                                                        "Infinity Cache until the job runs again)")
    ap.add_argument("--group", type=int, default=32,
                    help="batches per launch group: the windows of GROUP consecutive steps are featurised and run "
                         "through the model in one set of kernel launches (each window keeps its own batch's padding)")
    ap.add_argument("--precision", type=int, default=None,
                    help="GEMM operand format (herro_set_precision); default = the tier herro_load_model's calibration chooses for the model "
                         "(the cheapest f16 tier within 5e-4 of the f32 mode on the calibration batch; reported as config.precision / precision_choice)")
    ap.add_argument("--streams", type=int, default=2,
                    help="independent contexts (HIP streams) per GPU, each driven by its own host thread, like the "
                         "reference's concurrent feature / inference threads per device (lib.rs:154-200)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--windows", type=int, default=0, help="--scaling strong: total windows of the fixed job (default steps * batch)")
    ap.add_argument("--strong-windows", type=int, default=100000,
                    help="size of the fixed job of the 'strong' leg that follows the weak measurement (the sharded data path; BASELINE configs[2..3]: "
                         "100000 windows — generated as copies of at most --strong-base-targets targets, every copy a target read of its own); 0: skip")
    ap.add_argument("--strong-base-targets", type=int, default=4200, help="targets the generator makes for the 'strong' leg (the job is copies of them)")
    ap.add_argument("--strong-ingest", choices=["local", "rank0"], default="local",
                    help="'local': every rank holds its own share of the parsed alignments and one all-to-all takes targets to their owners "
                         "(scales); 'rank0': rank 0 ingests everything and scatters the work (the literal north_star path)")
    ap.add_argument("--strict-exit", action="store_true", help="exit status 3 when the 'strong' leg failed, also on several GPUs (default there: 0, strong_ok = false)")
    ap.add_argument("--strong-timeout", type=float, default=420.0, help="seconds the 'strong' leg may take before the line goes out without it")
    ap.add_argument("--settle", type=float, default=0.2, help="seconds of untimed steps on top of --warmup before the timed pass")
    ap.add_argument("--repeats", type=int, default=3, help="further timed passes of the same K steps after the measured one (spread only)")
    ap.add_argument("--min-jobs", type=int, default=1,
                    help="launch grouping for short runs (steps < 2 x group): the timed steps are split into at least this many jobs. "
                         "Measured at --steps 20: one job of 2560 windows 0.96 M windows/s, two jobs of 1280 on two streams 0.87 M "
                         "(every kernel fills the GPU, so the second stream buys no overlap and the smaller launches pay larger tails)")
    ap.add_argument("--e2e-mode", choices=["serial", "producer"], default="serial",
                    help="end_to_end feeders: 'serial' = one thread per context (create k+1, then execute k); 'producer' = a second "
                         "thread per context builds jobs ahead")
    ap.add_argument("--e2e-feeders", type=int, default=6, help="feeder threads (one context each) of the end_to_end leg")
    ap.add_argument("--e2e-jobs", type=int, default=None, help="jobs per feeder thread in the end_to_end leg (default: 32 on one GPU — 8 distinct jobs per feeder, cycled —, 0 = skipped on several)")
    ap.add_argument("--self-check", type=int, default=6, help="targets compared with the oracle after the timing (0: skip)")
    ap.add_argument("--long-run-steps", type=int, default=None,
                    help="after a SHORT measurement (steps < 64, one GPU) the same device-resident leg is run once more with this many steps in a "
                         "child process and reported as 'long_run' in the same line (default 256; 0: skip)")
    ap.add_argument("--sustained", type=float, default=None,
                    help="seconds of the `sustained` leg: the device-resident loop repeated for at least this long, windows/s of its first and last half second and the "
                         "shader clock sampled every half second (default 2 on one GPU when steps >= 64 — a short run gets it from its long_run child; 0: skip)")
    ap.add_argument("--sensitivity", type=int, default=None,
                    help="1: `value` re-measured, one pass each, on data sets with more informative rows per window (p_snp 8e-3, 3e-2 next to the workload's 2e-3): "
                         "the throughput is roughly inversely proportional to that mean (default 1 on one GPU, 0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    # the ranks of a node share its usable CPUs (hardware threads, affinity, cgroup quota): each takes its share for the
    # library's host pool and the input generator, so that N ranks do not throttle each other
    from herro_amd import synth
    cpus_rank = max(2, synth.usable_cpus() // max(1, world))
    if world > 1:
        os.environ.setdefault("HERRO_HOST_THREADS", str(cpus_rank))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if args.e2e_jobs is None:   # the end_to_end leg is a single-GPU figure (6 feeder contexts per GPU would have 8 ranks fight for the host)
        # 32 jobs per feeder, 8 distinct ones cycled (round 6; 6 before): with 6 the timed region was ~60 ms, much of it the pipeline filling and draining — 1.06-1.45 M windows/s where, on the same
        # box, 12 jobs measure 1.42-1.74 M, 16 1.73-1.77 M, 24 1.88-1.93 M, 32 1.96 M, 48 1.91 M (profiles/r6_e2e_jobs_per_feeder.txt: the leg levels off near 0.8 x value);
        # the line says how many it ran (end_to_end.jobs_per_feeder)
        args.e2e_jobs = 32 if world == 1 else 0

    from herro_amd import api, model_io, synth
    path, _ = model_io.default_model_file(os.path.join(ROOT, "tests", "_cache"))
    precision_choice = {"how": "requested (--precision)"}
    if args.scaling == "strong":
        from herro_amd import shard
        return shard.bench_strong(args, rank, world, local)   # (its contexts keep the load-time choice when --precision is not given)
    if args.precision is None:   # the library's own choice for THIS model: every context is then set to it explicitly (all legs, child runs, the sharded path)
        c0 = api.Context(local)
        c0.load_model(path)
        args.precision = c0.precision()
        precision_choice = {"how": "calibrated by herro_load_model (cheapest f16 tier within 5e-4 of mode 0 on 256 pileup-shaped rows)",
                            "calibration_error": {str(m): c0.calibration_error(m) for m in (5, 7, 8, 4)}}
        c0.close()
    targets_per_step = args.batch // WINS_PER_TARGET
    # at least two launch groups per timed region when possible, so that featurize(k+1) can overlap infer(k)
    G = max(1, min(args.group, args.steps // max(1, args.min_jobs)))
    n_full, rem = divmod(args.steps, G)
    NS = max(1, min(args.streams, n_full)) if n_full else 1
    # distinct jobs per stream, cycled: never fewer than two, so that a short run (the driver's --steps 20 is ONE launch group)
    # does not time a re-run of the very job that warmed it up — successive passes rotate through the pool
    pool = max(2 if n_full else 1, min(args.pool, (n_full + NS - 1) // NS if n_full else 1))
    n_jobs = NS * pool
    tpj = G * targets_per_step                                  # targets per job
    NF = max(NS, args.e2e_feeders)                              # feeder threads (= contexts) of the end_to_end leg
    E2E_DISTINCT = 8   # distinct jobs per feeder of the end_to_end leg (cycled when it runs more: 8 x 6 x 31 MB of CIGAR text and ~100 MB of reads per job are far beyond any cache)
    n_e2e = min(args.e2e_jobs, E2E_DISTINCT) * NF if (args.e2e_jobs > 0 and n_full) else 0
    n_t = (n_jobs + n_e2e) * tpj + rem * targets_per_step

    def job_targets(i):
        return range(i * tpj, (i + 1) * tpj)

    def prepare(parallel: bool):
        """synthetic reads + alignments -> read stores in HBM + jobs (descriptors uploaded); outside the timed region"""
        gen = synth.generate_parallel if parallel else synth.generate   # parallel: chunks generated concurrently, merged
        kw_ = dict(workers=min(cpus_rank, 64)) if parallel else {}
        sb_ = gen(n_t, WINS_PER_TARGET * W, N_OVL, seed=synth.SEED + 2 + 1000 * rank, **kw_)
        ctxs_ = []
        for s_i in range(NS):
            c = api.Context(local)
            c.load_model(path)
            c.set_precision(args.precision)
            if ctxs_:
                c.share_reads(ctxs_[0])          # one read store per device (herro_share_reads), not one per context
            else:
                c.set_reads(sb_.seq, sb_.qual, sb_.off)
            ctxs_.append(c)
        jobs_ = [[api.job_from_synth(ctxs_[s_i], sb_, W, job_targets(s_i * pool + i)) for i in range(pool)] for s_i in range(NS)]
        rem_ = api.job_from_synth(ctxs_[0], sb_, W, range((n_jobs + n_e2e) * tpj, n_t)) if rem else None
        assert all(j.n_windows == G * args.batch for js in jobs_ for j in js)
        return sb_, ctxs_, jobs_, rem_

    try:
        sb, ctxs, jobs, rem_job = prepare(True)
    except Exception as e:  # pragma: no cover — input preparation only; the serial generator is the tested baseline
        print(f"bench: input preparation from parallel chunks failed ({e!r}); generating serially", file=sys.stderr)
        sb, ctxs, jobs, rem_job = prepare(False)
    ctx = ctxs[0]

    def run_job(j):
        j.featurize()
        j.infer(args.batch, 1)
        j.consensus()      # corrected bases stay in HBM (≈4 KB/window); only they would cross PCIe

    rot = [0]   # launch groups issued so far per stream: the pool is walked on across passes

    def run_steps(n_steps):
        """exactly n_steps batches of `batch` windows, launch groups dealt round-robin to the streams"""
        nf, r = divmod(n_steps, G)
        base = rot[0]
        rot[0] += (nf + NS - 1) // NS

        def worker(s_i):
            seq = [jobs[s_i][(base + i // NS) % pool] for i in range(s_i, nf, NS)]
            if pool < 2:
                for j in seq:
                    run_job(j)
            else:
                # software pipeline on one in-order stream: the featurize kernels of job k+1 are queued
                # before the host waits for job k's per-window counts (needed to plan its batches), so
                # the GPU works through featurize(k+1) while the host builds the descriptors of infer(k).
                # Same work per step as run_job, only the enqueue order differs.
                if seq:
                    seq[0].featurize()
                for k, j in enumerate(seq):
                    if k + 1 < len(seq):
                        seq[k + 1].featurize()
                    j.infer(args.batch, 1)
                    j.consensus()
            ctxs[s_i].synchronize()

        if NS == 1:
            worker(0)
        else:
            th = [threading.Thread(target=worker, args=(s_i,)) for s_i in range(NS)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if r:
            assert rem_job is not None and r == rem
            run_job(rem_job)
            ctx.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        for c in ctxs:
            c.synchronize()

    # the set-up above (data generation, job building) ran on every CPU the process may use; under a cgroup CPU quota the
    # current bandwidth period may be spent, and a throttled feeder thread would stall the timed region for the rest of
    # it (up to 100 ms).  Two periods of rest give the short, CPU-light timed region a fresh budget.
    time.sleep(0.25)
    # warm-up: at least W steps, and every pooled job at least once — a job's late buffers (logits, batch descriptors) are
    # allocated by its first herro_job_infer, which must not fall into the timed region
    for s_i in range(NS):
        for i in range(max(pool, (args.warmup + G * NS - 1) // (G * NS))):
            run_job(jobs[s_i][i % pool])
    # ... and at least `--settle` seconds of the same steps: W = 5 steps is 0.7 ms of GPU work, after which clocks and
    # caches are still moving (measured: the pass after a one-job warm-up runs 8 % slower than the following ones)
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < args.settle:
        run_steps(args.steps)
        for c in ctxs:
            c.synchronize()

    def timed_pass():
        barrier()
        t_ = time.perf_counter()
        run_steps(args.steps)
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()
        el_ = time.perf_counter() - t_
        barrier()
        return el_
    el = timed_pass()                                    # THE measurement: exactly K steps
    repeats = [timed_pass() for _ in range(max(0, args.repeats))]   # reported beside it (run-to-run spread), never used for `value`
    # ---- sustained: the same device-resident steps for seconds instead of milliseconds — is the figure above a transient of clocks and caches?
    sustained = None
    sus_s = args.sustained if args.sustained is not None else (2.0 if (world == 1 and args.steps >= 64) else 0.0)
    if sus_s > 0 and world == 1:
        marks, clocks = [], []
        t0_ = time.perf_counter()
        next_probe, done_w = 0.0, 0
        while True:
            run_steps(args.steps)
            for c in ctxs:
                c.synchronize()
            done_w += args.steps * args.batch
            now = time.perf_counter() - t0_
            marks.append((now, done_w))
            if now >= next_probe:
                try:
                    clocks.append((round(now, 3), round(ctx.clock_probe(), 1)))
                except Exception as e:   # a probe that fails costs the sample, not the leg
                    clocks.append((round(now, 3), repr(e)))
                next_probe = now + 0.5
            if now >= sus_s:
                break
        T_ = marks[-1][0]

        def rate(a, b):   # windows/s between the first pass boundaries at or behind a and b seconds
            ia = next((i for i, (t, _) in enumerate(marks) if t >= a), len(marks) - 1) if a > 0 else -1
            ib = next((i for i, (t, _) in enumerate(marks) if t >= b), len(marks) - 1)
            ta, wa_ = (0.0, 0) if ia < 0 else marks[ia]
            tb, wb_ = marks[ib]
            return (wb_ - wa_) / (tb - ta) if tb > ta else None
        mhz = [c_[1] for c_ in clocks if isinstance(c_[1], float)]
        sustained = {"seconds": T_, "steps": len(marks) * args.steps, "passes": len(marks), "windows_per_s": marks[-1][1] / T_,
                     "first_half_second_windows_per_s": rate(0.0, 0.5), "last_half_second_windows_per_s": rate(max(0.0, T_ - 0.5), T_),
                     "shader_clock_mhz": clocks, "shader_clock_mhz_min_max": [min(mhz), max(mhz)] if mhz else None,
                     "note": "the timed region of `value` repeated back to back (each pass synchronised); shader clock = s_memtime / s_memrealtime of a one-wave probe "
                             "queued behind a pass (herro_clock_probe)"}
    # ---- sensitivity: the workload's 2e-3 SNP rate gives ~15 informative rows per window; throughput falls with that mean (the transformer's share grows)
    sensitivity = None
    if (args.sensitivity if args.sensitivity is not None else 1) and world == 1:
        sensitivity = []
        s_steps = max(1, min(args.steps, 32))
        for p_snp in (8e-3, 3e-2):
            try:
                sb_s = synth.generate_parallel(s_steps * targets_per_step, WINS_PER_TARGET * W, N_OVL, seed=synth.SEED + 77, workers=min(cpus_rank, 64), p_snp=p_snp)
                c_s = api.Context(local)
                c_s.load_model(path)
                c_s.set_precision(args.precision)
                c_s.set_reads(sb_s.seq, sb_s.qual, sb_s.off)
                j_s = api.job_from_synth(c_s, sb_s, W, range(s_steps * targets_per_step))
                for _ in range(2):        # warm (arenas, late buffers)
                    j_s.featurize(); j_s.infer(args.batch, 1); j_s.consensus()
                c_s.synchronize()
                t_ = time.perf_counter()
                j_s.featurize(); j_s.infer(args.batch, 1); j_s.consensus()
                c_s.synchronize()
                e_ = time.perf_counter() - t_
                st_s = j_s.stats()
                sensitivity.append({"p_snp": p_snp, "mean_informative": st_s["sum_supported"] / j_s.n_windows, "windows_per_s": j_s.n_windows / e_,
                                    "ms_per_step": 1e3 * e_ / s_steps, "steps": s_steps, "windows": j_s.n_windows})
                j_s.close(); c_s.close()
            except Exception as e:   # never the measured line
                sensitivity.append({"p_snp": p_snp, "error": repr(e)})
    if world > 1:
        tt = torch.tensor([el], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())

    # ---- end to end: job creation from host alignments + D2H of the corrected bases inside the timed region, fresh
    # inputs for every job.  Per feeder thread: create(k+1) runs on the host while the GPU works on job k.
    e2e = None
    if n_e2e:
        per = args.e2e_jobs
        stats = [None] * NF
        # The device-resident leg is best on two streams (every kernel fills the GPU; more streams only add contention);
        # end to end a job also waits for its text to cross PCIe and for the scan's answer, so more jobs in flight pay:
        # measured 0.66 M windows/s with 2 feeders, 0.76 M with 3, 0.93 M with 4.  The extra contexts are made here.
        for _ in range(NF - len(ctxs)):
            c = api.Context(local)
            c.load_model(path)
            c.set_precision(args.precision)
            c.share_reads(ctxs[0])
            ctxs.append(c)
        prep = api.PreparedAlignments(sb)   # the parsed alignments of the data set, resident on the host (outside the timed region)
        if os.environ.get("HERRO_ZERO_COPY", "1") not in ("", "0"):
            prep.register(ctxs[0])          # ... its CIGAR blob pinned once: herro_job_create copies a job's texts up from where they are

        def feeder_serial(s_i, ids, timed):
            # one thread per context: create(k+1) on the host while the GPU works on job k
            c = ctxs[s_i]
            host, bases, prev = 0.0, 0, None
            for i in ids:
                t_h = time.perf_counter()
                j = prep.job(c, i * tpj, (i + 1) * tpj, W)
                host += time.perf_counter() - t_h
                j.featurize()
                if prev is not None:
                    prev.infer(args.batch, 1)
                    prev.consensus()
                    bases += prev.consensus_fetch()                                   # D2H of the corrected bases (synchronises)
                    prev.close()
                prev = j
            prev.infer(args.batch, 1)
            prev.consensus()
            bases += prev.consensus_fetch()
            prev.close()
            if timed:
                stats[s_i] = (host, bases)

        def feeder(s_i, ids, timed):
            if args.e2e_mode == "serial":
                return feeder_serial(s_i, ids, timed)
            # two threads per context: a producer builds jobs from the host alignments (CIGAR parse, windowing, one async
            # upload) ahead of time; this thread executes them.  Creation therefore overlaps both the GPU work and this
            # thread's waits (per-window counts, D2H of the corrected bases).
            import queue
            c = ctxs[s_i]
            q = queue.Queue(maxsize=1)   # at most four live jobs per context: executing, featurized, queued, being built
            host = [0.0]

            def producer():
                for i in ids:
                    t_h = time.perf_counter()
                    j = prep.job(c, i * tpj, (i + 1) * tpj, W)
                    host[0] += time.perf_counter() - t_h
                    q.put(j)
                q.put(None)
            pt = threading.Thread(target=producer)
            pt.start()
            bases = 0
            cur = q.get()
            cur.featurize()
            while cur is not None:
                nxt = q.get()
                if nxt is not None:
                    nxt.featurize()
                cur.infer(args.batch, 1)
                cur.consensus()
                bases += cur.consensus_fetch()                                        # D2H of the corrected bases (synchronises)
                cur.close()
                cur = nxt
            pt.join()
            if timed:
                stats[s_i] = (host[0], bases)

        def run_feeders(id_lists, timed):
            th = [threading.Thread(target=feeder, args=(s_i, id_lists[s_i], timed)) for s_i in range(NF)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            for c in ctxs:
                c.synchronize()
        # untimed pass over already-seen target ranges: a context holds up to four jobs at a time, and the first jobs of a
        # context pay for their arenas (hipMalloc of ~1.2 GB, page-locking ~70 MB each); a long-running host recycles them
        n_warm = 6
        run_feeders([[(s_i * pool + k) % n_jobs for k in range(n_warm)] for s_i in range(NF)], False)
        barrier()
        t1 = time.perf_counter()
        dist_j = min(per, E2E_DISTINCT)
        run_feeders([[n_jobs + s_i * dist_j + k % dist_j for k in range(per)] for s_i in range(NF)], True)
        el2 = time.perf_counter() - t1
        barrier()
        if world > 1:
            tt = torch.tensor([el2], device="cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el2 = float(tt.item())
        prep.unregister()
        n_w = per * NF * G * args.batch
        host_s = sum(s[0] for s in stats)
        e2e = {"windows_per_s": n_w * world / el2, "mbases_per_s": sum(s[1] for s in stats) * world / el2 / 1e6,
               "windows": n_w * world, "usable_cpus": synth.usable_cpus(), "jobs_per_feeder": per, "distinct_jobs_per_feeder": dist_j, "feeders_per_gpu": NF, "warmup_jobs_per_feeder": n_warm,
               "host_prepare_windows_per_s_per_feeder": n_w / NF / (host_s / NF) if host_s else None,
               "zero_copy_text": os.environ.get("HERRO_ZERO_COPY", "1") not in ("", "0"),
               "note": "herro_job_create from host alignments (CIGAR text copied up from the registered blob and scanned on the GPU, windows cut and descriptors written "
                       "on the device behind the scan — build_dev.hip, round 6; 64 bytes of totals and the window descriptors come back) + featurize + infer + consensus + "
                       "D2H of the corrected bases, all inside the timed region; "
                       "fresh inputs per job (a feeder cycles through its distinct jobs when it runs more than it has); the alignments are resident on the host as one parsed array (what the reference's reader thread "
                       "hands over, lib.rs:141-151); " + ("per context one thread builds jobs ahead, one executes them" if args.e2e_mode == "producer"
                                                     else "one feeder thread per context: create(k+1) runs on the host while the GPU works on job k"),
               "mode": args.e2e_mode}

    # ---- per-kernel durations with HIP events on the launch stream (separate pass, same jobs, single stream, so
    # that kernel durations are not inflated by the other stream's kernels)
    st = jobs[0][0].stats()
    ctx.timing_enable(True)
    ctx.timing_reset()
    n_timed = max(1, min(n_full, 4)) if n_full else 0
    for i in range(n_timed):
        run_job(jobs[0][i % pool])
    if not n_full:
        run_job(rem_job)
    ctx.synchronize()
    tm = ctx.timing()
    ctx.timing_enable(False)
    timed_steps = n_timed * G if n_full else rem

    check = None
    if rank == 0 and args.self_check > 0 and n_full:
        try:
            check = self_check(jobs[0][0], sb, list(job_targets(0)), args.self_check, 7)
        except Exception as e:   # the checker must never take the measurement down with it
            check = {"ok": None, "error": repr(e)}

    if rank == 0:
        total_windows = args.steps * args.batch * world
        kern = {k: {"ms_total": v[0], "calls": v[1], "avg_us": 1e3 * v[0] / max(v[1], 1)} for k, v in tm.items()}
        lean = os.environ.get("HERRO_FEATURIZE_PLANES", "0") in ("", "0")   # the library's default path (k_rows); HERRO_FEATURIZE_PLANES=1: the planes path (k_tokens), for the A/B
        feat_names = ["cols", "win", "layout", "rows", "tokens", "supgather", "rf_quals"]
        feat_ms = sum(tm[k][0] for k in feat_names if k in tm) / timed_steps
        model_ms = sum(v[0] for k, v in tm.items() if k not in feat_names) / timed_steps
        # ---- algorithmic work per launch (one launch = G steps = G*batch windows); DESIGN.md §4/§5
        per_job = {k: float(v) for k, v in st.items()}
        n_cols = 1 + N_OVL
        tokens = per_job["sum_supported"]
        D, FF, C1, C2, KW, NL = 256, 1024, 64, 128, 3, 4
        # What featurize has to move.  In: the 2-bit bases (1/5 of bases + qualities) and the binary ops.  Out, planes
        # path: the TOKEN planes (31 L' bytes per window) + the receptive-field records.  Out, lean path (round 5): no planes — the
        # receptive-field records (16 bytes per informative row and column, 5 quality bytes read for each), the decoder's votes (three
        # bit planes over the window's positions + a byte per insertion row) and the informative-row lists (8 bytes per row).
        n_launch_windows = G * args.batch
        rf_bytes = tokens * 31 * (16.0 + 5.0)
        vote_bytes = n_launch_windows * 3 * ((W + 31) // 32) * 4 + max(per_job["sum_len"] - n_launch_windows * W, 0.0) + 8.0 * tokens
        plane_bytes = per_job["out_bytes"] / 2.0
        feat_bytes = per_job["read_bytes"] / 5.0 + per_job["op_bytes"] + rf_bytes + (vote_bytes if lean else plane_bytes)
        alg = {  # name -> (bound, work per launch, unit)
            "tokens": ("hbm", per_job["out_bytes"] / 2.0, "B"),                                      # the token planes written
            "conv_fused": ("mfma", 2.0 * tokens * 31 * (KW * C1) * C2, "F"),
            "layers_fused": ("mfma", NL * (2.0 * tokens * D * 3 * D + 2.0 * tokens * D * D + 4.0 * tokens * D * FF), "F"),
            "cols": ("hbm", per_job["read_bytes"] / 5.0 + per_job["op_bytes"], "B"),              # 2-bit bases + ops read
            "rows": ("hbm", n_launch_windows * 30 * 3 * ((W + 31) // 32) * 4 + vote_bytes, "B"),  # the selected columns' planes read once more, votes + lists written
            "patch_conv1": ("hbm", tokens * 31 * KW * C1 * 4, "B"),                               # y1 hi/lo written
            "conv2_gemm": ("mfma", 2.0 * tokens * 31 * (KW * C1) * C2, "F"),
            "fc_gemm": ("mfma", 2.0 * tokens * (31 * C2) * D, "F"),
            "qkv_gemm": ("mfma", 2.0 * tokens * D * 3 * D, "F"),
            "proj_gemm": ("mfma", 2.0 * tokens * D * D, "F"),
            "ff1_gemm": ("mfma", 2.0 * tokens * D * FF, "F"),
            "ff2_gemm": ("mfma", 2.0 * tokens * FF * D, "F"),
        }
        tj = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")     # written by tools/pmc_traffic.py from a --pmc run
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("precision", 1) != args.precision:
                tj = None                                           # counters of another operand format: not this run's traffic

        def roof_of(k):
            avg_s = tm[k][0] / max(tm[k][1], 1) * 1e-3
            bound, work, _ = alg[k]
            # PMC bytes were collected per launch of tj["windows_per_launch"] windows; every kernel's traffic is proportional to
            # the windows it processes, so it is scaled to this run's launch size
            traffic = (tj["kernels"][k]["hbm_bytes_corrected"] / tj.get("windows_per_launch", tj["group"] * 128) * G * args.batch
                       if tj and k in tj.get("kernels", {}) else None)
            if bound == "hbm":
                return {"kernel": k, "bound": "hbm", "achieved": work / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": work / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                        "algorithmic_bytes_per_launch": work, "launch_us": avg_s * 1e6, "windows_per_launch": G * args.batch}
            terms = MFMA_TERMS.get(args.precision, 1)
            return {"kernel": k, "bound": "mfma", "achieved": work / avg_s / 1e12, "peak": MFMA_PEAK_TF,
                    "unit": "TFLOP/s", "frac": work / avg_s / 1e12 / MFMA_PEAK_TF, "traffic": traffic,
                    "algorithmic_flops_per_launch": work, "launch_us": avg_s * 1e6, "windows_per_launch": G * args.batch,
                    "note": f"algorithmic 2MNK flops; this precision issues {terms} MFMA product(s) per algorithmic product in the "
                            "encoder GEMMs (issued-MFMA fraction = frac x that)"}
        ranked = sorted((k for k in tm if k in alg), key=lambda k: -tm[k][0])
        roof = roof_of(ranked[0])                                   # the dominant kernel
        roof_next = [roof_of(k) for k in ranked[1:3]]               # the two behind it (the top two trade places between launch sizes)
        out = {
            "metric": "4096-bp windows corrected/sec at batch=128",
            "value": total_windows / el,
            "unit": "windows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps,
            "timed_region_s": el,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": DTYPE[args.precision],
            "data": "synthetic (SURVEY §8d generator, seed 0x48455252+2; random-init weights of the assumed architecture)",
            "config": {"workload": "synthetic windows, 4096 bp, 32 overlaps each, batch=128, 1xMI355X per rank "
                                   "(BASELINE configs[2])", "batch": args.batch, "window": W, "overlaps": N_OVL,
                       "mean_len": st["sum_len"] / (G * args.batch), "mean_informative": st["sum_supported"] / (G * args.batch),
                       "model_windows_per_batch": st["n_model_windows"] / G, "batches_per_launch_group": G,
                       "streams_per_gpu": NS, "distinct_windows_cycled": n_jobs * G * args.batch, "precision": args.precision, "precision_choice": precision_choice,
                       "total_windows": total_windows, "strong_leg_windows": args.strong_windows},
            "mbases_per_s": total_windows / el * W / 1e6,
            "roofline": roof,
            "roofline_featurize_group": {
                "kernels": feat_names, "path": "lean (no token planes)" if lean else "planes", "bound": "hbm", "achieved": feat_bytes / G / (feat_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": feat_bytes / G / (feat_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_step": feat_bytes / G, "ms_per_step": feat_ms,
                # the same time against what a design that writes the token planes has to move (rounds 3-4: 2-bit bases + ops in, planes + receptive
                # fields out): the lean path does not move those bytes — it removed them — so its own fraction above is the smaller number
                "planes_path_equivalent": {"algorithmic_bytes_per_step": (per_job["read_bytes"] / 5.0 + per_job["op_bytes"] + plane_bytes + tokens * 5 * 31 * 2.0) / G,
                                           "frac": (per_job["read_bytes"] / 5.0 + per_job["op_bytes"] + plane_bytes + tokens * 5 * 31 * 2.0) / G / (feat_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}},
            "roofline_next_kernels": roof_next, "repeat_ms_per_step": [r * 1e3 / args.steps for r in repeats], "stage_ms_per_step": {"featurize": feat_ms, "model": model_ms},
            "end_to_end": e2e,
            "sustained": sustained,
            "sensitivity": sensitivity,
            "self_check": check,
            "kernels": kern,
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(synth.SEED + 2)
    for j in [j for js in jobs for j in js] + ([rem_job] if rem_job else []):
        j.close()
    for c in ctxs:
        c.close()
    # ---- the sharded data path on ONE fixed job (every rank takes part; figures on rank 0, same JSON line)
    strong_failed = strong_hung = False
    if args.strong_windows > 0:
        from herro_amd import shard
        # The measured line must not be lost to the extra leg — neither to an exception nor to a hang (a rank that fails before a
        # send leaves its peers waiting in a receive for ever).  The leg runs on a worker thread under a deadline; a rank on which
        # it failed or ran out of time reports that (rank 0: in the line) and leaves without the collective shutdown.
        box = {}

        def leg():
            try:
                torch.cuda.set_device(local)   # the current device is per thread
                box["r"] = shard.strong_leg(args, rank, world, local, args.strong_windows, model_path=path)
            except BaseException as e:
                box["e"] = repr(e)
        th = threading.Thread(target=leg, daemon=True)
        th.start()
        th.join(args.strong_timeout)
        if th.is_alive():
            strong, strong_failed, strong_hung = {"error": f"no result within --strong-timeout {args.strong_timeout:.0f} s"}, True, True
        elif "e" in box:
            strong, strong_failed = {"error": box["e"]}, True
        else:
            strong = box.get("r")
        if rank == 0:
            out["strong"] = strong
            if isinstance(strong, dict) and "windows_per_s" in strong and out.get("end_to_end"):
                strong["vs_end_to_end"] = strong["windows_per_s"] / (out["end_to_end"]["windows_per_s"] or 1.0)
    # ---- a short run's timed region is a millisecond or two: the default-size figure of the same leg goes into the same line
    lr_steps = 256 if args.long_run_steps is None else args.long_run_steps
    if rank == 0 and world == 1 and lr_steps > 0 and args.steps < 64 and time.time() - T_PROCESS_START > 240:
        out["long_run"] = {"skipped": "the legs above took more than 240 s: the extra run is left out so that the line goes out in time"}
    elif rank == 0 and world == 1 and lr_steps > 0 and args.steps < 64:
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(lr_steps), "--warmup", "8", "--no-cpu-baseline", "--self-check", "0", "--e2e-jobs", "0",
               "--strong-windows", "0", "--repeats", "1", "--long-run-steps", "0", "--precision", str(args.precision), "--batch", str(args.batch),
               "--sustained", str(2.0 if args.sustained is None else args.sustained), "--sensitivity", "0"]
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=150)
            d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
            out["long_run"] = {k: d[k] for k in ("steps", "warmup", "value", "ms_per_step", "timed_region_s", "repeat_ms_per_step")}
            out["long_run"]["roofline_frac"] = d["roofline"]["frac"]
            out["long_run"]["streams_per_gpu"] = d["config"]["streams_per_gpu"]
            if d.get("sustained") and not out.get("sustained"):
                out["sustained"] = dict(d["sustained"], measured_in="the long_run child (256-step passes, two streams)")
        except Exception as e:   # never the measured line
            out["long_run"] = {"error": repr(e)}
    if rank == 0:
        out["strong_ok"] = (not strong_failed) if args.strong_windows > 0 else None
        print(json.dumps(out), flush=True)
    if strong_failed:
        # The measured line is out (with strong_ok = false).  On one GPU a broken sharded path also shows in the exit status (3); on
        # several — where this leg's collectives have never met real hardware from here — the status stays 0 unless --strict-exit, so
        # that a failure of the EXTRA leg cannot cost a driver the weak-scaling line it came for.  Peers may be stuck in a collective
        # of the failed leg, or this process's own worker thread inside a HIP call: leave without destroy_process_group and without
        # tearing the runtime down under it.
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(3 if (world == 1 or args.strict_exit) else 0)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
