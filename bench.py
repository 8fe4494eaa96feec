#!/usr/bin/env python
"""bench.py — throughput of the HERRO hot path (pileup featurisation + correction-model forward)
on synthetic overlap batches, one process per GPU.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of `--batch` (128) 4096-bp windows with 32 overlaps
each (BASELINE.json configs[2]): featurise the windows' reads on the GPU, batch the windows with >=1
informative position across reads, run the model.  Inputs (2-bit read store, window descriptors) are
resident in HBM before the timed region.  Weak scaling: every rank processes its own batches; there
is no data-path collective (windows are independent, SURVEY.md §8 e).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0


def cpu_baseline(seed: int) -> dict:
    """Reference algorithm on the host: oracle restatement (features) + PyTorch-CPU twin (model),
    on a bounded sample of the same workload.  Checker code is used ONLY here."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib as O
    import model_ref as MR
    import torch
    from herro_amd import model_io, synth
    import concurrent.futures as cf
    cores = os.cpu_count() or 1
    # feature generation: one target read per task on all host cores, like the reference's feature threads
    # (lib.rs:159-187); the sample is sized to keep every core busy for a few tasks
    n_tgt = max(24, min(4 * cores, 1024))
    sb = synth.generate_parallel(n_tgt, 4 * 4096, 32, seed=seed, chunk=32)
    store = O.store_from_synth(sb)
    tasks = [O.target_alignments(sb, t) for t in range(sb.n_targets)]
    workers = max(1, min(cores, len(tasks)))

    def feat(task):
        rid, rows, cigs = task
        return len(store.extract_features(rid, rows, cigs, 4096))   # result released at once: its memory is recycled by the next task
    with cf.ThreadPoolExecutor(workers) as ex:      # the oracle is a C++ library behind ctypes: the GIL is released
        list(ex.map(feat, tasks[:2 * workers]))     # warm-up: threads started, read store paged in, allocator arenas grown
        t0 = time.perf_counter()
        n_win = sum(ex.map(feat, tasks))
        t_feat = time.perf_counter() - t0
    res = [store.extract_features(*tasks[0], 4096)]   # one read's windows for the model leg below
    # model: dense twin on one read's windows (reference grouping), all host cores
    torch.set_num_threads(cores)
    _, raw = model_io.default_model_file(os.path.join(ROOT, "tests", "_cache"))
    twin = MR.build(raw, model_io.Hyper())
    nb, bt = res[0].collate(4, 0)
    t0 = time.perf_counter()
    MR.run_batch(twin, bt["bases"], bt["quals"], bt["lens"], bt["indices"])
    t_model = time.perf_counter() - t0
    n_mwin = len(bt["lens"])
    per_win = t_feat / n_win + t_model / n_mwin
    return {"value": 1.0 / per_win, "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"oracle extract_features on {n_win} windows ({workers} threads, {n_win / t_feat:.1f} win/s) + "
                      f"dense PyTorch-CPU fp32 twin on {n_mwin} windows ({cores} threads, {n_mwin / t_model:.2f} win/s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--pool", type=int, default=2, help="distinct synthetic jobs cycled through")
    ap.add_argument("--group", type=int, default=32,
                    help="batches per launch group: the windows of GROUP consecutive steps are featurised and run "
                         "through the model in one set of kernel launches (each window keeps its own batch's padding)")
    ap.add_argument("--precision", type=int, default=None,
                    help="GEMM operand format (herro_set_precision); default = herro_amd.api.DEFAULT_PRECISION, the mode the "
                         "end-to-end parity test holds to the 1e-3 logits contract")
    ap.add_argument("--streams", type=int, default=2,
                    help="independent contexts (HIP streams) per GPU, each driven by its own host thread, like the "
                         "reference's concurrent feature / inference threads per device (lib.rs:154-200)")
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from herro_amd import api, model_io, synth
    if args.precision is None:
        args.precision = api.DEFAULT_PRECISION
    W, n_ovl = 4096, 32
    targets_per_step = args.batch // 4
    path, _ = model_io.default_model_file(os.path.join(ROOT, "tests", "_cache"))
    # at least two launch groups per timed region when possible, so that featurize(k+1) can overlap infer(k)
    G = max(1, min(args.group, args.steps // 2 if args.steps >= 2 else 1))
    n_full, rem = divmod(args.steps, G)
    NS = max(1, min(args.streams, n_full)) if n_full else 1
    pool = max(1, min(args.pool, (n_full + NS - 1) // NS if n_full else 1))
    n_jobs = NS * pool
    n_t = n_jobs * G * targets_per_step + rem * targets_per_step
    def prepare(parallel: bool):
        """synthetic reads + alignments -> read stores in HBM + jobs (descriptors uploaded); outside the timed region"""
        gen = synth.generate_parallel if parallel else synth.generate   # parallel: chunks generated concurrently, merged
        sb_ = gen(n_t, 4 * W, n_ovl, seed=synth.SEED + 2 + 1000 * rank)
        ctxs_ = []
        for s_i in range(NS):
            c = api.Context(local)
            c.load_model(path)
            c.set_precision(args.precision)
            c.set_reads(sb_.seq, sb_.qual, sb_.off)
            ctxs_.append(c)
        t0 = time.perf_counter()
        jobs_ = [[api.job_from_synth(ctxs_[s_i], sb_, W, range((s_i * pool + i) * G * targets_per_step,
                                                                (s_i * pool + i + 1) * G * targets_per_step))
                  for i in range(pool)] for s_i in range(NS)]
        dt = time.perf_counter() - t0
        rem_ = api.job_from_synth(ctxs_[0], sb_, W, range(n_jobs * G * targets_per_step, n_t)) if rem else None
        assert all(j.n_windows == G * args.batch for js in jobs_ for j in js)
        return sb_, ctxs_, jobs_, rem_, dt

    try:
        sb, ctxs, jobs, rem_job, host_prepare_s = prepare(True)
    except Exception as e:  # pragma: no cover — input preparation only; the serial generator is the tested baseline
        print(f"bench: input preparation from parallel chunks failed ({e!r}); generating serially", file=sys.stderr)
        sb, ctxs, jobs, rem_job, host_prepare_s = prepare(False)
    ctx = ctxs[0]

    def run_job(j):
        j.featurize()
        j.infer(args.batch, 1)
        j.consensus()      # corrected bases stay in HBM (≈4 KB/window); only they would cross PCIe

    import threading

    def run_steps(n_steps):
        """exactly n_steps batches of `batch` windows, launch groups dealt round-robin to the streams"""
        nf, r = divmod(n_steps, G)

        def worker(s_i):
            seq = [jobs[s_i][(i // NS) % pool] for i in range(s_i, nf, NS)]
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

    for s_i in range(NS):
        for i in range(max(1, (args.warmup + G * NS - 1) // (G * NS))):
            run_job(jobs[s_i][i % pool])
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    for c in ctxs:
        c.synchronize()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    barrier()
    if world > 1:
        tt = torch.tensor([el], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())

    # ---- per-kernel durations with HIP events on the launch stream (second pass, same steps)
    # (single stream, so that kernel durations are not inflated by the other stream's kernels)
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

    if rank == 0:
        total_windows = args.steps * args.batch * world
        kern = {k: {"ms_total": v[0], "calls": v[1], "avg_us": 1e3 * v[0] / max(v[1], 1)} for k, v in tm.items()}
        feat_names = ["ow_stats", "win_rank", "pass1_pos", "select_layout", "tile_plan", "final_tiles", "sup_compact", "rf_quals"]
        feat_ms = sum(tm[k][0] for k in feat_names if k in tm) / timed_steps
        model_ms = sum(v[0] for k, v in tm.items() if k not in feat_names) / timed_steps
        # ---- algorithmic work per launch (one launch = G steps = G*batch windows); DESIGN.md §4/§5
        per_job = {k: float(v) for k, v in st.items()}
        n_cols = 1 + n_ovl
        tokens = per_job["sum_supported"]
        D, FF, C1, C2, KW = 256, 1024, 64, 128, 3
        # SURVEY §8 d, minus what this design never moves: featurize writes the TOKEN planes only (31 L' bytes per
        # window) and reads the 2-bit bases (1/5 of bases + qualities); the qualities are touched only inside the
        # model's receptive fields (rf_quals: 5 rows x 31 columns per informative row, read + written)
        NL = 4
        rf_bytes = tokens * 5 * 31 * 2.0
        feat_bytes = per_job["read_bytes"] / 5.0 + per_job["op_bytes"] + per_job["out_bytes"] / 2.0 + rf_bytes
        alg = {  # name -> (bound, work per launch, unit)
            "final_tiles": ("hbm", per_job["out_bytes"] / 2.0 + per_job["read_bytes"] / 5.0 * 31.0 / n_cols, "B"),
            "conv_fused": ("mfma", 2.0 * tokens * 31 * (KW * C1) * C2, "F"),
            "layers_fused": ("mfma", NL * (2.0 * tokens * D * 3 * D + 2.0 * tokens * D * D + 4.0 * tokens * D * FF), "F"),
            "ow_stats": ("hbm", per_job["read_bytes"] / 5.0 + per_job["op_bytes"], "B"),          # 2-bit only
            "pass1_pos": ("hbm", per_job["read_bytes"] / 5.0, "B"),
            "patch_conv1": ("hbm", tokens * 31 * KW * C1 * 4, "B"),                               # y1 hi/lo written
            "conv2_gemm": ("mfma", 2.0 * tokens * 31 * (KW * C1) * C2, "F"),
            "fc_gemm": ("mfma", 2.0 * tokens * (31 * C2) * D, "F"),
            "qkv_gemm": ("mfma", 2.0 * tokens * D * 3 * D, "F"),
            "proj_gemm": ("mfma", 2.0 * tokens * D * D, "F"),
            "ff1_gemm": ("mfma", 2.0 * tokens * D * FF, "F"),
            "ff2_gemm": ("mfma", 2.0 * tokens * FF * D, "F"),
        }
        dom = max((k for k in tm if k in alg), key=lambda k: tm[k][0])
        dom_avg_s = tm[dom][0] / max(tm[dom][1], 1) * 1e-3
        bound, work, _ = alg[dom]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")     # written by tools/pmc_traffic.py from a --pmc run
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if dom in tj.get("kernels", {}) and tj.get("group") == G:
                traffic = tj["kernels"][dom]["hbm_bytes_corrected"]
        if bound == "hbm":
            roof = {"kernel": dom, "bound": "hbm", "achieved": work / dom_avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": work / dom_avg_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": work, "launch_us": dom_avg_s * 1e6, "windows_per_launch": G * args.batch}
        else:
            roof = {"kernel": dom, "bound": "mfma", "achieved": work / dom_avg_s / 1e12, "peak": MFMA_BF16_PEAK_TF,
                    "unit": "TFLOP/s", "frac": work / dom_avg_s / 1e12 / MFMA_BF16_PEAK_TF, "traffic": traffic,
                    "algorithmic_flops_per_launch": work, "launch_us": dom_avg_s * 1e6,
                    "note": "algorithmic 2MNK flops; the bf16x3 split issues 3x that many MFMA flops"}
        out = {
            "metric": "4096-bp windows corrected/sec at batch=128",
            "value": total_windows / el,
            "unit": "windows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {0: "f32", 1: "bf16x3", 2: "f32-valu", 3: "bf16x3", 4: "f16 (activation hi+lo in the encoder GEMMs)", 5: "f16"}[args.precision],
            "data": "synthetic (SURVEY §8d generator, seed 0x48455252+2; random-init weights of the assumed architecture)",
            "config": {"workload": "synthetic windows, 4096 bp, 32 overlaps each, batch=128, 1xMI355X per rank "
                                   "(BASELINE configs[2])", "batch": args.batch, "window": W, "overlaps": n_ovl,
                       "mean_len": st["sum_len"] / (G * args.batch), "mean_informative": st["sum_supported"] / (G * args.batch),
                       "model_windows_per_batch": st["n_model_windows"] / G, "batches_per_launch_group": G,
                       "streams_per_gpu": NS},
            "mbases_per_s": total_windows / el * W / 1e6,
            "roofline": roof,
            "roofline_featurize_group": {
                "kernels": feat_names, "bound": "hbm", "achieved": feat_bytes / G / (feat_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": feat_bytes / G / (feat_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_step": feat_bytes / G, "ms_per_step": feat_ms},
            "stage_ms_per_step": {"featurize": feat_ms, "model": model_ms},
            "host_prepare": {"windows_per_s": n_jobs * G * args.batch / host_prepare_s,
                             "note": "herro_job_create (CIGAR parse + windowing on a host thread pool + descriptor upload), "
                                     "outside the timed region; includes the Python-side array packing"},
            "kernels": kern,
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(synth.SEED + 2)
        print(json.dumps(out))
    for j in [j for js in jobs for j in js] + ([rem_job] if rem_job else []):
        j.close()
    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
