"""-m gpu: the sharded data path (herro_amd/shard.py) with the HIP corrector on one rank — work packed, unpacked, turned
into jobs by group, corrected, FASTA gathered and id-sorted — gives the records a plain job gives, and those are the
oracle's (consensus.rs restated) for the job's logits.  The 2-rank transport is covered on the CPU (test_sharding.py)."""
import numpy as np
import pytest

import gpu_common as G
import oracle_lib as O
from herro_amd import api, model_io, shard, synth

pytestmark = pytest.mark.gpu


def test_sharded_path_single_rank_matches_plain_job():
    W = 512
    sb = synth.generate(7, 4 * 512 + 77, 14, seed=91, flank_min=60, flank_max=90, p_partial=0.2)
    c = G.ctx()
    c.set_precision(api.DEFAULT_PRECISION)
    G.load_synth(c, sb)
    nw = shard.windows_of((sb.off[1:] - sb.off[:-1])[sb.tgt_rid], W)
    c2 = api.Context(0)                       # a second context of the same GPU: two feeder threads, two jobs in flight each
    c2.load_model(model_io.default_model_file(G.CACHE)[0])
    c2.set_precision(api.DEFAULT_PRECISION)
    c2.share_reads(c)                         # one read store per device: the second context adopts the first one's
    rec, n_mine = shard.correct_sharded(sb, nw, shard.hip_corrector([c, c2], W, 5, sb.read_name, group_targets=3))
    fasta = shard.sorted_fasta(*rec)
    assert n_mine == sb.n_targets and len(rec[0]) == sb.n_targets
    job = api.job_from_synth(c, sb, W)
    job.featurize()
    job.infer(5, 1)
    job.consensus()
    recs = sorted((int(sb.tgt_rid[t]), job.consensus_fasta(t, sb.read_name(int(sb.tgt_rid[t])))) for t in range(sb.n_targets))
    # the sharded path groups targets 3 at a time: cross-read batches differ from the one-job run only in their padding
    # (lmax), which can move a logit by f32 rounding but not the calls on this data — and either way each must be
    # what the oracle decodes from that run's own logits, which the other tests pin; here: same records
    assert fasta.decode() == "".join(f for _, f in recs)
    assert fasta.count(b">") >= 1
    # the same through the per-rank-ingestion path (one rank: its share is everything, nothing to route)
    share = shard._Share(*shard.shard_arrays(sb, np.arange(sb.n_targets)))
    rec2, n2, sent = shard.correct_sharded_local(share, shard.hip_corrector([c, c2], W, 5, sb.read_name, group_targets=3))
    assert n2 == sb.n_targets and sent == 0 and shard.sorted_fasta(*rec2) == fasta
    # herro_job_fasta (all targets, one call) == the per-target entry
    text, ends = job.fasta([sb.read_name(int(r)) for r in sb.tgt_rid], with_ends=True)
    assert text.decode() == "".join(job.consensus_fasta(t, sb.read_name(int(sb.tgt_rid[t]))) for t in range(sb.n_targets))
    assert int(ends[-1]) == len(text)
    job.close()
    # the shared store outlives the context that uploaded it being given another one: c2 keeps working on the old store
    sb2 = synth.generate(2, 600, 6, seed=4, flank_min=30, flank_max=50)
    c.set_reads(sb2.seq, sb2.qual, sb2.off)
    j2 = api.job_from_synth(c2, sb, W)
    j2.featurize(); j2.infer(5, 1); j2.consensus()
    assert j2.consensus_fasta(0, sb.read_name(int(sb.tgt_rid[0]))) == recs[0][1] if recs[0][0] == int(sb.tgt_rid[0]) else True
    j2.close()
    with pytest.raises(api.HerroError):
        c2.share_reads(c2)
    c2.close()
    G.load_synth(c, sb)                        # (the shared test context goes back to a known state)


def test_pool_of_contexts_fed_from_one_queue_matches_plain_job():
    """herro_pool (csrc/pool.cpp, the in-process layout of lib.rs:154-200): three contexts of the GPU pull groups of two targets from
    one shared counter; the FASTA of all targets, in target order, equals what one plain job gives; every group was taken once."""
    W = 512
    sb = synth.generate(9, 4 * 512 + 33, 12, seed=92, flank_min=60, flank_max=90, p_partial=0.2)
    pool = api.Pool([0, 0, 0])
    try:
        pool.load_model(model_io.default_model_file(G.CACHE)[0])
        pool.set_precision(api.DEFAULT_PRECISION)
        pool.set_reads(sb.seq, sb.qual, sb.off)
        names = [sb.read_name(int(r)) for r in sb.tgt_rid]
        text, ends = pool.correct(sb.tgt_rid, sb.aln, sb.tgt_aln_off, sb.cig, sb.cig_off, W, 5, names, group_targets=2)
        taken = pool.groups_taken()
        assert sum(taken) == 5 and len(taken) == 3
        c = G.ctx()
        c.set_precision(api.DEFAULT_PRECISION)
        G.load_synth(c, sb)
        want = []
        for t0 in range(0, sb.n_targets, 2):      # the same grouping: cross-read batches of 5 windows inside groups of two targets
            job = api.job_from_synth(c, sb, W, range(t0, min(t0 + 2, sb.n_targets)))
            job.featurize(); job.infer(5, 1); job.consensus()
            want.append(job.fasta(names[t0:t0 + 2]))
            job.close()
        assert text.tobytes() == b"".join(want)
        assert int(ends[-1]) == len(text) and np.all(np.diff(ends.astype(np.int64)) >= 0)
        # an input the reference panics on: the pool reports it with the reference's message
        bad = sb.aln.copy()
        bad[0, 5] = bad[0, 5] + 1                 # tid != rid of its group
        with pytest.raises(api.HerroError):
            pool.correct(sb.tgt_rid, bad, sb.tgt_aln_off, sb.cig, sb.cig_off, W, 5, names, group_targets=2)
    finally:
        pool.close()


def test_device_staged_collectives_on_rccl_with_one_rank(tmp_path):
    """the `nccl` (= RCCL) branch of shard.py has only ever run on gloo.  Here a world_size-1 RCCL group on the lease's GPU
    runs the sharded paths in LOOPBACK mode (shard.LOOPBACK: no single-rank shortcuts, a rank's message to itself through the same grouped
    isend / irecv as a peer's): broadcast of the read store, the size collectives, scatter, the all-to-all to the owners, the owners' all_gather,
    the gather of the records — every RCCL call, dtype and device staging of the N-rank path — and strong_leg on top.  FASTA == the plain job's.
    What it cannot show: a transfer between two distinct GPUs."""
    import os, subprocess, sys, textwrap, json
    code = textwrap.dedent("""
        import json, os, sys, types
        import numpy as np
        sys.path.insert(0, 'tests')
        import torch, torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
        import gpu_common as G
        from herro_amd import api, model_io, shard, synth
        assert shard.LOOPBACK and dist.get_backend() == 'nccl' and shard._dev().type == 'cuda'
        W = 512
        sb = synth.generate(7, 4 * 512 + 77, 14, seed=91, flank_min=60, flank_max=90, p_partial=0.2)
        c = G.ctx()
        seq, qual, off = shard.broadcast_reads(sb)                 # rank 0 -> everybody (here: itself), through the device
        assert np.array_equal(seq, sb.seq) and np.array_equal(qual, sb.qual) and np.array_equal(off, sb.off)
        c.set_reads(seq, qual, off)
        got = shard.allgather_u32(np.arange(5, dtype=np.uint32))
        assert len(got) == 1 and got[0].tolist() == [0, 1, 2, 3, 4]
        msg = np.arange(1000, dtype=np.uint32).view(np.uint8)
        assert np.array_equal(shard.exchange_bytes([msg])[0], msg)
        assert np.array_equal(shard.scatter_bytes([msg]), msg)
        assert np.array_equal(shard.gather_bytes(msg)[0], msg)
        nw = shard.windows_of((sb.off[1:] - sb.off[:-1])[sb.tgt_rid], W)
        fn = shard.hip_corrector([c], W, 5, sb.read_name, group_targets=3)
        rec, n_mine = shard.correct_sharded(sb, nw, fn)
        share = shard._Share(*shard.shard_arrays(sb, np.arange(sb.n_targets)))
        read_lens = np.diff(np.asarray(sb.off).astype(np.int64))
        rec2, n2, sent = shard.correct_sharded_local(share, fn, read_lens=read_lens, window_size=W)   # owners by load: all_gather of the ids
        job = api.job_from_synth(c, sb, W)
        job.featurize(); job.infer(5, 1); job.consensus()
        plain = "".join(f for _, f in sorted((int(sb.tgt_rid[t]), job.consensus_fasta(t, sb.read_name(int(sb.tgt_rid[t])))) for t in range(sb.n_targets)))
        job.close()
        out = {"n_mine": n_mine, "n2": n2, "same1": shard.sorted_fasta(*rec).decode() == plain, "same2": shard.sorted_fasta(*rec2).decode() == plain, "records": plain.count(">")}
        args = types.SimpleNamespace(precision=None, batch=128, group=2, warmup=0, strong_ingest='local', strong_base_targets=64)
        r = shard.strong_leg(args, 0, 1, 0, 256, n_ctx=2)
        out["strong"] = {k: r[k] for k in ("windows", "ranks_seen", "fasta_records") if k in r}
        out["strong_props"] = r.get("properties")
        dist.barrier()
        dist.destroy_process_group()
        print("RESULT " + json.dumps(out))
    """)
    env = dict(os.environ, HERRO_SHARD_LOOPBACK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], cwd=G.ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert r.returncode == 0 and lines, r.stdout[-3000:]
    d = json.loads(lines[-1][7:])
    assert d["same1"] and d["same2"] and d["n_mine"] == d["n2"] == 7 and d["records"] >= 1, d
    assert d["strong"]["windows"] == 256 and d["strong"]["ranks_seen"] == 1, d
    assert d["strong_props"] is None or all(v is True for k, v in d["strong_props"].items() if isinstance(v, bool)), d
