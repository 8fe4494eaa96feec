"""Multi-GPU path on CPU: world_size-2 gloo processes agree on the partition of target reads, cover
every read exactly once, balance windows, and the summed per-rank statistics equal the single-process
totals (the path has no data-path collective — SURVEY.md §8 e)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from herro_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_is_exact_cover_and_balanced():
    rng = np.random.default_rng(0)
    nw = rng.integers(1, 40, 1000)
    for ws in (1, 2, 4, 8):
        parts = shard.partition_targets(nw, ws)
        allt = np.concatenate(parts)
        assert sorted(allt.tolist()) == list(range(len(nw)))
        loads = [int(nw[p].sum()) for p in parts]
        assert max(loads) - min(loads) <= int(nw.max())
    assert shard.windows_of(np.array([4096, 4097, 1, 8192]), 4096).tolist() == [1, 2, 1, 2]


WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch, torch.distributed as dist
    from herro_amd import shard, synth
    import oracle_lib as O
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
    rank = dist.get_rank()
    sb = synth.generate(6, 1024, 10, seed=5, flank_min=30, flank_max=60)        # same data on every rank
    lens = (sb.off[1:] - sb.off[:-1])[sb.tgt_rid]
    parts = shard.partition_targets(shard.windows_of(lens, 256), 2)
    mine = parts[rank]
    store = O.store_from_synth(sb)
    nwin = nsup = 0
    for t in mine:                                                                # this rank's reads only
        rid, rows, cigs = O.target_alignments(sb, int(t))
        res = store.extract_features(rid, rows, cigs, 256)
        nwin += len(res)
        nsup += sum(len(res.window(w).sup_pos) for w in range(len(res)))
    tot = shard.gather_counts({{"windows": nwin, "informative": nsup, "reads": len(mine)}})
    if rank == 0:
        print(json.dumps({{"tot": tot, "parts": [p.tolist() for p in parts]}}))
    dist.destroy_process_group()
""")


def test_two_rank_gloo(tmp_path):
    port = 29500 + os.getpid() % 2000
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, port=port))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=env) for r in range(2)]
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    got = json.loads(outs[0][0].strip().splitlines()[-1])
    # single-process reference
    import oracle_lib as O
    from herro_amd import synth
    sb = synth.generate(6, 1024, 10, seed=5, flank_min=30, flank_max=60)
    store = O.store_from_synth(sb)
    nwin = nsup = 0
    for t in range(sb.n_targets):
        rid, rows, cigs = O.target_alignments(sb, t)
        res = store.extract_features(rid, rows, cigs, 256)
        nwin += len(res)
        nsup += sum(len(res.window(w).sup_pos) for w in range(len(res)))
    assert got["tot"] == {"windows": nwin, "informative": nsup, "reads": sb.n_targets}
    assert sorted(got["parts"][0] + got["parts"][1]) == list(range(sb.n_targets))


WORKER2 = textwrap.dedent("""
    import os, sys, json, hashlib
    sys.path.insert(0, {root!r})
    import numpy as np, torch, torch.distributed as dist
    from herro_amd import shard, synth
    world = int(sys.argv[2])
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=world)
    rank = int(sys.argv[1])
    n_targets = int(sys.argv[3]) if len(sys.argv) > 3 else 9
    sb = synth.generate(n_targets, 1400, 7, seed=11, flank_min=30, flank_max=60) if rank == 0 else None   # ONLY rank 0 ingests
    nw = shard.windows_of((sb.off[1:] - sb.off[:-1])[sb.tgt_rid], 256) if rank == 0 else None
    if world > 1:                                         # the read store is replicated by broadcast
        seq, qual, off = shard.broadcast_reads(sb)
    else:
        seq, qual, off = sb.seq, sb.qual, sb.off

    def correct(rids, aln_off, rows, cig_off, cig):       # stand-in corrector: a digest of exactly what arrived + the store
        out = []
        assert len(rids) > 0                              # an empty shard never reaches the corrector
        for k, rid in enumerate(rids):
            a0, a1 = int(aln_off[k]), int(aln_off[k + 1])
            h = hashlib.sha1()
            h.update(rows[a0:a1].tobytes())
            for a in range(a0, a1):
                h.update(cig[int(cig_off[a]):int(cig_off[a]) + int(rows[a, 9])].tobytes())
            h.update(seq[int(off[rid]):int(off[rid + 1])].tobytes()); h.update(qual[int(off[rid]):int(off[rid + 1])].tobytes())
            out.append((int(rid), (">read%d \\n%s\\n" % (rid, h.hexdigest())).encode()))
        ends = np.cumsum([len(f) for _, f in out]).astype(np.uint64)
        return np.array([r for r, _ in out], np.uint32), ends, b"".join(f for _, f in out)
    rec, n_mine = shard.correct_sharded(sb, nw, correct)
    if rank == 0:
        print(json.dumps({{"fasta": shard.sorted_fasta(*rec).decode(), "mine": n_mine, "records": int(len(rec[0]))}}))
    if world > 1:
        dist.destroy_process_group()
""")


def test_scatter_work_gather_fasta_two_ranks(tmp_path):
    """north_star's only collectives: rank 0 scatters the window work, every rank corrects its shard, the corrected reads are
    gathered to rank 0.  The gathered FASTA (id-sorted) is byte-identical to the single-rank one; rank 1 never saw the
    alignments except through the scatter, nor the reads except through the broadcast."""
    port = 31500 + os.getpid() % 2000
    script = tmp_path / "worker2.py"
    script.write_text(WORKER2.format(root=ROOT, port=port))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    import json
    single = subprocess.run([sys.executable, str(script), "0", "1"], capture_output=True, text=True, env=env, timeout=240)
    assert single.returncode == 0, single.stderr
    want = json.loads(single.stdout.strip().splitlines()[-1])
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=env) for r in range(2)]
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    got = json.loads(outs[0][0].strip().splitlines()[-1])
    assert got["fasta"] == want["fasta"] and got["fasta"].count(">") == 9
    assert 0 < got["mine"] < 9 and want["mine"] == 9


def test_three_ranks_with_an_empty_shard(tmp_path):
    """Fewer target reads than ranks: one rank gets no work, sends no result, and nothing hangs; the gathered records are
    the single-rank ones."""
    port = 33500 + os.getpid() % 2000
    script = tmp_path / "worker3.py"
    script.write_text(WORKER2.format(root=ROOT, port=port))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    import json
    single = subprocess.run([sys.executable, str(script), "0", "1", "2"], capture_output=True, text=True, env=env, timeout=240)
    assert single.returncode == 0, single.stderr
    want = json.loads(single.stdout.strip().splitlines()[-1])
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "3", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=env) for r in range(3)]
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    got = json.loads(outs[0][0].strip().splitlines()[-1])
    assert got["fasta"] == want["fasta"] and got["records"] == 2 and got["mine"] == 1


WORKER3 = textwrap.dedent("""
    import os, sys, json, hashlib
    sys.path.insert(0, {root!r})
    import numpy as np, torch, torch.distributed as dist
    from herro_amd import shard, synth, api
    rank, world, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    if world > 1:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sb = synth.generate(7, 1300, 8, seed=13, flank_min=30, flank_max=60)      # the READS (every rank has the read set); alignments come from the file
    names = [sb.read_name(i).encode() for i in range(sb.n_reads)]
    seq, qual, off = sb.seq, sb.qual, sb.off

    def correct(rids, aln_off, rows, cig_off, cig):       # stand-in corrector: a digest of exactly what arrived
        out = []
        for k, rid in enumerate(rids):
            assert int(shard.owner_of([rid], world)[0]) == rank   # only targets this rank owns
            a0, a1 = int(aln_off[k]), int(aln_off[k + 1])
            h = hashlib.sha1()
            h.update(np.ascontiguousarray(rows[a0:a1]).tobytes())
            for a in range(a0, a1):
                h.update(cig[int(cig_off[a]):int(cig_off[a]) + int(rows[a, 9])].tobytes())
            h.update(seq[int(off[rid]):int(off[rid + 1])].tobytes())
            out.append((int(rid), (">read%d \\n%s\\n" % (rid, h.hexdigest())).encode()))
        ends = np.cumsum([len(f) for _, f in out]).astype(np.uint64)
        return np.array([r for r, _ in out], np.uint32), ends, b"".join(f for _, f in out)
    share = shard.ingest_paf_range(shard.paf_byte_range(path, rank, world), names)     # THIS rank's bytes only
    rec, n_mine, sent = shard.correct_sharded_local(share, correct)
    tot = shard.gather_counts({{"sent": sent, "alns_read": len(share.aln), "owned": n_mine}}) if world > 1 else {{"sent": sent, "alns_read": len(share.aln), "owned": n_mine}}
    if rank == 0:
        print(json.dumps({{"fasta": shard.sorted_fasta(*rec).decode(), "records": int(len(rec[0])), "tot": tot, "sent0": sent, "read0": len(share.aln)}}))
    if world > 1:
        dist.destroy_process_group()
""")


@pytest.mark.parametrize("world", [3, 8])   # 8: the node the driver scales to — more ranks than targets (7), ranks that own nothing and ranks whose byte range holds no line of their own targets
def test_per_rank_ingestion_three_ranks(tmp_path, world):
    """Round 4: no rank reads or ships the whole alignment set.  Every rank parses its own byte range of the PAF, one all-to-all
    takes every target to its owner (a hash of the read id), the owner merges the pieces in file order and drops a second alignment of a
    (query, target) pair ACROSS pieces as parse_paf does inside one file; the gathered FASTA equals the single-rank one, and the
    one the whole-file parser gives."""
    import json
    from herro_amd import synth, api
    sb = synth.generate(7, 1300, 8, seed=13, flank_min=30, flank_max=60)
    names = [sb.read_name(i).encode() for i in range(sb.n_reads)]

    def line(a):
        r = sb.aln[a]
        return b"\t".join([names[r[0]], b"%d" % r[1], b"%d" % r[2], b"%d" % r[3], b"-" if r[4] else b"+", names[r[5]],
                            b"%d" % r[6], b"%d" % r[7], b"%d" % r[8], b"60", b"60", b"255", b"cg:Z:" + sb.cigar(a)])
    order = np.random.default_rng(3).permutation(len(sb.aln))        # a PAF sorted by nothing: every target's alignments all over the file
    lines = [line(int(a)) for a in order]
    lines.append(line(int(order[0])))                                # the same (query, target) pair again, at the far end of the file: dropped
    lines.insert(len(lines) // 2, b"\t".join([names[3], b"1300", b"0", b"100", b"+", names[3], b"1300", b"0", b"100", b"60", b"60", b"255", b"cg:Z:100M"]))  # self overlap
    path = tmp_path / "all.paf"
    path.write_bytes(b"\n".join(lines) + b"\n")
    port = 35500 + os.getpid() % 2000
    script = tmp_path / "worker_local.py"
    script.write_text(WORKER3.format(root=ROOT, port=port))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    single = subprocess.run([sys.executable, str(script), "0", "1", str(path)], capture_output=True, text=True, env=env, timeout=240)
    assert single.returncode == 0, single.stderr
    want = json.loads(single.stdout.strip().splitlines()[-1])
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(world)]
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    got = json.loads(outs[0][0].strip().splitlines()[-1])
    assert got["fasta"] == want["fasta"] and got["records"] == want["records"] == sb.n_targets
    assert got["tot"]["alns_read"] == len(sb.aln) + 1 - 0 and got["tot"]["owned"] == sb.n_targets   # every line read once (the duplicate is read, then dropped by the owner; the self overlap by the parser)
    assert want["tot"]["sent"] == 0 and 0 < got["sent0"] < got["tot"]["sent"]          # rank 0 ships part of ITS share, nothing else
    assert got["read0"] < 0.6 * len(sb.aln)
    # the whole-file parser sees the same work: per target the same alignments in the same order
    paf = api.Paf(names, text=path.read_bytes())
    rows = paf.rows()
    merged = shard.merge_pieces([shard.shard_arrays(sh, np.arange(len(sh.tgt_rid))) for sh in
                                 (shard.ingest_paf_range(shard.paf_byte_range(str(path), r, world), names) for r in range(world))])
    assert merged[0].tolist() == paf.targets.tolist() and merged[1].tolist() == paf.aln_off.tolist()
    for a in range(len(rows)):
        assert merged[2][a, :9].tolist() == list(rows[a][:9]) and merged[4][int(merged[3][a]):int(merged[3][a]) + int(merged[2][a, 9])].tobytes() == rows[a][9]
    paf.close()


WORKER4 = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, {root!r})
    import numpy as np, torch, torch.distributed as dist
    from herro_amd import shard
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    W = 4096
    rng = np.random.default_rng(11)
    n_reads = 600
    lens = rng.integers(3000, 40000, n_reads)
    lens[rng.choice(n_reads, 25, replace=False)] = rng.integers(200000, 400000, 25)      # ultra-long reads: 10x the windows of the others
    targets = np.flatnonzero(rng.random(n_reads) < 0.7).astype(np.uint32)
    # every rank has read SOME alignments of some targets (a byte range of a PAF sorted by nothing): overlapping subsets
    seen = targets[rng.random((world, len(targets)))[rank] < 0.6]
    if rank == 1:
        seen = seen[:0]                                                                    # a rank that read nothing still takes part
    aln_off = np.arange(len(seen) + 1, dtype=np.uint64)
    rows = np.zeros((len(seen), 10), np.uint32); rows[:, 5] = seen; rows[:, 0] = 100000 + rank; rows[:, 9] = 2
    share = shard._Share(seen, aln_off, rows, np.arange(len(seen), dtype=np.uint64) * 2, np.frombuffer(b"1M" * max(len(seen), 1), np.uint8)[:2 * len(seen)].copy())

    def correct(rids, aln_off, rows, cig_off, cig):
        ends = np.cumsum([len(b">r%d \\nA\\n" % r) for r in rids]).astype(np.uint64) if len(rids) else np.zeros(0, np.uint64)
        return np.asarray(rids, np.uint32), ends, b"".join(b">r%d \\nA\\n" % r for r in rids)
    own_h = shard.owner_of(seen, world)
    own_b = shard.owners_by_load(seen, lens, W)
    rec, n_mine, sent = shard.correct_sharded_local(share, correct, read_lens=lens, window_size=W)
    # the load every rank ends up with under both rules (windows of the targets it owns), from the union of what the ranks saw
    union = np.unique(np.concatenate(shard.allgather_u32(seen)))
    wins = shard.windows_of(lens[union], W)
    ob = shard.owners_by_load(union, lens, W)
    oh = shard.owner_of(union, world)
    load_b = [int(wins[ob == r].sum()) for r in range(world)]
    load_h = [int(wins[oh == r].sum()) for r in range(world)]
    same = shard.gather_counts({{"agree": int((own_b == ob[np.searchsorted(union, seen)]).all())}})
    if rank == 0:
        print(json.dumps({{"load_balanced": load_b, "load_hash": load_h, "agree": same["agree"], "records": int(len(rec[0])), "targets": int(len(union))}}))
    dist.destroy_process_group()
""")


def test_owners_balanced_by_window_count_three_ranks(tmp_path):
    """across processes ownership was a hash of the read id — with UL reads (10x the windows) the load of a rank
    was left to chance.  owners_by_load: one all_gather of the target ids every rank has seen, longest-first assignment by window
    count; every rank computes the same owners, max / mean load <= 1.1, every target is corrected exactly once."""
    import json
    port = 37500 + os.getpid() % 2000
    script = tmp_path / "worker_lpt.py"
    script.write_text(WORKER4.format(root=ROOT, port=port))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(3)]
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    got = json.loads(outs[0][0].strip().splitlines()[-1])
    lb = np.array(got["load_balanced"], float)
    assert got["agree"] == 3 and got["records"] == got["targets"]
    assert lb.max() / lb.mean() <= 1.1, got
    lh = np.array(got["load_hash"], float)
    assert lb.max() <= lh.max()          # never worse than the hash


def test_record_messages_round_trip():
    a = (np.array([7, 3], np.uint32), np.array([5, 5], np.uint64), b">r7 \n")
    b = (np.array([1], np.uint32), np.array([6], np.uint64), b">r1 \nA")
    rids, ends, text = shard.unpack_records(shard.pack_records(*a))
    assert rids.tolist() == [7, 3] and ends.tolist() == [5, 5] and bytes(text) == a[2]
    m = shard.merge_records([a, (np.zeros(0, np.uint32), np.zeros(0, np.uint64), b""), b])
    assert m[0].tolist() == [7, 3, 1] and m[1].tolist() == [5, 5, 11] and bytes(m[2]) == a[2] + b[2]
    assert shard.sorted_fasta(*m) == b">r1 \nA>r7 \n"
    assert len(shard.unpack_records(b"")[0]) == 0


def test_work_message_round_trip():
    from herro_amd import synth
    sb = synth.generate(5, 900, 6, seed=3, flank_min=20, flank_max=40)
    msg = shard.shard_work(sb, [4, 1, 2])
    rids, aln_off, rows, cig_off, cig = shard.unpack_work(msg)
    assert rids.tolist() == sb.tgt_rid[[4, 1, 2]].tolist() and len(aln_off) == 4
    k = 0
    for t in (4, 1, 2):
        for a in range(int(sb.tgt_aln_off[t]), int(sb.tgt_aln_off[t + 1])):
            assert rows[k].tolist() == sb.aln[a].tolist()
            assert cig[int(cig_off[k]):int(cig_off[k]) + int(rows[k, 9])].tobytes() == sb.cigar(a)
            k += 1
    assert k == len(rows) == int(aln_off[-1])


def test_work_messages_are_packed_once_and_sized_ahead():
    """shard_work writes a shard's message straight into one buffer; work_size announces its length before anything is packed
    (rank 0 broadcasts the sizes first); unpack_work hands back views equal to the zero-copy arrays rank 0 uses for itself."""
    from herro_amd import shard, synth
    sb = synth.generate(40, 2048 + 77, 12, seed=5, p_partial=0.3, flank_min=60, flank_max=90)
    rng = np.random.default_rng(1)
    for k, tg in enumerate((np.arange(0, 40), np.arange(7, 19), np.array([3, 4, 5, 9, 10, 30]), rng.permutation(40)[:17],
                            np.zeros(0, np.int64), np.array([39]))):
        for slot in (None, ("t", k)):                       # fresh buffer / the destination's reusable one
            m = shard.shard_work(sb, tg, slot=slot)
            assert len(m) == shard.work_size(sb, tg)
            rids, aln_off, rows, cig_off, cig = shard.unpack_work(m)
            r2, a2, rw2, so2, blob = shard.shard_arrays(sb, tg)
            assert np.array_equal(rids, r2) and np.array_equal(aln_off, a2) and np.array_equal(rows, rw2)
            for a in range(len(rows)):
                n = int(rows[a, 9])
                assert cig[int(cig_off[a]):int(cig_off[a]) + n].tobytes() == blob[int(so2[a]):int(so2[a]) + n].tobytes()
    rec = shard.pack_records(np.array([5, 9], np.uint32), np.array([4, 9], np.uint64), b">a\nAC>b\nG")
    r, e, t = shard.unpack_records(rec)
    assert r.tolist() == [5, 9] and e.tolist() == [4, 9] and t.tobytes() == b">a\nAC>b\nG"
