"""Multi-GPU sharding of the hot path (SURVEY.md §8 e).

The independent unit of feature generation is the TARGET READ (the haplotype re-ranking couples all
windows of a read, reference features.rs:462-500; consensus needs all windows of a read on one worker,
consensus.rs:249).  The reference runs one replica per `-d` device pulling reads from one shared queue
(lib.rs:154-200) and never communicates between devices.  Here: one process per GPU; the read store is
replicated (one broadcast per data set); target reads are partitioned statically, balanced by window count
(longest first).  Two modes:
  * weak scaling (bench.py default): every rank generates / owns its own batches; torch.distributed only carries the
    barrier and the max-over-ranks time — no data-path collective;
  * the sharded data path of BASELINE.json's north_star (`correct_sharded`, bench.py's "strong" leg): rank 0 ingests,
    the window work is SCATTERED (sizes by one small collective, payloads by grouped point-to-point sends — RCCL has no
    scatterv), every rank corrects its shard with several jobs in flight, the corrected reads are GATHERED to rank 0 the
    same way.  Nothing is reduced anywhere.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

ctypes_u8 = ctypes.c_uint8


def partition_targets(n_windows_per_target: np.ndarray, world_size: int) -> list[np.ndarray]:
    """Greedy longest-first partition of target indices into `world_size` shards with balanced window
    counts.  Deterministic: every rank computes the same partition from the same lengths."""
    nwt = np.asarray(n_windows_per_target, dtype=np.int64)
    if len(nwt) and (nwt == nwt[0]).all():   # equal reads: contiguous blocks are as balanced as any assignment, and cost nothing to cut
        return [np.asarray(p, np.int64) for p in np.array_split(np.arange(len(nwt), dtype=np.int64), world_size)]
    order = np.argsort(-nwt, kind="stable")
    load = np.zeros(world_size, np.int64)
    shards: list[list[int]] = [[] for _ in range(world_size)]
    for t in order:
        r = int(np.argmin(load))
        shards[r].append(int(t))
        load[r] += int(n_windows_per_target[t])
    return [np.array(sorted(s), dtype=np.int64) for s in shards]


def windows_of(read_lens: np.ndarray, window_size: int) -> np.ndarray:
    """features.rs:338: n_windows = ceil(len / W)."""
    return (np.asarray(read_lens, dtype=np.int64) + window_size - 1) // window_size


def gather_counts(local: dict[str, int], group=None) -> dict[str, int]:
    """Sum small integer statistics over ranks (windows, informative positions, corrected bases)."""
    import torch
    import torch.distributed as dist
    keys = sorted(local)
    t = torch.tensor([int(local[k]) for k in keys], dtype=torch.int64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t = t.cpu()
    return {k: int(v) for k, v in zip(keys, t.tolist())}


# =====================================================================================================================
# Strong scaling: ONE fixed set of target reads, ingested on rank 0, sharded over the ranks.
#
# The reference feeds its per-device replicas from one shared queue (reference lib.rs:154-200), collects per device
# (consensus.rs:229-263) and writes through one writer (lib.rs:267-291).  Across processes that becomes (BASELINE.json
# north_star, SURVEY.md §8 e): the read store replicated on every GPU (broadcast once), the window WORK — target ids,
# PAF rows, CIGARs — scattered from rank 0 by the partition above, the corrected reads gathered back to rank 0 as FASTA
# text, which rank 0 writes.  RCCL has no scatterv / gatherv: sizes go through one small collective, payloads through
# grouped point-to-point sends (xGMI is point-to-point anyway).  Nothing is reduced: there is no all-reduce on the path.
# The payloads are small (~15 KB of descriptors and ~4 KB of corrected bases per window), so they are aggregated: one
# message per rank for the work, one per rank for the results.
# =====================================================================================================================
def _dist():
    import torch.distributed as dist
    return dist


# LOOPBACK (tests; HERRO_SHARD_LOOPBACK=1): with ONE rank every collective below is still issued — sizes, broadcasts, all_gathers on the
# transport's device — and a rank's message to ITSELF goes through the same grouped isend / irecv as a peer's instead of the local shortcut.
# A world_size-1 `nccl` group on one GPU then executes every RCCL call, tensor dtype and device staging of the N-rank path
# (tests/test_gpu_sharded.py::test_device_staged_collectives_on_rccl_with_one_rank); what it cannot show is a transfer between two GPUs.
LOOPBACK = os.environ.get("HERRO_SHARD_LOOPBACK", "0") not in ("", "0")


def _alone(world: int) -> bool:
    return world == 1 and not (LOOPBACK and _dist().is_initialized())


def _dev(group=None):
    import torch
    dist = _dist()
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def pack_work(rids, aln_off, rows, cig_off, cig) -> np.ndarray:
    """One contiguous u8 message: the targets of one shard with their alignments (rows u32 [m, 10], CIGAR blob rebased)."""
    rids = np.ascontiguousarray(rids, np.uint32)
    aln_off = np.ascontiguousarray(aln_off, np.uint64)
    rows = np.ascontiguousarray(rows, np.uint32).reshape(-1, 10)
    cig_off = np.ascontiguousarray(cig_off, np.uint64)
    cig = np.ascontiguousarray(cig, np.uint8)
    hdr = np.array([len(rids), len(rows), len(cig)], np.uint64)
    parts = [hdr.view(np.uint8), aln_off.view(np.uint8), cig_off.view(np.uint8), rids.view(np.uint8), rows.reshape(-1).view(np.uint8), cig]
    return np.concatenate(parts)


def unpack_work(buf: np.ndarray):
    buf = np.ascontiguousarray(buf, np.uint8)
    n, m, c = (int(x) for x in buf[:24].view(np.uint64))
    o = 24
    # views, not copies (every piece is naturally aligned: 24-byte header, then the u64 arrays, then the u32 ones)
    aln_off = buf[o:o + 8 * (n + 1)].view(np.uint64); o += 8 * (n + 1)
    cig_off = buf[o:o + 8 * m].view(np.uint64); o += 8 * m
    rids = buf[o:o + 4 * n].view(np.uint32); o += 4 * n
    rows = buf[o:o + 40 * m].view(np.uint32).reshape(m, 10); o += 40 * m
    cig = buf[o:o + c]
    return rids, aln_off, rows, cig_off, cig


def shard_arrays(sb, targets):
    """(rids, aln_off, rows, cig_off, cig) of the given target indices of a SynthBatch-like object (tgt_rid, tgt_aln_off, aln,
    cig_off, cig): the arguments of a corrector.  The CIGAR blob is the data set's own (offsets into it): no text is copied —
    what rank 0 uses for its own shard."""
    targets = np.asarray(targets, np.int64)
    a0 = np.asarray(sb.tgt_aln_off, np.int64)[targets] if len(targets) else np.zeros(0, np.int64)
    a1 = np.asarray(sb.tgt_aln_off, np.int64)[targets + 1] if len(targets) else np.zeros(0, np.int64)
    cnt = a1 - a0
    aln_off = np.zeros(len(targets) + 1, np.uint64)
    aln_off[1:] = np.cumsum(cnt)
    total = int(aln_off[-1])
    contiguous = len(targets) > 0 and total == int(a1[-1] - a0[0])            # consecutive targets: plain slices
    if contiguous:
        rows = sb.aln[int(a0[0]):int(a1[-1])]
        src_off = np.asarray(sb.cig_off, np.uint64)[int(a0[0]):int(a1[-1])]
    else:
        sel = np.repeat(a0 - aln_off[:-1].astype(np.int64), cnt) + np.arange(total, dtype=np.int64)
        rows = sb.aln[sel]
        src_off = np.asarray(sb.cig_off, np.uint64)[sel] if total else np.zeros(0, np.uint64)
    return np.asarray(sb.tgt_rid)[targets], aln_off, rows, src_off, sb.cig


def work_size(sb, targets) -> int:
    """len(shard_work(sb, targets)) without building the message (rank 0 announces all sizes before it packs anything)."""
    targets = np.asarray(targets, np.int64)
    n = len(targets)
    if n == 0:
        return 24 + 8
    off = np.asarray(sb.tgt_aln_off, np.int64)
    a0, a1 = off[targets], off[targets + 1]
    m = int((a1 - a0).sum())
    if m == int(a1[-1] - a0[0]):     # consecutive targets
        c = int(np.asarray(sb.aln[int(a0[0]):int(a1[-1]), 9], np.int64).sum())
    else:
        c = int(sum(int(np.asarray(sb.aln[int(x):int(y), 9], np.int64).sum()) for x, y in zip(a0, a1)))
    return 24 + 8 * (n + 1) + 8 * m + 4 * n + 40 * m + c


_SCRATCH: dict = {}


def _scratch(slot, nbytes: int) -> np.ndarray:
    """A reusable message buffer per destination (grow-only): the pages of a fresh 50 MB array are touched for the first time
    while it is filled, which costs more than the copy itself; the second job through the path finds them mapped."""
    b = _SCRATCH.get(slot)
    if b is None or len(b) < nbytes:
        b = _SCRATCH[slot] = np.empty(int(nbytes * 1.25) + 4096, np.uint8)
    return b[:nbytes]


def shard_work(sb, targets, slot=None) -> np.ndarray:
    """pack_work of the given target indices: the message for another rank, written ONCE into one buffer (header, offsets, read
    ids, PAF rows, then one slice of the CIGAR blob per run of alignments whose texts lie back to back — they do in every
    producer here)."""
    rids, aln_off, rows, src_off, blob = shard_arrays(sb, targets)
    n, total = len(rids), len(rows)
    lens = rows[:, 9].astype(np.uint64) if total else np.zeros(0, np.uint64)
    c = int(lens.sum()) if total else 0
    nbytes = 24 + 8 * (n + 1) + 8 * total + 4 * n + 40 * total + c
    buf = np.empty(nbytes, np.uint8) if slot is None else _scratch(slot, nbytes)   # slot: reuse this destination's buffer
    buf[:24].view(np.uint64)[:] = (n, total, c)
    o = 24
    buf[o:o + 8 * (n + 1)].view(np.uint64)[:] = aln_off; o += 8 * (n + 1)
    cig_off = buf[o:o + 8 * total].view(np.uint64); o += 8 * total
    if total:
        cig_off[0] = 0
        np.cumsum(lens[:-1], out=cig_off[1:])
    buf[o:o + 4 * n].view(np.uint32)[:] = rids; o += 4 * n
    buf[o:o + 40 * total].view(np.uint32).reshape(total, 10)[:] = rows; o += 40 * total
    if total:
        brk = np.ones(total, bool)
        if total > 1:
            brk[1:] = src_off[1:] != src_off[:-1] + lens[:-1]
        run0 = np.flatnonzero(brk)
        run1 = np.concatenate([run0[1:], [total]])
        for i, j in zip(run0, run1):
            b0, b1 = int(src_off[i]), int(src_off[j - 1] + lens[j - 1])
            buf[o:o + b1 - b0] = blob[b0:b1]
            o += b1 - b0
    assert o == len(buf)
    return buf


def _exchange_sizes(sizes_on_root, n_local: int, root_to_all: bool, group=None) -> list[int]:
    """root_to_all: rank 0 tells every rank the size of its message; else every rank tells rank 0 its size."""
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = _dev(group)
    if root_to_all:
        t = torch.tensor(sizes_on_root if rank == 0 else [0] * world, dtype=torch.int64, device=dev)
        dist.broadcast(t, 0, group=group)
        return [int(x) for x in t.cpu().tolist()]
    t = torch.zeros(world, dtype=torch.int64, device=dev)
    t[rank] = n_local
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)   # world integers: the one collective of the result path
    return [int(x) for x in t.cpu().tolist()]


def scatter_bytes(messages, group=None, sizes=None) -> np.ndarray:
    """rank 0 holds one u8 message per rank (or a callable that builds it, with `sizes` = their lengths); every rank gets its
    own.  Sizes by one broadcast, payloads by grouped point-to-point sends (RCCL has no scatterv)."""
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if _alone(world):
        return np.ascontiguousarray(messages[0], np.uint8)
    dev = _dev(group)
    if rank == 0 and sizes is None:
        sizes = [len(m) for m in messages]
    sizes = _exchange_sizes(sizes if rank == 0 else None, 0, True, group)
    if rank == 0 and LOOPBACK:   # rank 0's own message through the transport as well
        m0 = messages[0]() if callable(messages[0]) else messages[0]
        t = torch.from_numpy(np.array(m0, np.uint8)).to(dev)
        back = torch.empty(len(t), dtype=torch.uint8, device=dev)
        if len(t):
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, t, 0, group=group), dist.P2POp(dist.irecv, back, 0, group=group)]):
                w.wait()
        messages = [back.cpu().numpy()] + list(messages[1:])
    if rank == 0:
        # a message may be a callable that builds it (sizes given by the caller): the peers' messages are packed and staged on
        # the transport's device by a few threads at once (numpy copies and torch's H2D release the GIL), then sent in one group
        import concurrent.futures as cf

        def stage(r):
            m = messages[r]() if callable(messages[r]) else messages[r]
            m = np.ascontiguousarray(m, np.uint8)
            assert len(m) == sizes[r], (r, len(m), sizes[r])
            return torch.from_numpy(m).to(dev)
        todo = [r for r in range(1, world) if sizes[r]]
        with cf.ThreadPoolExecutor(max(1, min(len(todo), 8))) as ex:
            keep = dict(zip(todo, ex.map(stage, todo)))
        ops = [dist.P2POp(dist.isend, keep[r], r, group=group) for r in todo]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        m0 = messages[0]() if callable(messages[0]) else messages[0]
        return np.ascontiguousarray(m0, np.uint8)
    buf = torch.empty(sizes[rank], dtype=torch.uint8, device=dev)
    if sizes[rank]:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, 0, group=group)]):
            w.wait()
    return buf.cpu().numpy()


def gather_bytes(message, group=None):
    """every rank contributes one message (bytes or a u8 array); rank 0 returns the list of u8 arrays in rank order (others
    None).  gatherv by grouped send / recv; nothing is copied on the way except by the transport."""
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = np.frombuffer(message, np.uint8) if isinstance(message, (bytes, bytearray, memoryview)) else np.ascontiguousarray(message, np.uint8)
    if _alone(world):
        return [mine]
    dev = _dev(group)
    sizes = _exchange_sizes(None, len(mine), False, group)
    if rank == 0:
        bufs = [None] + [torch.empty(sizes[r], dtype=torch.uint8, device=dev) for r in range(1, world)]
        ops = [dist.P2POp(dist.irecv, bufs[r], r, group=group) for r in range(1, world) if sizes[r]]
        if LOOPBACK and len(mine):   # its own records through the transport as well
            bufs[0] = torch.empty(len(mine), dtype=torch.uint8, device=dev)
            ops += [dist.P2POp(dist.isend, torch.from_numpy(np.array(mine, np.uint8)).to(dev), 0, group=group), dist.P2POp(dist.irecv, bufs[0], 0, group=group)]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return [mine if bufs[0] is None else bufs[0].cpu().numpy()] + [bufs[r].cpu().numpy() for r in range(1, world)]
    if len(mine):
        t = torch.from_numpy(mine if mine.flags.writeable else mine.copy()).to(dev)   # (torch wants a writeable array to wrap)
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, t, 0, group=group)]):
            w.wait()
    return None


def broadcast_reads(sb, group=None):
    """The read store, replicated: rank 0's (seq, qual, off) to every rank (the one bulk transfer, once per data set)."""
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if _alone(world):
        return sb.seq, sb.qual, sb.off
    dev = _dev(group)
    n = torch.tensor([len(sb.seq), len(sb.off)] if rank == 0 else [0, 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, 0, group=group)
    nb, no = (int(x) for x in n.cpu().tolist())
    out = []
    for arr, cnt, dt in ((sb.seq if rank == 0 else None, nb, np.uint8), (sb.qual if rank == 0 else None, nb, np.uint8),
                         (sb.off if rank == 0 else None, no, np.uint64)):
        t = torch.from_numpy(np.ascontiguousarray(arr, dt).view(np.uint8)).to(dev) if rank == 0 else \
            torch.empty(cnt * np.dtype(dt).itemsize, dtype=torch.uint8, device=dev)
        dist.broadcast(t, 0, group=group)
        out.append(t.cpu().numpy().view(dt))
    return tuple(out)


def pack_records(rids, ends, text) -> np.ndarray:
    """One u8 message: the FASTA records of a set of targets — their read ids, the end offset of every target's records in
    `text`, the text.  No per-record Python objects anywhere on the result path."""
    rids = np.ascontiguousarray(rids, np.uint32)
    ends = np.ascontiguousarray(ends, np.uint64)
    text = np.frombuffer(text, np.uint8) if isinstance(text, (bytes, bytearray, memoryview)) else np.ascontiguousarray(text, np.uint8)
    n, nb = len(rids), len(text)
    buf = np.empty(16 + 12 * n + nb, np.uint8)
    buf[:16].view(np.uint64)[:] = (n, nb)
    buf[16:16 + 8 * n].view(np.uint64)[:] = ends
    buf[16 + 8 * n:16 + 12 * n].view(np.uint32)[:] = rids
    buf[16 + 12 * n:] = text
    return buf


def unpack_records(buf):
    buf = np.frombuffer(buf, np.uint8) if isinstance(buf, (bytes, bytearray, memoryview)) else np.ascontiguousarray(buf, np.uint8)
    if len(buf) < 16:
        return np.zeros(0, np.uint32), np.zeros(0, np.uint64), np.zeros(0, np.uint8)
    n, nb = (int(x) for x in buf[:16].view(np.uint64))
    o = 16
    ends = buf[o:o + 8 * n].view(np.uint64); o += 8 * n
    rids = buf[o:o + 4 * n].view(np.uint32); o += 4 * n
    return rids, ends, buf[o:o + nb]


def _concat_u8(arrays) -> np.ndarray:
    """np.concatenate for a few large u8 arrays, copied by several threads (numpy releases the GIL in the copy): the FASTA text of a
    32768-window job set is 130 MB — one thread takes 30 ms over it, which was half of what the whole sharded leg took."""
    arrays = [np.frombuffer(a, np.uint8) if isinstance(a, (bytes, bytearray, memoryview)) else a for a in arrays]
    total = sum(len(a) for a in arrays)
    if total < (8 << 20) or len(arrays) < 2:
        return np.concatenate(arrays) if len(arrays) > 1 else arrays[0]
    out = _scratch("concat", total)   # grow-only, reused: fresh pages cost more than the copy (the result is valid until the next call)
    offs = np.cumsum([0] + [len(a) for a in arrays])
    import concurrent.futures as cf

    def put(i):
        out[offs[i]:offs[i + 1]] = arrays[i]
    with cf.ThreadPoolExecutor(min(8, len(arrays))) as ex:
        list(ex.map(put, range(len(arrays))))
    return out


def merge_records(parts):
    """Concatenate (rids, ends, text) triples (ends rebased)."""
    parts = [p for p in parts if len(p[0])]
    if not parts:
        return np.zeros(0, np.uint32), np.zeros(0, np.uint64), np.zeros(0, np.uint8)
    if len(parts) == 1:
        return parts[0]
    base = np.cumsum([0] + [len(p[2]) for p in parts[:-1]]).astype(np.uint64)
    return (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] + b for p, b in zip(parts, base)]), _concat_u8([p[2] for p in parts]))


def sorted_fasta(rids, ends, text) -> bytes:
    """The records ordered by read id (the reference writes them in completion order, lib.rs:267-291: compare as sorted sets)."""
    text = bytes(text) if not isinstance(text, bytes) else text
    starts = np.concatenate([[0], ends[:-1]]).astype(np.int64) if len(ends) else np.zeros(0, np.int64)
    order = np.argsort(np.asarray(rids, np.int64), kind="stable")
    return b"".join(text[int(starts[i]):int(ends[i])] for i in order)


def correct_sharded(sb, n_windows_per_target, correct_fn, group=None):
    """The whole multi-GPU data path for one set of targets.  Rank 0 passes `sb` (targets + alignments) and the windows per
    target; other ranks pass None.  correct_fn(rids, aln_off, rows, cig_off, cig) -> (rids, ends, text): the FASTA records of
    its targets (any order), runs on every rank over its shard.  Returns (on rank 0; None elsewhere) the (rids, ends, text)
    of all targets in arrival order — sorted_fasta() orders them by read id — and this rank's shard size."""
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if rank == 0:   # its own shard needs no message: views into the data set
        parts = partition_targets(n_windows_per_target, world)
        if not _alone(world):
            scatter_bytes([np.zeros(0, np.uint8)] + [(lambda p=p, r=r: shard_work(sb, p, slot=("work", r))) for r, p in enumerate(parts[1:], 1)], group,
                          sizes=[0] + [work_size(sb, p) for p in parts[1:]])
        rids, aln_off, rows, cig_off, cig = shard_arrays(sb, parts[0])
    else:
        rids, aln_off, rows, cig_off, cig = unpack_work(scatter_bytes(None, group))
    rec = correct_fn(rids, aln_off, rows, cig_off, cig) if len(rids) else (np.zeros(0, np.uint32), np.zeros(0, np.uint64), b"")
    if _alone(world):
        return merge_records([rec]), len(rids)
    gathered = gather_bytes(pack_records(*rec), group)
    if rank != 0:
        return None, len(rids)
    return merge_records([unpack_records(g) for g in gathered]), len(rids)


# =====================================================================================================================
# Per-rank ingestion (round 4): no rank parses or ships the whole alignment set.
#
# correct_sharded above is the literal north_star path (rank 0 ingests, scatters) and cannot scale: 13 KB of CIGAR text per
# window leave one process.  Here every rank INGESTS ITS OWN SHARE of the alignments — a line-aligned byte range of the PAF
# (paf_byte_range + ingest_paf_range), or its own batch files — and keeps what it owns: a target read belongs to rank
# owner_of(rid) (a hash of its id; all windows of a read must meet on one rank: features.rs:462-500, consensus.rs:249).  What a rank has read of
# targets it does not own goes to their owners in ONE all-to-all of packed work messages (the path's only real exchange step:
# a share of the file is not a share of the targets); the owner merges the pieces in file order and applies parse_paf's rule
# for a second alignment of a (query, target) pair across pieces (overlaps.rs:175-185: the first one stays).  Rank 0 sends
# (world - 1) / world of ITS OWN share and nothing else; the read store is still replicated by one broadcast.
# =====================================================================================================================
def owner_of(rids, world: int) -> np.ndarray:
    """The rank a target read belongs to: a multiplicative hash of its id (read ids often come in strides — every k-th read a
    target — which a plain `rid % world` would send to one rank)."""
    h = (np.asarray(rids, np.uint64) * np.uint64(0x9E3779B1)) & np.uint64(0xffffffff)
    return ((h >> np.uint64(8)) % np.uint64(max(world, 1))).astype(np.int64)


def allgather_u32(local: np.ndarray, group=None) -> list[np.ndarray]:
    """Every rank's u32 array on every rank (sizes by one all_gather of an integer, payloads by one padded all_gather)."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    local = np.ascontiguousarray(local, np.uint32)
    if _alone(world):
        return [local]
    dev = _dev(group)
    n = torch.tensor([len(local)], dtype=torch.int64, device=dev)
    ns = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    sizes = [int(t.item()) for t in ns]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.int32, device=dev)
    if len(local):
        buf[:len(local)] = torch.from_numpy(local.view(np.int32)).to(dev)
    out = [torch.zeros(cap, dtype=torch.int32, device=dev) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return [o[:k].cpu().numpy().view(np.uint32) for o, k in zip(out, sizes)]


def owners_by_load(tgt_rid, read_lens, window_size: int, group=None) -> np.ndarray:
    """The owner of every target of this rank's share, balanced by WORK instead of hashed (SURVEY §8 e; the reference hands its
    reads out dynamically, lib.rs:154-200 — across processes the nearest thing is to balance what can be known up front):
    every rank announces the target ids it has seen (one all_gather of 4 bytes per target), the union is cut into shards by
    partition_targets — greedy longest-first over the targets' window counts ceil(len / W) (features.rs:338), which every rank
    knows from the replicated read store — and every rank computes the same assignment.  UL reads differ 10x in length: a
    hash of the id leaves the load of a rank to chance, this bounds max / mean by the longest read's share."""
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    tgt_rid = np.asarray(tgt_rid, np.uint32)
    if _alone(world):
        return np.zeros(len(tgt_rid), np.int64)
    union = np.unique(np.concatenate(allgather_u32(tgt_rid, group)))
    shards = partition_targets(windows_of(np.asarray(read_lens)[union], window_size), world)
    owner_u = np.empty(len(union), np.int64)
    for r, sh in enumerate(shards):
        owner_u[sh] = r
    return owner_u[np.searchsorted(union, tgt_rid)]


class _Share:
    """The arrays shard_arrays / shard_work / work_size read from a data set: tgt_rid, tgt_aln_off, aln, cig_off, cig."""

    def __init__(self, rids, aln_off, rows, cig_off, cig):
        self.tgt_rid = np.asarray(rids, np.uint32)
        self.tgt_aln_off = np.asarray(aln_off, np.uint64)
        self.aln = np.asarray(rows, np.uint32).reshape(-1, 10)
        self.cig_off = np.asarray(cig_off, np.uint64)
        self.cig = np.asarray(cig, np.uint8)


def paf_byte_range(path: str, rank: int, world: int) -> np.ndarray:
    """This rank's line-aligned share of a PAF file: from the first line that starts at or behind size * rank / world up to the
    first line that starts at or behind size * (rank + 1) / world.  Every line belongs to exactly one rank."""
    size = os.path.getsize(path)

    def line_start(pos):   # first line start >= pos
        if pos <= 0:
            return 0
        if pos >= size:
            return size
        with open(path, "rb") as f:
            f.seek(pos - 1)
            while True:
                b = f.read(1 << 16)
                if not b:
                    return size
                k = b.find(b"\n")
                if k >= 0:
                    return f.tell() - len(b) + k + 1
    lo, hi = line_start(size * rank // world), line_start(size * (rank + 1) // world)
    return np.fromfile(path, np.uint8, count=hi - lo, offset=lo)


def ingest_paf_range(text: np.ndarray, names) -> _Share:
    """herro_paf_parse_view over this rank's bytes (no copy of the text: the CIGARs stay where they are) -> the share's arrays.
    `names`: api.NameIndex (built once per read set) or the list of read ids."""
    from herro_amd import api
    own = None
    if not isinstance(names, api.NameIndex):
        names = own = api.NameIndex(names)
    buf = np.ascontiguousarray(text, np.uint8)
    paf = api.Paf(names, text=bytes(buf), view=True) if len(buf) else None   # (ctypes wants a bytes object to point into; Paf keeps it alive)
    if paf is None or paf.n_alns == 0:
        if own is not None:
            own.close()
        return _Share(np.zeros(0, np.uint32), np.zeros(1, np.uint64), np.zeros((0, 10), np.uint32), np.zeros(0, np.uint64), np.zeros(0, np.uint8))
    m = paf.n_alns
    raw = np.ctypeslib.as_array((ctypes_u8 * (m * 48)).from_address(paf._alns_ptr)).view(np.uint32).reshape(m, 12)   # herro_alignment: 10 u32 + a pointer
    rows = raw[:, :10].copy()
    ptr = raw[:, 10:12].copy().view(np.uint64).reshape(m)
    base = ptr.min()
    lens = rows[:, 9].astype(np.uint64)
    end = int((ptr + lens).max() - base)
    blob = np.ctypeslib.as_array((ctypes_u8 * end).from_address(int(base))).copy()   # the stretch of text the CIGARs lie in
    sh = _Share(paf.targets, paf.aln_off, rows, ptr - base, blob)
    paf.close()
    if own is not None:
        own.close()
    return sh


def exchange_bytes(messages, group=None):
    """All-to-all of one u8 message per destination rank (empty ones cost nothing); returns the messages received, by source
    rank.  Sizes by one all_gather of `world` integers, payloads by one group of point-to-point transfers."""
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if _alone(world):
        return [np.ascontiguousarray(messages[0], np.uint8)]
    dev = _dev(group)
    self_too = LOOPBACK   # the message to this rank itself through the transport as well
    mine = torch.tensor([len(m) for m in messages], dtype=torch.int64, device=dev)
    allsz = [torch.zeros(world, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(allsz, mine, group=group)
    sizes = [[int(x) for x in t.cpu().tolist()] for t in allsz]          # sizes[src][dst]
    send = {d: torch.from_numpy(np.array(messages[d], np.uint8)).to(dev) for d in range(world) if (d != rank or self_too) and sizes[rank][d]}
    recv = {r: torch.empty(sizes[r][rank], dtype=torch.uint8, device=dev) for r in range(world) if (r != rank or self_too) and sizes[r][rank]}
    ops = [dist.P2POp(dist.isend, t, d, group=group) for d, t in send.items()] + [dist.P2POp(dist.irecv, t, r, group=group) for r, t in recv.items()]
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    out = []
    for r in range(world):
        out.append(np.ascontiguousarray(messages[rank], np.uint8) if (r == rank and not self_too) else (recv[r].cpu().numpy() if r in recv else np.zeros(0, np.uint8)))
    return out


def merge_pieces(pieces):
    """Pieces of work (rids, aln_off, rows, cig_off, cig) in FILE ORDER (source rank order) -> one: the alignments of a target
    from all pieces back to back in that order, targets by first appearance, a second alignment of a (query, target) pair
    dropped (overlaps.rs:175-185 across pieces; inside a piece the parser has done it)."""
    pieces = [p for p in pieces if len(p[0])]
    if not pieces:
        return np.zeros(0, np.uint32), np.zeros(1, np.uint64), np.zeros((0, 10), np.uint32), np.zeros(0, np.uint64), np.zeros(0, np.uint8)
    if len(pieces) == 1:   # one source (a single rank, or nobody else read anything of ours): the parser's own output, untouched — no copy
        return pieces[0]
    rows = np.concatenate([p[2] for p in pieces])
    base = np.cumsum([0] + [len(p[4]) for p in pieces[:-1]]).astype(np.uint64)
    cig_off = np.concatenate([p[3] + b for p, b in zip(pieces, base)])
    cig = np.concatenate([p[4] for p in pieces])
    tid = rows[:, 5].astype(np.int64)
    uniq, first = np.unique(tid, return_index=True)
    order_t = uniq[np.argsort(first, kind="stable")]                     # targets by first appearance
    gi = np.searchsorted(uniq, tid)
    grank = np.empty(len(uniq), np.int64); grank[np.searchsorted(uniq, order_t)] = np.arange(len(uniq))
    g = grank[gi]
    key = (g << 32) | rows[:, 0].astype(np.int64)                        # (target group, query id)
    _, keep_first = np.unique(key, return_index=True)
    keep = np.zeros(len(rows), bool); keep[keep_first] = True
    sel = np.flatnonzero(keep)
    sel = sel[np.argsort(g[sel], kind="stable")]
    cnt = np.bincount(g[sel], minlength=len(uniq))
    aln_off = np.zeros(len(uniq) + 1, np.uint64); aln_off[1:] = np.cumsum(cnt)
    return order_t.astype(np.uint32), aln_off, rows[sel], cig_off[sel], cig


def route_to_owners(share: _Share, group=None, read_lens=None, window_size: int = 0):
    """One all-to-all: every rank packs, per destination, the targets of its share that the destination owns, and gets back the
    pieces of its own targets from every rank, merged.  Ownership: balanced by window count when the read lengths are given
    (owners_by_load: one small all_gather more), else the hash of the id (owner_of).  Returns (work arrays, bytes this rank sent)."""
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    owner = owners_by_load(share.tgt_rid, read_lens, window_size, group) if read_lens is not None and window_size else owner_of(share.tgt_rid, world)
    mine = np.flatnonzero(owner == rank)
    msgs = [np.zeros(0, np.uint8)] * world
    sent = 0
    for d in range(world):
        if d == rank and not LOOPBACK:
            continue
        tg = np.flatnonzero(owner == d)
        if len(tg):
            msgs[d] = shard_work(share, tg, slot=("route", d))
            sent += len(msgs[d]) if d != rank else 0
    got = exchange_bytes(msgs, group) if not _alone(world) else [msgs[0]]
    pieces = []
    for r in range(world):
        if r == rank and not LOOPBACK:
            if len(mine):
                rids, aln_off, rows, src_off, blob = shard_arrays(share, mine)
                pieces.append((rids, aln_off, rows, src_off, blob))
        elif len(got[r]):
            pieces.append(unpack_work(got[r]))
    return merge_pieces(pieces), sent


def correct_sharded_local(share: _Share, correct_fn, group=None, read_lens=None, window_size: int = 0):
    """The sharded data path with per-rank ingestion: `share` = what THIS rank has read (ingest_paf_range of its byte range, or
    its own batch files); one all-to-all routes every target to its owner (read_lens + window_size: owners balanced by window
    count, owners_by_load; without them: the hash of the id); correct_fn runs on the owned targets; the FASTA records are gathered
    to rank 0.  Returns ((rids, ends, text) on rank 0 else None, owned targets, bytes sent while routing)."""
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    (rids, aln_off, rows, cig_off, cig), sent = route_to_owners(share, group, read_lens, window_size)
    rec = correct_fn(rids, aln_off, rows, cig_off, cig) if len(rids) else (np.zeros(0, np.uint32), np.zeros(0, np.uint64), b"")
    if _alone(world):
        return merge_records([rec]), len(rids), sent
    gathered = gather_bytes(pack_records(*rec), group)
    if rank != 0:
        return None, len(rids), sent
    return merge_records([unpack_records(g) for g in gathered]), len(rids), sent


def hip_corrector(ctxs, window_size: int, batch: int, read_name, group_targets: int = 1024):
    """correct_fn for correct_sharded on the HIP path: jobs of at most `group_targets` targets, cross-read batches of `batch`
    windows, device consensus, one herro_job_fasta call per job.  `ctxs`: one or more contexts of this rank's GPU; each gets a
    feeder thread that keeps two jobs in flight (herro_job_create of job k+1 runs on the host while the GPU works on job k),
    like the reference's feature threads ahead of its inference thread (lib.rs:159-187)."""
    import threading
    if not isinstance(ctxs, (list, tuple)):
        ctxs = [ctxs]

    def fn(rids, aln_off, rows, cig_off, cig):
        groups = [(t0, min(t0 + group_targets, len(rids))) for t0 in range(0, len(rids), group_targets)]
        out = [None] * len(groups)
        err = []

        def finish(g, job):
            t0, t1 = groups[g]
            job.infer(batch, 1)
            job.consensus()
            job.consensus_fetch()
            text, ends = job.fasta([read_name(int(r)) for r in rids[t0:t1]], with_ends=True, as_array=True,
                                   out_alloc=lambda nb: _scratch(("fasta", g), nb))   # per group a reusable buffer (valid until the next pass)
            out[g] = (rids[t0:t1], ends, text)
            job.close()

        def feeder(k):
            try:
                ctx, prev = ctxs[k], None
                for g in range(k, len(groups), len(ctxs)):
                    t0, t1 = groups[g]
                    a0, a1 = int(aln_off[t0]), int(aln_off[t1])
                    job = ctx.create_job(rids[t0:t1], rows[a0:a1], aln_off[t0:t1 + 1] - aln_off[t0], None, window_size,
                                         cig_blob=cig, cig_off=cig_off[a0:a1])
                    left_out = job.skipped()   # alignments parse_paf would have dropped: an ingest that hands them over is broken
                    if left_out != (0, 0):
                        raise RuntimeError(f"herro_job_create left out {left_out[0]} alignment(s) / {left_out[1]} target(s) of targets "
                                           f"{int(rids[t0])}..{int(rids[t1 - 1])}: {ctx.last_error()}")
                    job.featurize()
                    if prev is not None:
                        finish(*prev)
                    prev = (g, job)
                if prev is not None:
                    finish(*prev)
            except Exception as e:   # surfaced by the caller: a thread must not die silently
                err.append(e)
        th = [threading.Thread(target=feeder, args=(k,)) for k in range(min(len(ctxs), max(1, len(groups))))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if err:
            raise err[0]
        return merge_records([o for o in out if o is not None])
    return fn


def strong_leg(args, rank: int, world: int, local: int, n_windows: int, n_ctx: int | None = None, model_path: str | None = None) -> dict | None:
    """ONE fixed set of `n_windows` synthetic windows (BASELINE configs[3]) sharded over the ranks.  Default (--strong-ingest local):
    every rank holds its own share of the parsed alignments (handed out once, outside the timed region — what per-rank ingestion of its
    own byte range would have produced), sends the targets it does not own to their owners in one all-to-all, corrects what it owns
    (n_ctx contexts = feeder threads per GPU sharing one read store, two jobs in flight each); the corrected reads are gathered to
    rank 0.  --strong-ingest rank0: rank 0 ingests everything and scatters the work (the round-3 path).  Timed: routing ->
    herro_job_create -> featurize -> infer -> consensus -> D2H -> FASTA text -> gather, i.e. the whole multi-GPU data path, host work
    included (max over ranks).  Returns the figures on rank 0."""
    import os
    import time
    import torch
    import torch.distributed as dist
    from herro_amd import api, model_io, synth
    W, n_ovl, wpt = 4096, 32, 4
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n_t = max(world, -(-n_windows // wpt))
    # the fixed job at BASELINE's size (configs[3]: 100 000 windows) without minutes of generator time and ~30 GB of host memory: at most
    # `base_cap` targets are GENERATED, the set is the needed number of copies of them — every copy a target read of its own (own id, own
    # windows, own FASTA record), the 32 query reads shared (synth.replicate_targets).  The GPU does the work of every window of every copy.
    base_cap = int(getattr(args, "strong_base_targets", 4200) or 4200)
    copies = max(1, -(-n_t // base_cap))
    n_base = -(-n_t // copies)
    n_t = n_base * copies
    path = model_path or model_io.default_model_file(os.path.join(root, "tests", "_cache"))[0]
    sb = synth.replicate_targets(synth.generate_parallel(n_base, wpt * W, n_ovl, seed=synth.SEED + 3), copies) if rank == 0 else None
    seq, qual, off = broadcast_reads(sb) if not _alone(world) else (sb.seq, sb.qual, sb.off)
    if n_ctx is None:
        # feeder contexts per GPU: what the end_to_end leg of bench.py measured best on one GPU (four -> six feeders: +19 %, eight: slower again,
        # profiles/r4_ab_runs.json r4_e2e_feeders); with several ranks on one host three each, and the host pools of a rank sized to its share of the CPUs
        n_ctx = 6 if world == 1 else 3
    if world > 1 and "HERRO_HOST_THREADS" not in os.environ:
        per_host = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        os.environ["HERRO_HOST_THREADS"] = str(max(2, synth.usable_cpus() // (per_host * max(1, n_ctx))))   # read when a context starts its pool
    ctxs = []
    for _ in range(max(1, n_ctx)):
        c = api.Context(local)
        c.load_model(path)
        if args.precision is None:
            args.precision = c.precision()       # the tier the load-time calibration chose for this model
        c.set_precision(args.precision)
        if ctxs:
            c.share_reads(ctxs[0])               # one read store per device
        else:
            c.set_reads(seq, qual, off)
        ctxs.append(c)
    read_lens = np.diff(np.asarray(off).astype(np.int64))   # every rank knows the read lengths from the replicated store
    fn = hip_corrector(ctxs, W, args.batch, lambda rid: f"read{rid}", group_targets=max(1, args.group * args.batch // wpt))
    nw = np.full(n_t, wpt, np.int64) if rank == 0 else None
    local_ingest = getattr(args, "strong_ingest", "local") != "rank0"
    share = None
    if local_ingest:
        # every rank's own share of the parsed alignments (what it would have read from its own byte range / batch files): handed
        # out once, OUTSIDE the timed region — a contiguous block of targets per rank, which is NOT the ownership (owner_of)
        if rank == 0:
            blocks = np.array_split(np.arange(n_t, dtype=np.int64), world)
            msgs = [np.zeros(0, np.uint8)] + [(lambda p=p, r=r: shard_work(sb, p, slot=("share", r))) for r, p in enumerate(blocks[1:], 1)]
            if not _alone(world):
                scatter_bytes(msgs, None, sizes=[0] + [work_size(sb, p) for p in blocks[1:]])
            share = _Share(*shard_arrays(sb, blocks[0]))
        else:
            share = _Share(*unpack_work(scatter_bytes(None, None)))

    reg_blob = None
    if local_ingest and world == 1 and os.environ.get("HERRO_ZERO_COPY", "1") not in ("", "0") and len(share.cig):
        reg_blob = np.ascontiguousarray(share.cig, np.uint8)     # on one rank nothing is routed: the corrector sees the share's own blob — pinned once, jobs are created zero-copy
        if reg_blob is share.cig:
            ctxs[0].register_host(reg_blob)
        else:
            reg_blob = None

    def sync():
        if not _alone(world):
            dist.barrier()
        torch.cuda.synchronize()

    def one_pass():
        if local_ingest:
            return correct_sharded_local(share, fn, read_lens=read_lens, window_size=W)   # owners balanced by window count
        rec_, n_ = correct_sharded(sb, nw, fn)
        return rec_, n_, None
    if args.warmup:                                  # one untimed pass over the same fixed job (arenas, clocks)
        one_pass()
    sync()
    t0 = time.perf_counter()
    rec, n_mine, sent = one_pass()
    for c in ctxs:
        c.synchronize()
    el = time.perf_counter() - t0
    sync()
    seen = world
    sent_all = sent
    if not _alone(world):
        tt = torch.tensor([el], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
        one = torch.tensor([1, int(sent or 0)], device="cuda", dtype=torch.int64)
        dist.all_reduce(one)
        seen, sent_all = int(one[0].item()), int(one[1].item())
    if reg_blob is not None:
        ctxs[0].unregister_host(reg_blob)
    for c in ctxs:
        c.close()
    if rank != 0:
        return None
    rids, ends, text = rec
    n_rec = int(np.count_nonzero(np.asarray(text) == ord(">")))
    # size-independent properties of the gathered records: one record per target (every synthetic target has >= 2 alignments in every
    # window), every copy of a target corrected to the same bases as the original (the copies were computed independently, on whatever
    # rank owned them), nothing but ACGT in the sequences
    props = {"one_record_per_target": bool(n_rec == n_t and len(rids) == n_t)}
    try:
        tarr = np.asarray(text)
        order = np.argsort(np.asarray(rids), kind="stable")
        starts = np.concatenate([[0], np.asarray(ends[:-1], np.int64)])
        by_rid = {int(rids[i]): (int(starts[i]), int(ends[i])) for i in order}

        def body(rid):
            a, b = by_rid[rid]
            rec_ = tarr[a:b].tobytes()
            return rec_[rec_.index(b"\n") + 1:]
        sample = list(range(0, n_base, max(1, n_base // 64)))
        ok_copies, ok_alpha = True, True
        for t in sample:
            b0 = body(int(sb.tgt_rid[t]))
            ok_alpha &= set(b0) <= set(b"ACGT\n")
            for c in range(1, copies):
                ok_copies &= body(int(sb.tgt_rid[c * n_base + t])) == b0
        props.update(copies_agree=bool(ok_copies), acgt_only=bool(ok_alpha), targets_sampled=len(sample))
    except Exception as e:   # a property that cannot be evaluated is reported, it never takes the measurement down
        props["error"] = repr(e)
    return {"windows_per_s": n_t * wpt / el, "windows": n_t * wpt, "targets": n_t, "generated_targets": n_base, "copies_of_each_target": copies,
            "properties": props, "seconds": el, "ranks_seen": seen, "contexts_per_gpu": len(ctxs),
            "mbases_per_s": (len(text) - 16 * n_rec) / el / 1e6, "fasta_records": n_rec, "fasta_bytes": int(len(text)),
            "ingest": "local" if local_ingest else "rank0",
            "routing_bytes_sent_by_rank0": sent, "routing_bytes_sent_all_ranks": sent_all,
            "routing_bytes_per_window_rank0": (sent or 0) / (n_t * wpt),
            "timed": ("all-to-all of the targets a rank read but does not own + herro_job_create + featurize + infer + consensus + D2H + herro_job_fasta "
                      "+ gather to rank 0 (whole sharded data path, host work included); every rank holds its own share of the parsed alignments "
                      "(per-rank ingestion), rank 0 ships nothing but part of its share") if local_ingest else
                     ("scatter + herro_job_create + featurize + infer + consensus + D2H + herro_job_fasta + gather to rank 0 "
                      "(whole sharded data path, host work included); rank 0 ingests")}


def bench_strong(args, rank: int, world: int, local: int):
    """bench.py --scaling strong: the line of the sharded data path alone (see strong_leg)."""
    import json
    import torch.distributed as dist
    n_windows = args.windows or args.steps * args.batch
    r = strong_leg(args, rank, world, local, n_windows)
    if rank == 0:
        steps = r["windows"] // args.batch
        print(json.dumps({
            "metric": "4096-bp windows corrected/sec at batch=128", "value": r["windows_per_s"], "unit": "windows/s", "n_gpus": world,
            "steps": steps, "warmup": args.warmup, "ms_per_step": 1e3 * r["seconds"] / max(steps, 1), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": {1: "bf16x3", 4: "f16 (encoder proj / FF GEMMs: activation hi+lo)", 5: "f16", 7: "f16 (proj hi+lo, FF single)", 8: "f16 (FF hi+lo, proj single)"}.get(args.precision, str(args.precision)),
            "data": "synthetic (SURVEY §8d generator, seed 0x48455252+3; random-init weights of the assumed architecture)",
            "config": {"workload": f"ONE fixed job of {r['windows']} synthetic 4096-bp windows (32 overlaps each, batch=128) sharded by target read over "
                                   f"{world} rank(s), ingest = {r.get('ingest', 'local')}; rank 0 gathers the corrected reads (BASELINE configs[3])",
                       "batch": args.batch, "window": 4096, "overlaps": 32, "timed": r["timed"], "precision": args.precision},
            "strong": r,
            "roofline": {"bound": "hbm", "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None,
                         "note": "per-kernel roofline: run the default (weak) mode; this mode times the sharded data path end to end"}}))
    if world > 1:
        dist.destroy_process_group()
