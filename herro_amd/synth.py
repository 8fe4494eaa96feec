"""Synthetic overlap batches (SURVEY.md §8 d) — thin ctypes wrapper over csrc/synth.cpp.

The generator stands in for what the reference holds in RAM after `parse_reads` +
`parse_paf` (reference lib.rs:133, overlaps.rs:117-202): reads (ASCII bases, phred+33
quals) and, per target read, PAF-style alignments with M/I/D CIGARs.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SEED = 0x48455252  # "HERR" — SURVEY.md §8 d


class _Params(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("n_targets", C.c_uint32), ("target_len", C.c_uint32), ("n_overlaps", C.c_uint32),
        ("flank_min", C.c_uint32), ("flank_max", C.c_uint32), ("min_partial_len", C.c_uint32),
        ("p_sub", C.c_double), ("p_ins", C.c_double), ("p_del", C.c_double),
        ("p_long_indel", C.c_double), ("p_partial", C.c_double), ("p_n_base", C.c_double),
        ("p_snp", C.c_double),
    ]


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libherro_synth.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing — run `python -c 'import __graft_entry__ as g; g.build()'`")
        _LIB = C.CDLL(path)
        _LIB.herro_synth_generate.restype = C.c_void_p
        _LIB.herro_synth_generate.argtypes = [C.POINTER(_Params)]
        _LIB.herro_synth_free.argtypes = [C.c_void_p]
        _LIB.herro_synth_sizes.argtypes = [C.c_void_p, C.c_void_p]
        _LIB.herro_synth_copy.argtypes = [C.c_void_p] + [C.c_void_p] * 8
        _LIB.herro_synth_copy_into.argtypes = [C.c_void_p] + [C.c_void_p] * 8 + [C.c_uint64] * 4
    return _LIB


@dataclass
class SynthBatch:
    """Flat arrays; read `i` is seq[off[i]:off[i+1]] (ASCII) with the same slice of qual."""
    seq: np.ndarray          # u8 [total_bases]
    qual: np.ndarray         # u8 [total_bases]
    off: np.ndarray          # u64 [n_reads+1]
    aln: np.ndarray          # u32 [n_aln, 10]: qid,qlen,qstart,qend,strand,tid,tlen,tstart,tend,cigar_len
    cig_off: np.ndarray      # u64 [n_aln]
    cig: np.ndarray          # u8 [cigar_bytes] ASCII
    tgt_aln_off: np.ndarray  # u64 [n_targets+1] — alignments of target t: aln[tgt_aln_off[t]:tgt_aln_off[t+1]]
    tgt_rid: np.ndarray      # u32 [n_targets]

    @property
    def n_reads(self) -> int:
        return len(self.off) - 1

    @property
    def n_targets(self) -> int:
        return len(self.tgt_rid)

    def read_name(self, rid: int) -> str:
        return f"read{rid}"

    def cigar(self, a: int) -> bytes:
        o = int(self.cig_off[a])
        return self.cig[o:o + int(self.aln[a, 9])].tobytes()

    def read_seq(self, rid: int) -> bytes:
        return self.seq[int(self.off[rid]):int(self.off[rid + 1])].tobytes()


def _params(n_targets, target_len, n_overlaps, seed, flank_min=500, flank_max=1000, p_sub=0.006, p_ins=0.004, p_del=0.006,
            p_long_indel=0.0, p_partial=0.0, p_n_base=0.0, min_partial_len=0, p_snp=0.0) -> _Params:
    return _Params(seed, n_targets, target_len, n_overlaps, flank_min, flank_max,
                   min_partial_len or max(1, target_len // 4), p_sub, p_ins, p_del, p_long_indel,
                   p_partial, p_n_base, p_snp)


def _sizes(h) -> tuple[int, int, int, int, int]:
    sizes = np.zeros(5, np.uint64)
    _lib().herro_synth_sizes(h, sizes.ctypes.data)
    return tuple(int(x) for x in sizes)   # n_reads, bases, alignments, cigar bytes, targets


def generate(n_targets: int, target_len: int = 4 * 4096, n_overlaps: int = 32, *, seed: int = SEED,
             flank_min: int = 500, flank_max: int = 1000, p_sub: float = 0.006, p_ins: float = 0.004,
             p_del: float = 0.006, p_long_indel: float = 0.0, p_partial: float = 0.0,
             p_n_base: float = 0.0, min_partial_len: int = 0, p_snp: float = 0.0) -> SynthBatch:
    lib = _lib()
    p = _params(n_targets, target_len, n_overlaps, seed, flank_min, flank_max, p_sub, p_ins, p_del, p_long_indel, p_partial,
                p_n_base, min_partial_len, p_snp)
    h = lib.herro_synth_generate(C.byref(p))
    try:
        n_reads, nb, n_aln, ncig, nt = _sizes(h)
        b = SynthBatch(
            seq=np.empty(nb, np.uint8), qual=np.empty(nb, np.uint8), off=np.empty(n_reads + 1, np.uint64),
            aln=np.empty((n_aln, 10), np.uint32), cig_off=np.empty(n_aln, np.uint64),
            cig=np.empty(ncig, np.uint8), tgt_aln_off=np.empty(nt + 1, np.uint64),
            tgt_rid=np.empty(nt, np.uint32))
        lib.herro_synth_copy(h, b.seq.ctypes.data, b.qual.ctypes.data, b.off.ctypes.data,
                             b.aln.ctypes.data, b.cig_off.ctypes.data, b.cig.ctypes.data,
                             b.tgt_aln_off.ctypes.data, b.tgt_rid.ctypes.data)
        return b
    finally:
        lib.herro_synth_free(h)


def merge(batches: list[SynthBatch]) -> SynthBatch:
    """Concatenate independent batches into one read store (read ids, alignment and CIGAR offsets rebased)."""
    if len(batches) == 1:
        return batches[0]
    seq = np.concatenate([b.seq for b in batches])
    qual = np.concatenate([b.qual for b in batches])
    offs, alns, cig_offs, tgt_offs, tgt_rids = [np.zeros(1, np.uint64)], [], [], [np.zeros(1, np.uint64)], []
    rbase = bbase = abase = cbase = 0
    for b in batches:
        offs.append(b.off[1:] + np.uint64(bbase))
        a = b.aln.copy()
        a[:, 0] += np.uint32(rbase)   # qid
        a[:, 5] += np.uint32(rbase)   # tid
        alns.append(a)
        cig_offs.append(b.cig_off + np.uint64(cbase))
        tgt_offs.append(b.tgt_aln_off[1:] + np.uint64(abase))
        tgt_rids.append(b.tgt_rid + np.uint32(rbase))
        rbase += b.n_reads; bbase += len(b.seq); abase += len(b.aln); cbase += len(b.cig)
    return SynthBatch(seq=seq, qual=qual, off=np.concatenate(offs), aln=np.concatenate(alns),
                      cig_off=np.concatenate(cig_offs), cig=np.concatenate([b.cig for b in batches]),
                      tgt_aln_off=np.concatenate(tgt_offs), tgt_rid=np.concatenate(tgt_rids))


def replicate_targets(sb: SynthBatch, copies: int) -> SynthBatch:
    """`copies` times the targets of `sb` at the cost of one more copy of each TARGET read: copy c > 0 of a target is a new read (its own
    id, the same bases and qualities) with the same alignments against the same query reads (CIGAR bytes shared).  The windows of every
    copy are computed on their own — nothing downstream knows they are equal — so a run over the replicated set does the work of `copies` x
    as many windows; what it saves is generator time (the noise model draws ~16 random numbers per base: 100 000 windows would take minutes
    to generate) and host memory (32 of a target's 33 reads are shared).  Like real data, where a read overlaps many targets.
    Targets are ordered copy by copy; corrected records of copy c equal those of copy 0 but for the id (a check the caller can make)."""
    if copies <= 1:
        return sb
    nt, nr = sb.n_targets, sb.n_reads
    off = sb.off.astype(np.int64)
    tl = (off[sb.tgt_rid.astype(np.int64) + 1] - off[sb.tgt_rid.astype(np.int64)])
    extra_len = int(tl.sum())
    nb = len(sb.seq)
    seq = np.empty(nb + (copies - 1) * extra_len, np.uint8)
    qual = np.empty_like(seq)
    seq[:nb] = sb.seq
    qual[:nb] = sb.qual
    tsel = np.concatenate([np.arange(off[r], off[r + 1]) for r in sb.tgt_rid.astype(np.int64)]) if nt else np.zeros(0, np.int64)
    tseq, tqual = sb.seq[tsel], sb.qual[tsel]
    for c in range(1, copies):
        seq[nb + (c - 1) * extra_len: nb + c * extra_len] = tseq
        qual[nb + (c - 1) * extra_len: nb + c * extra_len] = tqual
    new_off = np.empty(nr + 1 + (copies - 1) * nt, np.uint64)
    new_off[:nr + 1] = sb.off
    ends = np.cumsum(np.tile(tl, copies - 1)) + nb
    new_off[nr + 1:] = ends.astype(np.uint64)
    n_aln = len(sb.aln)
    aln = np.tile(sb.aln, (copies, 1))
    t_of_aln = np.repeat(np.arange(nt, dtype=np.int64), np.diff(sb.tgt_aln_off.astype(np.int64)))     # target index of every alignment
    for c in range(1, copies):
        aln[c * n_aln:(c + 1) * n_aln, 5] = (nr + (c - 1) * nt + t_of_aln).astype(np.uint32)
    tgt_rid = np.concatenate([sb.tgt_rid] + [np.arange(nr + (c - 1) * nt, nr + c * nt, dtype=np.uint32) for c in range(1, copies)])
    tao = np.concatenate([sb.tgt_aln_off[:1]] + [sb.tgt_aln_off[1:] + np.uint64(c * n_aln) for c in range(copies)])
    return SynthBatch(seq=seq, qual=qual, off=new_off, aln=aln, cig_off=np.tile(sb.cig_off, copies), cig=sb.cig, tgt_aln_off=tao, tgt_rid=tgt_rid)


def usable_cpus() -> int:
    """CPUs this process may use: hardware threads, the affinity mask and a cgroup CPU quota (a box that shows 256
    threads may grant 16 CPUs of run time per period; starting 256 workers there only buys throttling)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return n


def generate_parallel(n_targets: int, target_len: int = 4 * 4096, n_overlaps: int = 32, *, seed: int = SEED,
                      chunk: int = 128, workers: int | None = None, **kw) -> SynthBatch:
    """`generate` for large batches: chunks of `chunk` targets are generated concurrently (the generator is a
    single-threaded C++ call that releases the GIL; chunk i uses seed + 7919 * i) and merged.  The data differ
    from one `generate(n_targets, seed=seed)` call — same distribution, different draws."""
    import concurrent.futures as cf
    sizes = [min(chunk, n_targets - i) for i in range(0, n_targets, chunk)]
    if len(sizes) <= 1:
        return generate(n_targets, target_len, n_overlaps, seed=seed, **kw)
    workers = workers or min(len(sizes), usable_cpus(), 64)
    lib = _lib()
    handles: list = [None] * len(sizes)

    def make(iz):
        p = _params(iz[1], target_len, n_overlaps, seed + 7919 * iz[0], **kw)
        handles[iz[0]] = lib.herro_synth_generate(C.byref(p))     # (ctypes releases the GIL for the call)
        return _sizes(handles[iz[0]])
    try:
        with cf.ThreadPoolExecutor(workers) as ex:
            sz = np.array(list(ex.map(make, enumerate(sizes))), np.int64).reshape(len(sizes), 5)
            base = np.concatenate([np.zeros((1, 5), np.int64), np.cumsum(sz, axis=0)])    # reads, bases, alignments, cigar bytes, targets in front of part i
            n_reads, nb, n_aln, ncig, nt = (int(x) for x in base[-1])
            # the merged arrays are allocated once and every part is written into its slice by the thread pool (the first
            # touch of the pages included): `merge` of per-part arrays spent more time in page faults than the generator in all
            # its arithmetic
            b = SynthBatch(seq=np.empty(nb, np.uint8), qual=np.empty(nb, np.uint8), off=np.empty(n_reads + 1, np.uint64),
                           aln=np.empty((n_aln, 10), np.uint32), cig_off=np.empty(n_aln, np.uint64), cig=np.empty(ncig, np.uint8),
                           tgt_aln_off=np.empty(nt + 1, np.uint64), tgt_rid=np.empty(nt, np.uint32))
            b.off[0] = 0
            b.tgt_aln_off[0] = 0

            def place(i):
                r0, b0, a0, c0, t0 = (int(x) for x in base[i])
                lib.herro_synth_copy_into(handles[i], b.seq.ctypes.data + b0, b.qual.ctypes.data + b0, b.off.ctypes.data + 8 * r0,
                                          b.aln.ctypes.data + 40 * a0, b.cig_off.ctypes.data + 8 * a0, b.cig.ctypes.data + c0,
                                          b.tgt_aln_off.ctypes.data + 8 * t0, b.tgt_rid.ctypes.data + 4 * t0, r0, b0, a0, c0)
                lib.herro_synth_free(handles[i])
                handles[i] = None
            list(ex.map(place, range(len(sizes))))
        return b
    finally:
        for h in handles:
            if h is not None:
                lib.herro_synth_free(h)
