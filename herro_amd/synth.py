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


def generate(n_targets: int, target_len: int = 4 * 4096, n_overlaps: int = 32, *, seed: int = SEED,
             flank_min: int = 500, flank_max: int = 1000, p_sub: float = 0.006, p_ins: float = 0.004,
             p_del: float = 0.006, p_long_indel: float = 0.0, p_partial: float = 0.0,
             p_n_base: float = 0.0, min_partial_len: int = 0) -> SynthBatch:
    lib = _lib()
    p = _Params(seed, n_targets, target_len, n_overlaps, flank_min, flank_max,
                min_partial_len or max(1, target_len // 4), p_sub, p_ins, p_del, p_long_indel,
                p_partial, p_n_base)
    h = lib.herro_synth_generate(C.byref(p))
    try:
        sizes = np.zeros(5, np.uint64)
        lib.herro_synth_sizes(h, sizes.ctypes.data)
        n_reads, nb, n_aln, ncig, nt = (int(x) for x in sizes)
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
