"""ctypes driver for oracle/liborc.so (CPU restatement of the reference; TEST USE ONLY)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


class OrcAln(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("qid", "qlen", "qstart", "qend", "strand", "tid", "tlen", "tstart", "tend", "cigar_len")] + \
               [("cigar_off", C.c_uint64)]


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(ROOT, "oracle", "liborc.so")
        src = [os.path.join(ROOT, "oracle", f) for f in ("herro_oracle.hpp", "herro_oracle_capi.cpp")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src if os.path.exists(s)):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        L.orc_last_error.restype = C.c_char_p
        L.orc_encode.restype = C.c_long
        L.orc_encode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.orc_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
        L.orc_store_new.restype = C.c_void_p
        L.orc_store_new.argtypes = [C.c_uint32] + [C.c_void_p] * 5
        L.orc_store_free.argtypes = [C.c_void_p]
        L.orc_extract_windows.restype = C.c_long
        L.orc_extract_windows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_uint64]
        L.orc_extract_features.restype = C.c_void_p
        L.orc_extract_features.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_result_free.argtypes = [C.c_void_p]
        L.orc_result_n_windows.restype = C.c_uint32
        L.orc_result_n_windows.argtypes = [C.c_void_p]
        L.orc_window_dims.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_window_copy.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 5
        L.orc_window_copy_p1.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 6
        L.orc_collate_dims.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_collate_copy.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_void_p] * 5
        L.orc_normalise_qual.restype = C.c_float
        L.orc_normalise_qual.argtypes = [C.c_uint8]
        L.orc_consensus_fasta.restype = C.c_long
        L.orc_consensus_fasta.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        _LIB = L
    return _LIB


class OracleError(RuntimeError):
    """A reference panic reproduced by the oracle."""


def _err():
    return OracleError(lib().orc_last_error().decode())


def encode(seq: bytes) -> np.ndarray:
    words = np.zeros((len(seq) + 31) // 32 + 1, np.uint64)
    buf = np.frombuffer(seq, np.uint8)
    n = lib().orc_encode(buf.ctypes.data if len(seq) else None, len(seq), words.ctypes.data, len(words))
    if n < 0:
        raise _err()
    return words[:n].copy()


def decode(words: np.ndarray, length: int, start: int, end: int, reversed_: bool) -> bytes:
    out = np.zeros(max(end - start, 0), np.uint8)
    words = np.ascontiguousarray(words, np.uint64)
    if lib().orc_decode(words.ctypes.data, len(words), length, start, end, int(reversed_), out.ctypes.data) != 0:
        raise _err()
    return out.tobytes()


def make_alns(rows, cigars):
    """rows: iterable of 9-tuples (qid,qlen,qstart,qend,strand,tid,tlen,tstart,tend); cigars: list[bytes]."""
    arr = (OrcAln * len(cigars))()
    blob = b"".join(cigars)
    off = 0
    for i, (r, c) in enumerate(zip(rows, cigars)):
        a = arr[i]
        (a.qid, a.qlen, a.qstart, a.qend, a.strand, a.tid, a.tlen, a.tstart, a.tend) = (int(x) for x in r)
        a.cigar_len = len(c)
        a.cigar_off = off
        off += len(c)
    return arr, np.frombuffer(blob + b"\0", np.uint8).copy()


def extract_windows(row, cigar: bytes, n_windows: int, window_size: int, is_target: bool = True):
    arr, blob = make_alns([row], [cigar])
    out = np.zeros((4096, 8), np.uint64)
    n = lib().orc_extract_windows(C.byref(arr), blob.ctypes.data, n_windows, window_size, int(is_target),
                                  out.ctypes.data, len(out))
    if n < 0:
        raise _err()
    return out[:n].astype(np.int64)


@dataclass
class OracleWindow:
    bases: np.ndarray       # u8 [L', 31] ASCII
    quals: np.ndarray       # u8 [L', 31]
    sup_pos: np.ndarray     # u16 [k]
    sup_ins: np.ndarray     # u8 [k]
    qids: np.ndarray        # u32 — ranked overlap read ids
    n_alns: int
    # pass-1 intermediates (not reference API)
    p1_qids: np.ndarray = field(default=None)
    p1_acc: np.ndarray = field(default=None)
    scores: np.ndarray = field(default=None)
    max_ins: np.ndarray = field(default=None)
    p1_L: int = 0
    p1_sup_pos: np.ndarray = field(default=None)
    p1_sup_ins: np.ndarray = field(default=None)


class Store:
    """Oracle read store (Vec<HAECRecord>)."""

    def __init__(self, seq: np.ndarray, qual: np.ndarray, off: np.ndarray, names):
        ids = [n.encode() if isinstance(n, str) else n for n in names]
        id_off = np.zeros(len(ids) + 1, np.uint64)
        id_off[1:] = np.cumsum([len(i) for i in ids])
        id_cat = np.frombuffer(b"".join(ids) + b"\0", np.uint8).copy()
        self._keep = (np.ascontiguousarray(seq, np.uint8), np.ascontiguousarray(qual, np.uint8),
                      np.ascontiguousarray(off, np.uint64), id_cat, id_off)
        self.h = lib().orc_store_new(len(ids), self._keep[0].ctypes.data, self._keep[1].ctypes.data,
                                     self._keep[2].ctypes.data, id_cat.ctypes.data, id_off.ctypes.data)
        if not self.h:
            raise _err()

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_store_free(self.h)
            self.h = None

    def extract_features(self, rid: int, rows, cigars, window_size: int) -> "FeatResult":
        arr, blob = make_alns(rows, cigars)
        h = lib().orc_extract_features(self.h, rid, len(cigars), C.byref(arr) if len(cigars) else None,
                                       blob.ctypes.data, window_size)
        if not h:
            raise _err()
        return FeatResult(self, h)


class FeatResult:
    def __init__(self, store: Store, h):
        self.store, self.h = store, h

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_result_free(self.h)
            self.h = None

    def __len__(self):
        return lib().orc_result_n_windows(self.h)

    def window(self, w: int) -> OracleWindow:
        d = np.zeros(8, np.uint64)
        lib().orc_window_dims(self.h, w, d.ctypes.data)
        L, ns, nq, na, p1n, p1L, p1ns, wl = (int(x) for x in d)
        bases = np.zeros((L, 31), np.uint8)
        quals = np.zeros((L, 31), np.uint8)
        sp = np.zeros(ns, np.uint16)
        si = np.zeros(ns, np.uint8)
        qids = np.zeros(nq, np.uint32)
        lib().orc_window_copy(self.h, w, bases.ctypes.data, quals.ctypes.data, sp.ctypes.data, si.ctypes.data,
                              qids.ctypes.data)
        p1q = np.zeros(p1n, np.uint32)
        acc = np.zeros(p1n, np.float32)
        sc = np.zeros(p1n, np.float64)
        mi = np.zeros(wl, np.uint16)
        p1sp = np.zeros(p1ns, np.uint16)
        p1si = np.zeros(p1ns, np.uint8)
        lib().orc_window_copy_p1(self.h, w, p1q.ctypes.data, acc.ctypes.data, sc.ctypes.data, mi.ctypes.data,
                                 p1sp.ctypes.data, p1si.ctypes.data)
        return OracleWindow(bases, quals, sp, si, qids, na, p1q, acc, sc, mi, p1L, p1sp, p1si)

    def collate(self, batch_size: int, bi: int = 0):
        """prepare_examples + collate (inference.rs:73-145,214-253) over this read's windows."""
        d = np.zeros(5, np.uint64)
        if lib().orc_collate_dims(self.h, batch_size, bi, d.ctypes.data) != 0:
            raise _err()
        nb, B, L, R, N = (int(x) for x in d)
        if bi >= nb:
            return nb, None
        wids = np.zeros(B, np.uint32)
        bases = np.zeros((B, L, R), np.uint8)
        quals = np.zeros((B, L, R), np.uint8)
        lens = np.zeros(B, np.int32)
        idx = np.zeros(N, np.int32)
        if lib().orc_collate_copy(self.h, batch_size, bi, wids.ctypes.data, bases.ctypes.data, quals.ctypes.data,
                                  lens.ctypes.data, idx.ctypes.data) != 0:
            raise _err()
        return nb, dict(wids=wids, bases=bases, quals=quals, lens=lens, indices=idx)

    def consensus_fasta(self, logits: np.ndarray) -> str:
        """consensus (consensus.rs:86-227) + write_sequence (lib.rs:282-317).  logits: f32 [N_total, 5]
        for all supported positions of all windows in window order."""
        logits = np.ascontiguousarray(logits, np.float32)
        cap = 1 << 24
        out = C.create_string_buffer(cap)
        n = lib().orc_consensus_fasta(self.store.h, self.h, logits.ctypes.data if logits.size else None, out, cap)
        if n < 0:
            raise _err()
        return out.raw[:n].decode()


def store_from_synth(sb) -> Store:
    return Store(sb.seq, sb.qual, sb.off, [sb.read_name(i) for i in range(sb.n_reads)])


def target_alignments(sb, t: int):
    """(rid, rows, cigars) for target index t of a SynthBatch."""
    a0, a1 = int(sb.tgt_aln_off[t]), int(sb.tgt_aln_off[t + 1])
    rows = [tuple(int(x) for x in sb.aln[a, :9]) for a in range(a0, a1)]
    cigars = [sb.cigar(a) for a in range(a0, a1)]
    return int(sb.tgt_rid[t]), rows, cigars
