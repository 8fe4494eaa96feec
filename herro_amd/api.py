"""ctypes binding of libherro_amd.so (include/herro_amd.h).  Fails loudly if the HIP library is
missing or no GPU is present — there is no CPU fallback in this package."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HERRO_LIB: another build of the same library (e.g. the HERRO_PROF_BUILD one of tools/prof.sh), for A/B runs on one GPU box
LIB_PATH = os.environ.get("HERRO_LIB") or os.path.join(_HERE, "libherro_amd.so")
_LIB = None
# GEMM precision used by bench.py / smoke / the end-to-end tests (herro_set_precision): 1 = bf16 hi/lo x3,
# 4 = f16 (conv / FC / attention single, encoder GEMMs activation hi + lo), see csrc/model_h.hip; 6 = 4 with the activation remainder as e4m3 on the
# K = 128 MX MFMA (same contract, measured no faster on an MI355X: not the default)
DEFAULT_PRECISION = 4

EXPORTS = [
    "herro_version", "herro_create", "herro_destroy", "herro_last_error", "herro_set_stream", "herro_synchronize",
    "herro_encode_2bit", "herro_decode_2bit", "herro_set_reads", "herro_set_reads_packed", "herro_share_reads", "herro_load_model",
    "herro_set_precision", "herro_precision", "herro_clock_probe", "herro_calibration_error", "herro_debug_force_precision", "herro_model_describe", "herro_job_create", "herro_job_free", "herro_job_n_windows", "herro_job_skipped", "herro_job_featurize",
    "herro_job_infer", "herro_job_consensus", "herro_job_consensus_fetch", "herro_job_window_info", "herro_job_window_copy", "herro_job_window_logits",
    "herro_job_consensus_fasta", "herro_job_fasta", "herro_model_forward", "herro_timing_enable", "herro_timing_reset",
    "herro_timing_get", "herro_job_stats", "herro_debug_extract_windows",
    "herro_paf_parse", "herro_oec_read", "herro_paf_n_targets", "herro_paf_target_ids", "herro_paf_aln_off",
    "herro_paf_alignments", "herro_paf_free", "herro_name_index_create", "herro_name_index_free", "herro_paf_parse_indexed",
    "herro_oec_read_indexed", "herro_paf_parse_view", "herro_debug_host_ctx", "herro_debug_job_array", "herro_debug_tile_plan", "herro_debug_tile_plan_sib",
    "herro_debug_set_featurize_planes", "herro_debug_set_host_build", "herro_debug_job_dev_built", "herro_debug_job_rf", "herro_debug_job_rf_fused", "herro_debug_job_rf_left", "herro_debug_e4m3", "herro_debug_sib_fault", "herro_debug_sib_retries", "herro_debug_base_row_votes", "herro_debug_vote5",
    "herro_pool_create", "herro_pool_destroy", "herro_pool_last_error", "herro_pool_size", "herro_pool_ctx", "herro_pool_set_reads", "herro_pool_load_model",
    "herro_pool_correct", "herro_pool_result", "herro_pool_groups_taken", "herro_pool_skipped", "herro_debug_pool_fake", "herro_job_create_status", "herro_host_register", "herro_host_unregister", "herro_debug_zero_copy_jobs",
    "herro_fastx_read", "herro_reads_count", "herro_reads_seq", "herro_reads_qual", "herro_reads_off", "herro_reads_ids",
    "herro_reads_descs", "herro_reads_free", "herro_write_window_features", "herro_job_write_features",
]


class HerroError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"herro_amd error {code}: {msg}")
        self.code = code


class Alignment(C.Structure):  # herro_alignment
    _fields_ = [(n, C.c_uint32) for n in
                ("qid", "qlen", "qstart", "qend", "strand", "tid", "tlen", "tstart", "tend", "cigar_len")] + \
               [("cigar", C.c_void_p)]


class WindowInfo(C.Structure):  # herro_window_info
    _fields_ = [(n, C.c_uint32) for n in
                ("rid", "wid", "n_total_wins", "length", "n_alns", "n_overlaps", "n_supported", "win_len")]


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built — run __graft_entry__.build(); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        L.herro_version.restype = C.c_char_p
        L.herro_create.restype = vp
        L.herro_create.argtypes = [i32]
        L.herro_destroy.argtypes = [vp]
        L.herro_last_error.restype = C.c_char_p
        L.herro_last_error.argtypes = [vp]
        L.herro_set_stream.argtypes = [vp, vp]
        L.herro_synchronize.argtypes = [vp]
        L.herro_encode_2bit.restype = C.c_int64
        L.herro_encode_2bit.argtypes = [vp, u64, vp]
        L.herro_decode_2bit.argtypes = [vp, u64, u64, u64, i32, vp]
        L.herro_set_reads.argtypes = [vp, u32, vp, vp, vp, vp]
        L.herro_set_reads_packed.argtypes = [vp, u32, vp, vp, vp, vp, vp]
        L.herro_share_reads.argtypes = [vp, vp]
        L.herro_pool_create.restype = vp
        L.herro_pool_create.argtypes = [vp, u32]
        L.herro_pool_destroy.argtypes = [vp]
        L.herro_pool_last_error.restype = C.c_char_p
        L.herro_pool_last_error.argtypes = [vp]
        L.herro_pool_size.restype = u32
        L.herro_pool_size.argtypes = [vp]
        L.herro_pool_ctx.restype = vp
        L.herro_pool_ctx.argtypes = [vp, u32]
        L.herro_pool_set_reads.argtypes = [vp, u32, vp, vp, vp, vp]
        L.herro_pool_load_model.argtypes = [vp, C.c_char_p]
        L.herro_pool_correct.restype = C.c_int64
        L.herro_pool_correct.argtypes = [vp, u32, vp, vp, vp, u32, u32, i32, u32, vp, vp]
        L.herro_pool_result.restype = vp
        L.herro_pool_result.argtypes = [vp, vp]
        L.herro_pool_groups_taken.restype = u32
        L.herro_pool_groups_taken.argtypes = [vp, u32]
        L.herro_pool_skipped.argtypes = [vp, vp, vp]
        L.herro_debug_pool_fake.restype = vp
        L.herro_debug_pool_fake.argtypes = [u32, vp]
        L.herro_job_create_status.argtypes = [vp]
        L.herro_host_register.argtypes = [vp, vp, u64]
        L.herro_host_unregister.argtypes = [vp, vp]
        L.herro_debug_zero_copy_jobs.restype = u64
        L.herro_debug_zero_copy_jobs.argtypes = []
        L.herro_load_model.argtypes = [vp, C.c_char_p]
        L.herro_set_precision.argtypes = [vp, i32]
        try:   # (an older build of the library selected by HERRO_LIB for a same-box A/B lacks the round-6 entries)
            L.herro_precision.argtypes = [vp]
            L.herro_calibration_error.restype = C.c_float
            L.herro_calibration_error.argtypes = [vp, i32]
            L.herro_debug_force_precision.argtypes = [vp, i32]
            L.herro_clock_probe.argtypes = [vp, vp]
        except AttributeError:
            if not os.environ.get("HERRO_LIB"):
                raise
        L.herro_job_create.restype = vp
        L.herro_job_create.argtypes = [vp, u32, vp, vp, vp, u32]
        L.herro_job_free.argtypes = [vp]
        L.herro_job_n_windows.restype = u32
        L.herro_job_n_windows.argtypes = [vp]
        L.herro_job_skipped.argtypes = [vp, vp, vp]
        L.herro_job_featurize.argtypes = [vp]
        L.herro_job_infer.argtypes = [vp, u32, i32]
        L.herro_job_consensus.argtypes = [vp]
        L.herro_job_consensus_fetch.argtypes = [vp, vp]
        L.herro_job_window_info.argtypes = [vp, u32, vp]
        L.herro_job_window_copy.argtypes = [vp, u32, i32, vp, vp, vp, vp, vp]
        L.herro_job_window_logits.argtypes = [vp, u32, vp, vp]
        L.herro_job_consensus_fasta.restype = C.c_int64
        L.herro_job_consensus_fasta.argtypes = [vp, u32, C.c_char_p, C.c_char_p, vp, u64]
        L.herro_model_describe.restype = C.c_int64
        L.herro_model_describe.argtypes = [vp, vp, u64]
        L.herro_job_fasta.restype = C.c_int64
        L.herro_job_fasta.argtypes = [vp, vp, vp, vp, u64, vp]
        L.herro_model_forward.argtypes = [vp, u32, u32, vp, vp, vp, vp, vp, vp]
        L.herro_timing_enable.argtypes = [vp, i32]
        L.herro_timing_reset.argtypes = [vp]
        L.herro_timing_get.argtypes = [vp, vp, u64, vp, vp, vp]
        L.herro_job_stats.argtypes = [vp, vp]
        L.herro_debug_extract_windows.restype = C.c_int64
        L.herro_debug_extract_windows.argtypes = [vp, u32, u32, vp, u64, vp, u64]
        L.herro_paf_parse.restype = vp
        L.herro_paf_parse.argtypes = [C.c_char_p, u64, u32, C.c_char_p, vp, vp, i32, vp, u64]
        L.herro_oec_read.restype = vp
        L.herro_oec_read.argtypes = [C.c_char_p, u32, C.c_char_p, vp, vp, i32, vp, u64]
        L.herro_paf_n_targets.restype = u32
        L.herro_paf_n_targets.argtypes = [vp]
        for f in (L.herro_paf_target_ids, L.herro_paf_aln_off, L.herro_paf_alignments):
            f.restype = vp
            f.argtypes = [vp]
        L.herro_paf_free.restype = None
        L.herro_paf_free.argtypes = [vp]
        L.herro_name_index_create.restype = vp
        L.herro_name_index_create.argtypes = [u32, C.c_char_p, vp]
        L.herro_name_index_free.restype = None
        L.herro_name_index_free.argtypes = [vp]
        L.herro_paf_parse_indexed.restype = vp
        L.herro_paf_parse_indexed.argtypes = [C.c_char_p, u64, vp, vp, i32, vp, u64]
        L.herro_oec_read_indexed.restype = vp
        L.herro_oec_read_indexed.argtypes = [C.c_char_p, vp, vp, i32, vp, u64]
        L.herro_paf_parse_view.restype = vp
        L.herro_paf_parse_view.argtypes = [C.c_char_p, u64, vp, vp, i32, vp, u64]
        L.herro_debug_host_ctx.restype = vp
        L.herro_debug_host_ctx.argtypes = [u32, vp, vp]
        L.herro_debug_job_array.restype = C.c_int64
        L.herro_debug_job_array.argtypes = [vp, i32, vp, vp]
        L.herro_debug_set_featurize_planes.argtypes = [vp, i32]
        L.herro_debug_set_host_build.argtypes = [vp, i32]
        L.herro_debug_job_dev_built.argtypes = [vp]
        L.herro_debug_base_row_votes.argtypes = [vp, vp, vp, u32, vp, vp]
        L.herro_debug_vote5.restype = u32
        L.herro_debug_vote5.argtypes = [vp, u32]
        L.herro_debug_job_rf_fused.argtypes = [vp]
        L.herro_debug_job_rf_left.argtypes = [vp]
        L.herro_debug_sib_fault.argtypes = [vp]
        L.herro_debug_e4m3.argtypes = [C.c_float]
        L.herro_debug_e4m3.restype = C.c_uint32
        L.herro_debug_sib_retries.argtypes = [vp]
        L.herro_debug_job_rf.restype = C.c_int64
        L.herro_debug_job_rf.argtypes = [vp, u32, vp, u64]
        L.herro_debug_tile_plan.restype = C.c_int64
        L.herro_debug_tile_plan.argtypes = [vp, u32, i32, u32, vp, vp, vp]
        L.herro_debug_tile_plan_sib.restype = i32
        L.herro_debug_tile_plan_sib.argtypes = [vp, u32, i32, u32, vp, vp, vp, vp]
        _LIB = L
    return _LIB


def encode_2bit(seq: bytes) -> np.ndarray:
    """haec_io.rs:121-136 (host utility)."""
    words = np.zeros((len(seq) + 31) // 32 + 1, np.uint64)
    buf = np.frombuffer(seq, np.uint8)
    n = lib().herro_encode_2bit(buf.ctypes.data if len(seq) else None, len(seq), words.ctypes.data)
    if n < 0:
        raise HerroError(int(n), "byte >= 128 in sequence")
    return words[:n].copy()


def decode_2bit(words: np.ndarray, length: int, start: int, end: int, rc: bool) -> bytes:
    """haec_io.rs:138-173 (host utility)."""
    words = np.ascontiguousarray(words, np.uint64)
    out = np.zeros(max(end - start, 0), np.uint8)
    r = lib().herro_decode_2bit(words.ctypes.data, length, start, end, int(rc), out.ctypes.data)
    if r:
        raise HerroError(r, "Out of bounds for 2-bit sequence decoding.")
    return out.tobytes()


def debug_extract_windows(row, cigar: bytes, n_windows: int, window_size: int) -> np.ndarray:
    """Product windowing for one alignment (host only).  Rows: window,tstart,qstart,qend,op_lo,op_hi,so,eo."""
    a = Alignment()
    (a.qid, a.qlen, a.qstart, a.qend, a.strand, a.tid, a.tlen, a.tstart, a.tend) = (int(x) for x in row)
    buf = np.frombuffer(cigar + b"\0", np.uint8).copy()
    a.cigar_len, a.cigar = len(cigar), buf.ctypes.data
    out = np.zeros((65536, 8), np.uint64)
    err = C.create_string_buffer(512)
    n = lib().herro_debug_extract_windows(C.byref(a), n_windows, window_size, out.ctypes.data, len(out), err, 512)
    if n < 0:
        raise HerroError(int(n), err.value.decode())
    return out[:n].astype(np.int64)


def debug_tile_plan(counts, packed: bool = True, qmode: int = 0, n_cu: int = 256, bounds: bool = False):
    """(tiles, order) of one fused launch over windows of `counts` informative rows (host only, herro_debug_tile_plan).
    qmode 1 / 2: the plans with 32-token tiles (short last round / every small window) -> (tiles of 64, tiles of 32, order);
    bounds: append the first token of every tile (+ end)."""
    c = np.ascontiguousarray(counts, np.uint32)
    order = np.zeros(max(len(c), 1), np.uint32)
    n_half = C.c_uint32(0)
    tok = np.zeros(len(c) + 2, np.uint32)
    n = lib().herro_debug_tile_plan(c.ctypes.data, len(c), int(packed) | (qmode << 1), n_cu, order.ctypes.data, C.byref(n_half), tok.ctypes.data)
    if n < 0:
        raise HerroError(int(n), "herro_debug_tile_plan")
    out = (int(n), int(n_half.value), order[:len(c)]) if qmode else (int(n), order[:len(c)])
    return out + (tok[:int(n) + int(n_half.value) + 1],) if bounds else out


def debug_tile_plan_sib(counts, packed: bool = True, qmode: int = 1, n_cu: int = 256):
    """Token-tile plan with windows above 64 rows admitted (herro_debug_tile_plan_sib): (n_sibling, n64, n32, order, tile bounds, grp)."""
    c = np.ascontiguousarray(counts, np.uint32)
    order = np.zeros(max(len(c), 1), np.uint32)
    cap = int(((c.astype(np.int64) + 63) // 64).sum()) + 2
    n_tiles = np.zeros(3, np.uint32)
    tok = np.zeros(cap, np.uint32)
    grp = np.zeros(cap, np.uint32)
    rc = lib().herro_debug_tile_plan_sib(c.ctypes.data, len(c), int(packed) | (qmode << 1), n_cu, order.ctypes.data, n_tiles.ctypes.data, tok.ctypes.data, grp.ctypes.data)
    if rc < 0:
        raise HerroError(int(rc), "herro_debug_tile_plan_sib")
    nb, n64, n32 = (int(x) for x in n_tiles)
    return nb, n64, n32, order[:len(c)], tok[:nb + n64 + n32 + 1], grp[:nb]


class Pool:
    """herro_pool (csrc/pool.cpp): several contexts of one process fed from one queue of target groups — the in-process layout of
    lib.rs:154-200.  device_ids: one entry per context (several may name the same GPU: they share its read store)."""

    def __init__(self, device_ids, fake_us_per_aln=None):
        """fake_us_per_aln: test hook (herro_debug_pool_fake) — a device-free pool of len(fake_us_per_aln) stand-in contexts."""
        self._l = lib()
        if fake_us_per_aln is not None:
            us = np.ascontiguousarray(fake_us_per_aln, np.uint32)
            self.h = self._l.herro_debug_pool_fake(len(us), us.ctypes.data)
            self.n = len(us)
            if not self.h:
                raise HerroError(-1, "herro_debug_pool_fake")
            return
        ids = (C.c_int * len(device_ids))(*device_ids)
        self.h = self._l.herro_pool_create(ids, len(device_ids))
        if not self.h:
            raise HerroError(-1, self._l.herro_last_error(None).decode(errors="replace"))
        self.n = len(device_ids)

    def skipped(self) -> tuple[int, int]:
        a, t = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._l.herro_pool_skipped(self.h, C.byref(a), C.byref(t)))
        return a.value, t.value

    def _chk(self, rc):
        if rc < 0:
            raise HerroError(int(rc), self._l.herro_pool_last_error(self.h).decode(errors="replace"))
        return rc

    def set_reads(self, seq, qual, off, name_class=None):
        seq, qual, off = np.ascontiguousarray(seq, np.uint8), np.ascontiguousarray(qual, np.uint8), np.ascontiguousarray(off, np.uint64)
        nc = None if name_class is None else np.ascontiguousarray(name_class, np.uint32)
        self._chk(self._l.herro_pool_set_reads(self.h, len(off) - 1, seq.ctypes.data, qual.ctypes.data, off.ctypes.data, None if nc is None else nc.ctypes.data))

    def load_model(self, path: str):
        self._chk(self._l.herro_pool_load_model(self.h, path.encode()))

    def set_precision(self, mode: int):
        for i in range(self.n):
            rc = self._l.herro_set_precision(self._l.herro_pool_ctx(self.h, i), mode)
            if rc:
                raise HerroError(rc, self._l.herro_last_error(self._l.herro_pool_ctx(self.h, i)).decode(errors="replace"))

    def correct(self, rids, rows, aln_off, cig_blob, cig_off, window_size: int, batch: int, read_ids, batch_mode: int = 1, group_targets: int = 1024):
        """FASTA text (u8 array) of all targets in target order + the end offset of every target's records."""
        rids = np.ascontiguousarray(rids, np.uint32)
        aln_off = np.ascontiguousarray(aln_off, np.uint64)
        rows = np.ascontiguousarray(rows, np.uint32).reshape(-1, 10)
        blob = np.ascontiguousarray(cig_blob, np.uint8)
        m = len(rows)
        raw = np.zeros((max(m, 1), 12), np.uint32)            # herro_alignment: 10 u32 + the CIGAR pointer
        if m:
            raw[:m, :10] = rows
            raw[:m, 10:12] = (np.asarray(cig_off, np.uint64) + np.uint64(blob.ctypes.data)).view(np.uint32).reshape(m, 2)
        names = (C.c_char_p * max(len(read_ids), 1))(*[r if isinstance(r, bytes) else r.encode() for r in read_ids])
        if len(read_ids) != len(rids):
            raise ValueError("one read id per target")
        n = self._chk(self._l.herro_pool_correct(self.h, len(rids), rids.ctypes.data, aln_off.ctypes.data, raw.ctypes.data, window_size, batch, batch_mode,
                                                 group_targets, names, None))
        ends_p = C.c_void_p()
        tp = self._l.herro_pool_result(self.h, C.byref(ends_p))
        text = np.ctypeslib.as_array((C.c_uint8 * n).from_address(tp)).copy() if n else np.zeros(0, np.uint8)
        ends = np.ctypeslib.as_array((C.c_uint64 * len(rids)).from_address(ends_p.value)).copy() if len(rids) else np.zeros(0, np.uint64)
        return text, ends

    def groups_taken(self):
        return [int(self._l.herro_pool_groups_taken(self.h, i)) for i in range(self.n)]

    def close(self):
        if self.h:
            self._l.herro_pool_destroy(self.h)
            self.h = None


@dataclass
class Window:
    info: WindowInfo
    bases: np.ndarray     # u8 [L',31]
    quals: np.ndarray     # u8 [L',31]
    sup_pos: np.ndarray   # u16
    sup_ins: np.ndarray   # u8
    qids: np.ndarray      # u32 ranked overlap ids


class Context:
    def __init__(self, device: int = 0):
        self._l = lib()
        self.h = self._l.herro_create(device)
        if not self.h:
            raise HerroError(-2, self._l.herro_last_error(None).decode(errors="replace"))

    def close(self):
        if getattr(self, "h", None):
            self._l.herro_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc: int):
        if rc != 0:
            raise HerroError(rc, self._l.herro_last_error(self.h).decode(errors="replace"))

    def last_error(self) -> str:
        return self._l.herro_last_error(self.h).decode(errors="replace")

    def set_stream(self, hip_stream: int | None):
        self._chk(self._l.herro_set_stream(self.h, hip_stream))

    def synchronize(self):
        self._chk(self._l.herro_synchronize(self.h))

    def set_reads(self, seq: np.ndarray, qual: np.ndarray, off: np.ndarray, name_class: np.ndarray | None = None):
        seq = np.ascontiguousarray(seq, np.uint8)
        qual = np.ascontiguousarray(qual, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        nc = None if name_class is None else np.ascontiguousarray(name_class, np.uint32)
        self._chk(self._l.herro_set_reads(self.h, len(off) - 1, seq.ctypes.data, qual.ctypes.data, off.ctypes.data,
                                          None if nc is None else nc.ctypes.data))

    def share_reads(self, other: "Context"):
        """Adopt the read store of another context of the same device (herro_share_reads): one copy in HBM per device."""
        self._chk(self._l.herro_share_reads(self.h, other.h))

    def load_model(self, path: str):
        self._chk(self._l.herro_load_model(self.h, path.encode()))

    def set_precision(self, mode: int):
        self._chk(self._l.herro_set_precision(self.h, mode))

    def precision(self) -> int:
        """the mode in force: the caller's, or the tier herro_load_model's calibration chose"""
        return int(self._l.herro_precision(self.h))

    def calibration_error(self, mode: int) -> float:
        """max |logit(mode) - logit(mode 0)| on the load-time calibration batch (-1: not measured)"""
        return float(self._l.herro_calibration_error(self.h, mode))

    def clock_probe(self) -> float:
        """shader clock in MHz right behind the work queued on this context's stream (herro_clock_probe; synchronises the stream)"""
        v = C.c_double(0.0)
        self._chk(self._l.herro_clock_probe(self.h, C.byref(v)))
        return float(v.value)

    def force_precision(self, on: bool):
        """test hook (herro_debug_force_precision): set_precision / load_model skip the calibration gate"""
        self._chk(self._l.herro_debug_force_precision(self.h, int(on)))

    def job_arrays(self, job: "Job") -> dict:
        out = {}
        for which, (name, dt) in enumerate([("ops", np.dtype("<u4")), ("ow", OW_DTYPE), ("win", WIN_DTYPE), ("tile_win", np.dtype("<u4")),
                                            ("tile_r0", np.dtype("<u4")), ("tgt_win_off", np.dtype("<u4"))]):
            p, eb = C.c_void_p(), C.c_uint32()
            n = self._l.herro_debug_job_array(job.h, which, C.byref(p), C.byref(eb))
            assert n >= 0 and eb.value == dt.itemsize, (name, n, eb.value, dt.itemsize)
            out[name] = np.frombuffer((C.c_char * (n * dt.itemsize)).from_address(p.value), dt, n).copy() if n else np.zeros(0, dt)
        return out

    def host_build(self, on: bool):
        """test hook: jobs created from now on are windowed / described by the host (rounds 3-5) instead of on the device (build_dev.hip)"""
        self._chk(self._l.herro_debug_set_host_build(self.h, int(on)))

    def featurize_planes(self, on: bool):
        """test hook: jobs featurized from now on take the planes path (k_tokens) instead of the lean one (k_rows)"""
        self._chk(self._l.herro_debug_set_featurize_planes(self.h, int(on)))

    def register_host(self, arr: np.ndarray):
        """herro_host_register: jobs whose CIGAR pointers lie in `arr` (a contiguous u8 array the caller keeps alive) are created zero-copy."""
        self._chk(self._l.herro_host_register(self.h, arr.ctypes.data, arr.nbytes))

    def unregister_host(self, arr: np.ndarray):
        self._chk(self._l.herro_host_unregister(self.h, arr.ctypes.data))

    def describe_model(self) -> str:
        buf = C.create_string_buffer(4096)
        n = self._l.herro_model_describe(self.h, buf, 4096)
        if n < 0:
            self._chk(int(n))
        return buf.value.decode()

    def create_job(self, rids, aln_rows: np.ndarray, aln_off, cigars: list[bytes] | None, window_size: int,
                   cig_blob: np.ndarray | None = None, cig_off: np.ndarray | None = None) -> "Job":
        """aln_rows u32 [n,>=9] (qid,qlen,qstart,qend,strand,tid,tlen,tstart,tend[,cigar_len]); CIGARs either
        as a list of bytes or as blob + offsets (+ lengths in column 9)."""
        rids = np.ascontiguousarray(rids, np.uint32)
        aln_off = np.ascontiguousarray(aln_off, np.uint64)
        n = len(aln_rows)
        if cigars is not None:
            lens = np.array([len(c) for c in cigars], np.uint64)
            cig_off = np.zeros(n, np.uint64)
            if n:
                cig_off[1:] = np.cumsum(lens)[:-1]
            cig_blob = np.frombuffer(b"".join(cigars) + b"\0", np.uint8).copy()
            cl = lens
        else:
            cig_blob = np.ascontiguousarray(cig_blob, np.uint8)
            cl = aln_rows[:, 9].astype(np.uint64)
        arr = (Alignment * max(n, 1))()
        base = cig_blob.ctypes.data
        # vectorised fill through a numpy view of the ctypes array
        view = np.frombuffer(arr, dtype=np.dtype([("f", np.uint32, 10), ("p", np.uint64)], align=True), count=max(n, 1))
        if n:
            view["f"][:n, :9] = aln_rows[:, :9]
            view["f"][:n, 9] = cl
            view["p"][:n] = base + np.asarray(cig_off, np.uint64)
        h = self._l.herro_job_create(self.h, len(rids), rids.ctypes.data, aln_off.ctypes.data, C.byref(arr), window_size)
        if not h:
            msg = self._l.herro_last_error(self.h).decode(errors="replace")
            code = -1
            if "[code " in msg:
                code = int(msg.rsplit("[code ", 1)[1].rstrip("]"))
            raise HerroError(code, msg)
        return Job(self, h, len(rids))

    def create_job_from_paf(self, paf: "Paf", window_size: int) -> "Job":
        """herro_job_create straight from a parsed PAF batch (targets in its order); `paf` must outlive the call."""
        h = self._l.herro_job_create(self.h, len(paf.targets), paf.targets.ctypes.data, paf.aln_off.ctypes.data,
                                     paf._alns_ptr, window_size)
        if not h:
            msg = self._l.herro_last_error(self.h).decode(errors="replace")
            code = int(msg.rsplit("[code ", 1)[1].rstrip("]")) if "[code " in msg else -1
            raise HerroError(code, msg)
        return Job(self, h, len(paf.targets))

    def model_forward(self, bases: np.ndarray, quals: np.ndarray, lens: np.ndarray, indices: np.ndarray):
        """inference.rs:147-175: tokens u8 [B,L,31], raw quals u8 [B,L,31], lens, flat indices -> logits."""
        bases = np.ascontiguousarray(bases, np.uint8)
        quals = np.ascontiguousarray(quals, np.uint8)
        lens = np.ascontiguousarray(lens, np.int32)
        indices = np.ascontiguousarray(indices, np.int32)
        B, L, R = bases.shape
        assert R == 31 and quals.shape == bases.shape
        N = int(lens.sum())
        info = np.zeros(N, np.float32)
        base = np.zeros((N, 5), np.float32)
        self._chk(self._l.herro_model_forward(self.h, B, L, bases.ctypes.data, quals.ctypes.data, lens.ctypes.data,
                                              indices.ctypes.data if N else None, info.ctypes.data, base.ctypes.data))
        return info, base

    def sib_fault(self):
        """test hook: what a sibling tile that timed out leaves behind (the next fetch repeats its job's model pass)"""
        self._chk(self._l.herro_debug_sib_fault(self.h))

    def sib_retries(self) -> int:
        return int(self._l.herro_debug_sib_retries(self.h))

    def timing_enable(self, on: bool = True):
        self._chk(self._l.herro_timing_enable(self.h, int(on)))

    def timing_reset(self):
        self._chk(self._l.herro_timing_reset(self.h))

    def timing(self) -> dict[str, tuple[float, int]]:
        n = C.c_uint32(256)
        names = C.create_string_buffer(8192)
        ms = (C.c_double * 256)()
        calls = (C.c_uint64 * 256)()
        self._chk(self._l.herro_timing_get(self.h, names, 8192, ms, calls, C.byref(n)))
        nm = [x for x in names.value.decode().split("\n") if x]
        return {nm[i]: (ms[i], int(calls[i])) for i in range(min(n.value, len(nm)))}


class Job:
    def __init__(self, ctx: Context, h, n_targets: int):
        self.ctx, self.h, self.n_targets = ctx, h, n_targets
        self._l = ctx._l

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self._l.herro_job_free(self.h)
        self.h = None

    __del__ = close

    @property
    def n_windows(self) -> int:
        return self._l.herro_job_n_windows(self.h)

    def skipped(self) -> tuple[int, int]:
        """(alignments left out, targets left without overlaps) — see herro_job_skipped"""
        a, t = C.c_uint32(0), C.c_uint32(0)
        self.ctx._chk(self._l.herro_job_skipped(self.h, C.byref(a), C.byref(t)))
        return a.value, t.value

    def featurize(self):
        self.ctx._chk(self._l.herro_job_featurize(self.h))

    def infer(self, batch_size: int, batch_mode: int = 0):
        self.ctx._chk(self._l.herro_job_infer(self.h, batch_size, batch_mode))

    def consensus(self):
        """consensus.rs:86-227 on the device; consensus_fasta() then only concatenates windows."""
        self.ctx._chk(self._l.herro_job_consensus(self.h))

    def consensus_fetch(self) -> int:
        """corrected bases of the whole job -> host; returns how many (herro_job_consensus_fetch)"""
        n = C.c_uint64(0)
        self.ctx._chk(self._l.herro_job_consensus_fetch(self.h, C.byref(n)))
        return n.value

    def info(self, w: int) -> WindowInfo:
        wi = WindowInfo()
        self.ctx._chk(self._l.herro_job_window_info(self.h, w, C.byref(wi)))
        return wi

    def window(self, w: int, encoded: bool = False) -> Window:
        wi = self.info(w)
        bases = np.zeros((wi.length, 31), np.uint8)
        quals = np.zeros((wi.length, 31), np.uint8)
        sp = np.zeros(wi.n_supported, np.uint16)
        si = np.zeros(wi.n_supported, np.uint8)
        qids = np.zeros(wi.n_overlaps, np.uint32)
        self.ctx._chk(self._l.herro_job_window_copy(self.h, w, int(encoded), bases.ctypes.data, quals.ctypes.data,
                                                    sp.ctypes.data, si.ctypes.data, qids.ctypes.data))
        return Window(wi, bases, quals, sp, si, qids)

    def logits(self, w: int):
        wi = self.info(w)
        info = np.zeros(wi.n_supported, np.float32)
        base = np.zeros((wi.n_supported, 5), np.float32)
        self.ctx._chk(self._l.herro_job_window_logits(self.h, w, info.ctypes.data, base.ctypes.data))
        return info, base

    def consensus_fasta(self, t: int, read_id: str, desc: str | None = None) -> str:
        cap = 1 << 24
        out = getattr(self, "_fasta_buf", None)
        if out is None:                     # one buffer per job: allocating (and zeroing) 16 MB per target dominated multi-target runs
            out = self._fasta_buf = C.create_string_buffer(cap)
        n = self._l.herro_job_consensus_fasta(self.h, t, read_id.encode(), None if desc is None else desc.encode(), out, cap)
        if n < 0:
            self.ctx._chk(int(n))
        return C.string_at(out, n).decode()   # (out.raw would copy the whole 16 MB buffer per call)

    def fasta(self, read_ids, with_ends: bool = False, as_array: bool = False, out_alloc=None):
        """FASTA records of every target of the job, target order (herro_job_fasta: one sizing call, one call that writes the
        text straight into a numpy buffer; the library's thread pool assembles it).  read_ids: one str/bytes per target.
        with_ends: also the end offset of every target's records.  as_array: the text as a u8 array (no bytes copy).
        out_alloc(nbytes) -> u8 array: where the text goes (a reusable buffer: a fresh 16 MB array costs more in page faults than
        the library needs to fill it)."""
        n = len(read_ids)
        if n != self.n_targets:   # the library reads one id and writes one end offset per target of the job
            raise ValueError(f"read_ids has {n} entries, the job has {self.n_targets} targets")
        arr = (C.c_char_p * max(n, 1))(*[r if isinstance(r, bytes) else r.encode() for r in read_ids])
        ends = np.zeros(max(n, 1), np.uint64)
        need = self._l.herro_job_fasta(self.h, arr, None, None, 0, ends.ctypes.data)
        if need < 0:
            self.ctx._chk(int(need))
        out = np.empty(max(int(need), 1), np.uint8) if out_alloc is None else out_alloc(max(int(need), 1))
        got = self._l.herro_job_fasta(self.h, arr, None, out.ctypes.data, int(need), None)
        if got < 0:
            self.ctx._chk(int(got))
        text = out[:got] if as_array else out[:got].tobytes()
        return (text, ends[:n]) if with_ends else text

    def rf_left(self) -> int:
        """test hook: windows of the last infer whose receptive fields k_rfq filled BEHIND a fused gather (more informative rows than k_rows stages)"""
        rc = self._l.herro_debug_job_rf_left(self.h)
        if rc < 0:
            self.ctx._chk(rc)
        return rc

    def rf_fused(self) -> bool:
        """test hook: the last infer read the receptive fields k_rows gathered itself (False: k_rfq's)"""
        rc = self._l.herro_debug_job_rf_fused(self.h)
        if rc < 0:
            self.ctx._chk(rc)
        return rc == 1

    def rf_records(self, w: int) -> np.ndarray:
        """test hook: the receptive-field records the model read for window w, [n_supported, 31, 16] (bytes 0..7 tokens, 8..15 qualities)"""
        ns = self.info(w).n_supported
        out = np.zeros((ns, 31, 16), np.uint8)
        n = self._l.herro_debug_job_rf(self.h, w, out.ctypes.data, out.nbytes)
        if n < 0:
            self.ctx._chk(int(n))
        assert n == ns * 31
        return out

    def stats(self) -> dict[str, int]:
        o = np.zeros(6, np.uint64)
        self.ctx._chk(self._l.herro_job_stats(self.h, o.ctypes.data))
        k = ("read_bytes", "op_bytes", "out_bytes", "sum_len", "sum_supported", "n_model_windows")
        return {a: int(b) for a, b in zip(k, o)}


OW_DTYPE = np.dtype([(n, "<u4") for n in ("win", "qid", "cls", "tstart", "qbeg", "qlen", "op_begin", "op_cnt", "start_off",
                                            "end_off", "scr_off", "strand", "wtstart", "wlen")] +
                    [(n, "<u8") for n in ("t_woff", "q_woff", "q_qual_off")], align=True)   # OwDesc (csrc/pileup_core.h)
WIN_DTYPE = np.dtype([(n, "<u4") for n in ("rid", "wid", "n_wids", "tstart", "win_len", "ow_begin", "ow_cnt", "lub")] +
                     [(n, "<u8") for n in ("col_off", "fin_off", "row_off", "pos_off", "ev_off")], align=True)   # WinDesc


class HostContext(Context):
    """Device-free context for testing the host half of herro_job_create (herro_debug_host_ctx)."""

    def __init__(self, read_len, name_class=None):
        self._l = lib()
        rl = np.ascontiguousarray(read_len, np.uint32)
        nc = None if name_class is None else np.ascontiguousarray(name_class, np.uint32)
        self.h = self._l.herro_debug_host_ctx(len(rl), rl.ctypes.data, None if nc is None else nc.ctypes.data)

class NameIndex:
    """read name -> read id, built once per read set (herro_name_index_create; the reference's `name_to_id`, lib.rs:136-140)."""

    def __init__(self, names: list[bytes]):
        self.h = None
        L = lib()
        blob = b"".join(names)
        off = np.zeros(len(names) + 1, np.uint64)
        off[1:] = np.cumsum([len(n) for n in names])
        self._l, self.n = L, len(names)
        self.h = L.herro_name_index_create(len(names), blob, off.ctypes.data)
        if not self.h:
            raise HerroError(-1, "herro_name_index_create")

    def close(self):
        if self.h:
            self._l.herro_name_index_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Paf:
    """Parsed PAF / .oec.zst batch (host only; overlaps.rs:117-202, 292-323): targets in order of first
    appearance, their alignments in file order.  `targets`, `aln_off`, `alns` are what herro_job_create takes;
    the CIGAR pointers inside `alns` stay valid while this object lives.  `names`: the read ids, or a NameIndex built once."""

    def __init__(self, names, text: bytes | None = None, path: str | None = None, core=None, threads: int = 0, view: bool = False):
        """view=True (with a NameIndex and `text`): no copy of the text — this object keeps `text` alive instead."""
        self.h = None
        self._text = text if view else None
        L = lib()
        core_a = None if core is None else np.ascontiguousarray(core, np.uint8)
        err = C.create_string_buffer(512)
        cptr = None if core_a is None else core_a.ctypes.data
        if isinstance(names, NameIndex):
            if text is not None and view:
                h = L.herro_paf_parse_view(text, len(text), names.h, cptr, threads, err, 512)
            elif text is not None:
                h = L.herro_paf_parse_indexed(text, len(text), names.h, cptr, threads, err, 512)
            else:
                h = L.herro_oec_read_indexed(path.encode(), names.h, cptr, threads, err, 512)
        else:
            blob = b"".join(names)
            off = np.zeros(len(names) + 1, np.uint64)
            off[1:] = np.cumsum([len(n) for n in names])
            if text is not None:
                h = L.herro_paf_parse(text, len(text), len(names), blob, off.ctypes.data, cptr, threads, err, 512)
            else:
                h = L.herro_oec_read(path.encode(), len(names), blob, off.ctypes.data, cptr, threads, err, 512)
        if not h:
            raise HerroError(-3, err.value.decode())
        self._l, self.h = L, h
        n = L.herro_paf_n_targets(h)
        self.targets = np.ctypeslib.as_array(C.cast(L.herro_paf_target_ids(h), C.POINTER(C.c_uint32)), (n,)).copy() if n else np.zeros(0, np.uint32)
        self.aln_off = np.ctypeslib.as_array(C.cast(L.herro_paf_aln_off(h), C.POINTER(C.c_uint64)), (n + 1,)).copy()
        na = int(self.aln_off[-1])
        self.n_alns = na
        self._alns_ptr = L.herro_paf_alignments(h)
        self.alns = (Alignment * na).from_address(self._alns_ptr) if na else []

    def rows(self):
        """[(qid, qlen, qstart, qend, strand, tid, tlen, tstart, tend, cigar bytes)] in output order."""
        out = []
        for a in self.alns:
            out.append((a.qid, a.qlen, a.qstart, a.qend, a.strand, a.tid, a.tlen, a.tstart, a.tend,
                        C.string_at(a.cigar, a.cigar_len)))
        return out

    def close(self):
        if self.h:
            self._l.herro_paf_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


class PreparedAlignments:
    """All alignments of a data set as ONE herro_alignment array, grouped by target — what a host holds after parsing its
    PAF / .oec.zst input (the reference's reader thread hands `(tid, Vec<Alignment>)` to the feature threads,
    lib.rs:141-151).  Jobs over target ranges are then pure C calls: no per-job packing on the Python side."""

    def __init__(self, sb):
        n = len(sb.aln)
        self.rids = np.ascontiguousarray(sb.tgt_rid, np.uint32)
        self.aln_off = np.ascontiguousarray(sb.tgt_aln_off, np.uint64)
        self._cig = np.ascontiguousarray(sb.cig, np.uint8)
        self._arr = (Alignment * max(n, 1))()
        view = np.frombuffer(self._arr, dtype=np.dtype([("f", np.uint32, 10), ("p", np.uint64)], align=True), count=max(n, 1))
        if n:
            view["f"][:n, :10] = sb.aln[:, :10]
            view["p"][:n] = self._cig.ctypes.data + np.asarray(sb.cig_off, np.uint64)
        self.n_targets = len(self.rids)
        self._reg = None

    def register(self, ctx: "Context"):
        """pin the CIGAR blob for zero-copy job creation (herro_host_register); undo with unregister() before the arrays go away"""
        if self._reg is None and len(self._cig):
            try:
                ctx.register_host(self._cig)
            except HerroError as e:       # e.g. the memlock limit: jobs are staged instead (the zero-copy path is an optimisation, never a requirement)
                if e.code != -4:   # HERRO_E_UNSUPPORTED
                    raise
                return
            self._reg = True

    def unregister(self):
        if self._reg is not None:
            rc = lib().herro_host_unregister(None, self._cig.ctypes.data)   # the registry is process-wide: no context needed (the one that registered may be closed by now)
            if rc:
                raise HerroError(int(rc), "herro_host_unregister")
            self._reg = None

    def job(self, ctx: "Context", t0: int, t1: int, window_size: int) -> "Job":
        """herro_job_create over targets [t0, t1): rids / aln_off are passed as offsets into the resident arrays"""
        h = ctx._l.herro_job_create(ctx.h, t1 - t0, self.rids.ctypes.data + 4 * t0, self.aln_off.ctypes.data + 8 * t0,
                                    C.byref(self._arr), window_size)
        if not h:
            msg = ctx._l.herro_last_error(ctx.h).decode(errors="replace")
            code = int(msg.rsplit("[code ", 1)[1].rstrip("]")) if "[code " in msg else -1
            raise HerroError(code, msg)
        return Job(ctx, h, t1 - t0)


def job_from_synth(ctx: Context, sb, window_size: int, targets=None) -> Job:
    """Job over targets of a SynthBatch (all by default)."""
    ts = list(range(sb.n_targets)) if targets is None else list(targets)
    a0 = [int(sb.tgt_aln_off[t]) for t in ts]
    a1 = [int(sb.tgt_aln_off[t + 1]) for t in ts]
    sel = np.concatenate([np.arange(x, y) for x, y in zip(a0, a1)]).astype(np.int64) if ts else np.zeros(0, np.int64)
    off = np.zeros(len(ts) + 1, np.uint64)
    off[1:] = np.cumsum([y - x for x, y in zip(a0, a1)])
    rows = sb.aln[sel]
    return ctx.create_job(sb.tgt_rid[ts], rows, off, None, window_size, cig_blob=sb.cig, cig_off=sb.cig_off[sel])
