"""Host-side file formats on either side of the hot path: thin ctypes wrappers over the C ABI
(csrc/fastx.cpp, include/herro_amd.h).

* `read_fastx`  — herro_fastx_read = get_reads (haec_io.rs:37-75) over needletail's record rules: FASTA / FASTQ, gzip,
  multi-line records; records shorter than `min_length` dropped, the header split at the first blank or tab into id /
  description, qualities mandatory, the `core` / `neighbour` filter (a read is kept if it is in either set; applied only
  when both are given).
* `write_window_features` / `write_job_features` — the `herro features` sink (features.rs:724-764, 783-839):
  `<base>/<read id>/<wid>.features.npy` = u8 `[2, L', 31]` (ASCII bases, then qualities), `<wid>.supported.npy` = records
  `{pos: <u2, ins: u1}`, `<wid>.ids.txt` = ranked overlap read ids, one per line.  NPY format 1.0 with the header numpy
  itself writes (the reference goes through the `npyz` crate; byte equality with a real `herro features` run was not
  checked: no Rust here).
"""
from __future__ import annotations

import ctypes as C
import dataclasses

import numpy as np

from . import api

SUPPORTED_DTYPE = np.dtype([("pos", "<u2"), ("ins", "u1")])


@dataclasses.dataclass
class Reads:
    ids: list[bytes]
    descriptions: list[bytes | None]
    seq: np.ndarray   # u8, all reads back to back
    qual: np.ndarray  # u8, same layout
    off: np.ndarray   # u64 [n+1]


def _lib():
    L = api.lib()
    if not getattr(L, "_io_ready", False):
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        L.herro_fastx_read.restype = vp
        L.herro_fastx_read.argtypes = [C.c_char_p, u32, vp, u64, vp, u64]
        L.herro_reads_count.restype = u32
        L.herro_reads_count.argtypes = [vp]
        for n in ("herro_reads_seq", "herro_reads_qual", "herro_reads_off", "herro_reads_ids", "herro_reads_descs"):
            getattr(L, n).restype = vp
            getattr(L, n).argtypes = [vp]
        L.herro_reads_free.restype = None
        L.herro_reads_free.argtypes = [vp]
        L.herro_write_window_features.restype = C.c_int
        L.herro_write_window_features.argtypes = [C.c_char_p, u32, vp, u32, vp, vp, u32, vp, vp, u32]
        L.herro_job_write_features.restype = C.c_int64
        L.herro_job_write_features.argtypes = [vp, C.c_char_p, vp]
        L._io_ready = True
    return L


def read_fastx(path: str, min_length: int = 0, core: set[str] | None = None, neighbour: set[str] | None = None) -> Reads:
    L = _lib()
    keep, n_keep = None, 0
    if core is not None and neighbour is not None:             # haec_io.rs:63: the filter needs both sets
        names = sorted(set(core) | set(neighbour))
        keep = (C.c_char_p * max(len(names), 1))(*[n.encode() for n in names])
        n_keep = len(names)
    err = C.create_string_buffer(256)
    h = L.herro_fastx_read(path.encode(), min_length, keep, n_keep, err, 256)
    if not h:
        raise ValueError(err.value.decode())
    try:
        n = L.herro_reads_count(h)
        def take(addr, count, dt):       # one memcpy out of the library's buffer (np.ctypeslib.as_array builds a ctypes array TYPE of that length first: seconds per GB)
            out = np.empty(count, dt)
            if count:
                C.memmove(out.ctypes.data, addr, out.nbytes)
            return out
        off = take(L.herro_reads_off(h), n + 1, np.uint64)
        nb = int(off[-1])
        seq = take(L.herro_reads_seq(h), nb, np.uint8)
        qual = take(L.herro_reads_qual(h), nb, np.uint8)
        idp = C.cast(L.herro_reads_ids(h), C.POINTER(C.c_char_p))
        dp = C.cast(L.herro_reads_descs(h), C.POINTER(C.c_char_p))
        ids = [idp[i] for i in range(n)]
        descs = [dp[i] for i in range(n)]
    finally:
        L.herro_reads_free(h)
    return Reads(ids, descs, seq, qual, off)


read_fastq = read_fastx   # the name the first rounds used


def write_window_features(out_dir: str, wid: int, ids: list[str], bases: np.ndarray, quals: np.ndarray,
                          sup_pos: np.ndarray, sup_ins: np.ndarray) -> None:
    """bases: ASCII u8 [L',31]; quals u8 [L',31]; supported positions in order (features.rs:724-764)."""
    L = _lib()
    bases = np.ascontiguousarray(bases, np.uint8)
    quals = np.ascontiguousarray(quals, np.uint8)
    sp = np.ascontiguousarray(sup_pos, np.uint16)
    si = np.ascontiguousarray(sup_ins, np.uint8)
    arr = (C.c_char_p * max(len(ids), 1))(*[i.encode() for i in ids])
    rc = L.herro_write_window_features(out_dir.encode(), wid, arr, len(ids), bases.ctypes.data, quals.ctypes.data, bases.shape[0],
                                       sp.ctypes.data, si.ctypes.data, len(sp))
    if rc:
        raise OSError(f"herro_write_window_features failed ({rc}) under {out_dir}")


def write_job_features(job, base_dir: str, read_names: list[str]) -> int:
    """`herro features` for a featurized job (herro_job_write_features): read_names[rid] for every read of the store."""
    L = _lib()
    arr = (C.c_char_p * max(len(read_names), 1))(*[n.encode() for n in read_names])
    n = L.herro_job_write_features(job.h, base_dir.encode(), arr)
    if n < 0:
        job.ctx._chk(int(n))
    return int(n)
