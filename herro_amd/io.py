"""Host-side file formats on either side of the hot path.

* `read_fastq`  — what `get_reads` (haec_io.rs:37-75) hands to the path: records shorter than `min_length`
  dropped, the header split at the first blank or tab into id / description, qualities mandatory, and the
  `core` / `neighbour` filter (a read is kept if it is in either set; applied only when both are given).
  Plain or gzip FASTQ, four lines per record.
* `write_window_features` / `write_job_features` — the `herro features` sink (features.rs:724-764, 818-833):
  `<base>/<read id>/<wid>.features.npy` = u8 `[2, L', 31]` (ASCII bases, then qualities),
  `<wid>.supported.npy` = records `{pos: <u2, ins: u1}`, `<wid>.ids.txt` = ranked overlap read ids, one per line.
  The reference writes NPY through the `npyz` crate; these files carry the same dtype / shape / C order and load
  identically with numpy (header bytes were not compared against a real `herro features` run: no Rust here).
"""
from __future__ import annotations

import dataclasses
import gzip
import os

import numpy as np

SUPPORTED_DTYPE = np.dtype([("pos", "<u2"), ("ins", "u1")])


@dataclasses.dataclass
class Reads:
    ids: list[bytes]
    descriptions: list[bytes | None]
    seq: np.ndarray   # u8, all reads back to back
    qual: np.ndarray  # u8, same layout
    off: np.ndarray   # u64 [n+1]


def read_fastq(path: str, min_length: int = 0, core: set[str] | None = None, neighbour: set[str] | None = None) -> Reads:
    opener = gzip.open if path.endswith(".gz") else open
    ids, descs, seqs, quals = [], [], [], []
    with opener(path, "rb") as f:
        while True:
            h = f.readline()
            if not h:
                break
            s, plus, q = f.readline(), f.readline(), f.readline()
            if not h.startswith(b"@") or not plus.startswith(b"+"):
                raise ValueError("Error parsing fastx file.")          # needletail error -> expect() panic
            h, s, q = h.rstrip(b"\r\n")[1:], s.rstrip(b"\r\n"), q.rstrip(b"\r\n")
            if len(q) != len(s):
                raise ValueError("Error parsing fastx file.")
            if len(s) < min_length:                                     # haec_io.rs:48-50
                continue
            cut = min((i for i in (h.find(b" "), h.find(b"\t")) if i >= 0), default=-1)   # splitn(2, ' ' | '\t')
            rid, desc = (h, None) if cut < 0 else (h[:cut], h[cut + 1:])
            if core is not None and neighbour is not None:             # haec_io.rs:63-69
                name = rid.decode()
                if name not in neighbour and name not in core:
                    continue
            ids.append(rid); descs.append(desc); seqs.append(s); quals.append(q)
    off = np.zeros(len(ids) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    return Reads(ids, descs, np.frombuffer(b"".join(seqs), np.uint8).copy(), np.frombuffer(b"".join(quals), np.uint8).copy(), off)


def write_window_features(out_dir: str, wid: int, ids: list[str], bases: np.ndarray, quals: np.ndarray,
                          sup_pos: np.ndarray, sup_ins: np.ndarray) -> None:
    """bases: ASCII u8 [L',31]; quals u8 [L',31]; supported positions in order (features.rs:724-764)."""
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"{wid}.ids.txt"), "w") as f:
        for i in ids:
            f.write(i + "\n")
    feats = np.ascontiguousarray(np.stack([np.asarray(bases, np.uint8), np.asarray(quals, np.uint8)], axis=0))
    np.save(os.path.join(out_dir, f"{wid}.features.npy"), feats)
    sup = np.zeros(len(sup_pos), SUPPORTED_DTYPE)
    sup["pos"], sup["ins"] = sup_pos, sup_ins
    np.save(os.path.join(out_dir, f"{wid}.supported.npy"), sup)


def write_job_features(job, base_dir: str, read_name, target_windows) -> int:
    """`herro features` for a featurized job: target_windows = [(target read id, first window, n windows)];
    read_name(rid) -> str.  Pulls ASCII bases + full quality planes through herro_job_window_copy."""
    n = 0
    for rid, w0, nw in target_windows:
        d = os.path.join(base_dir, read_name(rid))
        for k in range(nw):
            win = job.window(w0 + k, encoded=False)
            write_window_features(d, win.info.wid, [read_name(int(q)) for q in win.qids], win.bases, win.quals,
                                  win.sup_pos, win.sup_ins)
            n += 1
    return n
