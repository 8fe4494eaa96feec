"""Multi-GPU sharding of the hot path (SURVEY.md §8 e).

The independent unit of feature generation is the TARGET READ (the haplotype re-ranking couples all
windows of a read, reference features.rs:462-500; consensus needs all windows of a read on one worker,
consensus.rs:249).  The reference runs one replica per `-d` device pulling reads from one shared queue
(lib.rs:154-200) and never communicates between devices.  Here: one process per GPU; the read store is
replicated; target reads are partitioned statically, balanced by window count (longest first); there is
NO data-path collective.  torch.distributed is used only to agree on the partition's totals and to
gather small result summaries (corrected bases per read are written by the rank that owns the read).
"""
from __future__ import annotations

import numpy as np


def partition_targets(n_windows_per_target: np.ndarray, world_size: int) -> list[np.ndarray]:
    """Greedy longest-first partition of target indices into `world_size` shards with balanced window
    counts.  Deterministic: every rank computes the same partition from the same lengths."""
    order = np.argsort(-np.asarray(n_windows_per_target, dtype=np.int64), kind="stable")
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
