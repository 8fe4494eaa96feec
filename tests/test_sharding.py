"""Multi-GPU path on CPU: world_size-2 gloo processes agree on the partition of target reads, cover
every read exactly once, balance windows, and the summed per-rank statistics equal the single-process
totals (the path has no data-path collective — SURVEY.md §8 e)."""
import os
import subprocess
import sys
import textwrap

import numpy as np

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
