"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding + all-gather plumbing that
bench.py / the clustering stage use over RCCL on the GPUs."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, out_dir):
    import importlib
    import sys
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D = importlib.import_module(f"{PKG}.dist")
    full = torch.from_numpy(np.random.default_rng(0).standard_normal((n_total, 192)).astype(np.float32))
    lo, hi = D.shard_range(n_total)
    gathered = D.all_gather_rows(full[lo:hi].clone(), n_total)
    ok = torch.equal(gathered, full)
    s = D.all_reduce_sum(torch.tensor([float(hi - lo)]))
    ok = ok and int(s.item()) == n_total
    # ragged + empty shards
    lo1, hi1 = D.shard_range(1)                       # rank 0 owns the only row, rank 1 an empty shard
    g2 = D.all_gather_rows(full[lo1:hi1].clone(), 1)
    ok = ok and torch.equal(g2, full[:1])
    open(os.path.join(out_dir, f"rank{rank}.ok" if ok else f"rank{rank}.bad"), "w").close()
    dist.destroy_process_group()


def test_shard_bounds_cover_rows_exactly():
    import importlib
    D = importlib.import_module(f"{PKG}.dist")
    for n in (0, 1, 7, 8, 1000, 1_000_003):
        for w in (1, 2, 3, 8):
            b = D.shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [h - l for l, h in b]
            assert max(sizes) - min(sizes) <= 1
    assert D.shard_bounds(1_000_000, 8)[3] == (375_000, 500_000)     # config #4: 125k segments per GPU


def test_all_gather_rows_world2_gloo(tmp_path):
    world = 2
    for n_total in (11, 64):
        d = tmp_path / f"n{n_total}"
        d.mkdir()
        mp.spawn(_worker, args=(world, _free_port(), n_total, str(d)), nprocs=world, join=True)
        assert sorted(os.listdir(d)) == ["rank0.ok", "rank1.ok"]
