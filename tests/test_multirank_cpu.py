"""world_size-2 gloo test of the batch sharding used by bench.py / decode_batch across ranks (no data-path collective)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, n_items):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jxl_coder_amd.shard import shard_indices, max_over_ranks
    mine = shard_indices(n_items, rank, world)
    t = torch.zeros(n_items, dtype=torch.int64)
    t[mine] = 1
    dist.all_reduce(t)
    assert bool((t == 1).all()), "every frame decoded by exactly one rank"
    m = max_over_ranks(float(rank + 1))
    assert m == float(world)
    dist.destroy_process_group()


def test_frames_shard_exactly_once_over_two_ranks():
    mp.spawn(_worker, args=(2, 29611, 37), nprocs=2, join=True)


def test_shard_balance():
    from jxl_coder_amd.shard import shard_indices
    for n in (0, 1, 7, 256):
        for world in (1, 2, 4, 8):
            parts = [shard_indices(n, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
