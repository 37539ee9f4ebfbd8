"""Multi-GPU: independent frames (BASELINE configs 3/5) shard round-robin over ranks; one process per GPU, no
data-path collective.  torch.distributed is used only for the barrier / max-over-ranks timing."""


def shard_indices(n_items: int, rank: int, world: int):
    return list(range(rank, n_items, world))


def max_over_ranks(value: float) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
