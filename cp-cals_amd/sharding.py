"""Multi-GPU sharding of independent CPD models.

Models are independent given X, so the path shards with NO data-path collective: rank g owns the
models m with m % world == g (round-robin, BASELINE.json config 5), X is replicated, every rank runs
the single-GPU engine.  torch.distributed (RCCL on GPU, gloo on CPU) is used only for the barrier
around the timed region, the MAX over ranks of the elapsed time and the gather of per-rank counts.
"""
import torch
import torch.distributed as dist


def shard_round_robin(n_models, world, rank):
    """Global model indices owned by `rank`."""
    return list(range(rank, n_models, world))


def weak_scaling_models(models_per_gpu, world):
    """Total model count of the weak-scaling workload (per-GPU work fixed)."""
    return models_per_gpu * world


def max_over_ranks(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value, device="cpu"):
    """[value of rank 0, value of rank 1, ...] on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [float(p.item()) for p in parts]


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def aggregate_rate(local_units, local_seconds, device="cpu"):
    """Whole-job rate = units processed by all ranks / MAX over ranks of the elapsed time."""
    total = sum_over_ranks(local_units, device)
    t = max_over_ranks(local_seconds, device)
    return total / t, t


def strong_rate(job_steps, local_seconds, device="cpu"):
    """Strong scaling (BASELINE config 5): the job's models are split over the ranks and one step advances
    the WHOLE job by one sweep, so the job rate is steps / MAX over ranks of the elapsed time."""
    t = max_over_ranks(local_seconds, device)
    return job_steps / t, t
