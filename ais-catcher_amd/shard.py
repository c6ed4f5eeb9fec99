"""Multi-GPU sharding of independent receivers: no data-path collective.

Receivers are closed systems, so rank r of W simply owns a contiguous range of receiver ids; the only
communication is the benchmark's barrier and the max-over-ranks of the elapsed time (torch.distributed on
gloo, on the GPU box as in the CPU tests: there is no RCCL traffic on this path).
"""


def receiver_range(rank, world, receivers_per_gpu):
    """Global receiver ids owned by `rank` (weak scaling: every rank owns receivers_per_gpu receivers)."""
    return range(rank * receivers_per_gpu, (rank + 1) * receivers_per_gpu)


def owner_of(receiver_id, receivers_per_gpu):
    return receiver_id // receivers_per_gpu


def max_over_ranks(seconds, dist=None, device="cpu"):
    """Job time = the slowest rank's time."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_msamples(samples_per_rank_step, world, steps, seconds):
    """Whole-job IQ Msamples/s over all ranks."""
    return samples_per_rank_step * world * steps / seconds / 1e6
