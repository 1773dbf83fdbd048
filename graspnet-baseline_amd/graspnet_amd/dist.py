"""Multi-GPU sharding of batched scenes.

Scenes are independent, so a global batch is cut into contiguous per-rank shards
(no data-path collective); the only exchange is ONE all-gather of the padded
grasp lists after pred_decode: grasps (B_local, M, 17) f32 + counts (B_local,) i32
per rank (557 KB at B_local=8), RCCL over xGMI via torch.distributed ("nccl"
backend on ROCm; "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(global_batch, rank, world_size):
    """Contiguous shard [lo, hi) of rank; the first (global_batch % world_size) ranks get one extra scene."""
    base, rem = divmod(global_batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_grasps(grasps, counts, group=None):
    """grasps (Bl,M,17), counts (Bl,) on every rank (same Bl) -> (world*Bl,M,17), (world*Bl,) on every rank."""
    world = dist.get_world_size(group)
    if world == 1:
        return grasps, counts
    g_all = torch.empty((world * grasps.shape[0],) + tuple(grasps.shape[1:]), dtype=grasps.dtype, device=grasps.device)
    c_all = torch.empty((world * counts.shape[0],), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(g_all, grasps.contiguous(), group=group)
    dist.all_gather_into_tensor(c_all, counts.contiguous(), group=group)
    return g_all, c_all


def to_list(grasps, counts):
    """padded (B,M,17) + counts -> the reference's list[Tensor (Ns_i,17)] (host sync on counts)."""
    return [grasps[i, :c] for i, c in enumerate(counts.cpu().tolist())]
