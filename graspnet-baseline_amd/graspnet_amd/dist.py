"""Multi-GPU sharding of batched scenes.

Scenes are independent, so a global batch is cut into contiguous per-rank shards
(no data-path collective); the only exchange is ONE all-gather of the padded
grasp lists after pred_decode: grasps (B_local, M, 17) f32 + counts (B_local,) i32
per rank (557 KB at B_local=8; the counts ride in the same buffer), RCCL over xGMI via torch.distributed ("nccl"
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
    """grasps (Bl,M,17) f32, counts (Bl,) i32 on every rank (same Bl) -> (world*Bl,M,17), (world*Bl,) on every rank.

    ONE collective per step: each scene's count rides behind its M*17 grasp floats (bit-cast), so a rank
    sends Bl*(M*17+1) floats; the results are views of the gathered buffer (grasps: rows contiguous,
    scene stride M*17+1)."""
    world = dist.get_world_size(group)
    if world == 1:
        return grasps, counts
    bl, m, w = grasps.shape
    send = torch.empty((bl, m * w + 1), dtype=torch.float32, device=grasps.device)
    send[:, :m * w] = grasps.reshape(bl, m * w)
    send[:, m * w] = counts.to(torch.int32).view(torch.float32)
    recv = torch.empty((world * bl, m * w + 1), dtype=torch.float32, device=grasps.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    return recv[:, :m * w].unflatten(1, (m, w)), recv[:, m * w].view(torch.int32)


def to_list(grasps, counts):
    """padded (B,M,17) + counts -> the reference's list[Tensor (Ns_i,17)] (host sync on counts)."""
    return [grasps[i, :c] for i, c in enumerate(counts.cpu().tolist())]
