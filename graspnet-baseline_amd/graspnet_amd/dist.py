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


class GatherHandle:
    """Completion handle of one side-stream all-gather: `wait()` makes the caller's CURRENT stream depend on the collective
    and registers the gathered buffer with that stream (it was allocated on the gatherer's side stream: without the
    registration the caching allocator could hand the block to the next gather while a consumer kernel on another stream
    is still reading it); `synchronize()` blocks the host instead.  `event` is the raw completion event."""

    def __init__(self, event, recv):
        self.event, self._recv = event, recv

    def wait(self, stream=None):
        cur = stream if stream is not None else torch.cuda.current_stream(self._recv.device)
        cur.wait_event(self.event)
        self._recv.record_stream(cur)
        return self

    def synchronize(self):
        self.event.synchronize()
        self._recv.record_stream(torch.cuda.current_stream(self._recv.device))
        return self

    def query(self):
        return self.event.query()


class GraspGatherer:
    """The all-gather of SURVEY.md 8e as specified there: on a SIDE stream, overlapped with the next batch's kernels.

    gather() packs the rank's padded lists on the caller's current stream, records an event, and issues the collective on
    the gatherer's own stream behind that event; the caller's stream is never blocked by the collective and goes straight
    on to its next batch (FPS of the next batch runs while RCCL moves 557 KB per rank over xGMI).  Returns the gathered
    views plus a `GatherHandle`: a consumer calls `handle.wait()` on the stream it reads them from (or
    `handle.synchronize()` on the host) before touching them.  The gathered buffer is also registered with the stream
    gather() was called on, so the common case -- the consumer is that same stream -- is safe after `wait()` or any
    device synchronize.  CPU tensors (gloo tests) gather inline and return no handle."""

    def __init__(self, device=None, group=None):
        self.group = group
        self.device = torch.device(device) if device is not None else None
        self.side = torch.cuda.Stream(device=self.device) if self.device is not None and self.device.type == "cuda" else None

    def gather(self, grasps, counts):
        world = dist.get_world_size(self.group)
        if world == 1 or self.side is None:
            g, c = all_gather_grasps(grasps, counts, self.group)
            return g, c, None
        bl, m, w = grasps.shape
        cur = torch.cuda.current_stream(grasps.device)
        send = torch.empty((bl, m * w + 1), dtype=torch.float32, device=grasps.device)
        send[:, :m * w] = grasps.reshape(bl, m * w)
        send[:, m * w] = counts.to(torch.int32).view(torch.float32)
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            recv = torch.empty((world * bl, m * w + 1), dtype=torch.float32, device=grasps.device)
            dist.all_gather_into_tensor(recv, send, group=self.group)
            done = torch.cuda.Event()
            done.record(self.side)
        send.record_stream(self.side)      # allocated on the caller's stream, last read on the side stream
        recv.record_stream(cur)            # allocated on the side stream, read by the caller's stream(s)
        return recv[:, :m * w].unflatten(1, (m, w)), recv[:, m * w].view(torch.int32), GatherHandle(done, recv)


def to_list(grasps, counts):
    """padded (B,M,17) + counts -> the reference's list[Tensor (Ns_i,17)] (host sync on counts)."""
    return [grasps[i, :c] for i, c in enumerate(counts.cpu().tolist())]
