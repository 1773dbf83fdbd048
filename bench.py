#!/usr/bin/env python3
"""bench.py -- scenes/s of end-to-end grasp inference (BASELINE.json metric).

One step = one pass of the hot path over one batch: (8, 20000, 3) synthetic clouds already
resident in HBM -> GraspNet.forward (4xSA, 2xFP, ApproachNet, CloudCrop, heads) -> pred_decode
(padded (8,1024,17) grasp rows + counts on device) [-> all-gather of the grasp lists when N>1].

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel, timed live with HIP events on
the launch stream inside the timed region; `cpu_baseline` is the oracle port timed on this host
(rank 0, N=1 only, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "graspnet-baseline_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

# algorithmic work per scene at N=20000 (SURVEY.md 8d / BASELINE.md section 3)
FLOP_PER_SCENE = {"sa1": 3.272e9, "sa2": 4.320e9, "sa3": 1.348e9, "sa4": 0.674e9, "fp1": 0.201e9, "fp2": 0.403e9,
                  "approach": 0.479e9, "crop": 21.576e9, "heads": 0.855e9}
PEAK_F32_MFMA = 157.3e12     # MI355X_MICROARCH.md chip table
PEAK_HBM = 8.0e12


def cpu_baseline(n_scenes, n_points):
    """Oracle port (C operators with OpenMP + torch-CPU MLPs) on this host: the reference has no
    CPU PointNet++ path of its own, so this is kind='port'."""
    from oracle import model_oracle as MO, oracle as O
    from graspnet_amd import synthetic as S
    O.build()
    # a one-GPU box's CPU share is 16 cores; hundreds of torch/OpenMP threads on these small convs thrash
    cores = min(os.cpu_count() or 1, 16)
    O.set_num_threads(cores)
    torch.set_num_threads(cores)
    sd = S.make_state_dict(0)
    clouds = S.make_batch(1000, n_scenes, n_points)
    MO.pred_decode(MO.forward({"point_clouds": clouds[:1].clone()}, sd))      # warm-up
    t0 = time.time()
    for i in range(n_scenes):
        MO.pred_decode(MO.forward({"point_clouds": clouds[i:i + 1].clone()}, sd))
    dt = time.time() - t0
    return {"value": n_scenes / dt, "unit": "scenes/s", "cores": cores, "kind": "port",
            "sample": "%d scenes x %d pts, B=1 per call, oracle C ops (OpenMP) + torch-CPU MLPs, %.1f s" % (n_scenes, n_points, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="scenes per GPU per step")
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams the steps are issued round-robin on (FPS is a serial chain on 8 CUs per "
                         "batch; other batches' kernels fill the rest of the chip meanwhile)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scenes", type=int, default=3)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from graspnet_amd import GraspNet, synthetic as S
    from graspnet_amd import dist as gdist
    B, N = args.batch, args.points
    net = GraspNet(is_training=False, device=dev).load_state_dict(S.make_state_dict(0))
    # weak scaling: every rank owns its own B scenes (global batch = world*B, contiguous shards)
    lo, _ = gdist.shard_range(world * B, rank, world)
    clouds = S.make_batch(1000 + lo, B, N, "uniform").to(dev)

    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]
    torch.cuda.synchronize()

    def step(i):
        with torch.cuda.stream(streams[i % len(streams)]):
            ep = net({"point_clouds": clouds})
            g, c = net.decode_padded(ep)
            if world > 1:
                g, c = gdist.all_gather_grasps(g, c)
        return g, c

    for i in range(args.warmup):
        step(i)
    net.timing = {}          # HIP-event brackets around every kernel launch, recorded in the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        g, c = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        stages = {}
        for name, evs in net.timing.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            stages[name] = sum(ms) / max(len(ms), 1)
        net.timing = None
        dom = max(stages, key=stages.get) if stages else None
        key = {"crop_group_mlp_pool": "crop", "sa_group_mlp_pool[1]": "sa1", "sa_group_mlp_pool[2]": "sa2"}.get(dom)
        # roofline: the grouped-MLP kernel that carries most of the FLOPs (CloudCrop), fp32 MFMA bound
        rk = "crop_group_mlp_pool"
        roof = None
        if rk in stages and N == 20000:
            ach = FLOP_PER_SCENE["crop"] * B / (stages[rk] * 1e-3)
            roof = {"bound": "mfma", "kernel": rk, "achieved": ach / 1e12, "peak": PEAK_F32_MFMA / 1e12,
                    "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA, "traffic": None,
                    "avg_launch_ms": stages[rk]}
        out = {
            "metric": "scenes/s grasp inference, 20k-pt clouds, batch 8",
            "value": world * B * args.steps / elapsed, "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config[1]: batch %d x %d-pt synthetic clouds per GPU, full SA->ApproachNet->"
                                   "CloudCrop->OperationNet/ToleranceNet->pred_decode, seeded random weights" % (B, N),
                       "batch_per_gpu": B, "points": N, "parallelism": "scene shards x%d" % world,
                       "streams": len(streams)},
            "roofline": roof,
            "dominant_kernel": dom, "dominant_kernel_roofline_key": key,
            "kernel_ms": {k: round(v, 4) for k, v in sorted(stages.items(), key=lambda kv: -kv[1])},
            "grasps_last_step": int(c.sum().item()),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_scenes, N)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
