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

# ROCm maps HIP streams onto this many hardware queues (default 4).  The bench overlaps up to 12
# batches (FPS is a serial chain on 8 CUs per batch); with too few queues two batches share a queue
# and serialise (measured with the fp32 kernels, 200 steps: 4 queues/3 streams 2050, 8/6 3900, 16/8 4150
# scenes/s; with the split kernels, 150 steps at 16 queues: 8 streams 6030, 10: 6110, 12: 6320, 14: 6280, 16: 5710).
# Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "graspnet-baseline_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

# algorithmic work per scene at N=20000 (SURVEY.md 8d / BASELINE.md section 3)
FLOP_PER_SCENE = {"sa1": 3.272e9, "sa2": 4.320e9, "sa3": 1.348e9, "sa4": 0.674e9, "fp1": 0.201e9, "fp2": 0.403e9,
                  "approach": 0.479e9, "crop": 21.576e9, "heads": 0.855e9}
PEAK_F32_MFMA = 157.3e12     # MI355X_MICROARCH.md chip table (v_mfma_f32_32x32x2_f32: the SIMDs' FP32 ALUs)
PEAK_BF16_MFMA = 2.5e15      # dense bf16 matrix-core peak, same table
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


def op_rooflines(dev, reps=20):
    """HBM-bound drop-in operators at SA sizes (B=8): algorithmic bytes / HIP-event time."""
    from graspnet_amd import pointnet2_ext as E, synthetic as S
    out = {}
    xyz = S.make_batch(1000, 8, 20000).to(dev)
    ctr = xyz[:, :2048].contiguous()

    def timed(fn):
        fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e-3

    t = timed(lambda: E.ball_query(ctr, xyz, 0.04, 64))
    out["ball_query_sa1"] = {"bytes": 788864 * 8, "ms": t * 1e3, "GBps": 788864 * 8 / t / 1e9, "frac_hbm": 788864 * 8 / t / PEAK_HBM,
                             "tests_per_s": 8 * 2048 * 20000 / t, "bound": "VALU issue (brute force), not HBM"}
    feat = torch.rand(8, 128, 2048, device=dev)
    idx = torch.randint(0, 2048, (8, 1024, 32), device=dev, dtype=torch.int32)
    t = timed(lambda: E.group_points(feat, idx))
    out["group_points_sa2"] = {"bytes": 17956864 * 8, "ms": t * 1e3, "GBps": 17956864 * 8 / t / 1e9, "frac_hbm": 17956864 * 8 / t / PEAK_HBM,
                               "bound": "hbm (write)"}
    xt = xyz.transpose(1, 2).contiguous()
    idx1 = torch.randint(0, 20000, (8, 2048, 64), device=dev, dtype=torch.int32)
    t = timed(lambda: E.group_points(xt, idx1))
    out["group_points_sa1"] = {"bytes": 2337152 * 8, "ms": t * 1e3, "GBps": 2337152 * 8 / t / 1e9, "frac_hbm": 2337152 * 8 / t / PEAK_HBM,
                               "bound": "hbm (write)"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch", type=int, default=8, help="scenes per GPU per step")
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: a fixed global batch cut into contiguous per-rank shards "
                         "(default 0 = weak scaling, --batch scenes on every rank)")
    ap.add_argument("--streams", type=int, default=12,
                    help="HIP streams the steps are issued round-robin on (FPS is a serial chain on 8 CUs per "
                         "batch; other batches' kernels fill the rest of the chip meanwhile)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N>1 (nccl = RCCL over xGMI; gloo only to rehearse the N>1 code "
                         "path with several ranks sharing one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scenes", type=int, default=16)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from graspnet_amd import GraspNet, synthetic as S
    from graspnet_amd import dist as gdist
    B, N = args.batch, args.points
    net = GraspNet(is_training=False, device=dev).load_state_dict(S.make_state_dict(0))
    # weak scaling: every rank owns its own B scenes (global batch = world*B, contiguous shards);
    # strong scaling (--global-batch G): the same G scenes however many ranks share them
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit("--global-batch must be a multiple of the number of ranks (equal shards for the all-gather)")
        lo, hi = gdist.shard_range(args.global_batch, rank, world)
        B = hi - lo
    else:
        lo, _ = gdist.shard_range(world * B, rank, world)
    clouds = S.make_batch(1000 + lo, B, N, "uniform").to(dev)

    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]
    # torch's caching allocator keeps one pool per stream: reserve each stream's pool up front (setup, like creating the
    # streams) so that a stream's first step does not spend its time in hipMalloc when --warmup is smaller than --streams
    per_step = 6 * B * (N * 12 + 4 * 1024 * 1024)        # ~2x the outputs + workspace of one step (0.2 GB at B=8)
    for st in streams:
        with torch.cuda.stream(st):
            blk = torch.empty((per_step,), dtype=torch.uint8, device=dev)
            del blk
    torch.cuda.synchronize()

    last = {}

    def step(i):
        with torch.cuda.stream(streams[i % len(streams)]):
            ep = net({"point_clouds": clouds})
            g, c = net.decode_padded(ep)
            if world > 1:
                g, c = gdist.all_gather_grasps(g, c)
        last["ep"] = ep
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

    def averages(timing):
        return {name: sum(a.elapsed_time(b) for a, b in evs) / max(len(evs), 1) for name, evs in timing.items()}

    overlapped = averages(net.timing)          # inside the timed region: kernels of up to `streams` batches overlap
    # isolated pass (outside the timed region): the same steps on ONE stream, so a kernel's event bracket is
    # its own duration -- these feed the per-kernel roofline; rocprofv3 of `--streams 1` matches them
    net.timing = {}
    for i in range(min(args.steps, 10)):
        with torch.cuda.stream(streams[0]):
            ep = net({"point_clouds": clouds})
            net.decode_padded(ep)
    torch.cuda.synchronize()
    if rank == 0:
        last_ep = last.get("ep", {})
        stages = averages(net.timing)
        net.timing = None
        # dominance by CU-time: FPS is the longest kernel but runs on B of the 256 CUs
        cu_share = {k: (B / 256.0 if k.startswith("furthest_point_sampling") else 1.0) for k in stages}
        dom = max(stages, key=lambda k: stages[k] * cu_share[k]) if stages else None
        rk = "crop_unique_mlp_pool" if "crop_unique_mlp_pool" in stages else "crop_group_mlp_pool"
        roof = None
        if rk in stages and N == 20000:
            sec = stages[rk] * 1e-3
            ach = FLOP_PER_SCENE["crop"] * B / sec          # SURVEY.md 8d algorithmic figure x scenes per launch
            roof = {"bound": "mfma", "kernel": rk, "achieved": ach / 1e12, "peak": PEAK_F32_MFMA / 1e12,
                    "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA, "traffic": None, "avg_launch_ms": stages[rk],
                    "avg_launch_ms_in_timed_region": overlapped.get(rk),
                    "measured": "HIP events on the launch stream; avg_launch_ms from a single-stream pass of the same "
                                "steps right after the timed region (inside it, kernels of up to %d batches share the "
                                "chip and a launch's bracket includes that contention)" % len(streams)}
            # HBM traffic of this kernel cannot be counted from inside the process: it comes from the committed PMC passes
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r1g_pmc_traffic.json")))[rk]
                if net.math_split and B == 8:
                    roof["traffic"] = pmc["traffic_bytes"]
                    roof["traffic_source"] = "profiles/r1g_pmc_traffic.json: " + pmc["how"]
            except (OSError, KeyError, ValueError):
                pass
            uc = last_ep.get("_gn", {}).get("ucount")
            if uc is not None:
                # the de-duplicated path executes fewer MACs than the reference algorithm: report the
                # MFMA work actually issued too (rows padded to 32-row tiles x 82,944 FLOP per row)
                rows = int((((uc + 31) // 32) * 32).sum().item())
                ex = rows * 82944.0 / sec
                roof["executed"] = {"f32_equiv_tflops": ex / 1e12, "rows_per_seed": rows / float(uc.numel()),
                                    "note": "algorithmic 'achieved' counts the reference's 4x64 rows per seed; the "
                                            "kernel evaluates each unique (seed,point) row once"}
                if net.math_split:
                    # layers 2-3 (81,920 of the 82,944 FLOP per row) run as 6 bf16 products per fp32 MAC on the matrix cores
                    exb = rows * 81920.0 * 6.0 / sec
                    roof["executed"].update({"bf16_mfma_tflops": exb / 1e12, "frac_of_bf16_mfma_peak": exb / PEAK_BF16_MFMA,
                                             "bf16_mfma_peak_tflops": PEAK_BF16_MFMA / 1e12})
                    roof["note"] = ("fp32 results from the bf16 matrix cores: each operand split exactly into 3 bf16 terms, 6 "
                                    "products per MAC, fp32 accumulate (csrc/mlp_split.h); 'achieved'/'frac' are fp32-equivalent "
                                    "algorithmic FLOP/s against the fp32 MFMA peak and exceed 1 because the fp32 MFMA is not the "
                                    "unit doing the work; 'executed' prices the issued bf16 MFMAs against the bf16 peak")
                else:
                    roof["executed"]["frac_of_peak"] = ex / PEAK_F32_MFMA
        mlp_ms = sum(v for k, v in stages.items() if "mlp" in k)
        mlp_flop = sum(FLOP_PER_SCENE.values()) * B if N == 20000 else None
        out = {
            "metric": "scenes/s grasp inference, 20k-pt clouds, batch 8",
            "value": world * B * args.steps / elapsed, "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak",
            "vs_baseline": None,
            "dtype": ("f32 (grouped layers: exact 3xbf16 operand split, 6 bf16 MFMA products per MAC, fp32 accumulate)"
                      if net.math_split else "f32"),
            "data": "synthetic",
            "config": {"workload": "config[1]: batch %d x %d-pt synthetic clouds per GPU, full SA->ApproachNet->"
                                   "CloudCrop->OperationNet/ToleranceNet->pred_decode, seeded random weights" % (B, N),
                       "batch_per_gpu": B, "points": N, "parallelism": "scene shards x%d" % world,
                       "streams": len(streams)},
            "roofline": roof,
            "dominant_kernel_by_cu_time": dom,
            "all_mlp_kernels": {"ms": mlp_ms, "algorithmic_tflops": (mlp_flop / (mlp_ms * 1e-3) / 1e12) if mlp_flop and mlp_ms else None},
            "op_rooflines": op_rooflines(dev) if world == 1 else None,
            "kernel_ms": {k: round(v, 4) for k, v in sorted(stages.items(), key=lambda kv: -kv[1])},
            "kernel_ms_in_timed_region": {k: round(v, 4) for k, v in sorted(overlapped.items(), key=lambda kv: -kv[1])},
            "end_to_end": {"algorithmic_mlp_tflops": (sum(FLOP_PER_SCENE.values()) * world * B * args.steps / elapsed / 1e12) if N == 20000 else None,
                           "frac_of_f32_mfma_peak": (sum(FLOP_PER_SCENE.values()) * B * args.steps / elapsed / PEAK_F32_MFMA) if N == 20000 else None},
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
