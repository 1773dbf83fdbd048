#!/usr/bin/env python3
"""bench.py -- scenes/s of end-to-end grasp inference (BASELINE.json metric).

One step = one pass of the hot path over one batch: (8, 20000, 3) synthetic clouds already
resident in HBM -> GraspNet.forward (4xSA, 2xFP, ApproachNet, CloudCrop, heads) -> pred_decode
(padded (8,1024,17) grasp rows + counts on device) [-> all-gather of the grasp lists when N>1].

  python bench.py --gpus N --steps K --warmup W        (N > 1: starts the N ranks itself, as child processes)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the driver's form)

Rank 0 prints ONE JSON line.  `value` is the headline leg: K timed steps issued round-robin on `--streams`
HIP streams over 4 distinct sets of 8 `uniform` clouds, nothing but the product path inside the
timed region.  Beside it (N=1 only, each its own timed leg AFTER the headline): the same on `surface` clouds
(duplicate points -> FPS ties -> the SA2-4 samplers really run), the single-stream rate (what a caller looping
`net(batch)` gets), the GN_MATH=f32 rate, and the hipGraph-replayed latency at B=1 and B=8.  `roofline` is for
the dominant kernel, timed with HIP events on its launch stream in an isolated single-stream pass;
`cpu_baseline` is the oracle port timed on this host (rank 0, N=1 only, bounded sample).
"""
import argparse
import json
import os
import statistics
import sys
import time

# ROCm maps HIP streams onto this many hardware queues (default 4).  The bench overlaps up to 12
# batches (FPS is a serial chain on 8 CUs per batch); with too few queues two batches share a queue
# and serialise (measured with the fp32 kernels, 200 steps: 4 queues/3 streams 2050, 8/6 3900, 16/8 4150
# scenes/s; with the split kernels, 150 steps at 16 queues: 8 streams 6030, 10: 6110, 12: 6320, 14: 6280, 16: 5710).
# Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# the pool's host driver only supports dmabuf IPC: without this RCCL's inter-process handles fail with
# `hipIpcGetMemHandle: invalid argument`.  Set for BOTH launch forms (ranks started by bench.py itself and ranks that
# arrive from an external torch.distributed.run), before anything initialises HIP.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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
CROP_FLOP_PER_ROW_SPLIT = 81920.0 * 6.0   # layers 2-3 (64->128->256): 6 bf16 products per fp32 MAC (csrc/mlp_split.h)
CROP_FLOP_PER_ROW_F32 = 82944.0           # all three layers as fp32 MACs
NSETS = 4                                  # distinct 8-cloud sets the timed steps rotate over


def log(msg):
    """progress on stderr (stdout carries the ONE JSON line): a run cut short says how far it got"""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.perf_counter()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(n_points, reps1, reps8):
    """Oracle port (C operators with OpenMP + torch-CPU MLPs) on this host: the reference has no CPU PointNet++
    path of its own, so this is kind='port'.  SURVEY.md 8d: warm-up 2, median of 10, on config[0]
    (one cloud per call) and config[1] (batch 8 per call; fewer repetitions to stay within ~30 s)."""
    from oracle import model_oracle as MO, oracle as O
    from graspnet_amd import synthetic as S
    O.build()
    # a one-GPU box's CPU share is 16 cores; hundreds of torch/OpenMP threads on these small convs thrash
    cores = min(os.cpu_count() or 1, 16)
    O.set_num_threads(cores)
    torch.set_num_threads(cores)
    sd = S.make_state_dict(0)
    clouds = S.make_batch(1000, 8, n_points)

    def timed(batch, warm, reps):
        for _ in range(warm):
            MO.pred_decode(MO.forward({"point_clouds": batch.clone()}, sd))
        ts = []
        MO.STAGE_TIMES = {}
        for _ in range(reps):
            t0 = time.perf_counter()
            MO.pred_decode(MO.forward({"point_clouds": batch.clone()}, sd))
            ts.append(time.perf_counter() - t0)
        stage_ms = {k: round(v / reps * 1e3, 2) for k, v in sorted(MO.STAGE_TIMES.items(), key=lambda kv: -kv[1])}
        MO.STAGE_TIMES = None
        return statistics.median(ts), stage_ms

    t_all = time.perf_counter()
    m1, st1 = timed(clouds[:1], 2, reps1)
    m8, st8 = timed(clouds, 1, reps8)
    total = time.perf_counter() - t_all
    return {"value": 8.0 / m8, "unit": "scenes/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "config0_scenes_per_s": 1.0 / m1,
            # mean wall-clock ms per call and stage (fps / ball_query / cylinder_query: the oracle's C
            # operators; grouping: group + gather + centre / rotate; *_mlp, fp, approach_net, heads: torch-CPU convs)
            "stage_ms_config1_batch8": st8, "stage_ms_config0_batch1": st1,
            "sample": "oracle C ops (OpenMP) + torch-CPU MLPs, %d threads: config[1] batch 8 x %d pts per call, warm-up 1, "
                      "median of %d calls (%.2f s each) -> value; config[0] one cloud per call, warm-up 2, median of %d "
                      "(%.3f s each); %.0f s in all" % (cores, n_points, reps8, m8, reps1, m1, total)}


def pmc_file():
    """The committed PMC passes of the bench command (HBM bytes per launch); newest round first."""
    for name in ("r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1g_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            try:
                return name, json.load(open(path))
            except ValueError:
                pass
    return None, {}


def op_rooflines(dev, reps=20, n_points=20000):
    """The drop-in operators BASELINE.json's metric names, at SA sizes (B=8): algorithmic bytes / HIP-event
    time, against the 8 TB/s HBM peak; PMC bytes per launch beside them where a committed pass has them.
    n_points=50000: config[3]'s stress shapes (M=2048 centres, nsample 64; the N-proportional byte terms scale x2.5)."""
    from graspnet_amd import pointnet2_ext as E, synthetic as S
    out = {}
    xyz = S.make_batch(1000, 8, n_points).to(dev)
    ctr = xyz[:, :2048].contiguous()
    _, pmc = pmc_file()
    if n_points != 20000:
        sfx = "_n%d" % n_points

        def timed3(fn):
            fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / reps * 1e-3

        def rec3(name, nbytes, t, bound, **extra):
            r = {"bytes": nbytes, "ms": t * 1e3, "GBps": nbytes / t / 1e9, "frac_hbm": nbytes / t / PEAK_HBM, "bound": bound,
                 "pmc_bytes": pmc.get(name, {}).get("traffic_bytes")}
            r.update(extra)
            out[name] = r
        nb = n_points * 12
        t = timed3(lambda: E.ball_query(ctr, xyz, 0.04, 64))
        rec3("ball_query_sa1" + sfx, (nb + 24576 + 524288) * 8, t,
             "latency / VALU issue (grid cells then a sorted 64-lane register), HBM-trivial",
             tests_per_s_brute_force_equivalent=8 * 2048 * n_points / t)
        xt = xyz.transpose(1, 2).contiguous()
        idx1 = E.ball_query(ctr, xyz, 0.04, 64)
        t = timed3(lambda: E.group_points(xt, idx1))
        rec3("group_points_sa1" + sfx, (nb + 524288 + 1572864) * 8, t, "hbm (write)")
        t = timed3(lambda: E.furthest_point_sampling(xyz, 2048))
        rec3("furthest_point_sampling" + sfx, (nb + 8192) * 8, t, "latency (serial chain of 2047 steps per cloud on one CU each)",
             us_per_iteration=t * 1e6 / 2047)
        return out

    def timed(fn):
        fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e-3

    def rec(name, nbytes, t, bound, **extra):
        r = {"bytes": nbytes, "ms": t * 1e3, "GBps": nbytes / t / 1e9, "frac_hbm": nbytes / t / PEAK_HBM, "bound": bound,
             "pmc_bytes": pmc.get(name, {}).get("traffic_bytes")}
        r.update(extra)
        out[name] = r

    # every drop-in query bins its cloud (round 3) unless the cloud was binned by the previous query on the same stream
    # (round 4: graspnet_amd/pointnet2_ext.py _grid_for; the reference model queries one cloud five times per forward):
    # "_cold" = grid build + query (GN_GRID_CACHE=0: what the first query on a cloud costs), plain = the query on the kept grid
    os.environ["GN_GRID_CACHE"] = "0"
    t = timed(lambda: E.ball_query(ctr, xyz, 0.04, 64))
    rec("ball_query_sa1_cold", 788864 * 8, t, "latency / VALU issue: grid build (one workgroup per cloud) + query, HBM-trivial",
        tests_per_s_brute_force_equivalent=8 * 2048 * 20000 / t)
    os.environ["GN_GRID_CACHE"] = "1"
    t = timed(lambda: E.ball_query(ctr, xyz, 0.04, 64))
    rec("ball_query_sa1", 788864 * 8, t, "latency / VALU issue (grid cells then a sorted 64-lane register), HBM-trivial; the cloud's grid "
        "is reused from the previous query on it", tests_per_s_brute_force_equivalent=8 * 2048 * 20000 / t)
    feat = torch.rand(8, 128, 2048, device=dev)
    idx = torch.randint(0, 2048, (8, 1024, 32), device=dev, dtype=torch.int32)
    t = timed(lambda: E.group_points(feat, idx))
    rec("group_points_sa2", 17956864 * 8, t, "hbm (write)")
    xt = xyz.transpose(1, 2).contiguous()
    idx1 = torch.randint(0, 20000, (8, 2048, 64), device=dev, dtype=torch.int32)
    t = timed(lambda: E.group_points(xt, idx1))
    rec("group_points_sa1", 2337152 * 8, t, "hbm (write)")
    rot = torch.eye(3, device=dev).reshape(1, 1, 9).repeat(8, 1024, 1).contiguous()
    seeds = xyz[:, :1024].contiguous()
    os.environ["GN_GRID_CACHE"] = "0"
    t = timed(lambda: E.cylinder_query(seeds, xyz, rot, 0.05, -0.02, 0.04, 64))
    rec("cylinder_query_cold", 551296 * 8, t, "latency / VALU issue: grid build + query, HBM-trivial")
    os.environ["GN_GRID_CACHE"] = "1"
    t = timed(lambda: E.cylinder_query(seeds, xyz, rot, 0.05, -0.02, 0.04, 64))
    rec("cylinder_query", 551296 * 8, t, "latency / VALU issue, HBM-trivial; grid reused (CloudCrop's four queries bin the cloud once)")
    E.clear_grid_cache()
    return out


def postprocess_leg(net, clouds, dev, reps=20):
    """config[4]'s steps around the network on the batch's decoded scenes, all on the device and batched: voxel down-sample of
    every scene's cloud (open3d's, restated; deterministic) + ModelFreeCollisionDetector.detect per scene against its own
    points (utils/collision_detector.py:69-129, the step README.md:72 calls slow) + sort + NMS -- ~12 launches per batch, no
    host read (graspnet_amd.postprocess.grasp_pipeline).  Beside it the numpy restatement (oracle/post_oracle.py, the
    reference's own arithmetic) on ONE scene as the stated CPU baseline."""
    from graspnet_amd import postprocess as P
    ep = net({"point_clouds": clouds})
    g, c = net.decode_padded(ep)
    torch.cuda.synchronize()
    B = clouds.shape[0]

    def ev_time(fn, n):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            r = fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n, r

    t_vox, det = ev_time(lambda: P.BatchedCollisionDetector(clouds, voxel_size=0.01), reps)
    nvox = det.scene_counts.cpu().tolist()
    t_col, cms = ev_time(lambda: det.detect_padded(g, c, approach_dist=0.05, collision_thresh=0.01), reps)
    t_nms, (out, oc) = ev_time(lambda: P.sort_and_nms(g, c, drop_mask=cms, nms=True), reps)
    t0 = time.perf_counter()
    for _ in range(reps):
        P.grasp_pipeline(clouds, g, c, voxel_size=0.01, approach_dist=0.05, collision_thresh=0.01)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    cl = c.cpu().tolist()
    res = {"workload": "batch of %d decoded scenes (%d-pt clouds, ~%d grasps each): voxel down-sample 0.01 m + collision detect "
                       "(approach_dist 0.05, thresh 0.01) of every scene against its own points, sort + NMS (0.03 m, 30 deg); "
                       "one batched call each, no host read" % (B, clouds.shape[1], int(sum(cl) / max(B, 1))),
           "ms_per_batch": wall, "voxel_down_sample_ms": t_vox, "collision_detect_ms": t_col, "grasp_nms_ms": t_nms,
           "host_syncs": 0, "scene_voxels": nvox[:4], "kept_after_collision_and_nms": oc.cpu().tolist()[:4],
           "bound": "voxel: latency (hash insert with integer atomics, two one-workgroup-per-cloud scans, ordered segment sums; "
                    "9 launches); collision: float64 VALU (transform + 5 box tests per scene point and grasp: %.3f G "
                    "point-grasp tests per batch); nms: LDS bitonic sort + a serial greedy sweep per scene"
                    % (sum(nvox[i] * cl[i] for i in range(B)) / 1e9)}
    try:
        import numpy as np
        from oracle import post_oracle as PO
        pts = clouds[0].cpu().numpy()
        garr = g[0, :cl[0]].cpu().numpy()
        t0 = time.perf_counter()
        sp = PO.voxel_down_sample(pts, 0.01)
        t1 = time.perf_counter()
        wcm, _, _ = PO.collision_detect(sp, garr, approach_dist=0.05, collision_thresh=0.01, voxel_size=0.01)
        t2 = time.perf_counter()
        keep = PO.nms_grasp(garr[~wcm], 0.03, 30.0 / 180.0 * np.pi)
        t3 = time.perf_counter()
        res["cpu_numpy_one_scene_ms"] = {"voxel_down_sample": (t1 - t0) * 1e3, "collision_detect": (t2 - t1) * 1e3,
                                         "nms": (t3 - t2) * 1e3, "total": (t3 - t0) * 1e3, "kept": int(len(keep)),
                                         "note": "oracle/post_oracle.py (numpy, the reference's arithmetic) on scene 0 of the batch; "
                                                 "x%d for the batch" % B}
        res["same_collision_mask_as_numpy"] = bool((cms[0, :cl[0]].cpu().numpy() == wcm).all())
        res["same_kept_rows_as_numpy"] = bool(int(oc[0]) == len(keep) and np.array_equal(out[0, :len(keep)].cpu().numpy(), keep))
    except Exception as ex:       # the CPU side is a reported baseline only
        res["cpu_numpy_one_scene_ms"] = {"error": str(ex)[:200]}
    return res


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """Parent of an N-rank run started WITHOUT a launcher: one child `python -m torch.distributed.run` (which starts the
    N ranks, one per GPU) on a free local port; the children's stdout (rank 0's ONE JSON line) and stderr pass straight
    through; returns the launcher's exit status (non-zero if any rank failed).  Never an exec: a process is only
    ever started fresh, before it has touched the GPU."""
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    log("launching %d ranks: %s" % (n, " ".join(cmd)))
    return subprocess.call(cmd, env=env)


def device_identity(index):
    """what tells two GPUs apart on one node: the PCI address (domain:bus:device) where torch exposes it, else the uuid"""
    try:
        pr = torch.cuda.get_device_properties(index)
        if hasattr(pr, "pci_bus_id"):
            return "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0))
        if hasattr(pr, "uuid"):
            return str(pr.uuid)
    except Exception as ex:       # reporting only
        return "unknown(%s)" % type(ex).__name__
    return "cuda:%d" % index


def check_local_ranks(world, local_world, visible, backend):
    """Before init_process_group: RCCL refuses two ranks on one device ('Duplicate GPU detected') deep inside the first
    collective; say it here, in words, instead.  Returns an error string or None."""
    if world > 1 and backend == "nccl" and local_world > visible:
        return ("bench.py: %d ranks on this node but only %d visible GPU(s): with --backend nccl (RCCL) every rank needs a GPU of "
                "its own (two ranks on one device are refused as 'Duplicate GPU detected'); lower --gpus / --nproc-per-node, "
                "widen HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES, or rehearse with --backend gloo" % (local_world, visible))
    return None


def check_rank_devices(records, world, backend):
    """After the rendezvous, on rank 0: records = [(rank, local_rank, device_index, identity)] gathered from every rank.
    Hard checks before any number is printed: every rank answered, and under RCCL no two ranks share a device."""
    ranks = sorted(r[0] for r in records)
    if len(records) != world or ranks != list(range(world)):
        return "bench.py: launched for %d ranks but %d answered the rendezvous (ranks %s)" % (world, len(records), ranks)
    if backend == "nccl":
        ids = [r[3] for r in records]
        if len(set(ids)) != world:
            dup = sorted({i for i in ids if ids.count(i) > 1})
            return ("bench.py: %d ranks share device(s) %s -- %d distinct GPUs are needed (rank -> device: %s)"
                    % (len(ids) - len(set(ids)) + len(dup), dup, world,
                       ", ".join("%d -> cuda:%d %s" % (r[0], r[2], r[3]) for r in sorted(records))))
    return None


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
    ap.add_argument("--variant", default="uniform", choices=("uniform", "surface"), help="synthetic clouds of the headline leg")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N>1 (nccl = RCCL over xGMI; gloo only to rehearse the N>1 code "
                         "path with several ranks sharing one GPU)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-step timed region is run this many times (each bracketed by barrier + synchronize); `value` is "
                         "the MEDIAN repeat, value_min / value_max the extremes: a 20-step region is 22 ms, one unhidden FPS "
                         "ramp or a clock dip moves a single sample by 10 %%")
    ap.add_argument("--legs", default="all",
                    help="all | headline (only the timed headline leg + the isolated per-kernel pass: what the rocprofv3 passes "
                         "run) | comma list of the legs beside the headline: single,other,f32,graph,ops,cpu,cfg3,service,post,full,dropin,knn,image")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, form the process group, print {n_gpus, ranks_seen, backend} and stop: checks the "
                         "N-rank launch without touching a GPU (CPU tests, with --backend gloo)")
    ap.add_argument("--cpu-reps", type=int, default=10, help="timed config[0] calls of the CPU baseline (config[1]: a third)")
    args = ap.parse_args()
    all_legs = {"single", "other", "f32", "graph", "ops", "cpu", "cfg3", "service", "post", "full", "dropin", "knn", "image"}
    legs = all_legs if args.legs == "all" else (set() if args.legs == "headline" else set(args.legs.split(",")))
    if not legs <= all_legs:
        raise SystemExit("--legs: unknown leg in %r" % (args.legs,))
    if args.no_cpu_baseline:
        legs.discard("cpu")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: this process has made no GPU call yet (importing torch does not
        # initialise HIP), so it only launches the N ranks as CHILD processes and relays rank 0's JSON line
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; start it as `python bench.py --gpus N` "
                         "(it launches the ranks itself) or under torch.distributed.run with --nproc-per-node equal to --gpus"
                         % (args.gpus, world))
    import torch.distributed as dist
    if args.rendezvous_only:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        dist.init_process_group(args.backend if args.backend != "nccl" or torch.cuda.is_available() else "gloo",
                                rank=rank, world_size=world)
        seen = [None] * world
        # a rehearsal has no GPUs: GN_BENCH_FAKE_DEVICE stands in for the device identity (CPU tests of the duplicate check)
        ident = os.environ.get("GN_BENCH_FAKE_DEVICE", "cpu:%d" % local)
        dist.all_gather_object(seen, (rank, local, local, ident))
        err = check_rank_devices(seen, world, os.environ.get("GN_BENCH_CHECK_AS", dist.get_backend())) if rank == 0 else None
        if rank == 0 and not err:
            print(json.dumps({"n_gpus": world, "ranks_seen": dist.get_world_size(), "backend": dist.get_backend(),
                              "ranks": sorted(r[0] for r in seen), "rank_devices": [r[3] for r in sorted(seen)],
                              "rendezvous_only": True}))
        dist.destroy_process_group()
        if err:
            raise SystemExit(err)
        return
    rank_records = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        visible = torch.cuda.device_count()          # counting devices does not initialise HIP
        err = check_local_ranks(world, int(os.environ.get("LOCAL_WORLD_SIZE", world)), visible, args.backend)
        if err:
            raise SystemExit(err)
        local = local % max(visible, 1)              # only a gloo rehearsal (ranks sharing a card) ever wraps
        torch.cuda.set_device(local)
        ident = device_identity(local)
        print("[bench] rank %d/%d (local %d) -> cuda:%d %s  visible=%d HIP_VISIBLE_DEVICES=%s ROCR_VISIBLE_DEVICES=%s HSA_ENABLE_IPC_MODE_LEGACY=%s"
              % (rank, world, int(os.environ.get("LOCAL_RANK", "0")), local, ident, visible, os.environ.get("HIP_VISIBLE_DEVICES"),
                 os.environ.get("ROCR_VISIBLE_DEVICES"), os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")), file=sys.stderr, flush=True)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
        rank_records = [None] * world
        dist.all_gather_object(rank_records, (rank, int(os.environ.get("LOCAL_RANK", "0")), local, ident))
        err = check_rank_devices(rank_records, world, args.backend)
        if err:      # every rank holds the same records and so reaches the same verdict: no rank is left waiting
            dist.destroy_process_group()
            raise SystemExit(err)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from graspnet_amd import GraspNet, GraphedGraspNet, synthetic as S
    from graspnet_amd import dist as gdist
    B, N = args.batch, args.points
    sd = S.make_state_dict(0)
    net = GraspNet(is_training=False, device=dev).load_state_dict(sd)
    # weak scaling: every rank owns its own B scenes (global batch = world*B, contiguous shards);
    # strong scaling (--global-batch G): the same G scenes however many ranks share them
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit("--global-batch must be a multiple of the number of ranks (equal shards for the all-gather)")
        lo, hi = gdist.shard_range(args.global_batch, rank, world)
        B = hi - lo
        gb = args.global_batch
    else:
        gb = world * B
        lo, _ = gdist.shard_range(gb, rank, world)

    def cloud_sets(variant):
        """NSETS distinct global batches; this rank's contiguous shard of each (scene s of set j: seed 1000 + j*gb + s)"""
        return [S.make_batch(1000 + j * gb + lo, B, N, variant).to(dev) for j in range(NSETS)]

    sets = cloud_sets(args.variant)
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]
    gatherer = gdist.GraspGatherer(dev) if world > 1 else None
    rank_devices = [dev.index]

    post_args = dict(voxel_size=0.01, approach_dist=0.05, collision_thresh=0.01)     # demo.py / test.py settings
    from graspnet_amd import postprocess as gpost

    def run_steps(model, batches, strms, nsteps, gather=True, post=False, scenes=None):
        """scenes: per batch set, (raw cloud (B,Nraw,3), n_valid (B,)) -- the cloud the collision filter checks against (the
        reference: the RAW masked cloud, test.py:92 return_raw_cloud=True / demo.py:481, not the points sampled for the network)"""
        last = None
        # several batches in flight: the per-point chains in their throughput tile shapes (fewer CU-microseconds per launch;
        # GraspNet.throughput / gn_set_point_tile_policy -- what PipelinedGraspNet does); one stream: the latency shapes
        model.throughput = len(strms) > 1 and os.environ.get("GN_POINT_TILES", "auto") != "latency"
        for i in range(nsteps):
            with torch.cuda.stream(strms[i % len(strms)]):
                ep = model({"point_clouds": batches[i % len(batches)]})
                g, c = model.decode_padded(ep)
                if post:      # config[4]: collision filter + sort + NMS of the batch, still on the device (test.py:88-96)
                    if scenes is not None:
                        sc, sn = scenes[i % len(scenes)]
                        g, c, _ = gpost.grasp_pipeline(sc, g, c, n_valid=sn, **post_args)
                    else:
                        g, c, _ = gpost.grasp_pipeline(batches[i % len(batches)], g, c, **post_args)
                if gather and gatherer is not None:
                    # the all-gather runs on the gatherer's side stream behind an event, so this stream goes straight on
                    # to its next batch; the final synchronize covers it
                    g, c, _ = gatherer.gather(g, c)
            last = (ep, g, c)
        return last

    def timed_leg(model, batches, strms, nsteps, warm, post=False, scenes=None):
        """warm: untimed steps (at least one per stream, so no stream meets its first kernels, allocator pool or code
        objects inside the timed region; 0 = a repeat of a leg that is already warm); then EXACTLY nsteps between
        barrier+synchronize brackets; max over ranks"""
        if warm:
            run_steps(model, batches, strms, max(warm, len(strms)), post=post, scenes=scenes)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        last = run_steps(model, batches, strms, nsteps, post=post, scenes=scenes)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        per_rank = [elapsed]
        if world > 1:
            t = torch.tensor([elapsed, float(dev.index)], dtype=torch.float64, device=dev)
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            per_rank = [float(x[0].item()) for x in allt]
            rank_devices[:] = [int(x[1].item()) for x in allt]
            elapsed = max(per_rank)
        return elapsed, per_rank, last

    log("setup done: B=%d N=%d world=%d streams=%d" % (B, N, world, len(streams)))
    # ---- headline leg: nothing but the product path in the timed region (no per-kernel events, no hooks)
    net.timing = None
    repeats = []
    for r in range(max(1, args.repeats)):
        repeats.append(timed_leg(net, sets, streams, args.steps, args.warmup if r == 0 else 0))
    order = sorted(range(len(repeats)), key=lambda i: repeats[i][0])
    elapsed, per_rank, last = repeats[order[len(order) // 2]]           # the median repeat (upper median for an even count)
    elapsed_all = [r[0] for r in repeats]
    last_ep, _, c = last
    grasps_last = int(c.sum().item())

    def averages(timing):
        return {name: sum(a.elapsed_time(b) for a, b in evs) / max(len(evs), 1) for name, evs in timing.items()}

    log("headline leg: %.3f ms/step" % (elapsed / args.steps * 1e3))
    # ---- isolated pass (outside every timed region): the same steps on ONE stream with HIP-event brackets around every
    # kernel launch on that stream, so a bracket is the kernel's own duration -- these feed the per-kernel roofline;
    # rocprofv3 --kernel-trace of `--streams 1 --legs headline` agrees with them (profiles/)
    run_steps(net, sets, streams[:1], 2, gather=False)      # the latency tile shapes' kernels may meet their first call here
    torch.cuda.synchronize()
    net.timing = {}
    run_steps(net, sets, streams[:1], min(args.steps, 12), gather=False)
    torch.cuda.synchronize()
    stages = averages(net.timing)
    net.timing = None

    log("isolated per-kernel pass done")
    extra = {}
    k1 = max(10, min(args.steps, 40))
    if world == 1 and "single" in legs:
        e1, _, _ = timed_leg(net, sets, streams[:1], k1, 3)
        extra["single_stream_scenes_per_s"] = B * k1 / e1
        extra["single_stream_ms_per_step"] = e1 / k1 * 1e3
        log("single-stream leg: %.3f ms/step" % (e1 / k1 * 1e3))
    if world == 1 and "other" in legs:
        other = "surface" if args.variant == "uniform" else "uniform"
        osets = cloud_sets(other)
        e2, _, l2 = timed_leg(net, osets, streams, args.steps, args.warmup)
        e2s, _, _ = timed_leg(net, osets, streams[:1], k1, 3)
        extra["%s_clouds" % other] = {"scenes_per_s": B * args.steps / e2, "ms_per_step": e2 / args.steps * 1e3,
                                      "single_stream_scenes_per_s": B * k1 / e2s, "steps": args.steps,
                                      "grasps_last_step": int(l2[2].sum().item()),
                                      "note": "%d distinct sets of %d `%s` clouds; surface scenes hold ~1000 duplicated points each "
                                              "(harmless to the sampler, not reported as ties) and box faces on which distinct "
                                              "points do tie exactly in about one scene in eight; a batch with such a scene runs "
                                              "the full SA2-4 FPS chains instead of the identity-prefix shortcut" % (NSETS, B, other)}
        del osets
        log("%s legs done" % other)
    if world == 1 and "f32" in legs and net.math_split:
        if True:
            net32 = GraspNet(is_training=False, device=dev).load_state_dict(sd)
            net32.math_split = False
            k3 = max(10, min(args.steps, 50))
            e3, _, _ = timed_leg(net32, sets, streams, k3, args.warmup)
            extra["gn_math_f32_scenes_per_s"] = B * k3 / e3
            del net32
        log("f32 leg done")
    if world == 1 and "graph" in legs:
        lat = {}
        for gbatch in sorted({1, B}):
            gnet = GraphedGraspNet(net, batch=gbatch, points=N)
            x = sets[0][:gbatch]
            for _ in range(3):
                gnet(x)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(20):
                a.record()
                gnet(x)
                b.record()
                b.synchronize()
                ts.append(a.elapsed_time(b))
            lat["B=%d" % gbatch] = statistics.median(ts)
            del gnet
        extra["hipgraph_latency_ms"] = lat
        log("hipGraph latency legs done: %s" % lat)

    if world == 1 and "cfg3" in legs:
        # config[3] of BASELINE.json: dense 50 000-pt clouds (M=2048 seeds, nsample-64 ball query), same network
        n3 = 50000
        sets3 = [S.make_batch(3000 + j * B, B, n3, "uniform").to(dev) for j in range(2)]
        k3s = max(6, min(args.steps, 24))
        e3m, _, l3 = timed_leg(net, sets3, streams, k3s, len(streams))
        e3s, _, _ = timed_leg(net, sets3, streams[:1], max(4, k3s // 3), 2)
        net.timing = {}
        run_steps(net, sets3, streams[:1], 4, gather=False)
        torch.cuda.synchronize()
        st3 = averages(net.timing)
        net.timing = None
        extra["config3_n50000"] = {
            "workload": "config[3]: batch %d x %d-pt synthetic `uniform` clouds, full forward + pred_decode" % (B, n3),
            "scenes_per_s": B * k3s / e3m, "ms_per_step": e3m / k3s * 1e3, "steps": k3s, "streams": len(streams),
            "single_stream_ms_per_forward": e3s / max(4, k3s // 3) * 1e3,
            "fps_sa1_ms": st3.get("furthest_point_sampling[1]"),
            "fps_sa1_us_per_iteration": (st3.get("furthest_point_sampling[1]") or 0.0) * 1e3 / 2047.0,
            "kernel_ms": {k: round(v, 4) for k, v in sorted(st3.items(), key=lambda kv: -kv[1])[:8]},
            "grasps_last_step": int(l3[2].sum().item()),
            "op_rooflines": op_rooflines(dev, reps=10, n_points=n3) if "ops" in legs else None}
        del sets3
        log("config[3] leg done: %.3f ms/step on %d streams, %.3f ms/forward on one" % (e3m / k3s * 1e3, len(streams),
                                                                                        extra["config3_n50000"]["single_stream_ms_per_forward"]))
    if world == 1 and "service" in legs:
        # the fork's own operating points (round-4 VERDICT item 2): its ROS services / demos sample ONE far larger cloud per
        # call -- demo.py:836 40 000, demo_pointcloud.py:31 / foundationstereo.py:55 85 000, grasp_service.py:289 95 000 points
        # (B = 1) -- so what they see is the latency of one forward + pred_decode: hipGraph replay, the SA1 sampler's share
        # (2047 serial steps on ONE CU) and its us per step beside it
        svc = {}
        for n_s in (20000, 40000, 85000, 95000):
            try:
                x = S.make_batch(5000 + n_s, 1, n_s, "uniform").to(dev)
                gnet = GraphedGraspNet(net, batch=1, points=n_s)
                for _ in range(3):
                    gnet(x)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ts = []
                for _ in range(15):
                    a.record()
                    gnet(x)
                    b.record()
                    b.synchronize()
                    ts.append(a.elapsed_time(b))
                del gnet
                net.timing = {}
                for _ in range(4):
                    net.decode_padded(net({"point_clouds": x}))
                torch.cuda.synchronize()
                st_s = averages(net.timing)
                net.timing = None
                fps_ms = st_s.get("furthest_point_sampling[1]") or 0.0
                svc[str(n_s)] = {"hipgraph_ms": statistics.median(ts), "eager_kernel_sum_ms": sum(st_s.values()),
                                 "fps_sa1_ms": fps_ms, "fps_sa1_us_per_iteration": fps_ms * 1e3 / 2047.0,
                                 "kernel_ms_top": {k: round(v, 4) for k, v in sorted(st_s.items(), key=lambda kv: -kv[1])[:4]}}
            except Exception as ex:          # a leg beside the headline never takes the line down
                svc[str(n_s)] = {"error": str(ex)[:200]}
        extra["service_latency_ms"] = {k: v.get("hipgraph_ms") for k, v in svc.items()}
        extra["service_latency"] = {"workload": "one cloud per call (B = 1), forward + pred_decode as one hipGraph replay incl. the "
                                                "copy of the cloud into the graph's input; `uniform` synthetic clouds of the sizes the "
                                                "fork's callers sample", "per_size": svc}
        log("service-latency leg done: %s" % extra["service_latency_ms"])
    if world == 1 and "post" in legs:
        extra["postprocess"] = postprocess_leg(net, sets[0], dev)
        log("postprocess leg done")
    if world == 1 and "full" in legs:
        # config[4]'s whole pipeline per step: forward -> pred_decode -> voxel down-sample -> collision filter -> sort + NMS,
        # every batch on its stream, nothing read back (what test.py:82-103 does per batch with a host round trip per scene)
        # the collision filter checks against the RAW scene, as the reference's callers do (test.py:92 return_raw_cloud=True,
        # demo.py:481 cloud_masked): RAW_X x the points the network samples from it (a 720 x 1280 frame's workspace holds a few
        # hundred thousand valid pixels for the 20 000 sampled); the network's input is a sample of that scene
        RAW_X = 5
        raws = [S.make_batch(7000 + j * gb + lo, B, RAW_X * N, args.variant).to(dev) for j in range(NSETS)]
        fsets = [r[:, :N].contiguous() for r in raws]
        fscenes = [(r, torch.full((B,), RAW_X * N, dtype=torch.int32, device=dev)) for r in raws]
        ef = [timed_leg(net, fsets, streams, args.steps, args.warmup if r == 0 else 0, post=True, scenes=fscenes)[0] for r in range(3)]
        ef_med = sorted(ef)[1]
        e1f, _, _ = timed_leg(net, fsets, streams[:1], k1, 3, post=True, scenes=fscenes)
        es = [timed_leg(net, sets, streams, args.steps, 0, post=True)[0] for r in range(2)]
        fp = {"scenes_per_s": B * args.steps / ef_med, "ms_per_step": ef_med / args.steps * 1e3, "steps": args.steps,
              "repeats": 3, "scenes_per_s_all": [round(B * args.steps / e, 1) for e in ef], "streams": len(streams),
              "single_stream_ms_per_step": e1f / k1 * 1e3, "fraction_of_value": (B * args.steps / ef_med) / (B * args.steps / elapsed),
              "scene_points": RAW_X * N,
              "scenes_per_s_with_the_sampled_cloud_as_scene": B * args.steps / min(es),
              "workload": "batches of %d x %d-pt clouds sampled from %d-pt raw scenes through forward + pred_decode + "
                          "postprocess.grasp_pipeline against the RAW scene (voxel 0.01 m, approach_dist 0.05, collision_thresh 0.01, "
                          "NMS 0.03 m / 30 deg), no host read; `..._sampled_cloud_as_scene`: round 4's form of this leg, the filter "
                          "run against the %d sampled points" % (B, N, RAW_X * N, N)}
        try:
            gnet = GraphedGraspNet(net, batch=B, points=N, post=post_args, scene_points=RAW_X * N)
            x, (sc, sn) = fsets[0], fscenes[0]
            for _ in range(3):
                gnet(x, scene=sc, scene_n_valid=sn)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(20):
                a.record()
                gnet(x, scene=sc, scene_n_valid=sn)
                b.record()
                b.synchronize()
                ts.append(a.elapsed_time(b))
            fp["hipgraph_latency_ms"] = statistics.median(ts)
            del gnet
        except Exception as ex:          # a leg beside the headline never takes the line down
            fp["hipgraph_latency_ms"] = None
            fp["hipgraph_error"] = str(ex)[:200]
        extra["full_pipeline_scenes_per_s"] = fp["scenes_per_s"]
        extra["full_pipeline"] = fp
        del raws, fsets, fscenes
        log("full-pipeline leg done: %.3f ms/step on %d streams" % (fp["ms_per_step"], len(streams)))

    if world == 1 and "image" in legs:
        # image -> grasps as ONE leg (round-4 VERDICT item 5; demo.py:449-499 / test.py:82-103 on the device): 720 x 1280 u16 depth
        # images + a workspace mask -> valid pixels -> the draw of N of them (without replacement) -> cloud -> forward ->
        # pred_decode -> voxel down-sample of the RAW masked cloud -> collision filter -> sort + NMS.  Synthetic images: a table
        # plane with smooth bumps 0.45-0.65 m from the camera, the workspace a centred rectangle (half the frame)
        try:
            import numpy as np
            H, W = 720, 1280
            intr = (631.5, 631.5, 640.0, 360.0, 1000.0)
            rng = np.random.default_rng(2024)
            yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)

            def image(seed):
                r = np.random.default_rng(seed)
                z = 550.0 + 40.0 * np.sin(xx / (40.0 + 30.0 * r.random()) + 6.28 * r.random()) * np.cos(yy / (35.0 + 30.0 * r.random()))
                for _ in range(8):      # boxes on the table
                    cx0, cy0, hw, hh = r.integers(300, 980), r.integers(150, 570), r.integers(20, 90), r.integers(20, 90)
                    z[cy0 - hh:cy0 + hh, cx0 - hw:cx0 + hw] -= r.uniform(20.0, 90.0)
                z += r.normal(0.0, 0.6, size=z.shape)
                z[r.random(z.shape) < 0.02] = 0.0                  # holes
                return np.clip(z, 0, 65535).astype(np.uint16)
            mask_np = np.zeros((H, W), np.uint8)
            mask_np[110:610, 180:1100] = 1
            nimg = B * 2
            depth_sets = [torch.from_numpy(np.stack([image(100 + j * B + i) for i in range(B)]).view(np.int16)).to(dev) for j in range(2)]
            masks = torch.from_numpy(np.broadcast_to(mask_np, (B, H, W)).copy()).to(dev)
            keysets = [gpost.draw_keys(31 + j, B, dev) for j in range(4)]

            def img_steps(strms, nsteps):
                net.throughput = len(strms) > 1
                for i in range(nsteps):
                    with torch.cuda.stream(strms[i % len(strms)]):
                        o = gpost.image_to_grasps(net, depth_sets[i % 2], masks, intr, num_point=N, keys=keysets[i % 4], post=post_args)
                return o
            o = img_steps(streams, len(streams))
            torch.cuda.synchronize()
            ki = max(12, min(args.steps, 36))
            t0 = time.perf_counter()
            o = img_steps(streams, ki)
            torch.cuda.synchronize()
            e_multi = time.perf_counter() - t0
            img_steps(streams[:1], 2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            img_steps(streams[:1], 8)
            torch.cuda.synchronize()
            e_one = (time.perf_counter() - t0) / 8
            lat = {}
            for gbatch in (1, B):
                gnet = GraphedGraspNet(net, batch=gbatch, points=N, post=post_args, front=dict(height=H, width=W, intrinsics=intr))
                dd, mm = depth_sets[0][:gbatch], masks[:gbatch]
                for _ in range(3):
                    gnet(dd, mm, seed=1)
                torch.cuda.synchronize()
                a, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ts = []
                for r in range(12):
                    a.record()
                    gnet(dd, mm, seed=r)
                    b2.record()
                    b2.synchronize()
                    ts.append(a.elapsed_time(b2))
                lat["B=%d" % gbatch] = statistics.median(ts)
                og = gnet.outputs
                del gnet
            extra["image_to_grasps"] = {
                "scenes_per_s": B * ki / e_multi, "ms_per_step": e_multi / ki * 1e3, "steps": ki, "streams": len(streams),
                "single_stream_ms_per_step": e_one * 1e3, "hipgraph_latency_ms": lat,
                "valid_pixels_per_image": o["n_valid_pixels"].cpu().tolist()[:4], "kept_last_step": o["kept"].cpu().tolist()[:4],
                "voxel_status_max": int(o["voxel_status"].max().item()),
                "workload": "%d synthetic 720 x 1280 u16 depth images + workspace mask per step (depth already in HBM) -> valid-pixel "
                            "compaction -> %d drawn without replacement -> cloud -> forward -> pred_decode -> collision filter "
                            "against the raw masked cloud (every valid pixel, voxel 0.01 m) -> sort + NMS; nothing read back" % (B, N)}
            del depth_sets, masks
            torch.cuda.empty_cache()
            log("image -> grasps leg done: %.3f ms/step on %d streams, hipGraph %s" % (e_multi / ki * 1e3, len(streams), lat))
        except Exception as ex:          # a leg beside the headline never takes the line down
            extra["image_to_grasps"] = {"error": str(ex)[:300]}

    if world == 1 and "dropin" in legs:
        # drop-in MODE: the reference's unfused python composition (tests/dropin_harness.py -- ~30 `_ext` calls per forward in
        # the reference's order, torch-on-GPU layers between them) over graspnet_amd.pointnet2_ext + graspnet_amd.pred_decode:
        # the figure a maintainer gets from replacing only pointnet2/_ext (pointnet2_utils.py:26).  Harness code, product
        # operators; one stream, like the reference's own loop (test.py:82-96).
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import dropin_harness as DH
            from graspnet_amd import pointnet2_ext as DE, pred_decode as gn_pred_decode
            sdd = DH.to_device(sd, dev)

            def dropin_step(x):
                with torch.no_grad():
                    ep_d = DH.forward(DE, x, sdd)
                return net.decode_padded(ep_d)

            def dropin_time(n):
                dropin_step(sets[0])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    dropin_step(sets[i % NSETS])
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n

            def ext_share():
                """one instrumented forward: HIP-event brackets around every `_ext` call -> ms inside the drop-in operators"""
                import types
                evs = []
                wrapped = types.SimpleNamespace()
                for name in ("furthest_point_sampling", "gather_points", "ball_query", "group_points", "three_nn",
                             "three_interpolate", "cylinder_query"):
                    def mk(fn):
                        def w(*a):
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                            r = fn(*a)
                            e1.record()
                            evs.append((e0, e1))
                            return r
                        return w
                    setattr(wrapped, name, mk(getattr(DE, name)))
                with torch.no_grad():
                    DH.forward(wrapped, sets[0], sdd)
                torch.cuda.synchronize()
                return sum(a.elapsed_time(b) for a, b in evs), len(evs)

            kd = max(5, min(args.steps, 20))
            td = dropin_time(kd)
            ext_ms, ext_calls = ext_share()
            os.environ["GN_GRID_CACHE"] = "0"
            td0 = dropin_time(kd)
            os.environ["GN_GRID_CACHE"] = "1"
            DE.clear_grid_cache()
            extra["dropin_unfused_scenes_per_s"] = B / td
            extra["dropin_unfused"] = {"scenes_per_s": B / td, "ms_per_forward": td * 1e3, "steps": kd,
                                       "ms_per_forward_without_grid_reuse": td0 * 1e3,
                                       "ms_inside_the_ext_operators": ext_ms, "ext_calls_per_forward": ext_calls,
                                       "fraction_of_single_stream_fused": (B / td) / extra.get("single_stream_scenes_per_s", float("nan")),
                                       "note": "unfused reference composition (harness) on torch-GPU layers over the drop-in "
                                               "operators; one stream; the fused path is `single_stream_scenes_per_s`"}
            del sdd
            torch.cuda.empty_cache()
            log("drop-in unfused leg done: %.2f ms per forward" % (td * 1e3))
        except Exception as ex:          # a leg beside the headline never takes the line down
            extra["dropin_unfused"] = None
            extra["dropin_unfused_error"] = str(ex)[:300]
    if world == 1 and "knn" in legs:
        # the knn operator (named in north_star; knn_pytorch.knn, knn/src/knn.h:12-65) at the reference's two call-site shapes
        # (utils/label_generation.py:58-60: 300 views x 300 views; :84-86: object points x 1024 seeds), k = 1: distance tests
        # per second against the chip's fp32 VALU rate (a test = D sub + D mul + D add + a compare-insert, ~12 lane-ops at D=3:
        # 256 CUs x 4 SIMDs x 16 lanes x ~2.4 GHz = 39.3 T lane-ops/s), beside the round-3 one-lane-per-query kernel (GN_KNN=scan)
        from graspnet_amd import knn_ext
        kn = {}
        for name, nr, nq in (("views_300x300", 300, 300), ("object_points_x_seeds_20000x1024", 20000, 1024)):
            r = torch.randn(1, 3, nr, device=dev)
            q = torch.randn(1, 3, nq, device=dev)
            out_i = torch.zeros(1, 1, nq, dtype=torch.int64, device=dev)

            def knn_time(reps=20):
                knn_ext.knn(r, q, out_i)
                a, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    knn_ext.knn(r, q, out_i)
                b2.record()
                torch.cuda.synchronize()
                return a.elapsed_time(b2) / reps
            t_new = knn_time()
            os.environ["GN_KNN"] = "scan"
            t_old = knn_time()
            del os.environ["GN_KNN"]
            tests = float(nr) * nq
            kn[name] = {"ms": t_new, "scan_kernel_ms": t_old, "speedup": t_old / t_new, "distance_tests": tests,
                        "tests_per_s": tests / (t_new * 1e-3), "frac_valu_peak": tests * 12.0 / (t_new * 1e-3) / 39.3e12,
                        "bound": "launch latency (a few us of work)" if tests < 1e6 else "VALU issue (wave per query, k-list in registers)"}
        extra["knn"] = kn
        log("knn leg done: %s" % {k2: round(v["ms"], 4) for k2, v in kn.items()})
    if rank == 0:
        # dominance by CU-time: FPS is the longest kernel but runs on B of the 256 CUs
        cu_share = {k: (B / 256.0 if k.startswith("furthest_point_sampling") else 1.0) for k in stages}
        dom = max(stages, key=lambda k: stages[k] * cu_share[k]) if stages else None
        rk = "crop_unique_mlp_pool" if "crop_unique_mlp_pool" in stages else "crop_group_mlp_pool"
        roof = None
        uc = last_ep.get("_gn", {}).get("ucount")
        if rk in stages and N == 20000 and uc is not None:
            sec = stages[rk] * 1e-3
            # MFMA work the kernel issues per launch: every unique (seed, point) row once; the packed kernel (default,
            # csrc/mlp_split.hip crop_packed_split_kernel) pads each WORKGROUP's row stream (spb consecutive seeds) to 32-row
            # M tiles, GN_CROP_PACKED=0 / GN_MATH=f32 pad every seed
            nseeds = int(uc.numel())
            spb = max(1, min(16, nseeds // 512))
            packed = net.math_split and os.environ.get("GN_CROP_PACKED", "1") != "0"
            if packed:
                flat = uc.reshape(-1).long()
                pad = (-nseeds) % spb
                if pad:
                    flat = torch.cat([flat, flat.new_zeros(pad)])
                rows = int((((flat.reshape(-1, spb).sum(1) + 31) // 32) * 32).sum().item())
            else:
                rows = int((((uc + 31) // 32) * 32).sum().item())
            rows_unique = int(uc.sum().item())
            alg = FLOP_PER_SCENE["crop"] * B          # the reference's 4 x 64 rows per seed, fp32 MACs
            pmc_name, pmc = pmc_file()
            traffic = pmc.get(rk, {}) if (net.math_split and B == 8) else {}
            if net.math_split:
                issued, peak, unit_note = rows * CROP_FLOP_PER_ROW_SPLIT, PEAK_BF16_MFMA, "bf16 MFMA (v_mfma_f32_32x32x16_bf16)"
            else:
                issued, peak, unit_note = rows * CROP_FLOP_PER_ROW_F32, PEAK_F32_MFMA, "fp32 MFMA (v_mfma_f32_32x32x2_f32)"
            roof = {"bound": "mfma", "kernel": rk, "achieved": issued / sec / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                    "frac": issued / sec / peak, "traffic": traffic.get("traffic_bytes"),
                    "priced": "issued %s FLOP per launch = %d evaluated rows (unique rows padded to 32-row tiles per %s) x %.0f, "
                              "against that unit's dense peak; useful_frac counts the unique rows only"
                              % (unit_note, rows, "workgroup stream" if packed else "seed", issued / rows),
                    "avg_launch_ms": stages[rk], "rows_per_seed": rows / float(uc.numel()),
                    "rows_unique_per_seed": rows_unique / float(uc.numel()),
                    "useful_frac": rows_unique * (issued / rows) / sec / peak,
                    "algorithmic": {"flop_per_launch": alg, "f32_equiv_tflops": alg / sec / 1e12,
                                    "x_f32_mfma_peak": alg / sec / PEAK_F32_MFMA,
                                    "note": "SURVEY.md 8d figure (21.576 GFLOP/scene: the reference evaluates 4 x 64 rows per seed as "
                                            "fp32 MACs) / the same launch time; a rate of useful work, not a fraction of any unit's "
                                            "peak: the kernel evaluates each unique row once and splits each fp32 MAC into 6 bf16 products"},
                    "measured": "HIP events on the launch stream, single-stream pass right after the timed region"}
            if traffic:
                roof["traffic_source"] = "profiles/%s: %s" % (pmc_name, traffic.get("how", ""))
        mlp_ms = sum(v for k, v in stages.items() if "mlp" in k)
        mlp_flop = sum(FLOP_PER_SCENE.values()) * B if N == 20000 else None
        out = {
            "metric": "scenes/s grasp inference, 20k-pt clouds, batch 8",
            "value": world * B * args.steps / elapsed, "unit": "scenes/s",
            "n_gpus": world, "ranks_seen": len(rank_records) if rank_records else 1,
            "backend": (dist.get_backend() if world > 1 else None), "rank_devices": rank_devices,
            "rank_device_ids": [r[3] for r in sorted(rank_records)] if rank_records else [device_identity(dev.index)],
            "visible_devices": torch.cuda.device_count(), "steps": args.steps, "warmup": args.warmup,
            "repeats": len(elapsed_all), "value_min": world * B * args.steps / max(elapsed_all),
            "value_max": world * B * args.steps / min(elapsed_all),
            "ms_per_step_all_repeats": [round(e / args.steps * 1e3, 4) for e in elapsed_all],
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak",
            "vs_baseline": None,
            "dtype": ("f32 (grouped layers: exact 3xbf16 operand split, 6 bf16 MFMA products per MAC, fp32 accumulate)"
                      if net.math_split else "f32"),
            "data": "synthetic",
            "config": {"workload": "config[1]: batch %d x %d-pt synthetic `%s` clouds per GPU (%d distinct sets in rotation), full "
                                   "SA->ApproachNet->CloudCrop->OperationNet/ToleranceNet->pred_decode, seeded random weights"
                                   % (B, N, args.variant, NSETS),
                       "batch_per_gpu": B, "points": N, "parallelism": "scene shards x%d" % world,
                       "streams": len(streams), "warmup_steps_run": max(args.warmup, len(streams))},
            "roofline": roof,
            "per_rank_ms_per_step": {"min": min(per_rank) / args.steps * 1e3, "max": max(per_rank) / args.steps * 1e3,
                                     "all": [round(t / args.steps * 1e3, 4) for t in per_rank]},
            "dominant_kernel_by_cu_time": dom,
            "all_mlp_kernels": {"ms": mlp_ms, "algorithmic_tflops": (mlp_flop / (mlp_ms * 1e-3) / 1e12) if mlp_flop and mlp_ms else None},
            "op_rooflines": op_rooflines(dev) if (world == 1 and "ops" in legs) else None,
            "kernel_ms": {k: round(v, 4) for k, v in sorted(stages.items(), key=lambda kv: -kv[1])},
            # round-4 VERDICT item 1: the per-point chains (FP1, FP2, ApproachNet, heads, the three hoisted linears) in the
            # single-stream pass (latency tile shapes: heads on 128-row fp32 tiles, ApproachNet on 64-row fp32 tiles)
            "per_point_chains": {"sum_ms": sum(v for k, v in stages.items() if k.startswith(("fp_interp_mlp", "approach_mlp", "heads_mlp", "point_linear"))),
                                 "kernels_ms": {k: round(v, 4) for k, v in stages.items()
                                                if k.startswith(("fp_interp_mlp", "approach_mlp", "heads_mlp", "point_linear"))},
                                 "tile_policy_of_the_timed_legs": "throughput (64- / 128-row fp32 tiles) on %d streams; latency shapes on one"
                                                                  % len(streams)},
            "grasps_last_step": grasps_last,
        }
        out.update(extra)
        log("op rooflines done; cpu baseline next")
        if world == 1 and "cpu" in legs:
            out["cpu_baseline"] = cpu_baseline(N, args.cpu_reps, max(1, args.cpu_reps // 3))
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
