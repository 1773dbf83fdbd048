"""GPU: bench.py's output contract (one JSON line, required keys) for N=1, and a rehearsal of the N>1 code path
(rank sharding, barrier + max-over-ranks timing, the all-gather of the grasp lists) with two ranks sharing the one GPU of
the test box over gloo -- the real N>1 runs use RCCL and are the driver's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _last_json(text):
    lines = [l for l in text.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "bench.py must print exactly one JSON line, got %d" % len(lines)
    return json.loads(lines[0])


def test_bench_single_gpu_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "12", "--cpu-reps", "3"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["scaling"] == "weak" and d["higher_is_better"] is True
    # fields exist and are positive / finite; speeds are compared with the same run's other legs, never with absolute
    # numbers (the box's other tests and the two-rank rehearsals share the card)
    assert d["unit"] == "scenes/s" and d["value"] > 0
    assert abs(d["value"] - 8 * 12 / (d["ms_per_step"] * 12e-3)) < 1e-6 * d["value"]
    assert d["repeats"] == 5 and len(d["ms_per_step_all_repeats"]) == 5
    assert d["value_min"] <= d["value"] <= d["value_max"]
    assert sorted(d["ms_per_step_all_repeats"])[2] == round(d["ms_per_step"], 4)          # value is the MEDIAN repeat
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["avg_launch_ms"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.05 < r["frac"] <= 1.0 and r["peak"] == 2500.0          # issued bf16 MFMA FLOP/s against the bf16 peak
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["cpu_model"]
    for key in ("stage_ms_config1_batch8", "stage_ms_config0_batch1"):       # per-stage ms of the CPU port
        assert {"fps", "ball_query", "grouping", "sa_mlp", "fp", "approach_net", "cylinder_query", "crop_mlp", "heads",
                "decode"} <= set(c[key]) and all(v >= 0 for v in c[key].values())
    # the legs beside the headline: single stream, the other cloud variant, the fp32-MFMA path, hipGraph latency
    assert 0 < d["single_stream_scenes_per_s"] <= d["value_max"] * 1.05
    assert d["surface_clouds"]["scenes_per_s"] > 0 and 0 < d["gn_math_f32_scenes_per_s"] <= d["value_max"] * 1.05
    assert set(d["hipgraph_latency_ms"]) == {"B=1", "B=8"} and d["hipgraph_latency_ms"]["B=1"] <= d["hipgraph_latency_ms"]["B=8"] * 1.05
    assert {"ball_query_sa1", "group_points_sa2", "group_points_sa1", "cylinder_query"} <= set(d["op_rooflines"])
    assert "workload" in d["config"] and "model" not in d["config"]
    # config[3] (dense 50 000-pt clouds) and config[4]'s steps around the network have driver-visible legs of their own
    c3 = d["config3_n50000"]
    assert 0 < c3["scenes_per_s"] <= d["value_max"] * 1.05 and c3["fps_sa1_us_per_iteration"] > 0 and c3["single_stream_ms_per_forward"] > 0
    assert {"ball_query_sa1_n50000", "group_points_sa1_n50000", "furthest_point_sampling_n50000"} <= set(c3["op_rooflines"])
    pp = d["postprocess"]
    assert pp["ms_per_batch"] > 0 and pp["collision_detect_ms"] > 0 and pp["grasp_nms_ms"] > 0 and pp["same_collision_mask_as_numpy"] is True
    assert pp["host_syncs"] == 0 and pp["same_kept_rows_as_numpy"] is True
    assert pp["cpu_numpy_one_scene_ms"]["total"] > 0
    fp = d["full_pipeline"]
    assert 0 < d["full_pipeline_scenes_per_s"] == fp["scenes_per_s"] <= d["value_max"] * 1.05
    assert fp["hipgraph_latency_ms"] and fp["hipgraph_latency_ms"] >= d["hipgraph_latency_ms"]["B=8"] * 0.95
    if d.get("dropin_unfused") is not None:
        assert 0 < d["dropin_unfused_scenes_per_s"] <= d["value_max"] * 1.05
    if d.get("knn") is not None:
        assert all(v["ms"] > 0 for v in d["knn"].values())


def test_bench_two_ranks_share_the_gpu_over_gloo():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "6", "--warmup", "4", "--backend", "gloo"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["cpu_baseline"] is None and d["config"]["parallelism"] == "scene shards x2"
    assert abs(d["value"] - 2 * 8 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]
    assert 16 * 300 < d["grasps_last_step"] < 16 * 1024          # every rank holds all 16 scenes' grasp lists
    pr = d["per_rank_ms_per_step"]
    assert len(pr["all"]) == 2 and pr["min"] <= pr["max"] and abs(pr["max"] - d["ms_per_step"]) < 1e-6 * d["ms_per_step"] + 1e-9


def test_bench_strong_scaling_two_ranks_over_gloo():
    """--global-batch: a fixed global batch of 16 cut into two contiguous shards of 8 (SURVEY.md 8e's strong-scaling run)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29519", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "6", "--warmup", "4", "--backend", "gloo", "--global-batch", "16"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["batch_per_gpu"] == 8
    assert abs(d["value"] - 16 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]
    assert 16 * 300 < d["grasps_last_step"] < 16 * 1024


def _self_launch(extra):
    """`python bench.py --gpus 2 ...` with NO launcher in front: bench.py starts its two ranks itself"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "6",
                          "--warmup", "4"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "launching 2 ranks" in out.stderr
    return _last_json(out.stdout)


def test_bench_gpus_2_starts_its_own_ranks_weak():
    d = _self_launch([])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["scaling"] == "weak"
    assert len(d["rank_devices"]) == 2 and all(0 <= i < d["visible_devices"] for i in d["rank_devices"])
    assert abs(d["value"] - 2 * 8 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]
    assert 16 * 300 < d["grasps_last_step"] < 16 * 1024


def test_bench_gpus_2_starts_its_own_ranks_strong():
    d = _self_launch(["--global-batch", "16"])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "strong" and d["config"]["batch_per_gpu"] == 8
    assert abs(d["value"] - 16 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]


def test_rccl_backend_runs_the_gather_collective_on_this_box():
    """the N>1 runs use torch.distributed's "nccl" backend (= RCCL on ROCm); two ranks cannot share the one GPU of the test box
    under RCCL, so this checks what can be checked here: the backend initialises (HSA_ENABLE_IPC_MODE_LEGACY=0 in the
    environment), and the very collective GraspGatherer issues -- all_gather_into_tensor of the packed (B, M*17+1) buffer on a
    side stream behind an event -- runs on it with a one-rank group and returns the rank's own lists"""
    code = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.getcwd(), "graspnet-baseline_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
g = torch.rand(8, 1024, 17, device="cuda"); c = torch.randint(0, 1024, (8,), device="cuda", dtype=torch.int32)
send = torch.empty((8, 1024 * 17 + 1), device="cuda"); send[:, :-1] = g.reshape(8, -1); send[:, -1] = c.view(torch.float32)
side = torch.cuda.Stream(); ev = torch.cuda.Event(); ev.record()
with torch.cuda.stream(side):
    side.wait_event(ev)
    recv = torch.empty_like(send)
    dist.all_gather_into_tensor(recv, send)
side.synchronize()
assert torch.equal(recv, send)
dist.barrier(); dist.destroy_process_group()
print("rccl ok", dist.is_nccl_available())
"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "rccl ok True" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])
