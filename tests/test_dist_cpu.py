"""CPU (gloo, world_size 2): scene sharding and the all-gather of padded grasp lists."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "graspnet-baseline_amd"))
    from graspnet_amd import dist as gdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B_global, M = 6, 16
        lo, hi = gdist.shard_range(B_global, rank, world)
        g = torch.zeros(hi - lo, M, 17)
        c = torch.zeros(hi - lo, dtype=torch.int32)
        for i, scene in enumerate(range(lo, hi)):      # scene s holds s+1 grasps whose score is s
            c[i] = scene + 1
            g[i, :scene + 1, 0] = float(scene)
        ga, ca = gdist.all_gather_grasps(g, c)
        gb_, cb_, done = gdist.GraspGatherer().gather(g, c)       # CPU tensors: gathered inline, no event
        assert done is None and torch.equal(gb_, ga) and torch.equal(cb_, ca)
        lst = gdist.to_list(ga, ca)
        ok = (ga.shape == (B_global, M, 17) and ca.tolist() == [s + 1 for s in range(B_global)]
              and all(t.shape == (s + 1, 17) and bool((t[:, 0] == s).all()) for s, t in enumerate(lst)))
        q.put((rank, ok, (lo, hi)))
    finally:
        dist.destroy_process_group()


def test_shard_ranges_partition_the_batch():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "graspnet-baseline_amd"))
    from graspnet_amd.dist import shard_range
    for gb in (1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            r = [shard_range(gb, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == gb
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_all_gather_grasps_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert sorted(r for _, _, r in res) == [(0, 3), (3, 6)]


def _bench(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, cwd=root, capture_output=True, text=True,
                          timeout=600, env=env)


def test_bench_gpus_n_launches_n_ranks_by_itself():
    """`python bench.py --gpus 3` with no launcher in front starts three ranks as child processes (no GPU call in the
    parent) and relays rank 0's one JSON line; --rendezvous-only stops after the process group has formed"""
    import json
    out = _bench(["--gpus", "3", "--backend", "gloo", "--rendezvous-only"])
    assert out.returncode == 0, out.stderr[-2000:]
    assert "launching 3 ranks" in out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["ranks_seen"] == 3 and d["ranks"] == [0, 1, 2] and d["backend"] == "gloo"


def test_bench_refuses_a_world_size_that_is_not_gpus():
    out = _bench(["--gpus", "4", "--rendezvous-only"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "--gpus 4" in out.stderr and "WORLD_SIZE=1" in out.stderr


def test_bench_parent_exits_nonzero_when_a_rank_fails():
    """no GPU in the build container: the ranks of a real run die at their first device call and the parent says so"""
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    out = _bench(["--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"])
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def _bench_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_sets_the_ipc_mode_for_both_launch_forms():
    """HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC: RCCL needs it on this pool) is a setdefault at import time, so ranks that
    arrive from an external torch.distributed.run get it too -- and a caller's explicit choice is kept"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import importlib.util, os; s = importlib.util.spec_from_file_location('b', 'bench.py'); m = importlib.util.module_from_spec(s); "
            "s.loader.exec_module(m); print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY'], os.environ['GPU_MAX_HW_QUEUES'])")
    for given, want in ((None, "0"), ("1", "1")):
        env = {k: v for k, v in os.environ.items() if k not in ("HSA_ENABLE_IPC_MODE_LEGACY", "GPU_MAX_HW_QUEUES")}
        if given is not None:
            env["HSA_ENABLE_IPC_MODE_LEGACY"] = given
        out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, out.stderr[-1500:]
        assert out.stdout.split() == [want, "16"]


def test_bench_rank_and_device_checks():
    b = _bench_module()
    # more local ranks than visible GPUs under RCCL: refused in words before init_process_group; gloo rehearsals may share
    assert "only 1 visible GPU" in b.check_local_ranks(2, 2, 1, "nccl")
    assert b.check_local_ranks(2, 2, 1, "gloo") is None and b.check_local_ranks(8, 8, 8, "nccl") is None
    assert b.check_local_ranks(1, 1, 0, "nccl") is None
    ok = [(r, r, r, "0000:%02x:00" % (5 + r)) for r in range(8)]
    assert b.check_rank_devices(ok, 8, "nccl") is None
    dup = ok[:7] + [(7, 7, 3, ok[3][3])]
    msg = b.check_rank_devices(dup, 8, "nccl")
    assert msg and "share device" in msg and ok[3][3] in msg and "7 -> cuda:3" in msg
    assert b.check_rank_devices(dup, 8, "gloo") is None              # ranks sharing a card: a gloo rehearsal
    assert "7 answered" in b.check_rank_devices(ok[:7], 8, "nccl")
    assert "answered" in b.check_rank_devices(ok[:7] + [ok[0]], 8, "nccl")


def test_bench_refuses_duplicate_devices_at_the_rendezvous():
    """three ranks that all claim one device: with the RCCL rule applied the run stops with the rank -> device table and
    prints no JSON line (GN_BENCH_FAKE_DEVICE / GN_BENCH_CHECK_AS stand in for GPUs on a CPU-only box)"""
    out = _bench(["--gpus", "3", "--backend", "gloo", "--rendezvous-only"],
                 {"GN_BENCH_FAKE_DEVICE": "0000:05:00", "GN_BENCH_CHECK_AS": "nccl"})
    assert out.returncode != 0 and "share device" in out.stderr and "0000:05:00" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
