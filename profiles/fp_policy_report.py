"""How far the nvcc FMA question reaches (round-4 VERDICT item 3) -> profiles/parity_r05.json.

The reference builds its CUDA kernels with plain `nvcc -O2` (pointnet2/setup.py:24-27), i.e. --fmad=true: its own binary
contracts the index-deciding expressions into fused multiply-adds, in an order that cannot be observed without nvcc.  This
build evaluates them as written by default and carries both plausible contracted orders behind gn_set_fp_policy
("contracted" = fma(c,c,fma(a,a,b*b)): LLVM / NVVM DAG-combine order; "contracted_b" = fma(c,c,fma(b,b,a*a))).  HIP == oracle
bit-exact under each (tests/test_ops_gpu.py, tests/test_model_gpu.py), so the counts below are a property of the
arithmetic, measured here on the HIP path: per operator, how many INDICES differ between a contracted policy and the
as-written one on BASELINE's config[1] (8 x 20 000 uniform clouds), config[3] (50 000 points) and the example scene; and how far
the decoded grasp rows move end to end.
    python profiles/fp_policy_report.py [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graspnet-baseline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
from graspnet_amd import GraspNet, pointnet2_ext as E, knn_ext, pred_decode, set_fp_policy, synthetic as S

dev = "cuda"
sd = S.make_state_dict(0)
net = GraspNet().load_state_dict(sd)
POL = ("as_written", "contracted", "contracted_b")


def example_cloud():
    z = np.load(os.path.join(ROOT, "tests", "golden", "example_scene_n20000.npz"))     # doc/example_data, 20 000 sampled pixels
    return torch.from_numpy(z["cloud"].astype(np.float32)).reshape(1, -1, 3)


def ops_on(cloud):
    """the operator chain of one forward on `cloud` (B,N,3) under the current policy"""
    x = cloud.to(dev)
    fps = E.furthest_point_sampling(x, 2048)
    ctr = torch.gather(x, 1, fps.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    return {"fps": fps, "ctr": ctr}


def downstream(cloud, ctr):
    """ball / cylinder / three_nn on FIXED centres (the as-written sampler's), so that a differing sample does not count twice"""
    x = cloud.to(dev)
    B = x.shape[0]
    rng = np.random.default_rng(7)
    v = torch.from_numpy(rng.standard_normal((B * 1024, 3)).astype(np.float32))
    from oracle import model_oracle as MO      # host-side table builder only (test infrastructure; this is a profiles/ script)
    rot = MO.viewpoint_to_matrix(v, torch.zeros(B * 1024)).view(B, 1024, 9).contiguous().to(dev)
    seeds = ctr[:, :1024].contiguous()
    out = {"ball_sa1": E.ball_query(ctr, x, 0.04, 64)}
    for d, h in enumerate((0.01, 0.02, 0.03, 0.04)):
        out["cylinder_d%d" % d] = E.cylinder_query(seeds, x, rot, 0.05, -0.02, h, 64)
    c2, c3 = ctr[:, :1024].contiguous(), ctr[:, :512].contiguous()
    d2, i3 = E.three_nn(c2, c3)
    out["three_nn_idx"], out["three_nn_dist2"] = i3, d2
    q = x[:, :1024].transpose(1, 2).contiguous()
    r = x.transpose(1, 2).contiguous()
    ki = torch.zeros(B, 1, 1024, dtype=torch.int64, device=dev)
    knn_ext.knn(r, q + 0.003, ki)
    out["knn_k1"] = ki
    return out


def rows_moved(pa, pb):
    """decoded rows matched by seed centre: rows only on one side, rows whose rotation / depth differ, max score change"""
    st = {"rows_a": 0, "rows_b": 0, "only_one_side": 0, "class_changes": 0, "score_max_abs": 0.0, "score_scale": 0.0}
    for a, b in zip(pa, pb):
        a, b = a.cpu(), b.cpu()
        ka = {tuple(r[13:16].tolist()): r for r in a}
        kb = {tuple(r[13:16].tolist()): r for r in b}
        st["rows_a"] += len(ka); st["rows_b"] += len(kb)
        st["only_one_side"] += len(set(ka) ^ set(kb))
        for k in set(ka) & set(kb):
            x, y = ka[k], kb[k]
            if not torch.allclose(x[4:13], y[4:13], atol=1e-5) or float(x[3]) != float(y[3]):
                st["class_changes"] += 1
                continue
            st["score_max_abs"] = max(st["score_max_abs"], abs(float(x[0]) - float(y[0])))
            st["score_scale"] = max(st["score_scale"], abs(float(y[0])))
    return st


cases = {"config[1]: 8 x 20000 uniform": S.make_batch(1000, 8, 20000, "uniform"),
         "config[1] surface: 8 x 20000": S.make_batch(1000, 8, 20000, "surface"),
         "config[3]: 2 x 50000 uniform": S.make_batch(3000, 2, 50000, "uniform")}
ex = example_cloud()
if ex is not None:
    cases["config[4] example scene: 1 x %d" % ex.shape[1]] = ex
report = {"what": __doc__.split("\n\n")[0], "policies": {"as_written": "a*a + b*b + c*c, one rounding per operation (default)",
                                                        "contracted": "fma(c, c, fma(a, a, b*b))  [LLVM / NVVM order]",
                                                        "contracted_b": "fma(c, c, fma(b, b, a*a))"}, "cases": {}}
for name, cloud in cases.items():
    rec = {}
    res = {}
    for pol in POL:
        set_fp_policy(pol)
        res[pol] = ops_on(cloud)
    ctr = res["as_written"]["ctr"]
    down = {}
    for pol in POL:
        set_fp_policy(pol)
        down[pol] = downstream(cloud, ctr)
        ep = net({"point_clouds": cloud.to(dev)})
        down[pol]["_rows"] = pred_decode(ep)
        down[pol]["_sa1"] = ep["sa1_inds"].clone()
        down[pol]["_view"] = ep["grasp_top_view_inds"].clone()
    torch.cuda.synchronize()
    for pol in POL[1:]:
        r = {}
        a, b = res["as_written"]["fps"], res[pol]["fps"]
        diff = (a != b)
        first = [int(torch.nonzero(diff[i])[0]) if diff[i].any() else None for i in range(a.shape[0])]
        r["fps_2048"] = {"indices": int(a.numel()), "differ": int(diff.sum()), "clouds_with_a_difference": int(diff.any(1).sum()),
                         "first_differing_step_per_cloud": first,
                         "note": "one different pick re-routes the rest of the chain: `differ` counts everything behind it"}
        for k in down["as_written"]:
            if k.startswith("_"):
                continue
            x, y = down["as_written"][k], down[pol][k]
            r[k] = {"entries": int(x.numel()), "differ": int((x != y).sum())}
        r["forward_sa1_inds_differ"] = int((down["as_written"]["_sa1"] != down[pol]["_sa1"]).sum())
        r["forward_top_view_differ"] = int((down["as_written"]["_view"] != down[pol]["_view"]).sum())
        r["decoded_rows"] = rows_moved(down["as_written"]["_rows"], down[pol]["_rows"])
        rec[pol + " vs as_written"] = r
    report["cases"][name] = rec
    print(name, json.dumps({k: {kk: (vv.get("differ") if isinstance(vv, dict) else vv) for kk, vv in v.items()} for k, v in rec.items()}))
set_fp_policy("as_written")
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_r05_fp_policy.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(report, open(out, "w"), indent=1)
print("written", out)
