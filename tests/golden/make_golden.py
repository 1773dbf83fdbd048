"""Generate tests/golden/*.npz by running the REFERENCE python model
(/root/reference/models/graspnet.py: GraspNet.forward + pred_decode) in the
build container on top of the CPU oracle operators.

Only runs where /root/reference exists (never on the GPU box).  The reference
source is imported from where it lies and nothing of it is written to the
repo: the fixtures hold seeds, inputs-by-seed and output tensors only.

What it pins:
  * oracle/model_oracle.py (our restatement of everything above the operator
    boundary) is bit-identical to the reference python on the same operators;
  * graspnet_amd.synthetic.make_state_dict() uses the reference's state-dict
    key layout (load_state_dict(strict=True) into the reference model);
  * the view template and viewpoint->matrix helper match loss_utils.py.
The seven pointnet2 operators have no CPU implementation in the reference
("CPU not supported"), so for them the fixtures are regression vectors of the
oracle restatement itself -- parity unpinned by upstream.

Usage: python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("GRASPNET_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))

from oracle import oracle as O            # noqa: E402
from oracle import model_oracle as MO     # noqa: E402
from graspnet_amd import synthetic as S   # noqa: E402


def import_reference():
    """Register the oracle as `pointnet2._ext` / `knn_pytorch.knn_pytorch` and
    import the reference model module."""
    pkg = types.ModuleType("pointnet2")
    pkg.__path__ = []
    pkg._ext = O
    sys.modules["pointnet2"] = pkg
    sys.modules["pointnet2._ext"] = O
    kp = types.ModuleType("knn_pytorch")
    kp.__path__ = []
    inner = types.ModuleType("knn_pytorch.knn_pytorch")
    inner.knn = O.knn
    kp.knn_pytorch = inner
    sys.modules["knn_pytorch"] = kp
    sys.modules["knn_pytorch.knn_pytorch"] = inner
    for sub in ("", "models", "pointnet2", "utils", "knn"):
        sys.path.append(os.path.join(REF, sub))
    import graspnet as ref_graspnet  # reference models/graspnet.py
    return ref_graspnet


def f64sum(t):
    return float(t.double().sum())


def pack_endpoints(ep, preds, tag, out):
    """Small tensors whole; big ones as float64 checksum + strided sample."""
    whole = ["sa1_inds", "sa2_inds", "fp2_inds", "fp2_xyz", "sa3_xyz", "sa4_xyz", "objectness_score",
             "grasp_top_view_inds", "grasp_top_view_score", "grasp_top_view_xyz", "grasp_top_view_rot"]
    for k in whole:
        out["%s/%s" % (tag, k)] = ep[k].numpy()
    big = ["sa1_features", "sa2_features", "sa3_features", "sa4_features", "fp2_features", "view_score",
           "grasp_score_pred", "grasp_angle_cls_pred", "grasp_width_pred", "grasp_tolerance_pred"]
    for k in big:
        t = ep[k].contiguous()
        out["%s/%s/sum" % (tag, k)] = np.float64(f64sum(t))
        out["%s/%s/abssum" % (tag, k)] = np.float64(float(t.double().abs().sum()))
        out["%s/%s/shape" % (tag, k)] = np.array(t.shape, dtype=np.int64)
        out["%s/%s/sample" % (tag, k)] = t.reshape(-1)[::97].numpy().copy()
    for i, p in enumerate(preds):
        out["%s/pred/%d" % (tag, i)] = p.numpy()


def main():
    ref = import_reference()
    torch.manual_seed(0)
    sd = S.make_state_dict(0)
    net = ref.GraspNet(input_feature_dim=0, num_view=300, num_angle=12, num_depth=4,
                       cylinder_radius=0.05, hmin=-0.02, hmax_list=[0.01, 0.02, 0.03, 0.04],
                       is_training=False)
    missing = net.load_state_dict(sd, strict=True)
    print("load_state_dict(strict=True):", missing)
    net.eval()
    manifest = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert set(manifest) == set(sd), "state-dict key layout mismatch"

    out = {}
    # ---- helpers of loss_utils.py
    import loss_utils as ref_lu
    tv = ref_lu.generate_grasp_views(300)
    assert torch.equal(tv, MO.grasp_views(300)), "view template differs from reference"
    out["views300"] = tv.numpy()
    rng = np.random.default_rng(7)
    tw = torch.from_numpy(rng.standard_normal((1000, 3)).astype(np.float32))
    tw[:8] = torch.tensor([[0, 0, 1.0], [0, 0, -1.0], [0, 0, 2.0], [0, 0, -0.5],
                           [1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, -1, 0]], dtype=torch.float32)
    ang = torch.from_numpy(rng.uniform(0, np.pi, 1000).astype(np.float32))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rm = ref_lu.batch_viewpoint_params_to_matrix(tw.clone(), ang)
    mm = MO.viewpoint_to_matrix(tw.clone(), ang)
    assert torch.equal(rm, mm), "viewpoint_to_matrix differs from reference"
    out["v2m/towards"] = tw.numpy(); out["v2m/angle"] = ang.numpy(); out["v2m/matrix"] = rm.numpy()

    # ---- model level: B=2 (one uniform scene, one surface scene with duplicates)
    clouds = torch.stack([torch.from_numpy(S.make_cloud(1000, 20000, "uniform")),
                          torch.from_numpy(S.make_cloud(1001, 20000, "surface"))], 0)
    out["model/cloud_seeds"] = np.array([1000, 1001]); out["model/state_seed"] = np.array([0])
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ep_ref = net({"point_clouds": clouds.clone()})
        preds_ref = ref.pred_decode(ep_ref)
    ep_or = MO.forward({"point_clouds": clouds.clone()}, sd)
    preds_or = MO.pred_decode(ep_or)
    nbad = 0
    for k, v in ep_ref.items():
        if v is None:
            continue
        same = torch.equal(v, ep_or[k])
        if not same:
            nbad += 1
            print("MISMATCH", k, (v.float() - ep_or[k].float()).abs().max().item())
    for a, b in zip(preds_ref, preds_or):
        if not torch.equal(a, b):
            nbad += 1
            print("MISMATCH pred_decode")
    assert nbad == 0, "model_oracle is not bit-identical to the reference python"
    print("model_oracle == reference python: bit-identical on", len(ep_ref), "end_points keys;",
          "grasps per scene:", [int(p.shape[0]) for p in preds_ref],
          "distinct top views:", [int(torch.unique(ep_ref["grasp_top_view_inds"][i]).numel()) for i in range(2)])
    pack_endpoints(ep_ref, preds_ref, "model", out)
    np.savez_compressed(os.path.join(HERE, "model_b2_n20000.npz"), **out)

    with open(os.path.join(HERE, "state_dict_manifest.txt"), "w") as f:
        for k in sorted(manifest):
            f.write("%s %s\n" % (k, "x".join(str(d) for d in manifest[k]) or "scalar"))

    # ---- operator level regression vectors (oracle restatement; small + adversarial)
    ops = {}
    rng = np.random.default_rng(11)
    for name, n in (("n1000", 1000), ("n300", 300), ("n4096dup", 4096)):
        pts = rng.uniform(-0.3, 0.3, size=(2, n, 3)).astype(np.float32)
        pts[..., 2] += 0.5
        if "dup" in name:
            pts[:, n // 2:] = pts[:, :n - n // 2]          # exact duplicates -> ties
            pts[0, 5] = 0.0; pts[0, 6] = [0.01, 0.01, 0.01]  # |p|^2 <= 1e-3 -> skipped by FPS
            pts[1, 0] = 0.0                                 # index 0 itself skipped
        x = torch.from_numpy(pts)
        m = min(256, n // 2)
        fi = O.furthest_point_sampling(x, m)
        xt = x.transpose(1, 2).contiguous()
        nx = O.gather_points(xt, fi).transpose(1, 2).contiguous()
        bq = O.ball_query(nx, x, 0.04, 16)
        bq2 = O.ball_query(nx, x, 0.0005, 8)     # mostly empty / tiny balls
        rot = MO.viewpoint_to_matrix(torch.from_numpy(rng.standard_normal((2 * m, 3)).astype(np.float32)),
                                     torch.zeros(2 * m)).view(2, m, 9).contiguous()
        cq = O.cylinder_query(nx, x, rot, 0.05, -0.02, 0.04, 16)
        d2, ti = O.three_nn(x[:, :200].contiguous(), nx)
        ops[name + "/pts"] = pts; ops[name + "/fps"] = fi.numpy(); ops[name + "/ball"] = bq.numpy()
        ops[name + "/ball_tiny"] = bq2.numpy(); ops[name + "/rot"] = rot.numpy()
        ops[name + "/cyl"] = cq.numpy(); ops[name + "/nn_d2"] = d2.numpy(); ops[name + "/nn_idx"] = ti.numpy()
    np.savez_compressed(os.path.join(HERE, "ops_small.npz"), **ops)
    for fn in ("model_b2_n20000.npz", "ops_small.npz"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)) // 1024, "KiB")


if __name__ == "__main__":
    main()
