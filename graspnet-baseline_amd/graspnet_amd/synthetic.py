"""Deterministic synthetic scenes and weights (numpy Generator streams, so the
same bytes are produced in the build container and on the GPU box).

Weights use the reference's state-dict key layout (keys are
those of models/graspnet.py:63-68 -> backbone.py:30-67, modules.py:22-232,
pytorch_utils.py:11-120), so the dict loads into the reference model with
strict=True as well as into graspnet_amd.GraspNet.
"""
import os

import numpy as np
import torch

# (npoint, radius, nsample, mlp incl. the +3 xyz channels) -- models/backbone.py:30-64
SA_CFG = (
    (2048, 0.04, 64, (3, 64, 64, 128)),
    (1024, 0.1, 32, (131, 128, 128, 256)),
    (512, 0.2, 16, (259, 128, 128, 256)),
    (256, 0.3, 16, (259, 128, 128, 256)),
)
FP_MLP = (512, 256, 256)          # models/backbone.py:66-67
CROP_MLP = (3, 64, 128, 256)      # models/modules.py:98
NUM_VIEW, NUM_ANGLE, NUM_DEPTH = 300, 12, 4   # models/graspnet.py:64


def make_cloud(seed, n=20000, variant="uniform"):
    """One (n,3) float32 table-top cloud in the camera frame.

    uniform : x,y ~ U(-0.3,0.3), z ~ U(0.4,0.6)
    surface : plane z=0.55 (+1 mm noise) plus 8 boxes/spheres sampled on their
              surfaces, then subsampled to n the way the reference's dataset
              does (dataset/graspnet_dataset.py:139-145: without replacement if
              enough points, else with replacement -> duplicates).
    """
    rng = np.random.default_rng(seed)
    if variant == "uniform":
        xy = rng.uniform(-0.3, 0.3, size=(n, 2))
        z = rng.uniform(0.4, 0.6, size=(n, 1))
        return np.concatenate([xy, z], 1).astype(np.float32)
    if variant == "surface":
        pts = []
        npl = int(n * 0.55)
        pl = np.stack([rng.uniform(-0.3, 0.3, npl), rng.uniform(-0.3, 0.3, npl),
                       0.55 + rng.normal(0, 0.001, npl)], 1)
        pts.append(pl)
        per = int(n * 0.4) // 8
        for o in range(8):
            c = np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), 0.5])
            if o % 2 == 0:   # sphere
                r = rng.uniform(0.02, 0.05)
                v = rng.normal(size=(per, 3))
                v /= np.linalg.norm(v, axis=1, keepdims=True)
                pts.append(c + r * v)
            else:            # box surface
                h = rng.uniform(0.02, 0.05, 3)
                u = rng.uniform(-1, 1, size=(per, 3))
                ax = rng.integers(0, 3, per)
                sg = rng.choice([-1.0, 1.0], per)
                u[np.arange(per), ax] = sg
                pts.append(c + u * h)
        src = np.concatenate(pts, 0)
        if len(src) >= n:
            idxs = rng.choice(len(src), n, replace=False)
        else:
            idxs1 = np.arange(len(src))
            idxs2 = rng.choice(len(src), n - len(src), replace=True)
            idxs = np.concatenate([idxs1, idxs2], 0)
        return src[idxs].astype(np.float32)
    raise ValueError("unknown cloud variant %r" % (variant,))


def make_batch(seed0, batch, n=20000, variant="uniform"):
    """(batch,n,3) float32 torch tensor; scene i uses seed seed0+i."""
    return torch.from_numpy(np.stack([make_cloud(seed0 + i, n, variant) for i in range(batch)], 0))


def _conv_bn(sd, rng, prefix_conv, prefix_bn, cout, cin, kdims, bias):
    w = rng.normal(0.0, np.sqrt(2.0 / cin), size=(cout, cin) + (1,) * kdims)
    sd[prefix_conv + ".weight"] = torch.from_numpy(w.astype(np.float32))
    if bias:
        sd[prefix_conv + ".bias"] = torch.from_numpy(rng.normal(0, 0.1, cout).astype(np.float32))
    if prefix_bn is not None:
        sd[prefix_bn + ".weight"] = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32))
        sd[prefix_bn + ".bias"] = torch.from_numpy(rng.normal(0, 0.1, cout).astype(np.float32))
        sd[prefix_bn + ".running_mean"] = torch.from_numpy(rng.normal(0, 0.1, cout).astype(np.float32))
        sd[prefix_bn + ".running_var"] = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32))
        sd[prefix_bn + ".num_batches_tracked"] = torch.tensor(1000, dtype=torch.int64)


def _shared_mlp(sd, rng, prefix, spec):
    for i in range(len(spec) - 1):
        _conv_bn(sd, rng, "%s.layer%d.conv" % (prefix, i), "%s.layer%d.bn.bn" % (prefix, i),
                 spec[i + 1], spec[i], 2, bias=False)


def make_state_dict(seed=0, calibrated=True, num_view=NUM_VIEW, variant=None):
    """Seeded, perturbed weights (random BN statistics/affine, non-zero biases):
    default-initialised BN makes every seed pick the same view.
    With calibrated=True and data/bn_calib_s<seed>.npz present (written by
    tests/golden/calibrate_bn.py), the BN running statistics are replaced by
    calibrated ones so the random network behaves like a trained one (diverse
    approach views, non-trivial objectness mask).

    variant (parity fixtures, tests/golden/make_golden.py):
      "unit": the two heads' last layers rescaled by powers of two (score rows x 1/4, tolerance x 1/128) so that the
              raw scores stay within ~[-1.2, 1.2], the tolerances within ~0.05 = GRASP_MAX_TOLERANCE (what training
              drives them to, loss_utils.py:9) and the decoded score*tolerance/0.05 (models/graspnet.py:121) is O(1)
              -- the regime in which "within 1e-4 on grasp scores" is a plain absolute statement;
      "tied": exact arg-max ties by construction -- the view rows of ApproachNet's last layer are equal in triples
              (views 3k, 3k+1, 3k+2) and the angle-class rows of OperationNet's last layer in pairs (2k, 2k+1), so
              torch.max / torch.argmax (modules.py:64, graspnet.py:93) decide by their first-maximum rule on every seed."""
    rng = np.random.default_rng(seed)
    sd = {}
    bb = "view_estimator.backbone"
    for i, (_, _, _, spec) in enumerate(SA_CFG):
        _shared_mlp(sd, rng, "%s.sa%d.mlp_module" % (bb, i + 1), spec)
    for i in (1, 2):
        _shared_mlp(sd, rng, "%s.fp%d.mlp" % (bb, i), FP_MLP)
    vp = "view_estimator.vpmodule"
    nv = 2 + num_view
    _conv_bn(sd, rng, vp + ".conv1", vp + ".bn1", 256, 256, 1, True)
    _conv_bn(sd, rng, vp + ".conv2", vp + ".bn2", nv, 256, 1, True)
    _conv_bn(sd, rng, vp + ".conv3", None, nv, nv, 1, True)
    _shared_mlp(sd, rng, "grasp_generator.crop.mlps", CROP_MLP)
    for name, cout in (("operation", 3 * NUM_ANGLE), ("tolerance", NUM_ANGLE)):
        p = "grasp_generator." + name
        _conv_bn(sd, rng, p + ".conv1", p + ".bn1", 128, 256, 1, True)
        _conv_bn(sd, rng, p + ".conv2", p + ".bn2", 128, 128, 1, True)
        _conv_bn(sd, rng, p + ".conv3", None, cout, 128, 1, True)
    if calibrated:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "bn_calib_s%d.npz" % seed)
        if os.path.exists(path):
            with np.load(path) as z:
                for k in z.files:
                    if num_view != NUM_VIEW and tuple(sd[k].shape) != z[k].shape:
                        continue            # the calibration file is for 300 views: layers sized by num_view keep their seeds
                    assert k in sd and tuple(sd[k].shape) == z[k].shape, k
                    sd[k] = torch.from_numpy(z[k].copy())
    if variant == "unit":
        op, tol = "grasp_generator.operation.conv3", "grasp_generator.tolerance.conv3"
        for k in (".weight", ".bias"):
            sd[op + k] = sd[op + k].clone()
            sd[op + k][0:NUM_ANGLE] *= 0.25
            sd[tol + k] = sd[tol + k] * (1.0 / 128.0)
    elif variant == "tied":
        vp3, op3 = "view_estimator.vpmodule.conv3", "grasp_generator.operation.conv3"
        for k in (".weight", ".bias"):
            t = sd[vp3 + k].clone()
            for v in range(num_view):
                t[2 + v] = t[2 + (v // 3) * 3]
            sd[vp3 + k] = t
            t = sd[op3 + k].clone()
            for a in range(NUM_ANGLE):
                t[NUM_ANGLE + a] = t[NUM_ANGLE + (a // 2) * 2]
            sd[op3 + k] = t
    elif variant is not None:
        raise ValueError("unknown weight variant %r" % (variant,))
    return sd
