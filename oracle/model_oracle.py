"""CPU restatement of the reference's python model (everything ABOVE the
operator boundary) on top of the C oracle operators.

TEST INFRASTRUCTURE ONLY (see oracle/gn_oracle.c header).  Used by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg ("port").

It is a functional re-expression -- weights come from a state-dict with the
reference's key layout -- that issues the same torch-CPU operations in the
same order and memory layout as the reference modules, so that on one machine
it is bit-identical to the reference python run on the same operators
(tests/golden/make_golden.py checks exactly that in the build container, where
the reference can be imported, and tests/golden/*.npz pin it elsewhere).

Reference locations restated here:
  models/backbone.py:78-130            -> backbone()
  pointnet2/pointnet2_modules.py:210-272 (PointnetSAModuleVotes.forward) -> sa_module()
  pointnet2/pointnet2_utils.py:313-372 (QueryAndGroup.forward)          -> inside sa_module()
  pointnet2/pointnet2_modules.py:371-416 (PointnetFPModule.forward)      -> fp_module()
  pointnet2/pytorch_utils.py:11-120 (SharedMLP = conv1x1(no bias)+BN+ReLU) -> shared_mlp()
  models/modules.py:41-77   (ApproachNet.forward)   -> approach_net()
  models/modules.py:109-141 (CloudCrop.forward) + pointnet2_utils.py:488-555 -> cloud_crop()
  models/modules.py:169-191, 214-232 (OperationNet / ToleranceNet)      -> head_net()
  utils/loss_utils.py:41-66, 68-96                                       -> grasp_views(), viewpoint_to_matrix()
  models/graspnet.py:70-75, 77-134                                       -> forward(), pred_decode()
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as ops

GRASP_MAX_WIDTH = 0.1        # utils/loss_utils.py:8
GRASP_MAX_TOLERANCE = 0.05   # utils/loss_utils.py:9
BN_EPS = 1e-5                # torch.nn.BatchNorm default, pytorch_utils.py:40-60

SA_CFG = ((2048, 0.04, 64), (1024, 0.1, 32), (512, 0.2, 16), (256, 0.3, 16))  # backbone.py:30-64


# When set to a numpy Generator (tests/golden/calibrate_bn.py only), every BN
# first overwrites its running statistics with (perturbed) batch statistics of
# its input, the way training-mode BN would have left them.  Never set during
# parity runs.
CALIBRATE_RNG = None

# bench.py's cpu_baseline leg sets this to a dict to collect wall-clock seconds per stage (BASELINE.md section 4.3:
# FPS, ball query, grouping, SA MLPs, FP, ApproachNet, cylinder query, CloudCrop MLP, heads, decode); None = no timing
STAGE_TIMES = None


class _stage:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if STAGE_TIMES is not None:
            import time
            self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        if STAGE_TIMES is not None:
            import time
            STAGE_TIMES[self.name] = STAGE_TIMES.get(self.name, 0.0) + time.perf_counter() - self.t0
        return False


def _bn(x, sd, p):
    if CALIBRATE_RNG is not None:
        dims = [d for d in range(x.dim()) if d != 1]
        m = x.mean(dim=dims)
        v = x.var(dim=dims, unbiased=False)
        c = m.numel()
        noise = torch.from_numpy(CALIBRATE_RNG.normal(0, 0.15, c).astype(np.float32))
        scale = torch.from_numpy(CALIBRATE_RNG.uniform(0.8, 1.25, c).astype(np.float32))
        sd[p + ".running_mean"] = (m + noise * v.sqrt()).contiguous()
        sd[p + ".running_var"] = (v * scale + 1e-6).contiguous()
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.1, BN_EPS)


def shared_mlp(x, sd, prefix, nlayers):
    """(B,C,np,ns) -> (B,C',np,ns); pytorch_utils.py:11-36."""
    for i in range(nlayers):
        x = F.conv2d(x, sd["%s.layer%d.conv.weight" % (prefix, i)])
        x = _bn(x, sd, "%s.layer%d.bn.bn" % (prefix, i))
        x = F.relu(x)
    return x


def _group(features, idx):
    """group_points for any dtype (float64 'truth' mode uses torch indexing; a pure copy either way)."""
    if features.dtype == torch.float32:
        return ops.group_points(features.contiguous(), idx)
    b, c, n = features.shape
    ix = idx.long().reshape(b, 1, -1).expand(-1, c, -1)
    return torch.gather(features, 2, ix).reshape(b, c, idx.shape[1], idx.shape[2])


def sa_module(xyz, features, sd, prefix, npoint, radius, nsample, dt=torch.float32):
    xyz_flipped = xyz.transpose(1, 2).contiguous()
    with _stage("fps"):
        inds = ops.furthest_point_sampling(xyz, npoint)
    with _stage("grouping"):
        new_xyz = ops.gather_points(xyz_flipped, inds).transpose(1, 2).contiguous()
    with _stage("ball_query"):
        idx = ops.ball_query(new_xyz, xyz, radius, nsample)
    with _stage("grouping"):
        grouped_xyz = ops.group_points(xyz_flipped, idx).to(dt)
        grouped_xyz -= new_xyz.to(dt).transpose(1, 2).unsqueeze(-1)
        grouped_xyz /= radius                      # normalize_xyz=True, backbone.py:36
        if features is not None:
            grouped = torch.cat([grouped_xyz, _group(features, idx)], dim=1)
        else:
            grouped = grouped_xyz
    with _stage("sa_mlp"):
        new_features = shared_mlp(grouped, sd, prefix + ".mlp_module", 3)
        new_features = F.max_pool2d(new_features, kernel_size=[1, new_features.size(3)]).squeeze(-1)
    return new_xyz, new_features, inds


def fp_module(unknown, known, unknow_feats, known_feats, sd, prefix):
    dist2, idx = ops.three_nn(unknown, known)
    dt = known_feats.dtype
    dist = torch.sqrt(dist2.to(dt))            # pointnet2_utils.py:138-140
    dist_recip = 1.0 / (dist + 1e-8)
    norm = torch.sum(dist_recip, dim=2, keepdim=True)
    weight = dist_recip / norm
    if dt == torch.float32:
        interpolated = ops.three_interpolate(known_feats.contiguous(), idx, weight.contiguous())
    else:
        b, c, m = known_feats.shape
        interpolated = sum(torch.gather(known_feats, 2, idx[:, :, i].long().unsqueeze(1).expand(-1, c, -1))
                           * weight[:, :, i].unsqueeze(1) for i in range(3))
    new_features = torch.cat([interpolated, unknow_feats], dim=1).unsqueeze(-1)
    return shared_mlp(new_features, sd, prefix + ".mlp", 2).squeeze(-1)


def backbone(pointcloud, sd, end_points, dt=torch.float32):
    p = "view_estimator.backbone"
    xyz = pointcloud[..., 0:3].contiguous()
    end_points["input_xyz"] = xyz
    end_points["input_features"] = None
    features = None
    for li, (npoint, radius, nsample) in enumerate(SA_CFG):
        xyz, features, inds = sa_module(xyz, features, sd, "%s.sa%d" % (p, li + 1), npoint, radius, nsample, dt)
        if li < 2:
            end_points["sa%d_inds" % (li + 1)] = inds
        end_points["sa%d_xyz" % (li + 1)] = xyz
        end_points["sa%d_features" % (li + 1)] = features
    with _stage("fp"):
        f = fp_module(end_points["sa3_xyz"], end_points["sa4_xyz"], end_points["sa3_features"],
                      end_points["sa4_features"], sd, p + ".fp1")
    end_points["_fp1_features"] = f            # private (not a reference key): stage-level error attribution
    with _stage("fp"):
        f = fp_module(end_points["sa2_xyz"], end_points["sa3_xyz"], end_points["sa2_features"], f, sd, p + ".fp2")
    end_points["fp2_features"] = f
    end_points["fp2_xyz"] = end_points["sa2_xyz"]
    num_seed = end_points["fp2_xyz"].shape[1]
    end_points["fp2_inds"] = end_points["sa1_inds"][:, 0:num_seed]
    return f, end_points["fp2_xyz"], end_points


def grasp_views(n=300):
    """Fibonacci-lattice view template, float64 math -> float32 (loss_utils.py:41-66)."""
    phi = (np.sqrt(5) - 1) / 2
    i = np.arange(n, dtype=np.float64)
    zi = (2 * i + 1) / n - 1
    xi = np.sqrt(1 - zi ** 2) * np.cos(2 * i * np.pi * phi)
    yi = np.sqrt(1 - zi ** 2) * np.sin(2 * i * np.pi * phi)
    return torch.from_numpy(np.stack([xi, yi, zi], axis=1).astype(np.float32))


def viewpoint_to_matrix(towards, angle):
    """loss_utils.py:68-96.  (torch.cross there relies on the deprecated default
    dim, which is dim=-1 whenever the row count is not 3.)"""
    axis_x = towards
    ones = torch.ones(axis_x.shape[0], dtype=axis_x.dtype)
    zeros = torch.zeros(axis_x.shape[0], dtype=axis_x.dtype)
    axis_y = torch.stack([-axis_x[:, 1], axis_x[:, 0], zeros], dim=-1)
    mask_y = (torch.norm(axis_y, dim=-1) == 0)
    axis_y[mask_y, 1] = 1
    axis_x = axis_x / torch.norm(axis_x, dim=-1, keepdim=True)
    axis_y = axis_y / torch.norm(axis_y, dim=-1, keepdim=True)
    axis_z = torch.linalg.cross(axis_x, axis_y, dim=-1)
    sin = torch.sin(angle)
    cos = torch.cos(angle)
    r1 = torch.stack([ones, zeros, zeros, zeros, cos, -sin, zeros, sin, cos], dim=-1).reshape([-1, 3, 3])
    r2 = torch.stack([axis_x, axis_y, axis_z], dim=-1)
    return torch.matmul(r2, r1)


def approach_net(seed_xyz, seed_features, sd, end_points, num_view=300, force_top_view_inds=None):
    p = "view_estimator.vpmodule"
    B, num_seed, _ = seed_xyz.size()
    f = F.relu(_bn(F.conv1d(seed_features, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"]), sd, p + ".bn1"))
    f = F.relu(_bn(F.conv1d(f, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"]), sd, p + ".bn2"))
    f = F.conv1d(f, sd[p + ".conv3.weight"], sd[p + ".conv3.bias"])
    end_points["objectness_score"] = f[:, :2, :]
    view_score = f[:, 2:2 + num_view, :].transpose(1, 2).contiguous()
    end_points["view_score"] = view_score
    top_view_scores, top_view_inds = torch.max(view_score, dim=2)
    if force_top_view_inds is not None:        # float64 'truth' mode follows the fp32 oracle's discrete choice
        top_view_inds = force_top_view_inds
        top_view_scores = torch.gather(view_score, 2, top_view_inds.unsqueeze(-1)).squeeze(-1)
    template = grasp_views(num_view)
    vp_xyz = template[top_view_inds.reshape(-1)].view(B, num_seed, 3)
    vp_rot = viewpoint_to_matrix(-vp_xyz.view(-1, 3), torch.zeros(B * num_seed)).view(B, num_seed, 3, 3)
    end_points["grasp_top_view_inds"] = top_view_inds
    end_points["grasp_top_view_score"] = top_view_scores
    end_points["grasp_top_view_xyz"] = vp_xyz
    end_points["grasp_top_view_rot"] = vp_rot
    return end_points


def cloud_crop(seed_xyz, pointcloud, vp_rot, sd, nsample=64, radius=0.05, hmin=-0.02,
               hmax_list=(0.01, 0.02, 0.03, 0.04), return_idx=False, dt=torch.float32):
    B, num_seed, _, _ = vp_rot.size()
    xyz_trans = pointcloud.transpose(1, 2).contiguous()
    grouped, idxs = [], []
    for hmax in hmax_list:
        with _stage("cylinder_query"):
            idx = ops.cylinder_query(seed_xyz, pointcloud, vp_rot.reshape(B, num_seed, 9).contiguous(),
                                     radius, hmin, hmax, nsample)
        idxs.append(idx)
        with _stage("grouping"):
            g = ops.group_points(xyz_trans, idx).to(dt)
            g -= seed_xyz.to(dt).transpose(1, 2).unsqueeze(-1)
            g_ = g.permute(0, 2, 3, 1).contiguous()
            g_ = torch.matmul(g_, vp_rot.to(dt))
            grouped.append(g_.permute(0, 3, 1, 2).contiguous())
    with _stage("crop_mlp"):
        x = torch.stack(grouped, dim=3).view(B, -1, num_seed * len(hmax_list), nsample)
        x = shared_mlp(x, sd, "grasp_generator.crop.mlps", 3)
        x = F.max_pool2d(x, kernel_size=[1, x.size(3)])
        x = x.view(B, -1, num_seed, len(hmax_list))
    if return_idx:
        return x, idxs
    return x


def head_net(vp_features, sd, p):
    B, _, num_seed, num_depth = vp_features.size()
    x = vp_features.view(B, -1, num_seed * num_depth)
    x = F.relu(_bn(F.conv1d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"]), sd, p + ".bn1"))
    x = F.relu(_bn(F.conv1d(x, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"]), sd, p + ".bn2"))
    x = F.conv1d(x, sd[p + ".conv3.weight"], sd[p + ".conv3.bias"])
    return x.view(B, -1, num_seed, num_depth)


def forward(end_points, sd, num_view=300, num_angle=12, keep_internal=False, f64=False, force_top_view_inds=None):
    """GraspNet.forward, is_training=False (graspnet.py:70-75).

    f64=True evaluates every conv / BN / interpolation in float64 on the SAME neighbour indices
    (index operators still see the float32 coordinates): the "exact arithmetic" yardstick that the
    parity tests measure both the fp32 oracle and the HIP path against."""
    with torch.no_grad():
        pc = end_points["point_clouds"]
        dt = torch.float64 if f64 else torch.float32
        if f64:
            sd = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        seed_features, seed_xyz, end_points = backbone(pc, sd, end_points, dt)
        with _stage("approach_net"):
            end_points = approach_net(seed_xyz, seed_features, sd, end_points, num_view, force_top_view_inds)
        vp, cyl_idx = cloud_crop(end_points["fp2_xyz"], end_points["input_xyz"],
                                 end_points["grasp_top_view_rot"], sd, return_idx=True, dt=dt)
        if keep_internal:
            end_points["_vp_features"] = vp
            end_points["_cylinder_idx"] = cyl_idx
        with _stage("heads"):
            op = head_net(vp, sd, "grasp_generator.operation")
            end_points["grasp_score_pred"] = op[:, 0:num_angle]
            end_points["grasp_angle_cls_pred"] = op[:, num_angle:2 * num_angle]
            end_points["grasp_width_pred"] = op[:, 2 * num_angle:3 * num_angle]
            end_points["grasp_tolerance_pred"] = head_net(vp, sd, "grasp_generator.tolerance")
    return end_points


def pred_decode(end_points):
    """graspnet.py:77-134 -> list of (Ns_i,17) float32 rows."""
    with _stage("decode"):
        return _pred_decode(end_points)


def _pred_decode(end_points):
    preds = []
    for i in range(len(end_points["point_clouds"])):
        objectness = end_points["objectness_score"][i].float()
        score = end_points["grasp_score_pred"][i].float()
        center = end_points["fp2_xyz"][i].float()
        approaching = -end_points["grasp_top_view_xyz"][i].float()
        angle_cls_score = end_points["grasp_angle_cls_pred"][i]
        width = torch.clamp(1.2 * end_points["grasp_width_pred"][i], min=0, max=GRASP_MAX_WIDTH)
        tol = end_points["grasp_tolerance_pred"][i]

        angle_cls = torch.argmax(angle_cls_score, 0)
        angle = angle_cls.float() / 12 * np.pi
        sel = angle_cls.unsqueeze(0)
        score = torch.gather(score, 0, sel).squeeze(0)
        width = torch.gather(width, 0, sel).squeeze(0)
        tol = torch.gather(tol, 0, sel).squeeze(0)

        depth_cls = torch.argmax(score, 1, keepdim=True)
        depth = (depth_cls.float() + 1) * 0.01
        score = torch.gather(score, 1, depth_cls)
        angle = torch.gather(angle, 1, depth_cls)
        width = torch.gather(width, 1, depth_cls)
        tol = torch.gather(tol, 1, depth_cls)

        mask = (torch.argmax(objectness, 0) == 1)
        score, width, depth = score[mask], width[mask], depth[mask]
        approaching, angle, center, tol = approaching[mask], angle[mask], center[mask], tol[mask]
        score = score * tol / GRASP_MAX_TOLERANCE

        ns = angle.size(0)
        rot = viewpoint_to_matrix(approaching.view(ns, 3), angle.view(ns)).view(ns, 9)
        height = 0.02 * torch.ones_like(score)
        obj_ids = -1 * torch.ones_like(score)
        preds.append(torch.cat([score, width, height, depth, rot, center, obj_ids], dim=-1))
    return preds
