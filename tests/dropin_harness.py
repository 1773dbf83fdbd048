"""Drop-in MODE, composed: the reference's UNFUSED python model (models/graspnet.py:70-75 over backbone.py:78-130,
pointnet2_modules.py:210-272 / 371-416, pointnet2_utils.py:313-372 / 488-555, modules.py:41-232) written out functionally for
DEVICE tensors on top of an operator module with the `pointnet2._ext` signatures -- i.e. what a maintainer runs after replacing
only `pointnet2/_ext` (pointnet2_utils.py:26) with graspnet_amd.pointnet2_ext.  HARNESS ONLY (tests + bench.py's
`dropin_unfused` leg): the product's model path is the fused one in graspnet_amd/graspnet.py.

The ~30 operator calls per forward are the reference's, in its order: 4 x FPS, 4 x gather_points, 4 x ball_query,
11 x group_points, 2 x three_nn, 2 x three_interpolate, 4 x cylinder_query.  The layers between them are
torch on the device; 1x1 convolutions are written as matmuls (same arithmetic as conv1d/conv2d with a 1x1 kernel; avoids
MIOpen's per-shape kernel search on a fresh box) and BN(eval) as its affine map.
"""
import numpy as np
import torch

BN_EPS = 1e-5
SA_CFG = ((2048, 0.04, 64), (1024, 0.1, 32), (512, 0.2, 16), (256, 0.3, 16))        # backbone.py:30-64


def to_device(sd, dev):
    return {k: v.to(dev) for k, v in sd.items()}


def _bn(x, sd, p):
    shape = [1, -1] + [1] * (x.dim() - 2)
    scale = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + BN_EPS)
    return (x - sd[p + ".running_mean"].view(shape)) * scale.view(shape) + sd[p + ".bias"].view(shape)


def _conv1x1(x, w, b=None):
    """x (B,C,...) , w (Cout,C[,1[,1]]) -> (B,Cout,...)"""
    w2 = w.reshape(w.shape[0], -1)
    y = torch.matmul(w2, x.flatten(2)).view(x.shape[0], w2.shape[0], *x.shape[2:])
    if b is not None:
        y = y + b.view([1, -1] + [1] * (x.dim() - 2))
    return y


def shared_mlp(x, sd, prefix, nlayers):                                              # pytorch_utils.py:11-36
    for i in range(nlayers):
        x = torch.relu(_bn(_conv1x1(x, sd["%s.layer%d.conv.weight" % (prefix, i)]), sd, "%s.layer%d.bn.bn" % (prefix, i)))
    return x


def sa_module(E, xyz, features, sd, prefix, npoint, radius, nsample):                # pointnet2_modules.py:210-272
    xyz_flipped = xyz.transpose(1, 2).contiguous()
    inds = E.furthest_point_sampling(xyz, npoint)
    new_xyz = E.gather_points(xyz_flipped, inds).transpose(1, 2).contiguous()
    idx = E.ball_query(new_xyz, xyz, radius, nsample)                                # pointnet2_utils.py:344
    grouped_xyz = E.group_points(xyz_flipped, idx)
    grouped_xyz = (grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)) / radius
    grouped = grouped_xyz if features is None else torch.cat([grouped_xyz, E.group_points(features.contiguous(), idx)], dim=1)
    f = shared_mlp(grouped, sd, prefix + ".mlp_module", 3)
    return new_xyz, f.max(dim=3).values, inds


def fp_module(E, unknown, known, unknow_feats, known_feats, sd, prefix):             # pointnet2_modules.py:371-416
    dist2, idx = E.three_nn(unknown, known)
    dist_recip = 1.0 / (torch.sqrt(dist2) + 1e-8)
    weight = dist_recip / dist_recip.sum(dim=2, keepdim=True)
    interp = E.three_interpolate(known_feats.contiguous(), idx, weight.contiguous())
    x = torch.cat([interp, unknow_feats], dim=1).unsqueeze(-1)
    return shared_mlp(x, sd, prefix + ".mlp", 2).squeeze(-1)


def grasp_views(n, dev):                                                             # loss_utils.py:41-66
    phi = (np.sqrt(5) - 1) / 2
    i = np.arange(n, dtype=np.float64)
    zi = (2 * i + 1) / n - 1
    xi = np.sqrt(1 - zi ** 2) * np.cos(2 * i * np.pi * phi)
    yi = np.sqrt(1 - zi ** 2) * np.sin(2 * i * np.pi * phi)
    return torch.from_numpy(np.stack([xi, yi, zi], axis=1).astype(np.float32)).to(dev)


def viewpoint_to_matrix(towards, angle):                                             # loss_utils.py:68-96
    ax = towards
    ones, zeros = torch.ones_like(angle), torch.zeros_like(angle)
    ay = torch.stack([-ax[:, 1], ax[:, 0], zeros], dim=-1)
    ay[torch.norm(ay, dim=-1) == 0, 1] = 1
    ax = ax / torch.norm(ax, dim=-1, keepdim=True)
    ay = ay / torch.norm(ay, dim=-1, keepdim=True)
    az = torch.linalg.cross(ax, ay, dim=-1)
    s, c = torch.sin(angle), torch.cos(angle)
    r1 = torch.stack([ones, zeros, zeros, zeros, c, -s, zeros, s, c], dim=-1).reshape(-1, 3, 3)
    return torch.matmul(torch.stack([ax, ay, az], dim=-1), r1)


def forward(E, point_clouds, sd, num_view=300, num_angle=12, hmax_list=(0.01, 0.02, 0.03, 0.04)):
    """-> end_points with the reference's keys (models/graspnet.py:70-75, is_training=False)"""
    ep = {"point_clouds": point_clouds}
    dev = point_clouds.device
    bb = "view_estimator.backbone"
    xyz = point_clouds[..., 0:3].contiguous()
    ep["input_xyz"] = xyz
    feats = None
    for li, (npoint, radius, nsample) in enumerate(SA_CFG):                          # backbone.py:105-121
        xyz, feats, inds = sa_module(E, xyz, feats, sd, "%s.sa%d" % (bb, li + 1), npoint, radius, nsample)
        if li < 2:
            ep["sa%d_inds" % (li + 1)] = inds
        ep["sa%d_xyz" % (li + 1)], ep["sa%d_features" % (li + 1)] = xyz, feats
    f = fp_module(E, ep["sa3_xyz"], ep["sa4_xyz"], ep["sa3_features"], ep["sa4_features"], sd, bb + ".fp1")
    f = fp_module(E, ep["sa2_xyz"], ep["sa3_xyz"], ep["sa2_features"], f, sd, bb + ".fp2")
    ep["fp2_features"], ep["fp2_xyz"] = f, ep["sa2_xyz"]
    M = ep["fp2_xyz"].shape[1]
    ep["fp2_inds"] = ep["sa1_inds"][:, 0:M]
    B = xyz.shape[0]
    # ApproachNet (modules.py:41-77)
    p = "view_estimator.vpmodule"
    v = torch.relu(_bn(_conv1x1(f, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"]), sd, p + ".bn1"))
    v = torch.relu(_bn(_conv1x1(v, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"]), sd, p + ".bn2"))
    v = _conv1x1(v, sd[p + ".conv3.weight"], sd[p + ".conv3.bias"])
    ep["objectness_score"] = v[:, :2, :]
    view_score = v[:, 2:2 + num_view, :].transpose(1, 2).contiguous()
    ep["view_score"] = view_score
    top_scores, top_inds = torch.max(view_score, dim=2)
    vp_xyz = grasp_views(num_view, dev)[top_inds.reshape(-1)].view(B, M, 3)
    vp_rot = viewpoint_to_matrix(-vp_xyz.view(-1, 3), torch.zeros(B * M, device=dev)).view(B, M, 3, 3)
    ep["grasp_top_view_inds"], ep["grasp_top_view_score"] = top_inds, top_scores
    ep["grasp_top_view_xyz"], ep["grasp_top_view_rot"] = vp_xyz, vp_rot
    # CloudCrop (modules.py:109-141; pointnet2_utils.py:488-555)
    seed_xyz, cloud = ep["fp2_xyz"], ep["input_xyz"]
    cloud_t = cloud.transpose(1, 2).contiguous()
    rot9 = vp_rot.reshape(B, M, 9).contiguous()
    grouped = []
    for hmax in hmax_list:
        idx = E.cylinder_query(seed_xyz, cloud, rot9, 0.05, -0.02, hmax, 64)
        g = E.group_points(cloud_t, idx) - seed_xyz.transpose(1, 2).unsqueeze(-1)
        g = torch.matmul(g.permute(0, 2, 3, 1), vp_rot)                               # (B,M,64,3) @ (B,M,3,3)
        grouped.append(g.permute(0, 3, 1, 2))
    x = torch.stack(grouped, dim=3).reshape(B, 3, M * len(hmax_list), 64)
    x = shared_mlp(x, sd, "grasp_generator.crop.mlps", 3).max(dim=3).values          # (B,256,M*D)
    D = len(hmax_list)

    def head(pfx):                                                                    # modules.py:169-191, 214-232
        h = torch.relu(_bn(_conv1x1(x, sd[pfx + ".conv1.weight"], sd[pfx + ".conv1.bias"]), sd, pfx + ".bn1"))
        h = torch.relu(_bn(_conv1x1(h, sd[pfx + ".conv2.weight"], sd[pfx + ".conv2.bias"]), sd, pfx + ".bn2"))
        return _conv1x1(h, sd[pfx + ".conv3.weight"], sd[pfx + ".conv3.bias"]).view(B, -1, M, D)

    op = head("grasp_generator.operation")
    ep["grasp_score_pred"] = op[:, 0:num_angle]
    ep["grasp_angle_cls_pred"] = op[:, num_angle:2 * num_angle]
    ep["grasp_width_pred"] = op[:, 2 * num_angle:3 * num_angle]
    ep["grasp_tolerance_pred"] = head("grasp_generator.tolerance")
    return ep
