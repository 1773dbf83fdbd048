"""View template and view/angle -> rotation tables, built once on the host.

Restates utils/loss_utils.py:41-66 (generate_grasp_views: Fibonacci lattice in
float64, cast to float32) and :68-96 (batch_viewpoint_params_to_matrix) with
torch CPU float32 ops in the same order.  The reference evaluates both on
every forward / decode (models/modules.py:66,71; models/graspnet.py:127); since
a view index takes 300 values and an angle class 12, the device path looks the
results up in tables instead.
"""
import numpy as np
import torch


def grasp_views(n=300):
    phi = (np.sqrt(5) - 1) / 2
    i = np.arange(n, dtype=np.float64)
    zi = (2 * i + 1) / n - 1
    xi = np.sqrt(1 - zi ** 2) * np.cos(2 * i * np.pi * phi)
    yi = np.sqrt(1 - zi ** 2) * np.sin(2 * i * np.pi * phi)
    return torch.from_numpy(np.stack([xi, yi, zi], axis=1).astype(np.float32))


def viewpoint_to_matrix(towards, angle):
    """(N,3) approach vectors, (N,) in-plane angles -> (N,3,3);  R = [x y z] . Rx(angle)."""
    axis_x = towards
    n = axis_x.shape[0]
    ones = torch.ones(n, dtype=axis_x.dtype)
    zeros = torch.zeros(n, dtype=axis_x.dtype)
    axis_y = torch.stack([-axis_x[:, 1], axis_x[:, 0], zeros], dim=-1)
    degenerate = (torch.norm(axis_y, dim=-1) == 0)
    axis_y[degenerate, 1] = 1
    axis_x = axis_x / torch.norm(axis_x, dim=-1, keepdim=True)
    axis_y = axis_y / torch.norm(axis_y, dim=-1, keepdim=True)
    axis_z = torch.linalg.cross(axis_x, axis_y, dim=-1)
    sin, cos = torch.sin(angle), torch.cos(angle)
    r1 = torch.stack([ones, zeros, zeros, zeros, cos, -sin, zeros, sin, cos], dim=-1).reshape(-1, 3, 3)
    r2 = torch.stack([axis_x, axis_y, axis_z], dim=-1)
    return torch.matmul(r2, r1)


def build_tables(num_view=300, num_angle=12):
    """views (V,3); view_rot (V,9) = matrix(-view, 0) [modules.py:70-71];
    decode_rot (V,A,9) = matrix(-view, a/12*pi) [graspnet.py:85,94,127]."""
    views = grasp_views(num_view)
    view_rot = viewpoint_to_matrix(-views, torch.zeros(num_view)).reshape(num_view, 9)
    cls = torch.arange(num_angle, dtype=torch.int64)
    ang = cls.float() / 12 * np.pi                       # graspnet.py:94 (12 is literal there)
    tw = (-views).unsqueeze(1).expand(-1, num_angle, -1).reshape(-1, 3).contiguous()
    aa = ang.unsqueeze(0).expand(num_view, -1).reshape(-1).contiguous()
    decode_rot = viewpoint_to_matrix(tw, aa).reshape(num_view, num_angle, 9)
    return views.contiguous(), view_rot.contiguous(), decode_rot.contiguous()
