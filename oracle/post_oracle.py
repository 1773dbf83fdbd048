"""numpy restatement of the host-side steps around the network (TEST INFRASTRUCTURE ONLY).

  depth_to_cloud      utils/data_utils.py:18-43
  collision_detect    utils/collision_detector.py:69-129 (line by line, same numpy dtypes)
  voxel_down_sample   open3d PointCloud.voxel_down_sample  -- UN-VENDORED third party (open3d >= 0.8, not
                      installed, not under /root/reference): restated from its documented algorithm
                      (voxel origin = min_bound - voxel/2; index = floor((p - origin)/voxel); output =
                      mean of the voxel's points, float64).  PARITY UNPINNED.
  nms_grasp           graspnetAPI GraspGroup.nms -> grasp_nms.nms_grasp -- UN-VENDORED (`graspnetAPI/` in the
                      reference is empty): restated as score-descending greedy suppression of grasps within
                      translation_thresh AND rotation_thresh (geodesic angle) of a kept grasp.  PARITY UNPINNED.
"""
import numpy as np


def depth_to_cloud(depth, fx, fy, cx, cy, scale):
    h, w = depth.shape
    xmap, ymap = np.meshgrid(np.arange(w), np.arange(h))
    points_z = depth / scale
    points_x = (xmap - cx) * points_z / fx
    points_y = (ymap - cy) * points_z / fy
    return np.stack([points_x, points_y, points_z], axis=-1).reshape(-1, 3)


def masked_cloud_sample(depth, mask, fx, fy, cx, cy, scale, idxs, depth_min=0.0, depth_max=np.inf):
    """dataset/graspnet_dataset.py:118-145 / demo.py:449-463: organised cloud from depth (utils/data_utils.py:18-43), keep
    `mask & (depth > 0)` (and depth_min < z < depth_max) in row-major order, take rows `idxs` of it (the caller's
    np.random.choice draw), float32 as the callers cast it.  -> (cloud_sampled (n,3) f32, number of valid pixels)"""
    cloud = depth_to_cloud(depth, fx, fy, cx, cy, scale)
    z = depth / scale
    m = (depth > 0) & (z > depth_min) & (z < depth_max)
    if mask is not None:
        m = m & (np.asarray(mask) != 0)
    cloud_masked = cloud[m.reshape(-1)]
    return cloud_masked[np.asarray(idxs)].astype(np.float32), int(len(cloud_masked))


def _hash32(x):
    """lowbias32 on uint32 arrays (csrc/frontend.hip fe_hash32)"""
    x = np.asarray(x, dtype=np.uint64) & 0xffffffff
    x ^= x >> 16
    x = (x * 0x7feb352d) & 0xffffffff
    x ^= x >> 15
    x = (x * 0x846ca68b) & 0xffffffff
    x ^= x >> 16
    return x


def draw_keys(seed, nimages):
    """(nimages, 4) uint32 round keys from a seed: key[b][r] = hash32(seed + 0x9e3779b9 * (4 b + r + 1)) -- what
    graspnet_amd.postprocess.draw_keys computes (host side, both)"""
    b = np.arange(nimages, dtype=np.uint64)[:, None]
    r = np.arange(4, dtype=np.uint64)[None, :]
    return _hash32((np.uint64(int(seed) & 0xffffffff) + 0x9e3779b9 * (4 * b + r + 1)) & 0xffffffff).astype(np.uint32)


def draw_positions(count, keys, n):
    """numpy restatement of csrc/frontend.hip fe_draw_kernel -- the device's stand-in for dataset/graspnet_dataset.py:139-145 /
    demo.py:458-463: count >= n: n DISTINCT positions (a keyed Feistel permutation of [0, count), cycle-walked), like
    np.random.choice(count, n, replace=False); count < n: arange(count) then draws with replacement, like the reference.
    count (B,), keys (B,4) uint32 -> (B,n) int32.  Integer arithmetic only: bit-identical to the kernel."""
    count = np.asarray(count).reshape(-1)
    keys = np.asarray(keys, dtype=np.uint32).reshape(-1, 4)
    out = np.zeros((len(count), n), np.int32)
    i = np.arange(n, dtype=np.uint64)
    for b, cnt in enumerate(int(c) for c in count):
        k = [np.uint64(v) for v in keys[b]]
        if cnt <= 0:
            continue
        if cnt < n:
            w = _hash32(((i * 0x9e3779b9) & 0xffffffff) ^ k[0])
            out[b] = np.where(i < cnt, i, (w * np.uint64(cnt)) >> 32).astype(np.int32)
            continue
        half = 1
        while (1 << (2 * half)) < cnt:
            half += 1
        hm = np.uint64((1 << half) - 1)
        x = i.copy()
        todo = np.ones(n, bool)
        for _ in range(1 << (2 * half)):
            l, r = x >> np.uint64(half), x & hm
            for kk in k:
                l, r = r, l ^ (_hash32(r ^ kk) & hm)
            y = (l << np.uint64(half)) | r
            x = np.where(todo, y, x)
            todo &= x >= cnt
            if not todo.any():
                break
        out[b] = x.astype(np.int32)
    return out


def voxel_down_sample(points, voxel_size):
    pts = np.asarray(points, dtype=np.float64)
    origin = pts.min(axis=0) - voxel_size * 0.5
    vi = np.floor((pts - origin) / voxel_size).astype(np.int64)
    _, inv, cnt = np.unique(vi, axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    out = np.zeros((len(cnt), 3))
    np.add.at(out, inv, pts)
    return out / cnt[:, None]


def voxel_down_sample_ordered(points, voxel_size, n_valid=None):
    """The same voxel means with the output ORDER pinned (open3d leaves it unspecified): voxels in order of first appearance
    (ascending lowest member index), each sum taken sequentially in ascending member index -- np.add.at is unbuffered and
    walks the points in index order -- so a deterministic device implementation can be compared bit for bit.
    Points past n_valid and non-finite points belong to no voxel."""
    pts = np.asarray(points, dtype=np.float64)
    if n_valid is not None:
        pts = pts[:int(n_valid)]
    pts = pts[np.isfinite(pts).all(axis=1)]
    if len(pts) == 0:
        return np.zeros((0, 3))
    origin = pts.min(axis=0) - voxel_size * 0.5
    vi = np.floor((pts - origin) / voxel_size).astype(np.int64)
    _, first, inv, cnt = np.unique(vi, axis=0, return_index=True, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    order = np.argsort(first, kind="stable")            # sorted-voxel id of the k-th voxel to appear
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    out = np.zeros((len(cnt), 3))
    np.add.at(out, rank[inv], pts)
    return out / cnt[order][:, None]


def collision_detect(scene_points, grasps, approach_dist=0.03, collision_thresh=0.05, empty_thresh=0.01, voxel_size=0.005,
                     finger_width=0.01, finger_length=0.06):
    """scene_points (Np,3) float64, grasps (M,17) float32 -> collision mask, empty mask, [5 ious]."""
    approach_dist = max(approach_dist, finger_width)
    T = grasps[:, 13:16]
    R = grasps[:, 4:13].reshape(-1, 3, 3)
    heights = grasps[:, 2][:, np.newaxis]
    depths = grasps[:, 3][:, np.newaxis]
    widths = grasps[:, 1][:, np.newaxis]
    targets = scene_points[np.newaxis, :, :] - T[:, np.newaxis, :]
    targets = np.matmul(targets, R)
    mask1 = ((targets[:, :, 2] > -heights / 2) & (targets[:, :, 2] < heights / 2))
    mask2 = ((targets[:, :, 0] > depths - finger_length) & (targets[:, :, 0] < depths))
    mask3 = (targets[:, :, 1] > -(widths / 2 + finger_width))
    mask4 = (targets[:, :, 1] < -widths / 2)
    mask5 = (targets[:, :, 1] < (widths / 2 + finger_width))
    mask6 = (targets[:, :, 1] > widths / 2)
    mask7 = ((targets[:, :, 0] <= depths - finger_length) & (targets[:, :, 0] > depths - finger_length - finger_width))
    mask8 = ((targets[:, :, 0] <= depths - finger_length - finger_width)
             & (targets[:, :, 0] > depths - finger_length - finger_width - approach_dist))
    left_mask = (mask1 & mask2 & mask3 & mask4)
    right_mask = (mask1 & mask2 & mask5 & mask6)
    bottom_mask = (mask1 & mask3 & mask5 & mask7)
    shifting_mask = (mask1 & mask3 & mask5 & mask8)
    global_mask = (left_mask | right_mask | bottom_mask | shifting_mask)
    left_right_volume = (heights * finger_length * finger_width / (voxel_size ** 3)).reshape(-1)
    bottom_volume = (heights * (widths + 2 * finger_width) * finger_width / (voxel_size ** 3)).reshape(-1)
    shifting_volume = (heights * (widths + 2 * finger_width) * approach_dist / (voxel_size ** 3)).reshape(-1)
    volume = left_right_volume * 2 + bottom_volume + shifting_volume
    global_iou = global_mask.sum(axis=1) / (volume + 1e-6)
    collision_mask = (global_iou > collision_thresh)
    inner_mask = (mask1 & mask2 & (~mask4) & (~mask6))
    inner_volume = (heights * finger_length * widths / (voxel_size ** 3)).reshape(-1)
    empty_mask = (inner_mask.sum(axis=-1) / inner_volume < empty_thresh)
    ious = [global_iou, left_mask.sum(axis=1) / (left_right_volume + 1e-6), right_mask.sum(axis=1) / (left_right_volume + 1e-6),
            bottom_mask.sum(axis=1) / (bottom_volume + 1e-6), shifting_mask.sum(axis=1) / (shifting_volume + 1e-6)]
    return collision_mask, empty_mask, ious


def sort_by_score(grasps):
    order = np.argsort(-grasps[:, 0].astype(np.float64), kind="stable")
    return grasps[order]


def nms_grasp(grasps, translation_thresh=0.03, rotation_thresh=30.0 / 180.0 * np.pi):
    g = sort_by_score(grasps)
    keep = []
    cos_rt = np.cos(rotation_thresh)
    tt2 = translation_thresh * translation_thresh
    for i in range(len(g)):
        ok = True
        for j in keep:
            d = g[i, 13:16].astype(np.float64) - g[j, 13:16].astype(np.float64)
            if float(d @ d) > tt2:
                continue
            tr = float((g[i, 4:13].astype(np.float64) * g[j, 4:13].astype(np.float64)).sum())
            cv = min(1.0, max(-1.0, (tr - 1.0) * 0.5))
            if cv >= cos_rt:
                ok = False
                break
        if ok:
            keep.append(i)
    return g[keep]
