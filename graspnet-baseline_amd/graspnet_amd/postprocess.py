"""Device-side front end and post-processing around the network.

  depth_to_cloud                 utils/data_utils.py:18-43 (+ picking the sampled pixels)
  valid_pixels / depth_to_cloud_masked   the whole host front end of dataset/graspnet_dataset.py:104-147 / demo.py:449-463
                                 (workspace mask & depth > 0, ordered compaction, sub-sampling) for a batch of images
  ModelFreeCollisionDetector     utils/collision_detector.py:10-129 (same constructor / detect() arguments)
  sort_and_nms                   graspnetAPI GraspGroup.sort_by_score() / .nms()  [un-vendored, parity unpinned]

Everything works on the padded (B,M,17) grasp rows + counts that GraspNet.decode_padded() leaves on
the device, so the reference's device->host copy before collision checking (test.py:88-96) disappears.
"""
import numpy as np
import torch

from . import _lib as L


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def depth_to_cloud(depth, fx, fy, cx, cy, factor_depth, pix_idx):
    """depth (H,W) uint16 CUDA tensor (or int16 view of it), pix_idx (N,) int32 flat pixel indices -> (N,3) f32."""
    if not depth.is_cuda:
        raise RuntimeError("CPU not supported")
    if depth.dtype not in (torch.uint16, torch.int16) or not depth.is_contiguous():
        raise RuntimeError("depth must be a contiguous 16-bit tensor")
    pix_idx = pix_idx.to(device=depth.device, dtype=torch.int32).contiguous()
    h, w = depth.shape
    out = torch.empty((pix_idx.numel(), 3), dtype=torch.float32, device=depth.device)
    with torch.cuda.device(depth.device):
        L.check(L.lib().gn_depth_to_cloud(depth.data_ptr(), h, w, float(fx), float(fy), float(cx), float(cy),
                                          float(factor_depth), pix_idx.data_ptr(), pix_idx.numel(), out.data_ptr(),
                                          _stream(depth)), "depth_to_cloud")
    return out


def _as_u16(depth):
    if depth.dtype not in (torch.uint16, torch.int16) or not depth.is_contiguous():
        raise RuntimeError("depth must be a contiguous 16-bit tensor")
    return depth


def valid_pixels(depth, mask, factor_depth, depth_min=0.0, depth_max=float("inf")):
    """depth (B,H,W) 16-bit CUDA, mask (B,H,W) uint8 / bool or None -> (valid_pix (B,H*W) i32, count (B,) i32) on the device:
    the flat indices of the pixels with mask != 0, depth > 0 and depth_min < depth / factor_depth < depth_max in row-major
    order -- numpy's cloud[mask] order (dataset/graspnet_dataset.py:118-133, demo.py:453-455) -- and how many there are."""
    if not depth.is_cuda:
        raise RuntimeError("CPU not supported")
    depth = _as_u16(depth)
    if depth.dim() == 2:
        depth = depth.unsqueeze(0)
        mask = None if mask is None else mask.unsqueeze(0)
    B, h, w = depth.shape
    dev = depth.device
    if mask is not None:
        if mask.shape != depth.shape:
            raise RuntimeError("mask must have the depth image's shape")
        mask = (mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)).to(dev).contiguous()
    lib = L.lib()
    ws = torch.empty((lib.gn_frontend_workspace_bytes(B, h, w),), dtype=torch.uint8, device=dev)
    vpix = torch.empty((B, h * w), dtype=torch.int32, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        L.check(lib.gn_depth_valid_compact(depth.data_ptr(), L.ptr(mask), B, h, w, float(factor_depth), float(depth_min),
                                           float(depth_max), ws.data_ptr(), vpix.data_ptr(), cnt.data_ptr(), _stream(depth)),
                "depth_valid_compact")
    return vpix, cnt


def depth_to_cloud_masked(depth, mask, fx, fy, cx, cy, factor_depth, num_point, positions=None, random_words=None, seed=None,
                          depth_min=0.0, depth_max=float("inf"), return_pix=False):
    """The reference's host-side front end (dataset/graspnet_dataset.py:104-147, demo.py:449-463) for a batch of images on the
    device: cloud from depth, keep workspace-mask & depth > 0 pixels, take num_point of them.
      depth (B,H,W) or (H,W) 16-bit CUDA tensor, mask the same shape (uint8 / bool) or None
      positions (B,num_point) int: the caller's own draw into the valid list -- with the reference's `idxs` the cloud is the
          reference's bit for bit; or random_words (B,num_point) 32-bit: all valid pixels first when there are fewer than
          num_point (as the reference does), then floor(word * count / 2^32), i.e. with replacement; or seed: the words come
          from torch's device generator seeded with it.  Nothing is read back.
    -> cloud (B,num_point,3) f32, count (B,) i32 [, pix (B,num_point) i32]"""
    if not depth.is_cuda:
        raise RuntimeError("CPU not supported")
    single = depth.dim() == 2
    if single:
        depth = depth.unsqueeze(0)
        mask = None if mask is None else mask.unsqueeze(0)
        positions = None if positions is None else positions.reshape(1, -1)
        random_words = None if random_words is None else random_words.reshape(1, -1)
    depth = _as_u16(depth)
    B, h, w = depth.shape
    dev = depth.device
    n = int(num_point)
    vpix, cnt = valid_pixels(depth, mask, factor_depth, depth_min, depth_max)
    if positions is not None:
        positions = positions.to(device=dev, dtype=torch.int32).contiguous()
        if positions.shape != (B, n):
            raise RuntimeError("positions must be (B, num_point)")
    elif random_words is None:
        gen = torch.Generator(device=dev)
        gen.manual_seed(0 if seed is None else int(seed))
        r64 = torch.randint(0, 2 ** 32, (B, n), dtype=torch.int64, device=dev, generator=gen)
        random_words = torch.where(r64 >= 2 ** 31, r64 - 2 ** 32, r64).to(torch.int32)       # the same 32 bits, as int32
    if random_words is not None:
        random_words = random_words.to(device=dev).contiguous()
        if random_words.dtype not in (torch.int32, torch.uint32) or random_words.shape != (B, n):
            raise RuntimeError("random_words must be (B, num_point) 32-bit integers")
    out = torch.empty((B, n, 3), dtype=torch.float32, device=dev)
    pix = torch.empty((B, n), dtype=torch.int32, device=dev) if return_pix else None
    with torch.cuda.device(dev):
        L.check(L.lib().gn_depth_sample_cloud(depth.data_ptr(), B, h, w, float(fx), float(fy), float(cx), float(cy),
                                              float(factor_depth), vpix.data_ptr(), cnt.data_ptr(), L.ptr(positions),
                                              L.ptr(random_words), n, out.data_ptr(), L.ptr(pix), _stream(depth)),
                "depth_sample_cloud")
    if single:
        out, pix = out[0], (None if pix is None else pix[0])
    return (out, cnt, pix) if return_pix else (out, cnt)


def voxel_down_sample_batched(points, voxel_size, n_valid=None, check=False):
    """(B,N,3) f32 CUDA clouds -> (means (B,N,3) f64 padded with zero rows, counts (B,) i32, status (B,) i32), all on the
    device, no host read (hipGraph-capturable).  Voxels come out in order of first appearance, every mean adds its members
    in ascending index order: run-to-run bit-identical.  n_valid (B,) i32: points past it are padding.
    check=True reads `status` back (one sync) and raises on a grid that exceeds 2^21 cells per axis."""
    if not points.is_cuda:
        raise RuntimeError("CPU not supported")
    if points.dim() != 3 or points.size(-1) != 3:
        raise RuntimeError("points must be (B,N,3)")
    points = points.to(torch.float32).contiguous()
    B, n = points.shape[0], points.shape[1]
    dev = points.device
    lib = L.lib()
    ws = torch.empty((max(lib.gn_voxel_batched_workspace_bytes(B, n), 256),), dtype=torch.uint8, device=dev)
    out = torch.empty((B, n, 3), dtype=torch.float64, device=dev)
    cnt = torch.empty((B,), dtype=torch.int32, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    nv = None if n_valid is None else n_valid.to(device=dev, dtype=torch.int32).contiguous()
    with torch.cuda.device(dev):
        L.check(lib.gn_voxel_down_sample_batched(points.data_ptr(), L.ptr(nv), B, n, float(voxel_size), ws.data_ptr(),
                                                 out.data_ptr(), cnt.data_ptr(), status.data_ptr(), _stream(points)),
                "voxel_down_sample")
    if check:
        e = int(status.max().item()) if B else 0
        if e:
            raise RuntimeError("voxel_down_sample: voxel grid exceeds 2^21 cells per axis or table overflow (code %d)" % e)
    return out, cnt, status


def voxel_down_sample(points, voxel_size):
    """(N,3) f32 CUDA -> (Nv,3) f64 CUDA voxel means in order of first appearance (one host sync for Nv, as open3d's
    call is synchronous too)."""
    if not points.is_cuda:
        raise RuntimeError("CPU not supported")
    out, cnt, status = voxel_down_sample_batched(points.reshape(1, -1, 3), voxel_size)
    nv, e = cnt.cpu().tolist()[0], status.cpu().tolist()[0]
    if e:
        raise RuntimeError("voxel_down_sample: voxel grid exceeds 2^21 cells per axis or table overflow (code %d)" % e)
    return out[0, :nv]


def _detect_outputs(B, M, dev, return_empty_grasp, return_ious):
    cm = torch.empty((B, M), dtype=torch.uint8, device=dev)
    em = torch.empty((B, M), dtype=torch.uint8, device=dev) if return_empty_grasp else None
    io = torch.empty((B, M, 5), dtype=torch.float32, device=dev) if return_ious else None
    return cm, em, io


def _detect_result(cm, em, io):
    ret = [cm.view(torch.bool)]
    if em is not None:
        ret.append(em.view(torch.bool))
    if io is not None:
        ret.append(io)
    return ret[0] if len(ret) == 1 else ret


class BatchedCollisionDetector:
    """test.py:88-96 builds one ModelFreeCollisionDetector per scene of the batch and copies the grasps to the host first;
    this is the same arithmetic for the whole batch in two sequences of launches with nothing read back: every scene's cloud
    is voxel-down-sampled at construction (padded (B,N,3) f64 + per-scene counts on the device), detect_padded() checks the
    padded grasp rows of scene b against scene b's points."""

    def __init__(self, clouds, voxel_size=0.005, n_valid=None):
        self.finger_width = 0.01          # collision_detector.py:29-30
        self.finger_length = 0.06
        self.voxel_size = voxel_size
        self.scene_points, self.scene_counts, self.status = voxel_down_sample_batched(clouds, voxel_size, n_valid)

    def detect_padded(self, grasps, counts=None, approach_dist=0.03, collision_thresh=0.05, return_empty_grasp=False,
                      empty_thresh=0.01, return_ious=False):
        if not grasps.is_cuda:
            raise RuntimeError("CPU not supported")
        grasps = grasps.contiguous()
        B, M = grasps.shape[0], grasps.shape[1]
        if B != self.scene_points.shape[0]:
            raise RuntimeError("grasps hold %d scenes, the detector %d" % (B, self.scene_points.shape[0]))
        dev = grasps.device
        cm, em, io = _detect_outputs(B, M, dev, return_empty_grasp, return_ious)
        with torch.cuda.device(dev):
            L.check(L.lib().gn_collision_detect_batched(self.scene_points.data_ptr(), self.scene_points.shape[1],
                                                        self.scene_counts.data_ptr(), grasps.data_ptr(), L.ptr(counts), B, M,
                                                        float(approach_dist), float(collision_thresh), float(empty_thresh),
                                                        float(self.voxel_size), self.finger_width, self.finger_length,
                                                        cm.data_ptr(), L.ptr(em), L.ptr(io), _stream(grasps)),
                    "collision_detect")
        return _detect_result(cm, em, io)


class ModelFreeCollisionDetector:
    """Same interface as utils/collision_detector.py:10-129; scene and grasps stay on the GPU."""

    def __init__(self, scene_points, voxel_size=0.005, device="cuda"):
        self.finger_width = 0.01          # collision_detector.py:29-30
        self.finger_length = 0.06
        self.voxel_size = voxel_size
        if isinstance(scene_points, np.ndarray):
            scene_points = torch.from_numpy(np.ascontiguousarray(scene_points, dtype=np.float32)).to(device)
        self.scene_points = voxel_down_sample(scene_points, voxel_size).contiguous()

    def detect_padded(self, grasps, counts=None, approach_dist=0.03, collision_thresh=0.05, return_empty_grasp=False,
                      empty_thresh=0.01, return_ious=False):
        """grasps (B,M,17) f32 CUDA, counts (B,) i32 or None -> collision mask (B,M) bool [, empty (B,M) bool][, ious (B,M,5)]
        every scene of `grasps` is checked against THIS detector's one scene (the reference's object holds one scene)"""
        if not grasps.is_cuda:
            raise RuntimeError("CPU not supported")
        grasps = grasps.contiguous()
        B, M = grasps.shape[0], grasps.shape[1]
        dev = grasps.device
        cm, em, io = _detect_outputs(B, M, dev, return_empty_grasp, return_ious)
        with torch.cuda.device(dev):
            L.check(L.lib().gn_collision_detect(self.scene_points.data_ptr(), None, self.scene_points.shape[0],
                                                grasps.data_ptr(), L.ptr(counts), B, M, float(approach_dist),
                                                float(collision_thresh), float(empty_thresh), float(self.voxel_size),
                                                self.finger_width, self.finger_length, cm.data_ptr(), L.ptr(em), L.ptr(io),
                                                _stream(grasps)), "collision_detect")
        return _detect_result(cm, em, io)

    def detect(self, grasp_group, approach_dist=0.03, collision_thresh=0.05, return_empty_grasp=False, empty_thresh=0.01,
               return_ious=False):
        """grasp_group: (M,17) tensor / ndarray, or an object with `.grasp_group_array` (graspnetAPI.GraspGroup)."""
        arr = getattr(grasp_group, "grasp_group_array", grasp_group)
        was_numpy = isinstance(arr, np.ndarray)
        g = torch.as_tensor(arr, dtype=torch.float32).to(self.scene_points.device).reshape(1, -1, 17)
        res = self.detect_padded(g, None, approach_dist, collision_thresh, return_empty_grasp, empty_thresh, return_ious)
        res = res if isinstance(res, list) else [res]
        res = [r[0] for r in res]
        if return_ious:      # reference returns a list of 5 (M,) arrays
            res[-1] = [res[-1][:, i] for i in range(5)]
        if was_numpy:
            res = [([x.cpu().numpy() for x in r] if isinstance(r, list) else r.cpu().numpy()) for r in res]
        return res[0] if len(res) == 1 else res


def sort_and_nms(grasps, counts=None, drop_mask=None, translation_thresh=0.03, rotation_thresh=30.0 / 180.0 * np.pi,
                 nms=True):
    """(B,M,17) padded rows -> rows sorted by score (descending), optionally NMS-suppressed, padded + counts."""
    if not grasps.is_cuda:
        raise RuntimeError("CPU not supported")
    grasps = grasps.contiguous()
    B, M = grasps.shape[0], grasps.shape[1]
    dev = grasps.device
    lib = L.lib()
    out = torch.empty_like(grasps)
    oc = torch.empty((B,), dtype=torch.int32, device=dev)
    ws = torch.empty((lib.gn_grasp_nms_workspace_bytes(B) if nms else 8,), dtype=torch.uint8, device=dev)
    dm = None if drop_mask is None else (drop_mask.view(torch.uint8) if drop_mask.dtype == torch.bool else drop_mask.to(torch.uint8)).contiguous()
    with torch.cuda.device(dev):
        L.check(lib.gn_grasp_nms(grasps.data_ptr(), L.ptr(counts), L.ptr(dm), B, M, float(translation_thresh),
                                 float(rotation_thresh), 1 if nms else 0, ws.data_ptr(), out.data_ptr(), oc.data_ptr(),
                                 _stream(grasps)), "grasp_nms")
    return out, oc


def grasp_pipeline(clouds, grasps, counts, voxel_size=0.01, approach_dist=0.05, collision_thresh=0.01,
                   translation_thresh=0.03, rotation_thresh=30.0 / 180.0 * np.pi, nms=True, n_valid=None):
    """What test.py:88-96 / demo_pointcloud.py:228-249 run on the host per scene after pred_decode -- voxel down-sample of the
    scene, ModelFreeCollisionDetector.detect, drop the collisions, sort by score (+ NMS) -- for the whole batch on the device:
    ~12 launches, no host read, no allocation besides torch's caching allocator, hipGraph-capturable.
    clouds (B,N,3) f32, grasps (B,M,17) + counts (B,) as GraspNet.decode_padded() leaves them ->
    (rows (B,M,17) sorted by score with zero padding, kept (B,) i32, collision mask (B,M) bool)."""
    if not (clouds.is_cuda and grasps.is_cuda):
        raise RuntimeError("CPU not supported")
    clouds = clouds.to(torch.float32).contiguous()
    grasps = grasps.contiguous()
    B, n = clouds.shape[0], clouds.shape[1]
    M = grasps.shape[1]
    dev = grasps.device
    lib = L.lib()
    ws = torch.empty((max(lib.gn_grasp_postprocess_workspace_bytes(B, n, M), 256),), dtype=torch.uint8, device=dev)
    out = torch.empty_like(grasps)
    oc = torch.empty((B,), dtype=torch.int32, device=dev)
    cm = torch.empty((B, M), dtype=torch.uint8, device=dev)
    nv = None if n_valid is None else n_valid.to(device=dev, dtype=torch.int32).contiguous()
    with torch.cuda.device(dev):      # ONE C call: voxel down-sample -> collision detect -> sort (+ NMS)
        L.check(lib.gn_grasp_postprocess(clouds.data_ptr(), L.ptr(nv), B, n, grasps.data_ptr(), L.ptr(counts), M, float(voxel_size),
                                         float(approach_dist), float(collision_thresh), float(translation_thresh),
                                         float(rotation_thresh), 1 if nms else 0, ws.data_ptr(), out.data_ptr(), oc.data_ptr(),
                                         cm.data_ptr(), None, _stream(grasps)), "grasp_postprocess")
    return out, oc, cm.view(torch.bool)
