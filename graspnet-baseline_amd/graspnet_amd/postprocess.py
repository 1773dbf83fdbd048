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


def _hash32(x):
    x &= 0xffffffff
    x ^= x >> 16
    x = (x * 0x7feb352d) & 0xffffffff
    x ^= x >> 15
    x = (x * 0x846ca68b) & 0xffffffff
    x ^= x >> 16
    return x


def draw_keys(seed, nimages, device=None):
    """(nimages, 4) round keys of the device-side draw from one seed (host arithmetic, 16 bytes per image):
    key[b][r] = hash32(seed + 0x9e3779b9 * (4 b + r + 1)); int32 tensor holding the 32-bit words."""
    seed = int(seed) & 0xffffffff
    words = [[_hash32(seed + 0x9e3779b9 * (4 * b + r + 1)) for r in range(4)] for b in range(int(nimages))]
    t = torch.tensor([[w - (1 << 32) if w >= (1 << 31) else w for w in row] for row in words], dtype=torch.int32).reshape(-1, 4)
    return t if device is None else t.to(device)


def draw_positions(count, num_point, keys):
    """The reference's draw (dataset/graspnet_dataset.py:139-145, demo.py:458-463) on the device: count (B,) i32 valid pixels
    per image, keys (B,4) 32-bit words (draw_keys) -> positions (B,num_point) i32 into the valid list -- num_point DISTINCT
    positions when count >= num_point (np.random.choice(..., replace=False)), else all of them followed by draws with
    replacement.  No host read; hipGraph-capturable."""
    if not count.is_cuda:
        raise RuntimeError("CPU not supported")
    B = count.numel()
    keys = keys.to(device=count.device).contiguous()
    if keys.dtype not in (torch.int32, torch.uint32) or keys.numel() != 4 * B:
        raise RuntimeError("keys must be (B, 4) 32-bit words")
    pos = torch.empty((B, int(num_point)), dtype=torch.int32, device=count.device)
    with torch.cuda.device(count.device):
        L.check(L.lib().gn_depth_draw_positions(count.data_ptr(), keys.data_ptr(), B, int(num_point), pos.data_ptr(), _stream(count)),
                "depth_draw_positions")
    return pos


def masked_cloud(depth, mask, fx, fy, cx, cy, factor_depth, max_points=None, depth_min=0.0, depth_max=float("inf"),
                 valid=None):
    """cloud_masked of the reference (`cloud[mask]`: dataset/graspnet_dataset.py:133, demo.py:455) for a batch of images, on the
    device: -> (cloud (B,max_points,3) f32 with zero rows behind each image's points, n_valid (B,) i32).  This is the scene the
    reference's callers give to the collision detector (test.py:92-96 return_raw_cloud=True; demo.py:481), several times
    larger than the num_point points the network sees.  max_points: rows kept per image (default H*W; an image with more
    valid pixels is cut there).  valid: (valid_pix, count) from valid_pixels() to share the compaction with the sampler."""
    if not depth.is_cuda:
        raise RuntimeError("CPU not supported")
    if depth.dim() == 2:
        depth = depth.unsqueeze(0)
        mask = None if mask is None else mask.unsqueeze(0)
    depth = _as_u16(depth)
    B, h, w = depth.shape
    vpix, cnt = valid if valid is not None else valid_pixels(depth, mask, factor_depth, depth_min, depth_max)
    nmax = h * w if max_points is None else min(int(max_points), h * w)
    out = torch.empty((B, nmax, 3), dtype=torch.float32, device=depth.device)
    nv = torch.empty((B,), dtype=torch.int32, device=depth.device)
    with torch.cuda.device(depth.device):
        L.check(L.lib().gn_depth_masked_cloud(depth.data_ptr(), B, h, w, float(fx), float(fy), float(cx), float(cy),
                                              float(factor_depth), vpix.data_ptr(), cnt.data_ptr(), nmax, out.data_ptr(),
                                              nv.data_ptr(), _stream(depth)), "depth_masked_cloud")
    return out, nv


def depth_to_cloud_masked(depth, mask, fx, fy, cx, cy, factor_depth, num_point, positions=None, random_words=None, seed=None,
                          depth_min=0.0, depth_max=float("inf"), return_pix=False, keys=None, valid=None):
    """The reference's host-side front end (dataset/graspnet_dataset.py:104-147, demo.py:449-463) for a batch of images on the
    device: cloud from depth, keep workspace-mask & depth > 0 pixels, take num_point of them.
      depth (B,H,W) or (H,W) 16-bit CUDA tensor, mask the same shape (uint8 / bool) or None
      positions (B,num_point) int: the caller's own draw into the valid list -- with the reference's `idxs` the cloud is the
          reference's bit for bit; or seed / keys (B,4): the device's own draw (draw_positions) -- WITHOUT replacement when
          the image has at least num_point valid pixels, like the reference's np.random.choice(..., replace=False), all valid
          pixels followed by draws with replacement when it has fewer; or random_words (B,num_point) 32-bit: all valid
          pixels first when there are fewer than num_point, then floor(word * count / 2^32) -- WITH replacement whatever the
          count (1-4 % duplicated points at 20 000 of a few hundred thousand: NOT what the reference feeds its sampler; kept
          for callers that bring their own random words).  Default (nothing given): seed 0.  Nothing is read back.
      valid: (valid_pix, count) from valid_pixels(), to share the compaction with masked_cloud().
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
    vpix, cnt = valid if valid is not None else valid_pixels(depth, mask, factor_depth, depth_min, depth_max)
    if positions is not None:
        positions = positions.to(device=dev, dtype=torch.int32).contiguous()
        if positions.shape != (B, n):
            raise RuntimeError("positions must be (B, num_point)")
    elif random_words is None:
        if keys is None:
            keys = draw_keys(0 if seed is None else seed, B, dev)
        positions = draw_positions(cnt, n, keys)
    if random_words is not None and positions is None:
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
                   translation_thresh=0.03, rotation_thresh=30.0 / 180.0 * np.pi, nms=True, n_valid=None, return_status=False,
                   check=False):
    """What test.py:88-96 / demo_pointcloud.py:228-249 run on the host per scene after pred_decode -- voxel down-sample of the
    scene, ModelFreeCollisionDetector.detect, drop the collisions, sort by score (+ NMS) -- for the whole batch on the device:
    ~12 launches, no host read, no allocation besides torch's caching allocator, hipGraph-capturable.
    clouds (B,N,3) f32: the SCENE the grasps are checked against -- the reference uses the raw masked cloud (test.py:92
    return_raw_cloud=True, demo.py:481 cloud_masked; masked_cloud() builds it on the device), not the num_point points sampled
    for the network; n_valid (B,) i32: rows of `clouds` past it are padding.  grasps (B,M,17) + counts (B,) as
    GraspNet.decode_padded() leaves them ->
    (rows (B,M,17) sorted by score with zero padding, kept (B,) i32, collision mask (B,M) bool [, voxel status (B,) i32]).
    return_status: also return the voxel step's per-scene status, on the device (0 ok; 1 = the scene spans more than 2^21
    voxels on an axis -- e.g. one far outlier: its points were left out and every grasp comes back collision-free; 2 = table
    overflow); check=True reads it back (one sync) and raises instead."""
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
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    nv = None if n_valid is None else n_valid.to(device=dev, dtype=torch.int32).contiguous()
    with torch.cuda.device(dev):      # ONE C call: voxel down-sample -> collision detect -> sort (+ NMS)
        L.check(lib.gn_grasp_postprocess(clouds.data_ptr(), L.ptr(nv), B, n, grasps.data_ptr(), L.ptr(counts), M, float(voxel_size),
                                         float(approach_dist), float(collision_thresh), float(translation_thresh),
                                         float(rotation_thresh), 1 if nms else 0, ws.data_ptr(), out.data_ptr(), oc.data_ptr(),
                                         cm.data_ptr(), status.data_ptr(), _stream(grasps)), "grasp_postprocess")
    if check:
        e = int(status.max().item()) if B else 0
        if e:
            raise RuntimeError("grasp_pipeline: a scene's voxel grid exceeds 2^21 cells per axis or its table overflowed (code %d): "
                               "its points were left out of the collision check" % e)
    if return_status:
        return out, oc, cm.view(torch.bool), status
    return out, oc, cm.view(torch.bool)


def image_to_grasps(net, depth, mask, intrinsics, num_point=20000, keys=None, seed=None, positions=None, post=None,
                    scene_points=None, depth_min=0.0, depth_max=float("inf")):
    """A caller's whole loop body (demo.py:449-499, test.py:82-103) for a batch of depth images, entirely on the device:
        depth (B,H,W) u16 + workspace mask  ->  valid pixels (mask & depth > 0)  ->  num_point of them drawn (without
        replacement)  ->  cloud  ->  GraspNet.forward  ->  pred_decode  ->  [post: voxel down-sample of the RAW masked cloud,
        collision filter, sort + NMS]
    intrinsics = (fx, fy, cx, cy, factor_depth); keys (B,4) / seed: the device's draw (draw_positions), or positions (B,num_point):
    the caller's own (with the reference's idxs the cloud is the reference's bit for bit).  post: None or a dict of
    grasp_pipeline() keyword arguments; scene_points: rows of the raw masked cloud kept per image (default H*W).
    -> dict(end_points, grasps (B,M,17), counts (B,), cloud (B,num_point,3), n_valid_pixels (B,)
            [, final_rows, kept, collision_mask, voxel_status, scene (B,scene_points,3), scene_n_valid (B,)]).
    ~25 + ~20 launches, no host read, no allocation besides torch's caching allocator: hipGraph-capturable
    (GraphedGraspNet(front=...))."""
    fx, fy, cx, cy, factor = (float(v) for v in intrinsics)
    if depth.dim() == 2:
        depth, mask = depth.unsqueeze(0), (None if mask is None else mask.unsqueeze(0))
    valid = valid_pixels(depth, mask, factor, depth_min, depth_max)
    cloud, cnt = depth_to_cloud_masked(depth, mask, fx, fy, cx, cy, factor, num_point, positions=positions, keys=keys, seed=seed,
                                       depth_min=depth_min, depth_max=depth_max, valid=valid)
    ep = net({"point_clouds": cloud})
    grasps, counts = net.decode_padded(ep)
    out = {"end_points": ep, "grasps": grasps, "counts": counts, "cloud": cloud, "n_valid_pixels": cnt}
    if post is not None:
        scene, scene_n = masked_cloud(depth, mask, fx, fy, cx, cy, factor, scene_points, depth_min, depth_max, valid=valid)
        rows, kept, cm, status = grasp_pipeline(scene, grasps, counts, n_valid=scene_n, return_status=True, **post)
        out.update(final_rows=rows, kept=kept, collision_mask=cm, voxel_status=status, scene=scene, scene_n_valid=scene_n)
    return out
