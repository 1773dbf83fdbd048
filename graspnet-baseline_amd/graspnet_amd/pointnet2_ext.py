"""Drop-in for the reference's native module `pointnet2._ext`
(pointnet2/_ext_src/src/bindings.cpp:12-27): same function names, positional
signatures, return types and error behaviour, implemented by hand-written
gfx950 kernels behind the C ABI in include/graspnet_hip.h.

Error behaviour mirrored from pointnet2/_ext_src/include/utils.h:10-30 and the
wrappers in pointnet2/_ext_src/src/*.cpp: a failed check raises RuntimeError
("<arg> must be a contiguous tensor" / "... a float tensor" / "... an int
tensor" / "... a CUDA tensor"); CPU tensors raise "CPU not supported" exactly
like the reference (there is no CPU path upstream either).
Outputs are freshly allocated on the input's device; kernels are enqueued on
the current torch stream; nothing synchronises.
"""
import torch

from . import _lib as L


def _contig(t, name):
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)


def _float(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be a float tensor" % name)


def _int(t, name):
    if t.dtype != torch.int32:
        raise RuntimeError("%s must be an int tensor" % name)


def _cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)


GRID_MIN_POINTS = 4096      # below this the index-order scan (query.hip) is as fast as binning the cloud first


def _use_grid(n, m, nsample, radius):
    """the grid kernels keep the nsample smallest hit indices in one 64-lane register: nsample <= 64"""
    import os
    return (n >= GRID_MIN_POINTS and m > 0 and 1 <= int(nsample) <= 64 and float(radius) > 0.0
            and os.environ.get("GN_QUERY_PATH", "grid") != "brute")


# ---- one grid per cloud, not per query --------------------------------------------------------------------------------
# The reference model queries ONE cloud tensor five times per forward (SA1's ball_query, then CloudCrop's four
# cylinder_query calls: backbone.py:105, modules.py:127-131); binning it again each time was 40 us of the 64-76 us a
# drop-in query took and 4.9-6.4x the compulsory HBM bytes.  The last grid built on a (device, stream) is kept together with
# a reference to the tensor it was built from and that tensor's version counter; a later query on the same storage
# (same pointer, shape and version -- torch bumps the counter, shared by all views, on every in-place write; holding the
# tensor keeps the allocator from handing its memory to anyone else) on the same stream reuses it.  The query kernels take
# any cell size (a grid built for a smaller reach only widens their cell window), so the first query's radius decides it,
# exactly as in the fused path (one grid at SA1's radius for the ball query and the cylinder queries).  GN_GRID_CACHE=0
# turns it off.  Memory written behind torch's back (a raw kernel on xyz's pointer, a write through `.data`) is not seen:
# inputs are borrowed, never modified, on both sides of this boundary -- a caller that does rewrite a cloud in place that way
# calls clear_grid_cache() (or sets GN_GRID_CACHE=0).
# hipGraph capture: an entry also carries the id of the capture it was enqueued in (0 = eagerly; gn_stream_capture_id) and is
# only reused while THAT capture is running (or, for id 0, outside any capture).  A grid built eagerly during warm-up is thus
# never reused inside a capture on the same stream -- the build would be missing from the graph and its replays would read the
# warm-up cloud's cells -- and a grid built inside a capture (its buffer lives in the graph's memory pool, the build is part
# of the graph) serves the later queries of the same capture, as it does eagerly: one build per captured forward.
_GRID_CACHE = {}


def _grid_for(lib, xyz, b, n, reach, st, what):
    import os
    dev = xyz.device
    key = (dev.index, int(st or 0))
    use = os.environ.get("GN_GRID_CACHE", "1") != "0"
    cap = L.capture_id(st) if use else 0
    hit = _GRID_CACHE.get(key) if use else None
    if hit is not None:
        src, ver, grid, built, hit_cap = hit
        # (a grid built for a very different reach stays correct but walks many small / few huge cells: rebuild then)
        if (hit_cap == cap and src.data_ptr() == xyz.data_ptr() and src.shape == xyz.shape and src._version == ver
                and xyz._version == ver and 0.25 * reach <= built <= 1.5 * reach):
            return grid
    grid = torch.empty((lib.gn_grid_workspace_bytes(b, n),), dtype=torch.uint8, device=dev)
    L.check(lib.gn_grid_build(xyz.data_ptr(), b, n, float(reach), grid.data_ptr(), st), what + " (grid build)")
    if use:
        _GRID_CACHE[key] = (xyz, xyz._version, grid, float(reach), cap)
    return grid


def clear_grid_cache():
    """drop the cached grids (and the references to the clouds they were built from)"""
    _GRID_CACHE.clear()


def _deterministic_grads():
    """The gradient operators run as sorted-segment sums (csrc/grad_det.hip: bit-identical to a sequential loop, the same
    every run) unless GN_GRAD=atomic asks for the reference's float atomicAdd scatter."""
    import os
    return os.environ.get("GN_GRAD", "deterministic") != "atomic"


def _grad_ws(lib, b, ndest, entries, device):
    nbytes = lib.gn_grad_workspace_bytes(b, ndest, entries)
    return torch.empty((nbytes,), dtype=torch.uint8, device=device) if nbytes else None


def _dispatch(first, others):
    """First tensor decides the device (e.g. ball_query.cpp:20-34)."""
    if not first.is_cuda:
        raise RuntimeError("CPU not supported")
    for t, name in others:
        _cuda(t, name)
    return L.lib(), L.stream_of(first)


# ---- sampling the sampled points again -------------------------------------------------------------------------------------
# The reference backbone samples 20 000 -> 2048 points and then samples THOSE again, 2048 -> 1024 -> 512 -> 256
# (backbone.py:105-121, its comments note that the result is the identity prefix).  It is -- provably -- as long as the first
# call met no exact tie between distinct points before the step in question (a unique maximum is the same point in any tie
# order), which the fused path knows and uses (csrc/forward.cpp).  A drop-in call only sees tensors, so the last call per
# (device, stream) is remembered -- its input, its indices, its tie step, with the tensors' version counters -- and a later call
# whose input is bit for bit the points that call picked (gn_fps_prefix_check, one small launch) inherits the tie step as the
# kernel's `prefix_ok`: the identity prefix is written at once, or the chain warm-starts behind the tie.  An input that is
# anything else gets no promise and runs in full.  GN_FPS_CHAIN=0 turns it off.  Like the grid cache, a remembered call is
# only consulted inside the hipGraph capture it was enqueued in (or eagerly, if it was enqueued eagerly): a captured
# gn_fps_prefix_check would otherwise read, on every replay, tensors of a warm-up call that the allocator is free to reuse.
_FPS_LAST = {}


def furthest_point_sampling(points, nsamples):
    """(B,N,3) f32 -> (B,nsamples) i32   [sampling.cpp:70-91]"""
    import os
    _contig(points, "points"); _float(points, "points")
    lib, st = _dispatch(points, [])
    b, n = points.size(0), points.size(1)
    dev = points.device
    out = torch.empty((b, nsamples), dtype=torch.int32, device=dev)
    wsb = lib.gn_fps_workspace_bytes(b, n)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev) if wsb else None
    chain = os.environ.get("GN_FPS_CHAIN", "1") != "0" and b > 0 and n > 0 and nsamples > 0
    key = (dev.index, int(st or 0))
    # the tie report is asked for where it is free, or where a chained call behind this one would save more than it costs
    # (re-sampling >= 512 picked points: 0.1-0.9 ms of chains against 0.12 ms for the tie-tracking kernel)
    want_tie = chain and (int(nsamples) >= 512 or bool(lib.gn_fps_tie_is_free(n)))
    tie = torch.empty((b,), dtype=torch.int32, device=dev) if want_tie else None
    prefix = None
    with torch.cuda.device(dev):
        cap = L.capture_id(st) if chain else 0
        last = _FPS_LAST.get(key) if chain else None
        if last is not None:
            src, src_ver, inds, inds_ver, t1, last_cap = last
            if (last_cap == cap and src.size(0) == b and inds.size(1) >= n and src._version == src_ver
                    and inds._version == inds_ver and src.data_ptr() != points.data_ptr()):
                prefix = torch.empty((b,), dtype=torch.int32, device=dev)
                L.check(lib.gn_fps_prefix_check(points.data_ptr(), src.data_ptr(), inds.data_ptr(), t1.data_ptr(), b, n,
                                                src.size(1), inds.size(1), prefix.data_ptr(), st), "fps_prefix_check")
        L.check(lib.gn_furthest_point_sampling(points.data_ptr(), b, n, int(nsamples), L.ptr(ws), out.data_ptr(), None,
                                               L.ptr(tie) if prefix is None else None, L.ptr(prefix), st),
                "furthest_point_sampling")
    if chain:
        # a call that ran on a promise passes the promise on (its own kernel does not track ties); one that did not reports
        # its own tie step (0 = "unknown" from kernels that do not track it: no promise for the next link)
        link = prefix if prefix is not None else tie
        if link is None:
            _FPS_LAST.pop(key, None)            # no report was asked for: nothing to inherit
        else:
            _FPS_LAST[key] = (points, points._version, out, out._version, link, cap)
    return out


def clear_fps_chain():
    """forget the last sampling call of every stream (and the references to its tensors)"""
    _FPS_LAST.clear()


def gather_points(points, idx):
    """(B,C,N) f32, (B,M) i32 -> (B,C,M)   [sampling.cpp:20-42]"""
    _contig(points, "points"); _contig(idx, "idx"); _float(points, "points"); _int(idx, "idx")
    lib, st = _dispatch(points, [(idx, "idx")])
    b, c, n = points.shape
    m = idx.size(1)
    out = torch.empty((b, c, m), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        L.check(lib.gn_gather_points(points.data_ptr(), idx.data_ptr(), b, c, n, m, out.data_ptr(), st),
                "gather_points")
    return out


def gather_points_grad(grad_out, idx, n):
    """(B,C,M) f32, (B,M) i32, n -> (B,C,n)   [sampling.cpp:44-68]"""
    _contig(grad_out, "grad_out"); _contig(idx, "idx"); _float(grad_out, "grad_out"); _int(idx, "idx")
    lib, st = _dispatch(grad_out, [(idx, "idx")])
    b, c, m = grad_out.shape
    out = torch.empty((b, c, int(n)), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        if _deterministic_grads():
            ws = _grad_ws(lib, b, int(n), m, grad_out.device)
            L.check(lib.gn_gather_points_grad_det(grad_out.data_ptr(), idx.data_ptr(), b, c, int(n), m, L.ptr(ws),
                                                  out.data_ptr(), st), "gather_points_grad")
        else:
            L.check(lib.gn_gather_points_grad(grad_out.data_ptr(), idx.data_ptr(), b, c, int(n), m,
                                              out.data_ptr(), st), "gather_points_grad")
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """(B,M,3), (B,N,3), r, ns -> (B,M,ns) i32   [ball_query.cpp:13-37]"""
    _contig(new_xyz, "new_xyz"); _contig(xyz, "xyz"); _float(new_xyz, "new_xyz"); _float(xyz, "xyz")
    lib, st = _dispatch(new_xyz, [(xyz, "xyz")])
    b, m = new_xyz.size(0), new_xyz.size(1)
    n = xyz.size(1)
    out = torch.empty((b, m, int(nsample)), dtype=torch.int32, device=new_xyz.device)
    with torch.cuda.device(new_xyz.device):
        if _use_grid(n, m, nsample, radius):
            # the cloud binned into cells (once per cloud: _grid_for), then only the cells a ball can touch are tested
            # (same index lists as the scan: tests/test_ops_gpu.py)
            grid = _grid_for(lib, xyz, b, n, float(radius), st, "ball_query")
            L.check(lib.gn_ball_query_grid(new_xyz.data_ptr(), grid.data_ptr(), b, n, m, float(radius), int(nsample),
                                           out.data_ptr(), st), "ball_query")
        else:
            L.check(lib.gn_ball_query(new_xyz.data_ptr(), xyz.data_ptr(), b, n, m, float(radius), int(nsample),
                                      out.data_ptr(), st), "ball_query")
    return out


def cylinder_query(new_xyz, xyz, rot, radius, hmin, hmax, nsample):
    """(B,M,3), (B,N,3), (B,M,9), r, hmin, hmax, ns -> (B,M,ns) i32   [cylinder_query.cpp:10-37]"""
    _contig(new_xyz, "new_xyz"); _contig(xyz, "xyz"); _contig(rot, "rot")
    _float(new_xyz, "new_xyz"); _float(xyz, "xyz"); _float(rot, "rot")
    lib, st = _dispatch(new_xyz, [(xyz, "xyz"), (rot, "rot")])
    b, m = new_xyz.size(0), new_xyz.size(1)
    n = xyz.size(1)
    out = torch.empty((b, m, int(nsample)), dtype=torch.int32, device=new_xyz.device)
    with torch.cuda.device(new_xyz.device):
        if _use_grid(n, m, nsample, radius):
            reach = (float(radius) ** 2 + max(abs(float(hmin)), abs(float(hmax))) ** 2) ** 0.5
            grid = _grid_for(lib, xyz, b, n, reach, st, "cylinder_query")
            L.check(lib.gn_cylinder_query_grid(new_xyz.data_ptr(), rot.data_ptr(), grid.data_ptr(), b, n, m, float(radius),
                                               float(hmin), float(hmax), int(nsample), out.data_ptr(), st), "cylinder_query")
        else:
            L.check(lib.gn_cylinder_query(new_xyz.data_ptr(), xyz.data_ptr(), rot.data_ptr(), b, n, m,
                                          float(radius), float(hmin), float(hmax), int(nsample),
                                          out.data_ptr(), st), "cylinder_query")
    return out


def group_points(points, idx):
    """(B,C,N) f32, (B,NP,NS) i32 -> (B,C,NP,NS)   [group_points.cpp:17-40]"""
    _contig(points, "points"); _contig(idx, "idx"); _float(points, "points"); _int(idx, "idx")
    lib, st = _dispatch(points, [(idx, "idx")])
    b, c, n = points.shape
    npoints, nsample = idx.size(1), idx.size(2)
    out = torch.empty((b, c, npoints, nsample), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        L.check(lib.gn_group_points(points.data_ptr(), idx.data_ptr(), b, c, n, npoints, nsample,
                                    out.data_ptr(), st), "group_points")
    return out


def group_points_grad(grad_out, idx, n):
    """(B,C,NP,NS) f32, (B,NP,NS) i32, n -> (B,C,n)   [group_points.cpp:42-66]"""
    _contig(grad_out, "grad_out"); _contig(idx, "idx"); _float(grad_out, "grad_out"); _int(idx, "idx")
    lib, st = _dispatch(grad_out, [(idx, "idx")])
    b, c, npoints, nsample = grad_out.shape
    out = torch.empty((b, c, int(n)), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        if _deterministic_grads():
            ws = _grad_ws(lib, b, int(n), npoints * nsample, grad_out.device)
            L.check(lib.gn_group_points_grad_det(grad_out.data_ptr(), idx.data_ptr(), b, c, int(n), npoints, nsample, L.ptr(ws),
                                                 out.data_ptr(), st), "group_points_grad")
        else:
            L.check(lib.gn_group_points_grad(grad_out.data_ptr(), idx.data_ptr(), b, c, int(n), npoints, nsample,
                                             out.data_ptr(), st), "group_points_grad")
    return out


def three_nn(unknowns, knows):
    """(B,n,3), (B,m,3) -> [dist2 (B,n,3) f32 (squared), idx (B,n,3) i32]   [interpolate.cpp:19-45]"""
    _contig(unknowns, "unknowns"); _contig(knows, "knows"); _float(unknowns, "unknowns"); _float(knows, "knows")
    lib, st = _dispatch(unknowns, [(knows, "knows")])
    b, n = unknowns.size(0), unknowns.size(1)
    m = knows.size(1)
    dist2 = torch.empty((b, n, 3), dtype=torch.float32, device=unknowns.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=unknowns.device)
    with torch.cuda.device(unknowns.device):
        L.check(lib.gn_three_nn(unknowns.data_ptr(), knows.data_ptr(), b, n, m, dist2.data_ptr(),
                                idx.data_ptr(), st), "three_nn")
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    """(B,C,m) f32, (B,n,3) i32, (B,n,3) f32 -> (B,C,n)   [interpolate.cpp:47-75]"""
    _contig(points, "points"); _contig(idx, "idx"); _contig(weight, "weight")
    _float(points, "points"); _int(idx, "idx"); _float(weight, "weight")
    lib, st = _dispatch(points, [(idx, "idx"), (weight, "weight")])
    b, c, m = points.shape
    n = idx.size(1)
    out = torch.empty((b, c, n), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        L.check(lib.gn_three_interpolate(points.data_ptr(), idx.data_ptr(), weight.data_ptr(), b, c, m, n,
                                         out.data_ptr(), st), "three_interpolate")
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """(B,C,n) f32, (B,n,3) i32, (B,n,3) f32, m -> (B,C,m)   [interpolate.cpp:77-104]"""
    _contig(grad_out, "grad_out"); _contig(idx, "idx"); _contig(weight, "weight")
    _float(grad_out, "grad_out"); _int(idx, "idx"); _float(weight, "weight")
    lib, st = _dispatch(grad_out, [(idx, "idx"), (weight, "weight")])
    b, c, n = grad_out.shape
    out = torch.empty((b, c, int(m)), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        if _deterministic_grads():
            ws = _grad_ws(lib, b, int(m), 3 * n, grad_out.device)
            L.check(lib.gn_three_interpolate_grad_det(grad_out.data_ptr(), idx.data_ptr(), weight.data_ptr(), b, c, n, int(m),
                                                      L.ptr(ws), out.data_ptr(), st), "three_interpolate_grad")
        else:
            L.check(lib.gn_three_interpolate_grad(grad_out.data_ptr(), idx.data_ptr(), weight.data_ptr(), b, c, n,
                                                  int(m), out.data_ptr(), st), "three_interpolate_grad")
    return out
