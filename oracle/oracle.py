"""ctypes front end of the CPU oracle (oracle/gn_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (graspnet_amd) never imports
this module.

The functions mirror the positional signatures of the reference's
`pointnet2._ext` pybind module (pointnet2/_ext_src/src/bindings.cpp:12-27) and
of `knn_pytorch.knn_pytorch.knn` (knn/src/vision.cpp:3-5), for torch CPU
tensors, so that the reference python -- or our own restatement of it -- can
run on top of them.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgn_oracle.so")
_lib = None


def build(force=False):
    """Compile libgn_oracle.so with gcc (seconds)."""
    srcs = [os.path.join(_HERE, "gn_oracle.c"), os.path.join(_HERE, "gn_oracle_fp.inc")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src) for src in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.gno_opt_n_threads.restype = ctypes.c_int
        _lib.gno_num_threads.restype = ctypes.c_int
    return _lib


FP_POLICIES = {"as_written": 0, "contracted": 1, "contracted_b": 2}


def set_fp_policy(policy):
    """How the index-deciding float expressions are evaluated (gn_oracle_fp.inc): "as_written" (default), "contracted"
    (fma(c,c,fma(a,a,b*b)): LLVM / NVVM order) or "contracted_b" (fma(c,c,fma(b,b,a*a))) -- the same switch as the HIP
    library's gn_set_fp_policy.  Returns the previous policy's name."""
    names = {v: k for k, v in FP_POLICIES.items()}
    l = lib()
    l.gno_get_fp_policy.restype = ctypes.c_int
    prev = names[int(l.gno_get_fp_policy())]
    l.gno_set_fp_policy(ctypes.c_int(FP_POLICIES[policy] if isinstance(policy, str) else int(policy)))
    return prev


def num_threads():
    return int(lib().gno_num_threads())


def set_num_threads(n):
    lib().gno_set_num_threads(ctypes.c_int(int(n)))


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _chk(t, dtype, name):
    if t.device.type != "cpu":
        raise RuntimeError("oracle: %s must be a CPU tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be a %s tensor" % (name, dtype))


_f = ctypes.c_float
_i = ctypes.c_int


def furthest_point_sampling(points, nsamples):
    _chk(points, torch.float32, "points")
    b, n, _ = points.shape
    out = torch.zeros(b, nsamples, dtype=torch.int32)
    tmp = torch.full((b, n), 1e10, dtype=torch.float32)
    lib().gno_furthest_point_sampling(_i(b), _i(n), _i(nsamples), _p(points), _p(tmp), _p(out))
    return out


def gather_points(points, idx):
    _chk(points, torch.float32, "points"); _chk(idx, torch.int32, "idx")
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.zeros(b, c, m, dtype=torch.float32)
    lib().gno_gather_points(_i(b), _i(c), _i(n), _i(m), _p(points), _p(idx), _p(out))
    return out


def gather_points_grad(grad_out, idx, n):
    _chk(grad_out, torch.float32, "grad_out"); _chk(idx, torch.int32, "idx")
    b, c, m = grad_out.shape
    out = torch.zeros(b, c, n, dtype=torch.float32)
    lib().gno_gather_points_grad(_i(b), _i(c), _i(n), _i(m), _p(grad_out), _p(idx), _p(out))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    _chk(new_xyz, torch.float32, "new_xyz"); _chk(xyz, torch.float32, "xyz")
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = torch.zeros(b, m, nsample, dtype=torch.int32)
    lib().gno_ball_query(_i(b), _i(n), _i(m), _f(radius), _i(nsample), _p(new_xyz), _p(xyz), _p(idx))
    return idx


def cylinder_query(new_xyz, xyz, rot, radius, hmin, hmax, nsample):
    _chk(new_xyz, torch.float32, "new_xyz"); _chk(xyz, torch.float32, "xyz")
    _chk(rot, torch.float32, "rot")
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = torch.zeros(b, m, nsample, dtype=torch.int32)
    lib().gno_cylinder_query(_i(b), _i(n), _i(m), _f(radius), _f(hmin), _f(hmax), _i(nsample),
                             _p(new_xyz), _p(xyz), _p(rot), _p(idx))
    return idx


def group_points(points, idx):
    _chk(points, torch.float32, "points"); _chk(idx, torch.int32, "idx")
    b, c, n = points.shape
    npoints, nsample = idx.shape[1], idx.shape[2]
    out = torch.zeros(b, c, npoints, nsample, dtype=torch.float32)
    lib().gno_group_points(_i(b), _i(c), _i(n), _i(npoints), _i(nsample), _p(points), _p(idx), _p(out))
    return out


def group_points_grad(grad_out, idx, n):
    _chk(grad_out, torch.float32, "grad_out"); _chk(idx, torch.int32, "idx")
    b, c, npoints, nsample = grad_out.shape
    out = torch.zeros(b, c, n, dtype=torch.float32)
    lib().gno_group_points_grad(_i(b), _i(c), _i(n), _i(npoints), _i(nsample), _p(grad_out), _p(idx), _p(out))
    return out


def three_nn(unknowns, knows):
    _chk(unknowns, torch.float32, "unknowns"); _chk(knows, torch.float32, "knows")
    b, n, _ = unknowns.shape
    m = knows.shape[1]
    dist2 = torch.zeros(b, n, 3, dtype=torch.float32)
    idx = torch.zeros(b, n, 3, dtype=torch.int32)
    lib().gno_three_nn(_i(b), _i(n), _i(m), _p(unknowns), _p(knows), _p(dist2), _p(idx))
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    _chk(points, torch.float32, "points"); _chk(idx, torch.int32, "idx")
    _chk(weight, torch.float32, "weight")
    b, c, m = points.shape
    n = idx.shape[1]
    out = torch.zeros(b, c, n, dtype=torch.float32)
    lib().gno_three_interpolate(_i(b), _i(c), _i(m), _i(n), _p(points), _p(idx), _p(weight), _p(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    _chk(grad_out, torch.float32, "grad_out"); _chk(idx, torch.int32, "idx")
    _chk(weight, torch.float32, "weight")
    b, c, n = grad_out.shape
    out = torch.zeros(b, c, m, dtype=torch.float32)
    lib().gno_three_interpolate_grad(_i(b), _i(c), _i(n), _i(m), _p(grad_out), _p(idx), _p(weight), _p(out))
    return out


def knn(ref, query, idx):
    """knn_pytorch.knn(ref (B,D,Nr), query (B,D,Nq), idx (B,k,Nq) int64) -> 1; idx 1-based."""
    _chk(ref, torch.float32, "ref"); _chk(query, torch.float32, "query")
    _chk(idx, torch.int64, "idx")
    b, d, nr = ref.shape
    nq = query.shape[2]
    k = idx.shape[1]
    lib().gno_knn(_i(b), _i(d), _i(nr), _i(nq), _i(k), _p(ref), _p(query), _p(idx))
    return 1
