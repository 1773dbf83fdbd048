"""ctypes loader for libgraspnet_hip.so (the C-ABI of include/graspnet_hip.h).

There is NO fallback: if the HIP library has not been built, or a call
returns an error, this raises.  The product path never routes through the CPU
oracle or through torch re-implementations of the operators.
"""
import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)                      # graspnet-baseline_amd/
# GN_LIB_PATH: an alternative build of the same library (profiles/build_probe.sh: instrumented kernels)
LIB_PATH = os.environ.get("GN_LIB_PATH") or os.path.join(_ROOT, "lib", "libgraspnet_hip.so")
CSRC = os.path.join(_ROOT, "csrc")

_lib = None
ABI_VERSION = 3          # include/graspnet_hip.h: GN_ABI_VERSION

_c_int, _c_float, _c_void_p, _c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t



class MlpWeights(ctypes.Structure):          # gn_mlp_weights
    _fields_ = [("w", _c_void_p * 3), ("b", _c_void_p * 3)]


class Weights(ctypes.Structure):             # gn_weights
    _fields_ = [("sa", MlpWeights * 4), ("sa_pre", MlpWeights * 4), ("fp", MlpWeights * 2), ("approach", MlpWeights),
                ("crop", MlpWeights), ("heads", MlpWeights), ("views", _c_void_p), ("view_rot", _c_void_p),
                ("decode_rot", _c_void_p), ("cylinder_radius", _c_float), ("hmin", _c_float), ("hmax", _c_float * 4),
                ("use_split", _c_int), ("sa_split", MlpWeights * 4), ("crop_split", MlpWeights),
                ("sa_pre_split", MlpWeights * 4), ("fp_split", MlpWeights * 2), ("approach_split", MlpWeights),
                ("heads_split", MlpWeights), ("num_view", _c_int)]


class Outputs(ctypes.Structure):             # gn_outputs
    _fields_ = [("sa_inds", _c_void_p * 4), ("sa_xyz", _c_void_p * 4), ("sa_features", _c_void_p * 4),
                ("fp2_features", _c_void_p), ("approach", _c_void_p), ("top_view_inds", _c_void_p),
                ("top_view_score", _c_void_p), ("top_view_xyz", _c_void_p), ("top_view_rot", _c_void_p),
                ("vp_features", _c_void_p), ("head", _c_void_p), ("crop_uidx", _c_void_p), ("crop_ucount", _c_void_p),
                ("grasps", _c_void_p), ("counts", _c_void_p), ("fork_streams", ctypes.c_int32)]


STAGE_HOOK = ctypes.CFUNCTYPE(None, ctypes.c_char_p, _c_int, _c_void_p)   # gn_stage_hook

# name -> (restype, argtypes); must list every symbol include/graspnet_hip.h declares
SIGNATURES = {
    "gn_last_error": (ctypes.c_char_p, []),
    "gn_version": (ctypes.c_char_p, []),
    "gn_abi_version": (_c_int, []),
    "gn_set_fp_policy": (_c_int, [_c_int]),
    "gn_get_fp_policy": (_c_int, []),
    "gn_stream_capture_id": (_c_int, [_c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]),
    "gn_fps_workspace_bytes": (_c_size_t, [_c_int, _c_int]),
    "gn_fps_tie_is_free": (_c_int, [_c_int]),
    "gn_fps_prefix_check": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p,
                                     _c_void_p]),
    "gn_furthest_point_sampling": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                           _c_void_p, _c_void_p]),
    "gn_gather_points": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "gn_gather_points_grad": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "gn_ball_query": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_float, _c_int, _c_void_p, _c_void_p]),
    "gn_cylinder_query": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _c_int, _c_void_p, _c_void_p]),
    "gn_group_points": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "gn_group_points_grad": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "gn_three_nn": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "gn_three_interpolate": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "gn_three_interpolate_grad": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "gn_grad_workspace_bytes": (_c_size_t, [_c_int, _c_int, ctypes.c_long]),
    "gn_gather_points_grad_det": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "gn_group_points_grad_det": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p,
                                          _c_void_p]),
    "gn_three_interpolate_grad_det": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p,
                                               _c_void_p, _c_void_p]),
    "gn_knn": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "gn_grid_workspace_bytes": (_c_size_t, [_c_int, _c_int]),
    "gn_grid_build": (_c_int, [_c_void_p, _c_int, _c_int, _c_float, _c_void_p, _c_void_p]),
    "gn_ball_query_grid": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_float, _c_int, _c_void_p, _c_void_p]),
    "gn_cylinder_query_grid": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float,
                                        _c_int, _c_void_p, _c_void_p]),
    "gn_cylinder_query_unique_grid": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_float, _c_float,
                                               ctypes.POINTER(_c_float), _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "gn_cylinder_query_multi": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_float, _c_float,
                                         ctypes.POINTER(_c_float), _c_int, _c_int, _c_void_p, _c_void_p]),
    "gn_cylinder_query_unique": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_float, _c_float,
                                          ctypes.POINTER(_c_float), _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "gn_crop_unique_mlp_pool": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                                         _c_int, _c_int, ctypes.POINTER(_c_int), ctypes.POINTER(_c_void_p),
                                         ctypes.POINTER(_c_void_p), _c_void_p, _c_void_p]),
    "gn_crop_unique_mlp_pool_split": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                                               _c_int, _c_int, ctypes.POINTER(_c_int), ctypes.POINTER(_c_void_p),
                                               ctypes.POINTER(_c_void_p), _c_void_p, _c_void_p]),
    "gn_sa_group_mlp_pool_split": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_float,
                                            ctypes.POINTER(_c_int), ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p),
                                            _c_void_p, _c_void_p]),
    "gn_sa_hoisted_mlp_pool_split": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                                              _c_float, ctypes.POINTER(_c_int), ctypes.POINTER(_c_void_p),
                                              ctypes.POINTER(_c_void_p), _c_void_p, _c_void_p]),
    "gn_sa_group_mlp_pool": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                      _c_float, ctypes.POINTER(_c_int), ctypes.POINTER(_c_void_p),
                                      ctypes.POINTER(_c_void_p), _c_void_p, _c_void_p]),
    "gn_point_linear": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "gn_sa_hoisted_mlp_pool": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                                        _c_float, ctypes.POINTER(_c_int), ctypes.POINTER(_c_void_p),
                                        ctypes.POINTER(_c_void_p), _c_void_p, _c_void_p]),
    "gn_crop_group_mlp_pool": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                        ctypes.POINTER(_c_int), ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p),
                                        _c_void_p, _c_void_p]),
    "gn_fp_interp_mlp": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                  ctypes.POINTER(_c_int), ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p),
                                  _c_void_p, _c_void_p]),
    "gn_fp_interp_mlp_split": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                        ctypes.POINTER(_c_int), ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p),
                                        _c_void_p, _c_void_p]),
    "gn_approach_mlp_split": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, ctypes.POINTER(_c_void_p),
                                       ctypes.POINTER(_c_void_p), _c_void_p, _c_void_p]),
    "gn_heads_mlp_split": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p),
                                    _c_void_p, _c_void_p]),
    "gn_set_point_tile_policy": (_c_int, [_c_int]),
    "gn_point_linear_split": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "gn_approach_mlp": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p),
                                 _c_void_p, _c_void_p]),
    "gn_heads_mlp": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p),
                              _c_void_p, _c_void_p]),
    "gn_approach_post": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                  _c_void_p, _c_void_p, _c_void_p]),
    "gn_depth_to_cloud": (_c_int, [_c_void_p, _c_int, _c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                   ctypes.c_double, ctypes.c_double, _c_void_p, _c_int, _c_void_p, _c_void_p]),
    "gn_frontend_workspace_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "gn_depth_valid_compact": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, ctypes.c_double, ctypes.c_double,
                                        ctypes.c_double, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "gn_depth_sample_cloud": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                       ctypes.c_double, ctypes.c_double, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int,
                                       _c_void_p, _c_void_p, _c_void_p]),
    "gn_depth_masked_cloud": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                       ctypes.c_double, ctypes.c_double, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "gn_depth_draw_positions": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p]),
    "gn_voxel_batched_workspace_bytes": (_c_size_t, [_c_int, _c_int]),
    "gn_voxel_down_sample_batched": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, ctypes.c_double, _c_void_p, _c_void_p,
                                             _c_void_p, _c_void_p, _c_void_p]),
    "gn_collision_detect_batched": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_float,
                                             ctypes.c_double, ctypes.c_double, ctypes.c_double, _c_float, _c_float, _c_void_p,
                                             _c_void_p, _c_void_p, _c_void_p]),
    "gn_voxel_workspace_bytes": (_c_size_t, [_c_int]),
    "gn_voxel_down_sample": (_c_int, [_c_void_p, _c_int, ctypes.c_double, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "gn_collision_detect": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_int, _c_float,
                                     ctypes.c_double, ctypes.c_double, ctypes.c_double, _c_float, _c_float, _c_void_p, _c_void_p,
                                     _c_void_p, _c_void_p]),
    "gn_grasp_nms_workspace_bytes": (_c_size_t, [_c_int]),
    "gn_grasp_nms": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, ctypes.c_double, ctypes.c_double, _c_int,
                              _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "gn_grasp_postprocess_workspace_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "gn_grasp_postprocess": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_int, ctypes.c_double,
                                      _c_float, ctypes.c_double, ctypes.c_double, ctypes.c_double, _c_int, _c_void_p, _c_void_p,
                                      _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "gn_graspnet_workspace_bytes": (_c_size_t, [_c_int, _c_int]),
    "gn_graspnet_forward": (_c_int, [ctypes.POINTER(Weights), _c_void_p, _c_int, _c_int, _c_void_p, ctypes.POINTER(Outputs),
                                     STAGE_HOOK, _c_void_p, _c_void_p]),
    "gn_pred_decode": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                                _c_int, _c_void_p, _c_void_p, _c_void_p]),
}


def build(verbose=False):
    """Compile every HIP source for gfx950 into lib/libgraspnet_hip.so (hipcc, in-tree)."""
    cmd = ["make", "-C", CSRC, "-j8", "all"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libgraspnet_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C graspnet-baseline_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if l.gn_abi_version() != ABI_VERSION:
            raise ImportError("%s has ABI revision %d, this package binds revision %d (include/graspnet_hip.h: GN_ABI_VERSION): "
                              "rebuild it" % (LIB_PATH, l.gn_abi_version(), ABI_VERSION))
        _lib = l
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().gn_last_error().decode("utf-8", "replace")
        raise RuntimeError("%s failed (gn_status %d): %s" % (what, status, msg))


FP_POLICIES = {"as_written": 0, "contracted": 1, "contracted_b": 2}


def set_fp_policy(policy):
    """How the index-deciding float expressions are evaluated (include/graspnet_hip.h gn_set_fp_policy): "as_written" (default:
    no FMA contraction), "contracted" (LLVM / NVVM order) or "contracted_b"; or 0 / 1 / 2.  Returns the previous policy's name."""
    names = {v: k for k, v in FP_POLICIES.items()}
    prev = names[lib().gn_get_fp_policy()]
    check(lib().gn_set_fp_policy(FP_POLICIES[policy] if isinstance(policy, str) else int(policy)), "set_fp_policy")
    return prev


def capture_id(stream):
    """0 when `stream` (the void* the C ABI takes) is not being captured into a hipGraph, else the capture's unique id."""
    cid = ctypes.c_ulonglong(0)
    check(lib().gn_stream_capture_id(stream, ctypes.byref(cid)), "stream_capture_id")
    return int(cid.value)


def ptr(t):
    return None if t is None else t.data_ptr()


def stream_of(t):
    """The current torch stream of t's device, as the void* the C ABI takes."""
    import torch
    return torch.cuda.current_stream(t.device).cuda_stream
