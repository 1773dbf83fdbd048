"""GraspNet inference on MI355X: host-side mirror of the reference model API.

Same constructor arguments, `forward(end_points) -> end_points` keys/shapes and
`pred_decode(end_points)` result as models/graspnet.py:63-134 of the reference, executed as ~25
hand-written gfx950 kernels behind the C ABI (include/graspnet_hip.h).  torch is used for device
memory and streams only.

Differences a caller can observe:
  * inference only (is_training=True is refused: label generation / losses are out of scope);
  * feature tensors in `end_points` have the reference's SHAPES but are strided views of
    position-major buffers (call .contiguous() if a dense (B,C,N) layout is needed);
  * weights are given as a state-dict with the reference's key layout via load_state_dict().
"""
import ctypes
import os

import torch

from . import _lib as L
from . import geometry, weights

# models/backbone.py:30-64 : (npoint, radius, nsample, C_in_features, mlp)
SA_CFG = (
    (2048, 0.04, 64, 0, (64, 64, 128)),
    (1024, 0.1, 32, 128, (128, 128, 256)),
    (512, 0.2, 16, 256, (128, 128, 256)),
    (256, 0.3, 16, 256, (128, 128, 256)),
)
GRASP_ROW = 17   # [score,width,height,depth,R(9),center(3),obj_id]  models/graspnet.py:131-133


def _ci(*vals):
    return (ctypes.c_int * len(vals))(*vals)


class GraspNet:
    """Drop-in for `models.graspnet.GraspNet` (inference)."""

    def __init__(self, input_feature_dim=0, num_view=300, num_angle=12, num_depth=4, cylinder_radius=0.05,
                 hmin=-0.02, hmax_list=(0.01, 0.02, 0.03, 0.04), is_training=False, device="cuda"):
        if is_training:
            raise NotImplementedError("graspnet_amd.GraspNet is an inference path (is_training=False)")
        if input_feature_dim != 0:
            raise NotImplementedError("input_feature_dim != 0 is not used by the reference model")
        if num_depth != 4 or len(hmax_list) != 4 or num_angle != 12 or not 1 <= int(num_view) <= 318:
            raise NotImplementedError("built for 12 angles, 4 depths (the reference's heads) and 1..318 approach views")
        self.num_view, self.num_angle, self.num_depth = num_view, num_angle, num_depth
        self.cylinder_radius, self.hmin = float(cylinder_radius), float(hmin)
        self.hmax = (ctypes.c_float * 4)(*[float(h) for h in hmax_list])
        self.crop_nsample = 64                     # models/graspnet.py:44
        self.is_training = False
        self.device = torch.device(device)
        self._packed = None
        self._tables = None
        self._sd = None
        self.timing = None        # set to {} to collect (start,end) HIP events per kernel launch
        # True: gn_graspnet_forward may run the grid build and the deeper levels' geometry on a side stream of its own
        # (gn_outputs.fork_streams).  A steady gain only inside a captured hipGraph (GraphedGraspNet sets it while it
        # captures); between eagerly issued streams it depends on the hardware-queue mapping, so the default is off
        self.fork = False
        # Tile shapes of the per-point chains (FP, ApproachNet, heads; include/graspnet_hip.h gn_set_point_tile_policy).  False:
        # every launch spreads over as many CUs as it can -- the shortest launch for ONE batch at a time.  True: 64- / 128-row
        # fp32 tiles (csrc/mlp_point_wide.hip) that stream the weights once per 64 / 128 rows on half as many CUs: a third fewer
        # CU-microseconds per batch, which other batches in flight use (PipelinedGraspNet sets it).  Outputs are bit-identical.
        self.throughput = os.environ.get("GN_POINT_TILES", "latency").startswith("t")
        # CloudCrop over de-duplicated (seed, point) rows (default) or over the four full index lists
        self.crop_unique = os.environ.get("GN_CROP_PATH", "unique") != "lists"
        # SA2-4: layer 1's per-point part hoisted out of the per-neighbour MLP (bit-identical, fewer MFMAs)
        self.sa_hoist = os.environ.get("GN_SA_PATH", "hoist") != "plain"
        # neighbour queries on the input cloud through a uniform grid (exact) or by brute-force scan
        self.query_grid = os.environ.get("GN_QUERY_PATH", "grid") != "brute"
        # the default configuration is sequenced by the library itself (gn_graspnet_forward: one host call per
        # forward); GN_FORWARD=stages issues the same stage calls one by one from here (always used for the
        # non-default A/B paths above)
        self.native_forward = os.environ.get("GN_FORWARD", "native") != "stages"
        # grouped layers on the bf16 matrix cores by exact operand splitting (csrc/mlp_split.h; as accurate as fp32,
        # 1.5-1.6x faster kernels) or on the fp32 MFMA (GN_MATH=f32)
        self.math_split = os.environ.get("GN_MATH", "split") != "f32"

    def _run(self, name, status_fn):
        """Issue one C-ABI call; when self.timing is a dict, bracket it with events on the launch stream."""
        if self.timing is None:
            L.check(status_fn(), name)
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        L.check(status_fn(), name)
        b.record()
        self.timing.setdefault(name, []).append((a, b))

    # ---- nn.Module-like surface used by the reference's callers (demo.py:get_net etc.)
    def load_state_dict(self, state_dict, strict=True):
        want = _expected_keys()
        missing = [k for k in want if k not in state_dict]
        unexpected = [k for k in state_dict if k not in want]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for GraspNet: missing %s unexpected %s"
                               % (missing[:5], unexpected[:5]))
        k3 = "view_estimator.vpmodule.conv3.weight"
        if k3 in state_dict and int(state_dict[k3].shape[0]) != 2 + self.num_view:
            raise RuntimeError("Error(s) in loading state_dict for GraspNet: size mismatch for %s: the checkpoint holds %d views, "
                               "the model was built with num_view=%d" % (k3, int(state_dict[k3].shape[0]) - 2, self.num_view))
        self._sd = {k: v.detach().clone() for k, v in state_dict.items()}
        self._build()
        return self

    def load_checkpoint(self, path, strict=True):
        """What every caller of the reference does with a checkpoint file (test.py:57-58, demo.py:429, demo_pointcloud.py:47-49,
        zividtest.py:88-98): `torch.load(path)['model_state_dict']` -> load_state_dict, here with a loader that executes
        nothing from the file (`weights_only=True`) and onto the CPU first (the fold / packing of weights.py runs in float64 on
        the host).  Accepts the dict train.py:211-219 saves ({'epoch', 'optimizer_state_dict', 'loss', 'model_state_dict'}),
        a bare state-dict, and keys carrying the `module.` prefix of an nn.DataParallel model (train.py:215-218,
        zividtest.py:93-97).  Returns the checkpoint's epoch (None if it holds none), like the callers print it."""
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        if not isinstance(ckpt, dict):
            raise RuntimeError("load_checkpoint: %s does not hold a dict" % path)
        sd = ckpt.get("model_state_dict", ckpt)
        if not isinstance(sd, dict) or not all(torch.is_tensor(v) for v in sd.values()):
            raise RuntimeError("load_checkpoint: %s holds no state-dict ('model_state_dict' or a bare one)" % path)
        if sd and all(k.startswith("module.") for k in sd):
            sd = {k[len("module."):]: v for k, v in sd.items()}
        self.load_state_dict(sd, strict=strict)
        return ckpt.get("epoch") if sd is not ckpt else None

    def state_dict(self):
        return dict(self._sd or {})

    def to(self, device):
        self.device = torch.device(device)
        if self._sd is not None:
            self._build()
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def eval(self):
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("training is out of scope")
        return self

    def _build(self):
        if self.device.type != "cuda":
            raise RuntimeError("graspnet_amd.GraspNet needs a GPU device (there is no CPU path)")
        L.lib()   # fail loudly here if the HIP library is missing
        self._packed = weights.prepare(self._sd, self.device, self.num_view, self.num_angle)
        v, vr, dr = geometry.build_tables(self.num_view, self.num_angle)
        self._tables = (v.to(self.device), vr.to(self.device), dr.to(self.device))
        self._cweights = self._make_cweights()

    def _make_cweights(self):
        """gn_weights for gn_graspnet_forward (device pointers into self._packed / self._tables)."""
        w, pk = L.Weights(), self._packed

        def fill(dst, packed):
            for i, (wt, bt) in enumerate(zip(packed.w, packed.b)):
                dst.w[i], dst.b[i] = wt.data_ptr(), bt.data_ptr()

        fill(w.sa[0], pk["sa1"])
        for l in (1, 2, 3):
            fill(w.sa[l], pk["sa%d_hoist" % (l + 1)])
            fill(w.sa_pre[l], pk["sa%d_pre" % (l + 1)])
        fill(w.fp[0], pk["fp1"])
        fill(w.fp[1], pk["fp2"])
        fill(w.approach, pk["approach"])
        fill(w.crop, pk["crop"])
        fill(w.heads, pk["heads"])
        w.views, w.view_rot, w.decode_rot = (t.data_ptr() for t in self._tables)
        w.cylinder_radius, w.hmin = self.cylinder_radius, self.hmin
        for i in range(4):
            w.hmax[i] = self.hmax[i]
        fill(w.sa_split[0], pk["sa1_split"])
        for l in (1, 2, 3):
            fill(w.sa_split[l], pk["sa%d_hoist_split" % (l + 1)])
        fill(w.crop_split, pk["crop_split"])
        for l in (1, 2, 3):
            fill(w.sa_pre_split[l], pk["sa%d_pre_split" % (l + 1)])
        fill(w.fp_split[0], pk["fp1_split"])
        fill(w.fp_split[1], pk["fp2_split"])
        fill(w.approach_split, pk["approach_split"])
        fill(w.heads_split, pk["heads_split"])
        w.use_split = 1 if self.math_split else 0
        w.num_view = int(self.num_view)
        return w

    def _forward_native(self, end_points, xyz):
        """One gn_graspnet_forward call: outputs allocated here, scratch in one workspace tensor."""
        lib = L.lib()
        dev = xyz.device
        B, N = xyz.size(0), xyz.size(1)
        f32, i32 = torch.float32, torch.int32
        M, D, A, nv = SA_CFG[1][0], self.num_depth, self.num_angle, 2 + self.num_view

        def new(shape, dt=f32):
            return torch.empty(shape, dtype=dt, device=dev)

        o = L.Outputs()
        inds, xyzs, feats = [], [], []
        for li, (npoint, _, _, _, mlp) in enumerate(SA_CFG):
            inds.append(new((B, npoint), i32)); xyzs.append(new((B, npoint, 3))); feats.append(new((B, npoint, mlp[2])))
            o.sa_inds[li], o.sa_xyz[li], o.sa_features[li] = inds[li].data_ptr(), xyzs[li].data_ptr(), feats[li].data_ptr()
        seed_feat, feat302 = new((B, M, 256)), new((B, M, nv))
        top_inds, top_score = new((B, M), torch.int64), new((B, M))
        top_xyz, top_rot = new((B, M, 3)), new((B, M, 3, 3))
        vp, head = new((B, M * D, 256)), new((B, M, D, 4 * A))
        uidx, ucount = new((B, M, D * self.crop_nsample), i32), new((B, M), i32)
        ws = new((lib.gn_graspnet_workspace_bytes(B, N),), torch.uint8)
        o.fp2_features, o.approach, o.top_view_inds, o.top_view_score = (
            seed_feat.data_ptr(), feat302.data_ptr(), top_inds.data_ptr(), top_score.data_ptr())
        o.top_view_xyz, o.top_view_rot, o.vp_features, o.head = (
            top_xyz.data_ptr(), top_rot.data_ptr(), vp.data_ptr(), head.data_ptr())
        o.crop_uidx, o.crop_ucount = uidx.data_ptr(), ucount.data_ptr()
        o.grasps = o.counts = None
        o.fork_streams = 1 if self.fork else 0
        hook = _NO_HOOK
        if self.timing is not None:
            pending = {}

            def on_stage(name, phase, _user):
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                if phase == 0:
                    pending[name] = ev
                else:
                    self.timing.setdefault(name.decode(), []).append((pending.pop(name), ev))
            hook = L.STAGE_HOOK(on_stage)
        st = torch.cuda.current_stream(dev).cuda_stream
        self._cweights.use_split = 1 if self.math_split else 0
        L.check(lib.gn_graspnet_forward(ctypes.byref(self._cweights), xyz.data_ptr(), B, N, ws.data_ptr(),
                                        ctypes.byref(o), hook, None, st), "graspnet_forward")
        for li in range(4):
            name = "sa%d" % (li + 1)
            if li < 2:
                end_points[name + "_inds"] = inds[li]
            end_points[name + "_xyz"] = xyzs[li]
            end_points[name + "_features"] = feats[li].transpose(1, 2)
        end_points["fp2_features"] = seed_feat.transpose(1, 2)
        end_points["fp2_xyz"] = xyzs[1]
        end_points["fp2_inds"] = inds[0][:, 0:M]
        end_points["objectness_score"] = feat302[:, :, 0:2].transpose(1, 2)
        end_points["view_score"] = feat302[:, :, 2:]
        end_points["grasp_top_view_inds"] = top_inds
        end_points["grasp_top_view_score"] = top_score
        end_points["grasp_top_view_xyz"] = top_xyz
        end_points["grasp_top_view_rot"] = top_rot
        hv = head.permute(0, 3, 1, 2)
        end_points["grasp_score_pred"] = hv[:, 0:A]
        end_points["grasp_angle_cls_pred"] = hv[:, A:2 * A]
        end_points["grasp_width_pred"] = hv[:, 2 * A:3 * A]
        end_points["grasp_tolerance_pred"] = hv[:, 3 * A:4 * A]
        end_points["_gn"] = {"feat": feat302, "head": head, "model": self, "vp": vp, "uidx": uidx, "ucount": ucount}
        return end_points

    def __call__(self, end_points):
        return self.forward(end_points)

    # ---- the hot path -----------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, end_points):
        if self._packed is None:
            raise RuntimeError("GraspNet.forward: load_state_dict() first")
        pc = end_points["point_clouds"]
        if not pc.is_cuda:
            raise RuntimeError("CPU not supported")
        if pc.dtype != torch.float32 or pc.dim() != 3 or pc.size(-1) < 3:
            raise RuntimeError("point_clouds must be a (B,N,3) float tensor")
        lib = L.lib()
        lib.gn_set_point_tile_policy(1 if self.throughput else 0)
        dev = pc.device
        B, N = pc.size(0), pc.size(1)
        f32, i32 = torch.float32, torch.int32
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            xyz = pc[..., 0:3].contiguous()                      # backbone.py:69-75 _break_up_pc
            end_points["input_xyz"] = xyz
            end_points["input_features"] = None
            if self.native_forward and self.crop_unique and self.sa_hoist and self.query_grid and N >= SA_CFG[0][0]:
                return self._forward_native(end_points, xyz)

            # ---- 4 set-abstraction levels (backbone.py:105-121)
            lv_xyz, lv_feat, lv_n, lv_inds = xyz, None, N, []
            feats = []
            # SA1's sampler reports whether any arg-max was an exact tie between distinct points; if not, the
            # deeper levels' samplers (which run on FPS-ordered input) are the identity prefix and say so at once
            tie_flag = torch.empty((B,), dtype=i32, device=dev)
            grid = None
            if self.query_grid and N >= 4096:
                # one grid over the input cloud (cells >= SA1's ball radius) serves SA1's ball query and CloudCrop's
                # cylinder query; it only needs the coordinates, so it is built before anything else
                grid = torch.empty((lib.gn_grid_workspace_bytes(B, N),), dtype=torch.uint8, device=dev)
                self._run("grid_build", lambda: lib.gn_grid_build(xyz.data_ptr(), B, N, SA_CFG[0][1], grid.data_ptr(), st))
            for li, (npoint, radius, nsample, c, mlp) in enumerate(SA_CFG):
                inds = torch.empty((B, npoint), dtype=i32, device=dev)
                new_xyz = torch.empty((B, npoint, 3), dtype=f32, device=dev)
                wsb = lib.gn_fps_workspace_bytes(B, lv_n)
                ws = torch.empty((wsb,), dtype=torch.uint8, device=dev) if wsb else None
                self._run("furthest_point_sampling[%d]" % (li + 1), lambda: lib.gn_furthest_point_sampling(lv_xyz.data_ptr(), B, lv_n, npoint, L.ptr(ws), inds.data_ptr(),
                                                       new_xyz.data_ptr(), tie_flag.data_ptr() if li == 0 else None,
                                                       tie_flag.data_ptr() if li > 0 else None, st))
                idx = torch.empty((B, npoint, nsample), dtype=i32, device=dev)
                if li == 0 and grid is not None:
                    self._run("ball_query_grid[1]", lambda: lib.gn_ball_query_grid(
                        new_xyz.data_ptr(), grid.data_ptr(), B, lv_n, npoint, radius, nsample, idx.data_ptr(), st))
                else:
                    self._run("ball_query[%d]" % (li + 1), lambda: lib.gn_ball_query(
                        new_xyz.data_ptr(), lv_xyz.data_ptr(), B, lv_n, npoint, radius, nsample, idx.data_ptr(), st))
                out = torch.empty((B, npoint, mlp[2]), dtype=f32, device=dev)
                if c > 0 and self.sa_hoist:
                    # layer 1's feature part once per point, then gathered as the accumulator seed
                    pre, pk = self._packed["sa%d_pre" % (li + 1)], self._packed["sa%d_hoist" % (li + 1)]
                    pfeat = torch.empty((B, lv_n, mlp[0]), dtype=f32, device=dev)
                    if self.math_split:
                        pres = self._packed["sa%d_pre_split" % (li + 1)]
                        self._run("point_linear[%d]" % (li + 1), lambda: lib.gn_point_linear_split(
                            lv_feat.data_ptr(), B * lv_n, c, mlp[0], pres.w[0].data_ptr(), pres.b[0].data_ptr(),
                            pfeat.data_ptr(), st))
                    else:
                        self._run("point_linear[%d]" % (li + 1), lambda: lib.gn_point_linear(
                            lv_feat.data_ptr(), B * lv_n, c, mlp[0], pre.w[0].data_ptr(), pre.b[0].data_ptr(),
                            pfeat.data_ptr(), st))
                    if self.math_split:
                        pks = self._packed["sa%d_hoist_split" % (li + 1)]
                        self._run("sa_hoisted_mlp_pool[%d]" % (li + 1), lambda: lib.gn_sa_hoisted_mlp_pool_split(
                            lv_xyz.data_ptr(), pfeat.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), B, lv_n, npoint,
                            nsample, radius, _ci(*mlp), pks.wptr, pks.bptr, out.data_ptr(), st))
                    else:
                        self._run("sa_hoisted_mlp_pool[%d]" % (li + 1), lambda: lib.gn_sa_hoisted_mlp_pool(
                            lv_xyz.data_ptr(), pfeat.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), B, lv_n, npoint,
                            nsample, radius, _ci(*mlp), pk.wptr, pk.bptr, out.data_ptr(), st))
                elif c == 0 and self.math_split:
                    pks = self._packed["sa1_split"]
                    self._run("sa_group_mlp_pool[1]", lambda: lib.gn_sa_group_mlp_pool_split(
                        lv_xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), B, lv_n, npoint, nsample, radius, _ci(*mlp),
                        pks.wptr, pks.bptr, out.data_ptr(), st))
                else:
                    pk = self._packed["sa%d" % (li + 1)]
                    self._run("sa_group_mlp_pool[%d]" % (li + 1), lambda: lib.gn_sa_group_mlp_pool(
                        lv_xyz.data_ptr(), L.ptr(lv_feat), new_xyz.data_ptr(), idx.data_ptr(), B, lv_n, npoint, nsample, c,
                        radius, _ci(*mlp), pk.wptr, pk.bptr, out.data_ptr(), st))
                name = "sa%d" % (li + 1)
                if li < 2:
                    end_points[name + "_inds"] = inds
                end_points[name + "_xyz"] = new_xyz
                end_points[name + "_features"] = out.transpose(1, 2)       # (B,C,np) view
                lv_inds.append(inds)
                feats.append(out)
                lv_xyz, lv_feat, lv_n = new_xyz, out, npoint

            # ---- 2 feature-propagation levels (backbone.py:124-125)
            def fp(unknown_xyz, known_xyz, skip, known, key):
                n, m = unknown_xyz.size(1), known_xyz.size(1)
                d2 = torch.empty((B, n, 3), dtype=f32, device=dev)
                i3 = torch.empty((B, n, 3), dtype=i32, device=dev)
                self._run("three_nn[%s]" % key, lambda: lib.gn_three_nn(unknown_xyz.data_ptr(), known_xyz.data_ptr(), B, n, m, d2.data_ptr(),
                                        i3.data_ptr(), st))
                o = torch.empty((B, n, 256), dtype=f32, device=dev)
                pk = self._packed[key + "_split" if self.math_split else key]
                fn = lib.gn_fp_interp_mlp_split if self.math_split else lib.gn_fp_interp_mlp
                self._run("fp_interp_mlp[%s]" % key, lambda: fn(known.data_ptr(), skip.data_ptr(), d2.data_ptr(), i3.data_ptr(), B, n,
                                                                  m, 256, 256, _ci(256, 256), pk.wptr, pk.bptr, o.data_ptr(), st))
                return o

            g1 = fp(end_points["sa3_xyz"], end_points["sa4_xyz"], feats[2], feats[3], "fp1")
            seed_feat = fp(end_points["sa2_xyz"], end_points["sa3_xyz"], feats[1], g1, "fp2")
            seed_xyz = end_points["sa2_xyz"]
            M = seed_xyz.size(1)
            end_points["fp2_features"] = seed_feat.transpose(1, 2)
            end_points["fp2_xyz"] = seed_xyz
            end_points["fp2_inds"] = end_points["sa1_inds"][:, 0:M]             # backbone.py:129

            # ---- ApproachNet (modules.py:41-77)
            nv = 2 + self.num_view
            feat302 = torch.empty((B, M, nv), dtype=f32, device=dev)
            pk = self._packed["approach_split" if self.math_split else "approach"]
            fn_a = lib.gn_approach_mlp_split if self.math_split else lib.gn_approach_mlp
            self._run("approach_mlp", lambda: fn_a(seed_feat.data_ptr(), B * M, 256, nv, pk.wptr, pk.bptr, feat302.data_ptr(), st))
            views, view_rot, decode_rot = self._tables
            top_inds = torch.empty((B, M), dtype=torch.int64, device=dev)
            top_score = torch.empty((B, M), dtype=f32, device=dev)
            top_xyz = torch.empty((B, M, 3), dtype=f32, device=dev)
            top_rot = torch.empty((B, M, 3, 3), dtype=f32, device=dev)
            self._run("approach_post", lambda: lib.gn_approach_post(feat302.data_ptr(), nv, B * M, self.num_view, views.data_ptr(),
                                         view_rot.data_ptr(), top_inds.data_ptr(), top_score.data_ptr(),
                                         top_xyz.data_ptr(), top_rot.data_ptr(), st))
            end_points["objectness_score"] = feat302[:, :, 0:2].transpose(1, 2)      # (B,2,M)
            end_points["view_score"] = feat302[:, :, 2:]                             # (B,M,300)
            end_points["grasp_top_view_inds"] = top_inds
            end_points["grasp_top_view_score"] = top_score
            end_points["grasp_top_view_xyz"] = top_xyz
            end_points["grasp_top_view_rot"] = top_rot

            # ---- CloudCrop (modules.py:109-141), inference branch of GraspNetStage2 (graspnet.py:53-57)
            D, NS = self.num_depth, self.crop_nsample
            vp = torch.empty((B, M * D, 256), dtype=f32, device=dev)
            pk = self._packed["crop"]
            if self.crop_unique:
                # the 4 nested cylinders as one de-duplicated row list per seed (exact, ~2.4x fewer MFMAs)
                uidx = torch.empty((B, M, D * NS), dtype=i32, device=dev)
                ucount = torch.empty((B, M), dtype=i32, device=dev)
                if grid is not None:
                    self._run("cylinder_query_unique_grid", lambda: lib.gn_cylinder_query_unique_grid(
                        seed_xyz.data_ptr(), top_rot.data_ptr(), grid.data_ptr(), B, N, M, self.cylinder_radius, self.hmin,
                        self.hmax, D, NS, uidx.data_ptr(), ucount.data_ptr(), st))
                else:
                    self._run("cylinder_query_unique", lambda: lib.gn_cylinder_query_unique(
                        seed_xyz.data_ptr(), xyz.data_ptr(), top_rot.data_ptr(), B, N, M, self.cylinder_radius, self.hmin,
                        self.hmax, D, NS, uidx.data_ptr(), ucount.data_ptr(), st))
                if self.math_split:
                    pks = self._packed["crop_split"]
                    self._run("crop_unique_mlp_pool", lambda: lib.gn_crop_unique_mlp_pool_split(
                        xyz.data_ptr(), seed_xyz.data_ptr(), top_rot.data_ptr(), uidx.data_ptr(), ucount.data_ptr(), B, N,
                        M, D, NS, _ci(64, 128, 256), pks.wptr, pks.bptr, vp.data_ptr(), st))
                else:
                    self._run("crop_unique_mlp_pool", lambda: lib.gn_crop_unique_mlp_pool(
                        xyz.data_ptr(), seed_xyz.data_ptr(), top_rot.data_ptr(), uidx.data_ptr(), ucount.data_ptr(), B, N,
                        M, D, NS, _ci(64, 128, 256), pk.wptr, pk.bptr, vp.data_ptr(), st))
                crop_priv = {"uidx": uidx, "ucount": ucount}
            else:
                cyl_idx = torch.empty((B, M, D, NS), dtype=i32, device=dev)
                self._run("cylinder_query_multi", lambda: lib.gn_cylinder_query_multi(
                    seed_xyz.data_ptr(), xyz.data_ptr(), top_rot.data_ptr(), B, N, M, self.cylinder_radius, self.hmin,
                    self.hmax, D, NS, cyl_idx.data_ptr(), st))
                self._run("crop_group_mlp_pool", lambda: lib.gn_crop_group_mlp_pool(
                    xyz.data_ptr(), seed_xyz.data_ptr(), top_rot.data_ptr(), cyl_idx.data_ptr(), B, N, M, D, NS,
                    _ci(64, 128, 256), pk.wptr, pk.bptr, vp.data_ptr(), st))
                crop_priv = {"cyl_idx": cyl_idx}

            # ---- OperationNet + ToleranceNet (modules.py:169-191, 214-232)
            A = self.num_angle
            head = torch.empty((B, M, D, 4 * A), dtype=f32, device=dev)
            pk = self._packed["heads_split" if self.math_split else "heads"]
            fn_h = lib.gn_heads_mlp_split if self.math_split else lib.gn_heads_mlp
            self._run("heads_mlp", lambda: fn_h(vp.data_ptr(), B * M * D, 256, 4 * A, pk.wptr, pk.bptr, head.data_ptr(), st))
            hv = head.permute(0, 3, 1, 2)                                            # (B,48,M,D) view
            end_points["grasp_score_pred"] = hv[:, 0:A]
            end_points["grasp_angle_cls_pred"] = hv[:, A:2 * A]
            end_points["grasp_width_pred"] = hv[:, 2 * A:3 * A]
            end_points["grasp_tolerance_pred"] = hv[:, 3 * A:4 * A]
            # private handles for the fused decode / tests
            end_points["_gn"] = dict({"feat": feat302, "head": head, "model": self, "vp": vp}, **crop_priv)
        return end_points

    @torch.no_grad()
    def decode_padded(self, end_points):
        """graspnet.py:77-134 for the whole batch, no host sync:
        -> grasps (B,M,17) f32 padded with zero rows, counts (B,) i32."""
        lib = L.lib()
        priv = end_points.get("_gn")
        if priv is not None:
            feat, head = priv["feat"], priv["head"]
        else:   # end_points produced elsewhere (e.g. the reference model on the drop-in operators)
            feat = end_points["objectness_score"].transpose(1, 2).contiguous()      # (B,M,2)
            head = torch.cat([end_points["grasp_score_pred"], end_points["grasp_angle_cls_pred"],
                              end_points["grasp_width_pred"], end_points["grasp_tolerance_pred"]], dim=1)
            head = head.permute(0, 2, 3, 1).contiguous()
        top_inds = end_points["grasp_top_view_inds"].contiguous()
        seed_xyz = end_points["fp2_xyz"].contiguous()
        B, M = seed_xyz.size(0), seed_xyz.size(1)
        dev = seed_xyz.device
        grasps = torch.empty((B, M, GRASP_ROW), dtype=torch.float32, device=dev)
        counts = torch.empty((B,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            self._run("pred_decode", lambda: lib.gn_pred_decode(feat.data_ptr(), feat.size(2), head.data_ptr(), top_inds.data_ptr(),
                                       seed_xyz.data_ptr(), self._tables[2].data_ptr(), B, M, self.num_depth,
                                       self.num_angle, grasps.data_ptr(), counts.data_ptr(), st))
        return grasps, counts


class GraphedGraspNet:
    """The whole forward + decode captured once as a hipGraph (torch.cuda.CUDAGraph) for a fixed (B,N):
    a replay costs ~20 us of host time instead of ~25 launches.  Inputs are copied into a static buffer;
    the returned end_points / grasps / counts are the graph's static output tensors (overwritten by the
    next call).  Nothing in the C ABI allocates or synchronises, which is what makes the capture legal."""

    def __init__(self, net, batch, points, warmup=2, fork_streams=True, post=None, scene_points=None, front=None):
        """post: None, or a dict of postprocess.grasp_pipeline() keyword arguments -- the graph then also holds the steps every
        caller of the reference runs on the host after pred_decode (voxel down-sample of each scene, collision filter, sort +
        NMS: test.py:88-96, demo_pointcloud.py:228-249) and a call returns
        (end_points, grasps, counts, final_rows (B,M,17), kept (B,), collision_mask (B,M), voxel_status (B,)).
        scene_points: with post, the graph checks the grasps against a SEPARATE scene cloud of up to that many points per
        scene -- the reference's callers use the raw masked cloud there (test.py:92 return_raw_cloud=True, demo.py:481), several
        times larger than the `points` sampled for the network; pass it to __call__(clouds, scene=, scene_n_valid=).  Without
        scene_points the sampled input cloud itself is the scene (fewer occupied voxels, lower IoUs: NOT the reference's check).
        front: None, or dict(height=, width=, intrinsics=(fx, fy, cx, cy, factor_depth)[, depth_min=, depth_max=,
        scene_points=]) -- the graph then starts from the IMAGES: __call__(depth (B,H,W) u16, mask, seed=) runs
        postprocess.image_to_grasps (valid-pixel compaction, the draw without replacement, cloud, forward, decode and, with post,
        the collision filter against the raw masked cloud) as ONE replay and returns its dict."""
        from . import postprocess
        self.net = net
        self.post = None if post is None else dict(post)
        self.front = None if front is None else dict(front)
        dev = net.device
        self.static_scene = self.static_scene_n = None
        if self.front is not None:
            h, w = int(self.front["height"]), int(self.front["width"])
            self.static_depth = torch.full((batch, h, w), 600, dtype=torch.int16, device=dev)      # 16-bit words (uint16 view)
            self.static_mask = torch.ones((batch, h, w), dtype=torch.uint8, device=dev)
            self.static_keys = postprocess.draw_keys(0, batch, dev)
            self.static_in = None
        else:
            self.static_in = torch.zeros((batch, points, 3), dtype=torch.float32, device=dev)
            # a cloud of all-equal points is a legal input but not a typical one: warm up on spread-out coordinates
            self.static_in.uniform_(-0.3, 0.3)
            self.static_in[..., 2] += 0.5
            if self.post is not None and scene_points is not None:
                self.static_scene = torch.zeros((batch, int(scene_points), 3), dtype=torch.float32, device=dev)
                n0 = min(points, int(scene_points))
                self.static_scene[:, :n0] = self.static_in[:, :n0]
                self.static_scene_n = torch.full((batch,), n0, dtype=torch.int32, device=dev)

        def run():
            if self.front is not None:
                f = self.front
                return postprocess.image_to_grasps(net, self.static_depth, self.static_mask, f["intrinsics"], num_point=points,
                                                   keys=self.static_keys, post=self.post, scene_points=f.get("scene_points"),
                                                   depth_min=f.get("depth_min", 0.0), depth_max=f.get("depth_max", float("inf")))
            ep = net({"point_clouds": self.static_in})
            g, c = net.decode_padded(ep)
            tail = ()
            if self.post is not None:
                if self.static_scene is not None:
                    tail = postprocess.grasp_pipeline(self.static_scene, g, c, n_valid=self.static_scene_n, return_status=True,
                                                      **self.post)
                else:
                    tail = postprocess.grasp_pipeline(self.static_in, g, c, return_status=True, **self.post)
            return (ep, g, c) + tuple(tail)

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):       # first calls set kernel attributes / fill allocator pools
                run()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        fork, net.fork = net.fork, fork_streams      # static edges inside the graph: the side stream is a steady gain here
        try:
            with torch.cuda.graph(self.graph):
                self.outputs = run()
        finally:
            net.fork = fork
        if self.front is None:
            self.end_points, self.grasps, self.counts = self.outputs[:3]
        else:
            self.end_points, self.grasps, self.counts = (self.outputs[k] for k in ("end_points", "grasps", "counts"))

    def __call__(self, point_clouds, mask=None, seed=None, keys=None, scene=None, scene_n_valid=None):
        """clouds (B,N,3) [, scene=(B,n<=scene_points,3), scene_n_valid=(B,)]; or, built with front=: depth (B,H,W) 16-bit,
        mask (B,H,W) uint8 / bool or None (= every pixel), seed or keys (B,4) for the draw (default: the previous keys)."""
        if self.front is not None:
            from . import postprocess
            d = point_clouds
            self.static_depth.copy_(d.view(torch.int16) if d.dtype == torch.uint16 else d, non_blocking=True)
            if mask is None:
                self.static_mask.fill_(1)
            else:
                self.static_mask.copy_(mask.view(torch.uint8) if mask.dtype == torch.bool else mask, non_blocking=True)
            if keys is None and seed is not None:
                keys = postprocess.draw_keys(seed, self.static_keys.shape[0])
            if keys is not None:
                self.static_keys.copy_(keys.view(torch.int32) if keys.dtype == torch.uint32 else keys, non_blocking=True)
            self.graph.replay()
            return self.outputs
        self.static_in.copy_(point_clouds[..., 0:3], non_blocking=True)
        if self.static_scene is not None:
            if scene is None:       # no separate scene given: the sampled cloud (as without scene_points)
                scene, scene_n_valid = point_clouds[..., 0:3], None
            n = min(int(scene.shape[1]), int(self.static_scene.shape[1]))
            self.static_scene[:, :n].copy_(scene[:, :n, 0:3], non_blocking=True)
            if scene_n_valid is None:
                self.static_scene_n.fill_(n)
            else:
                self.static_scene_n.copy_(torch.clamp(scene_n_valid.to(torch.int32), max=n), non_blocking=True)
        self.graph.replay()
        return self.outputs


class PipelinedGraspNet:
    """Throughput for a caller that simply loops over batches.

    One forward keeps 8 CUs busy for 2.2 ms with the serial FPS chain of its batch and the whole chip for ~1.1 ms with
    everything else, so a loop `for x in batches: net(x)` on ONE stream gets ~2 200 scenes/s of the ~7 000 the card
    delivers when several batches are in flight (bench.py: `single_stream_scenes_per_s` vs `value`).  This wrapper issues
    successive batches round-robin on `depth` HIP streams and hands back a ticket per batch:

        pipe = PipelinedGraspNet(net, depth=12)
        tickets = [pipe.submit(x) for x in batches]          # returns at once; batch i+1's FPS overlaps batch i's MLPs
        for t in tickets: end_points, grasps, counts = t.result()

    `result()` waits for that batch only.  ROCm maps HIP streams onto 4 hardware queues unless GPU_MAX_HW_QUEUES is set
    (16 for depth 12) BEFORE the HIP runtime initialises; with fewer queues batches share a queue and serialise.
    Outputs of a ticket are ordinary torch tensors allocated on the ticket's stream; `wait()` (device-side dependency) and
    `result()` (host-side wait) both register them with the caller's current stream (`record_stream`), after which they
    can be used -- and dropped -- like any other tensor."""

    class Ticket:
        def __init__(self, end_points, grasps, counts, event, stream):
            self.end_points, self.grasps, self.counts, self.event, self.stream = end_points, grasps, counts, event, stream

        final_rows = kept = collision_mask = voxel_status = None      # set when the pipeline was built with post=...

        def _tensors(self):
            priv = self.end_points.get("_gn") or {}
            vals = list(self.end_points.values()) + list(priv.values()) + [self.final_rows, self.kept, self.collision_mask,
                                                                                self.voxel_status]
            return [self.grasps, self.counts] + [v for v in vals if torch.is_tensor(v) and v.is_cuda]

        def _register(self, cur):
            # the outputs were allocated on the ticket's stream: tell the caching allocator that `cur` reads them too, or
            # a tensor dropped right after an async kernel on `cur` returns its block to the ticket stream's pool at once
            # and the next submit() on that stream may overwrite it under the kernel
            if cur != self.stream:
                for t in self._tensors():
                    t.record_stream(cur)

        def wait(self):
            """make the caller's current stream depend on this batch (no host sync)"""
            cur = torch.cuda.current_stream(self.grasps.device)
            cur.wait_event(self.event)
            self._register(cur)
            return self

        def result(self):
            """block the host until this batch is done; the outputs are then safe to use on the current stream"""
            self.event.synchronize()
            self._register(torch.cuda.current_stream(self.grasps.device))
            return self.end_points, self.grasps, self.counts

    def __init__(self, net, depth=12, post=None, throughput_tiles=True):
        """post: None, or a dict of postprocess.grasp_pipeline() keyword arguments: every submitted batch then also goes through
        the collision filter + sort + NMS on its stream (no host read), and the ticket carries `final_rows` (B,M,17), `kept` (B,),
        `collision_mask` (B,M) and `voxel_status` (B,) beside the raw grasps.  submit(clouds, scene=, scene_n_valid=) checks the
        grasps against a separate scene cloud -- the reference's callers use the raw masked cloud (test.py:92, demo.py:481), not
        the points sampled for the network; without it the sampled cloud is the scene.
        throughput_tiles: run the per-point chains in their throughput tile shapes (GraspNet.throughput) while submitting."""
        self.net = net
        self.throughput_tiles = bool(throughput_tiles) and int(depth) > 1
        self.post = None if post is None else dict(post)
        self.streams = [torch.cuda.Stream(device=net.device) for _ in range(max(1, int(depth)))]
        self._i = 0

    def submit(self, point_clouds, scene=None, scene_n_valid=None):
        st = self.streams[self._i % len(self.streams)]
        self._i += 1
        st.wait_stream(torch.cuda.current_stream(point_clouds.device))     # the input may still be in flight there
        with torch.cuda.stream(st):
            keep, self.net.throughput = self.net.throughput, (self.net.throughput or self.throughput_tiles)
            try:
                ep = self.net({"point_clouds": point_clouds})
            finally:
                self.net.throughput = keep
            grasps, counts = self.net.decode_padded(ep)
            tail = None
            if self.post is not None:
                from . import postprocess
                tail = postprocess.grasp_pipeline(point_clouds[..., 0:3] if scene is None else scene, grasps, counts,
                                                  n_valid=None if scene is None else scene_n_valid, return_status=True, **self.post)
            ev = torch.cuda.Event()
            ev.record(st)
        point_clouds.record_stream(st)
        for t_in in (scene, scene_n_valid):
            if t_in is not None:
                t_in.record_stream(st)
        t = PipelinedGraspNet.Ticket(ep, grasps, counts, ev, st)
        if tail is not None:
            t.final_rows, t.kept, t.collision_mask, t.voxel_status = tail
        return t


_NO_HOOK = L.STAGE_HOOK()      # NULL function pointer


def load_checkpoint(path, device="cuda", **model_kwargs):
    """get_net() of the reference's callers in one line (demo_pointcloud.py:40-53): build the model, load the checkpoint file
    (GraspNet.load_checkpoint), eval mode.  -> (net, epoch)"""
    net = GraspNet(is_training=False, device=device, **model_kwargs)
    epoch = net.load_checkpoint(path)
    return net.eval(), epoch


def _expected_keys():
    from . import synthetic
    return set(synthetic.make_state_dict(0, calibrated=False).keys())


_DEFAULT_DECODERS = {}     # device -> GraspNet holding only the decode tables (end_points produced elsewhere)


def pred_decode(end_points):
    """Drop-in for models/graspnet.py:77 pred_decode: list of B tensors (Ns_i, 17)."""
    priv = end_points.get("_gn")
    if priv is not None:
        model = priv["model"]
    else:
        dev = end_points["fp2_xyz"].device
        nview = int(end_points["view_score"].shape[-1]) if "view_score" in end_points else 300
        model = _DEFAULT_DECODERS.get((dev, nview))
        if model is None:
            model = GraspNet()
            model.device = dev
            model.num_view = int(end_points["view_score"].shape[-1]) if "view_score" in end_points else 300
            v, vr, dr = geometry.build_tables(model.num_view, 12)
            model._tables = (v.to(dev), vr.to(dev), dr.to(dev))
            _DEFAULT_DECODERS[(dev, nview)] = model
    grasps, counts = model.decode_padded(end_points)
    cnt = counts.cpu().tolist()          # the only host sync; the reference's callers sync here too (test.py:88)
    return [grasps[i, :c] for i, c in enumerate(cnt)]
