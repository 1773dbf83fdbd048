"""GPU: the split-bf16 matrix-core layers (csrc/mlp_split.h) against float64 and against the fp32 MFMA path.

Each fp32 operand is split exactly into three bf16 terms and six of the nine products are accumulated in fp32; the
claim tested here is that this is AS ACCURATE as fp32 arithmetic: error against a float64 evaluation of the same
layers no larger than the error of the fp32 CPU oracle (and of the fp32 MFMA kernel)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def clouds():
    from graspnet_amd import synthetic as S
    return S.make_batch(1000, 2, 20000, "uniform")


@pytest.fixture(scope="module")
def ref(model_oracle, state_dict, clouds):
    return model_oracle.forward({"point_clouds": clouds.clone()}, state_dict, keep_internal=True)


@pytest.fixture(scope="module")
def truth(model_oracle, state_dict, clouds, ref):
    return model_oracle.forward({"point_clouds": clouds.clone()}, state_dict, f64=True, keep_internal=True,
                                force_top_view_inds=ref["grasp_top_view_inds"])


@pytest.fixture(scope="module")
def net(state_dict):
    from graspnet_amd import GraspNet
    return GraspNet(is_training=False).load_state_dict(state_dict)


def _errs(x, t):
    e = (x.double().cpu() - t.double()).abs()
    return float(e.max()), float(e.pow(2).mean().sqrt())


def test_split_weights_are_exact(net):
    from graspnet_amd import weights as W
    w = torch.randn(256, 128) * torch.logspace(-4, 2, 128)[None, :]
    hi, mid, lo = W.split3(w)
    back = hi.double() + mid.double() + lo.double()
    assert float(((back - w.double()).abs() / w.double().abs().clamp_min(1e-30)).max()) <= 2.0 ** -22


def test_crop_split_matches_float64_as_well_as_fp32(ref, truth, net):
    from graspnet_amd import _lib as L
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, N, M = 2, 20000, 1024
    xyz = ref["input_xyz"].cuda()
    seed = ref["fp2_xyz"].cuda()
    rot = ref["grasp_top_view_rot"].reshape(B, M, 9).contiguous().cuda()
    uidx = torch.empty((B, M, 256), dtype=torch.int32, device="cuda")
    ucount = torch.empty((B, M), dtype=torch.int32, device="cuda")
    hmax = (ctypes.c_float * 4)(0.01, 0.02, 0.03, 0.04)
    L.check(lib.gn_cylinder_query_unique(seed.data_ptr(), xyz.data_ptr(), rot.data_ptr(), B, N, M, 0.05, -0.02, hmax, 4, 64,
                                         uidx.data_ptr(), ucount.data_ptr(), st), "cyl")
    mlp = (ctypes.c_int * 3)(64, 128, 256)
    v32 = torch.empty((B, M * 4, 256), device="cuda")
    vsp = torch.empty((B, M * 4, 256), device="cuda")
    pk, pks = net._packed["crop"], net._packed["crop_split"]
    L.check(lib.gn_crop_unique_mlp_pool(xyz.data_ptr(), seed.data_ptr(), rot.data_ptr(), uidx.data_ptr(), ucount.data_ptr(),
                                        B, N, M, 4, 64, mlp, pk.wptr, pk.bptr, v32.data_ptr(), st), "crop fp32")
    L.check(lib.gn_crop_unique_mlp_pool_split(xyz.data_ptr(), seed.data_ptr(), rot.data_ptr(), uidx.data_ptr(),
                                              ucount.data_ptr(), B, N, M, 4, 64, mlp, pks.wptr, pks.bptr, vsp.data_ptr(), st),
            "crop split")
    torch.cuda.synchronize()
    want = ref["_vp_features"].reshape(B, 256, M * 4).transpose(1, 2)
    tru = truth["_vp_features"].reshape(B, 256, M * 4).transpose(1, 2)
    e_or, e_32, e_sp = _errs(want, tru), _errs(v32, tru), _errs(vsp, tru)
    print("crop features vs float64 (max, rms): oracle fp32 %s  fp32 MFMA %s  split bf16 %s" % (e_or, e_32, e_sp))
    assert e_sp[0] <= 2.0 * max(e_or[0], e_32[0]) + 1e-6
    assert e_sp[1] <= 2.0 * max(e_or[1], e_32[1]) + 1e-7
    assert float((vsp - v32).abs().max()) <= 1e-4


def test_crop_packed_rows_equal_padded_tiles(ref, net, monkeypatch):
    """the packed-row CloudCrop kernel (a workgroup's seeds as one row stream, no per-seed padding, mixed tiles pooled once
    per seed) == the per-seed padded kernel, bit for bit -- incl. seeds without neighbours (one zero-row entry each, so
    several seeds share one 32-row tile) and a batch whose last workgroup is ragged"""
    from graspnet_amd import _lib as L
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, N, M = 2, 20000, 1024
    xyz = ref["input_xyz"].cuda()
    seed = ref["fp2_xyz"].clone().cuda()
    seed[0, 100:140] += 5.0              # 40 consecutive seeds far from the cloud: ucount 1 each
    seed[1, 1000:] += 5.0                # ... and the tail of the second cloud
    rot = ref["grasp_top_view_rot"].reshape(B, M, 9).contiguous().cuda()
    uidx = torch.empty((B, M, 256), dtype=torch.int32, device="cuda")
    ucount = torch.empty((B, M), dtype=torch.int32, device="cuda")
    hmax = (ctypes.c_float * 4)(0.01, 0.02, 0.03, 0.04)
    L.check(lib.gn_cylinder_query_unique(seed.data_ptr(), xyz.data_ptr(), rot.data_ptr(), B, N, M, 0.05, -0.02, hmax, 4, 64,
                                         uidx.data_ptr(), ucount.data_ptr(), st), "cyl")
    assert int(ucount[0, 100:140].max()) == 1
    mlp = (ctypes.c_int * 3)(64, 128, 256)
    pks = net._packed["crop_split"]
    outs = {}
    for mode, nseeds_b in (("1", B), ("0", B)):
        monkeypatch.setenv("GN_CROP_PACKED", mode)
        v = torch.full((B, M * 4, 256), float("nan"), device="cuda")
        L.check(lib.gn_crop_unique_mlp_pool_split(xyz.data_ptr(), seed.data_ptr(), rot.data_ptr(), uidx.data_ptr(),
                                                  ucount.data_ptr(), nseeds_b, N, M, 4, 64, mlp, pks.wptr, pks.bptr, v.data_ptr(), st),
                "crop split packed=" + mode)
        torch.cuda.synchronize()
        outs[mode] = v
    assert not torch.isnan(outs["1"]).any()
    assert torch.equal(outs["1"], outs["0"])
    # a ragged seed count (b*m not a multiple of the seeds per workgroup): m = 1000 seeds of cloud 0 only
    res = []
    for mode in ("1", "0"):
        monkeypatch.setenv("GN_CROP_PACKED", mode)
        v = torch.full((1, 1000 * 4, 256), float("nan"), device="cuda")
        L.check(lib.gn_crop_unique_mlp_pool_split(xyz.data_ptr(), seed.data_ptr(), rot.data_ptr(), uidx.data_ptr(),
                                                  ucount.data_ptr(), 1, N, 1000, 4, 64, mlp, pks.wptr, pks.bptr, v.data_ptr(), st),
                "crop split ragged packed=" + mode)
        torch.cuda.synchronize()
        res.append(v)
    # (with m = 1000 the kernel reads seed / rot / uidx rows 0..999 of the same buffers: row stride is per seed)
    assert not torch.isnan(res[0]).any() and torch.equal(res[0], res[1])
    # the 8-wave form (two waves per SIMD, one n-tile of each layer resident per wave): the same rows through the same MFMAs
    # in the same k order -> the same bits, on the full batch, the ragged seed count and the 3-seeds-per-workgroup case
    monkeypatch.setenv("GN_CROP_PACKED", "1")
    for waves in ("8", "4"):
        monkeypatch.setenv("GN_CROP_WAVES", waves)
        for (bb, mm, want) in ((B, M, outs["1"]), (1, 1000, res[0])):
            v = torch.full((bb, mm * 4, 256), float("nan"), device="cuda")
            L.check(lib.gn_crop_unique_mlp_pool_split(xyz.data_ptr(), seed.data_ptr(), rot.data_ptr(), uidx.data_ptr(),
                                                      ucount.data_ptr(), bb, N, mm, 4, 64, mlp, pks.wptr, pks.bptr, v.data_ptr(), st),
                    "crop split waves=" + waves)
            torch.cuda.synchronize()
            assert torch.equal(v, want), (waves, bb, mm)
    monkeypatch.delenv("GN_CROP_WAVES")
    # 2047 seeds as ONE cloud: 3 seeds per workgroup, the last workgroup holds a single seed
    res = []
    for mode in ("1", "0"):
        monkeypatch.setenv("GN_CROP_PACKED", mode)
        v = torch.full((1, 2047 * 4, 256), float("nan"), device="cuda")
        L.check(lib.gn_crop_unique_mlp_pool_split(xyz.data_ptr(), seed.data_ptr(), rot.data_ptr(), uidx.data_ptr(),
                                                  ucount.data_ptr(), 1, N, 2047, 4, 64, mlp, pks.wptr, pks.bptr, v.data_ptr(), st),
                "crop split 2047 seeds packed=" + mode)
        torch.cuda.synchronize()
        res.append(v)
    assert not torch.isnan(res[0]).any() and torch.equal(res[0], res[1])


def test_heads_and_approach_split_on_oracle_inputs(ref, truth, net):
    """per-point chains on the matrix-core path, fed with the ORACLE's inputs: within 1e-4 of the oracle's raw scores
    and as close to float64 as the fp32 kernels"""
    from graspnet_amd import _lib as L
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    B, M = 2, 1024
    x = ref["_vp_features"].reshape(B, 256, M * 4).transpose(1, 2).contiguous().cuda()
    outs = {}
    for name, fn, pk in (("f32", lib.gn_heads_mlp, net._packed["heads"]), ("split", lib.gn_heads_mlp_split, net._packed["heads_split"])):
        head = torch.empty((B, M * 4, 48), device="cuda")
        L.check(fn(x.data_ptr(), B * M * 4, 256, 48, pk.wptr, pk.bptr, head.data_ptr(), st), "heads " + name)
        outs[name] = head.reshape(B, M, 4, 48).permute(0, 3, 1, 2)
    torch.cuda.synchronize()
    for lo, key in ((0, "grasp_score_pred"), (12, "grasp_angle_cls_pred"), (24, "grasp_width_pred"), (36, "grasp_tolerance_pred")):
        want = ref[key]
        d_sp = float((outs["split"][:, lo:lo + 12].cpu() - want).abs().max())
        d_32 = float((outs["f32"][:, lo:lo + 12].cpu() - want).abs().max())
        scale = float(want.abs().max())
        assert d_sp <= 1e-4 + 2e-5 * scale, (key, d_sp, d_32, scale)

    seed_feat = ref["fp2_features"].transpose(1, 2).contiguous().cuda()          # (B,M,256)
    got = {}
    for name, fn, pk in (("f32", lib.gn_approach_mlp, net._packed["approach"]),
                         ("split", lib.gn_approach_mlp_split, net._packed["approach_split"])):
        o = torch.empty((B, M, 302), device="cuda")
        L.check(fn(seed_feat.data_ptr(), B * M, 256, 302, pk.wptr, pk.bptr, o.data_ptr(), st), "approach " + name)
        got[name] = o
    torch.cuda.synchronize()
    want = ref["view_score"]
    tru = truth["view_score"]
    e_or = _errs(want, tru)
    # truth's view_score follows truth's own (float64) seed features, so compare on equal inputs: oracle vs kernels
    for name in ("f32", "split"):
        d = float((got[name][:, :, 2:].cpu() - want).abs().max())
        assert d <= 1e-4 + 2e-5 * float(want.abs().max()), (name, d)
    assert float((got["split"] - got["f32"]).abs().max()) <= 1e-4 + 2e-5 * float(want.abs().max())
    assert e_or[0] >= 0.0


@pytest.mark.parametrize("rows", [8192, 1000, 33, 32768, 4133])
def test_point_chain_tile_shapes_are_bit_identical(net, monkeypatch, rows):
    """the per-point chains in their tile shapes (GN_POINT_WPC=1: 64-row tiles, one workgroup per CU; 2: 32-row tiles,
    two per CU, idle waves in the 64-wide last layer of the heads; GN_POINT_WIDE=1: the heads on 128-row fp32 tiles split at
    fragment load, mlp_point_wide.hip -- the default from 24 576 rows): the k order of every dot product and every bf16 cut
    are the same, so the outputs must be identical -- ragged last tiles included"""
    from graspnet_amd import _lib as L
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, 256, generator=g).cuda()
    for fn, pk, nout in ((lib.gn_heads_mlp_split, net._packed["heads_split"], 48),
                         (lib.gn_approach_mlp_split, net._packed["approach_split"], 302)):
        res = []
        for wpc, dense, wide in (("1", "0", "0"), ("2", "0", "0"), ("1", "1", "0"), ("2", "1", "0"), ("1", "0", "1")):
            # GN_HEADS_DENSE=1: the heads' block-diagonal layer 2 multiplies its zero blocks too (adds exact zeros)
            monkeypatch.setenv("GN_POINT_WPC", wpc)
            monkeypatch.setenv("GN_HEADS_DENSE", dense)
            monkeypatch.setenv("GN_POINT_WIDE", wide)
            o = torch.full((rows + 1, nout), float("nan"), device="cuda")
            L.check(fn(x.data_ptr(), rows, 256, nout, pk.wptr, pk.bptr, o.data_ptr(), st), "chain")
            torch.cuda.synchronize()
            assert torch.isnan(o[rows]).all() and not torch.isnan(o[:rows]).any()     # nothing beyond the last row
            res.append(o[:rows])
        assert all(torch.equal(res[0], r) for r in res[1:])
    monkeypatch.delenv("GN_POINT_WPC"); monkeypatch.delenv("GN_HEADS_DENSE"); monkeypatch.delenv("GN_POINT_WIDE")
    # the feature-propagation chain (gathered + interpolated + skip rows, K = 512) on 32-row plane tiles against 64-row
    # fp32 tiles, and the policy switch (gn_set_point_tile_policy / GN_POINT_TILES) that picks the forms
    import ctypes
    b, n, m = 2, max(rows // 2, 3), 17
    known, skip = torch.randn(b, m, 256, generator=g).cuda(), torch.randn(b, n, 256, generator=g).cuda()
    d2 = (torch.rand(b, n, 3, generator=g) * 0.01).cuda()
    i3 = torch.randint(0, m, (b, n, 3), generator=g, dtype=torch.int32).cuda()
    pk = net._packed["fp2_split"]
    mlp = (ctypes.c_int * 2)(256, 256)
    res = []
    for tiles in ("latency", "throughput"):
        monkeypatch.setenv("GN_POINT_TILES", tiles)
        o = torch.full((b * n + 1, 256), float("nan"), device="cuda")
        L.check(lib.gn_fp_interp_mlp_split(known.data_ptr(), skip.data_ptr(), d2.data_ptr(), i3.data_ptr(), b, n, m, 256, 256, mlp,
                                           pk.wptr, pk.bptr, o.data_ptr(), st), "fp")
        torch.cuda.synchronize()
        assert torch.isnan(o[b * n]).all() and not torch.isnan(o[:b * n]).any()
        res.append(o[:b * n])
    assert torch.equal(res[0], res[1])
    monkeypatch.delenv("GN_POINT_TILES")
    prev = lib.gn_set_point_tile_policy(1)
    try:
        o = torch.empty((b * n, 256), device="cuda")
        L.check(lib.gn_fp_interp_mlp_split(known.data_ptr(), skip.data_ptr(), d2.data_ptr(), i3.data_ptr(), b, n, m, 256, 256, mlp,
                                           pk.wptr, pk.bptr, o.data_ptr(), st), "fp")
        assert lib.gn_set_point_tile_policy(prev) == 1 and torch.equal(o, res[0])
    finally:
        lib.gn_set_point_tile_policy(0)


def test_sa_split_equals_fp32_kernels_to_rounding(net, clouds):
    """set-abstraction levels: split path vs fp32 MFMA path on the same clouds (identical sampling and neighbour lists)"""
    from graspnet_amd import GraspNet
    f32 = GraspNet().load_state_dict(net.state_dict())
    f32.math_split = False
    sp = GraspNet().load_state_dict(net.state_dict())
    sp.math_split = True
    a = f32({"point_clouds": clouds.cuda()})
    b = sp({"point_clouds": clouds.cuda()})
    for k in ("sa1_inds", "sa2_inds", "sa1_xyz", "sa4_xyz"):
        assert torch.equal(a[k], b[k]), k
    for k, tol in (("sa1_features", 2e-5), ("sa2_features", 5e-5), ("sa3_features", 1e-4), ("sa4_features", 1e-4),
                   ("fp2_features", 2e-4)):
        d = float((a[k] - b[k]).abs().max())
        assert d <= tol * max(1.0, float(a[k].abs().max())), (k, d, float(a[k].abs().max()))
