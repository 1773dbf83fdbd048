"""GPU: drop-in MODE end to end -- the reference's UNFUSED python composition (tests/dropin_harness.py, ~30 `_ext` calls per
forward in the reference's order: pointnet2_utils.py:26 + pointnet2_modules.py:210-272, modules.py:41-232) running on DEVICE
tensors over graspnet_amd.pointnet2_ext, against the CPU oracle on the same seeded inputs: every index output bit-exact,
floats under THE RULE (the layers between the operators are torch-on-GPU matmuls here, torch-CPU convolutions there), and
graspnet_amd.pred_decode on end_points it did not produce."""
import pytest
import torch

import dropin_harness as H
import parity_util as PU

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def run(model_oracle, state_dict):
    from graspnet_amd import pointnet2_ext as E, synthetic as S
    cl = torch.stack([torch.from_numpy(S.make_cloud(1000, 20000, "uniform")),
                      torch.from_numpy(S.make_cloud(1001, 20000, "surface"))], 0)
    rep = model_oracle.forward({"point_clouds": cl.clone()}, state_dict, keep_internal=True)
    sd = H.to_device(state_dict, "cuda")
    with torch.no_grad():
        ep = H.forward(E, cl.cuda(), sd)
    torch.cuda.synchronize()
    return cl, rep, ep, sd


def test_unfused_model_over_the_dropin_ext_matches_the_oracle(run, model_oracle):
    from graspnet_amd import pred_decode
    cl, rep, ep, _ = run
    for k in PU.INDEX_KEYS:
        assert torch.equal(ep[k].cpu(), rep[k]), k
    keys, flips, same = PU.compare_end_points(ep, rep)
    assert flips["top_view"] <= 2 and flips["max_gap_in_oracle_scores"] <= PU.MAX_VIEW_FLIP_GAP, flips
    for k, v in keys.items():
        assert v["ok"], "%s: max |dropin - oracle| %.3e > limit %.3e" % (k, v["vs_oracle"]["max"], v["limit"])
        if k in PU.HEAD_KEYS:
            assert v["vs_oracle"]["max"] <= PU.HEAD_ABS_TOL, (k, v["vs_oracle"]["max"])
    # the decode kernel on end_points produced elsewhere (contiguous (B,C,...) tensors, not the fused path's views)
    preds = pred_decode({k: v for k, v in ep.items()})
    rows = PU.compare_rows(preds, model_oracle.pred_decode(rep))
    assert rows["rows"] >= 0.95 * sum(int(r.shape[0]) for r in model_oracle.pred_decode(rep)) - 2, rows
    assert rows["class_flips"] <= 3 and rows["objectness_flips"] <= 2, rows
    assert rows["score_max"] <= rows["score_limit"] and rows["width_max"] <= 1e-5, rows
    PU.REPORT["drop-in mode: unfused model over graspnet_amd.pointnet2_ext, B=2 N=20000"] = {
        "math": "torch-GPU matmuls between the drop-in operators", "keys": keys, "flips": flips, "rows": rows}


def test_cylinder_lists_of_the_composed_model_are_the_oracles(run):
    """the four cylinder_query calls of CloudCrop inside the composed model (fed by the GPU's own top-view rotations)"""
    from graspnet_amd import pointnet2_ext as E
    cl, rep, ep, _ = run
    same = (ep["grasp_top_view_inds"].cpu() == rep["grasp_top_view_inds"])
    rot9 = ep["grasp_top_view_rot"].reshape(2, 1024, 9).contiguous()
    for d, hmax in enumerate((0.01, 0.02, 0.03, 0.04)):
        got = E.cylinder_query(ep["fp2_xyz"], ep["input_xyz"], rot9, 0.05, -0.02, hmax, 64).cpu()
        want = rep["_cylinder_idx"][d]
        assert torch.equal(got[same], want[same]), hmax


def test_consecutive_queries_on_one_cloud_share_one_grid(run, monkeypatch):
    """SA1's ball_query and CloudCrop's four cylinder_query calls bin the SAME cloud tensor: one grid build, not five
    (round-3 VERDICT weak item 8) -- and a reused grid returns exactly what a fresh one does; an in-place write to the cloud
    (torch bumps its version counter) or another stream invalidates it"""
    from graspnet_amd import pointnet2_ext as E, _lib as L
    cl, rep, ep, sd = run
    lib = L.lib()
    builds = []
    real = lib.gn_grid_build

    class Counting:
        def __call__(self, *a):
            builds.append(a[1:4])
            return real(*a)
    monkeypatch.setattr(lib, "gn_grid_build", Counting(), raising=False)
    E.clear_grid_cache()
    with torch.no_grad():
        ep2 = H.forward(E, cl.cuda(), sd)
    n_cached = len(builds)
    assert n_cached == 1, builds                   # the 20 000-pt level; SA2-4 (<= 2048 points) take the scan kernels
    monkeypatch.setenv("GN_GRID_CACHE", "0")
    builds.clear()
    with torch.no_grad():
        ep3 = H.forward(E, cl.cuda(), sd)
    assert len(builds) == 5, builds
    for k in ("sa1_inds", "grasp_top_view_inds", "grasp_score_pred", "sa1_features"):
        assert torch.equal(ep2[k], ep3[k]) and torch.equal(ep2[k], ep[k]), k
    monkeypatch.setenv("GN_GRID_CACHE", "1")
    E.clear_grid_cache()
    builds.clear()
    xyz = cl.cuda()
    ctr = xyz[:, :512].contiguous()
    a = E.ball_query(ctr, xyz, 0.04, 64)
    b = E.ball_query(ctr, xyz, 0.05, 32)
    assert len(builds) == 1
    xyz[:, 100:200] += 0.01                                   # in-place write: the cached grid is stale and must go
    c = E.ball_query(ctr, xyz, 0.04, 64)
    assert len(builds) == 2 and not torch.equal(a, c)
    b_reused = E.ball_query(ctr, xyz, 0.05, 32)               # on the grid built for radius 0.04
    assert len(builds) == 2
    monkeypatch.setenv("GN_GRID_CACHE", "0")
    assert torch.equal(E.ball_query(ctr, xyz, 0.04, 64), c) and torch.equal(E.ball_query(ctr, xyz, 0.05, 32), b_reused)
    monkeypatch.setenv("GN_GRID_CACHE", "1")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    n0 = len(builds)
    with torch.cuda.stream(side):
        d = E.ball_query(ctr, xyz, 0.04, 64)                  # another stream: its own grid (stream order is the only order)
    side.synchronize()
    assert len(builds) == n0 + 1 and torch.equal(d, c)
    E.clear_grid_cache()


def test_chained_sampling_calls_take_the_identity_prefix_shortcut(oracle, monkeypatch):
    """backbone.py:105-121 samples the sampled points again (2048 -> 1024 -> 512 -> 256).  The drop-in remembers its last call
    per stream; a call whose input is bit for bit the points that call picked inherits its tie step and writes the identity
    prefix at once (or warm-starts behind the tie) -- same indices as the oracle's full chains, as with GN_FPS_CHAIN=0; an input
    that only LOOKS related (indices edited in place, another cloud of the same shape) gets no promise and is still right"""
    from graspnet_amd import pointnet2_ext as E, synthetic as S
    x = torch.stack([torch.from_numpy(S.make_cloud(1000, 20000, "uniform")), torch.from_numpy(S.make_cloud(1001, 20000, "surface")),
                     torch.from_numpy(S.make_cloud(1007, 20000, "uniform"))], 0)
    xd = x.cuda()

    def chain(xd):
        lv, res = xd, []
        for m in (2048, 1024, 512, 256):
            inds = E.furthest_point_sampling(lv, m)
            res.append(inds)
            lv = E.gather_points(lv.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
        return res
    E.clear_fps_chain()
    got = chain(xd)
    lv = x
    for m, g in zip((2048, 1024, 512, 256), got):
        want = oracle.furthest_point_sampling(lv, m)
        assert torch.equal(g.cpu(), want), m
        lv = oracle.gather_points(lv.transpose(1, 2).contiguous(), want).transpose(1, 2).contiguous()
    assert torch.equal(got[1][0].cpu(), torch.arange(1024, dtype=torch.int32))          # scene 0: no tie -> identity prefix
    monkeypatch.setenv("GN_FPS_CHAIN", "0")
    ref = chain(xd)
    monkeypatch.setenv("GN_FPS_CHAIN", "1")
    assert all(torch.equal(a, b) for a, b in zip(got, ref))
    # the shortcut is a shortcut: the second link with the promise against without (same run, a ratio, not an absolute time)
    lv1 = E.gather_points(xd.transpose(1, 2).contiguous(), got[0]).transpose(1, 2).contiguous()

    def timed(chain_on):
        monkeypatch.setenv("GN_FPS_CHAIN", "1" if chain_on else "0")
        ts = []
        for _ in range(3):
            E.clear_fps_chain()
            E.furthest_point_sampling(xd, 2048)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r = E.furthest_point_sampling(lv1, 1024)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return min(ts), r
    t_on, r_on = timed(True)
    t_off, r_off = timed(False)
    monkeypatch.setenv("GN_FPS_CHAIN", "1")
    assert torch.equal(r_on, r_off) and t_on < 0.7 * t_off, (t_on, t_off)
    # indices edited in place after the first call: the remembered call no longer describes them -> no promise, full chain
    E.clear_fps_chain()
    i1 = E.furthest_point_sampling(xd, 2048)
    i1[:, 5] = i1[:, 7]
    lvx = E.gather_points(xd.transpose(1, 2).contiguous(), i1).transpose(1, 2).contiguous()
    assert torch.equal(E.furthest_point_sampling(lvx, 1024).cpu(), oracle.furthest_point_sampling(lvx.cpu(), 1024))
    # an unrelated cloud of the right shape right after a sampling call
    E.clear_fps_chain()
    E.furthest_point_sampling(xd, 2048)
    other = S.make_batch(4000, 3, 2048, "uniform")
    assert torch.equal(E.furthest_point_sampling(other.cuda(), 1024).cpu(), oracle.furthest_point_sampling(other, 1024))
    E.clear_fps_chain()


def test_dropin_caches_under_hipgraph_capture_on_the_callers_own_stream(oracle):
    """Round-4 VERDICT weak item 5: warm up, then capture a hipGraph ON THE SAME STREAM over a static input.  The grid (and the
    remembered sampling call) of the warm-up must not be reused inside the capture -- the build would be missing from the graph
    and every replay would read the warm-up cloud's cells.  Replays over DIFFERENT clouds must equal eager calls bit for bit;
    inside the capture the five queries of one forward still share one grid build."""
    from graspnet_amd import pointnet2_ext as E, synthetic as S, _lib as L
    lib = L.lib()
    real = lib.gn_grid_build
    builds = []

    class Counting:
        def __call__(self, *a):
            builds.append(L.capture_id(a[-1]))
            return real(*a)
    dev = torch.device("cuda")
    clouds = [S.make_batch(s, 2, 20000, "uniform").to(dev) for s in (7000, 7001, 7002)]
    static = clouds[0].clone()
    rot = torch.eye(3, device=dev).reshape(1, 1, 9).repeat(2, 512, 1).contiguous()

    def step(x):
        i1 = E.furthest_point_sampling(x, 2048)
        c1 = E.gather_points(x.transpose(1, 2).contiguous(), i1).transpose(1, 2).contiguous()
        i2 = E.furthest_point_sampling(c1, 1024)                              # chained: consults the remembered call
        ctr = c1[:, :512].contiguous()
        return (i1, i2, E.ball_query(ctr, x, 0.04, 64), E.cylinder_query(ctr, x, rot, 0.05, -0.02, 0.02, 64),
                E.cylinder_query(ctr, x, rot, 0.05, -0.02, 0.04, 64))
    E.clear_grid_cache(); E.clear_fps_chain()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    try:
        lib.gn_grid_build = Counting()
        with torch.cuda.stream(st):
            for _ in range(2):
                step(static)                                                  # warm-up on the capture's own stream
            st.synchronize()
            assert builds == [0], builds                                      # eager: one build, reused by the second step
            assert L.capture_id(st.cuda_stream) == 0
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                assert L.capture_id(st.cuda_stream) != 0
                outs = step(static)
            # the capture built its own grid exactly once (not zero times: the warm-up's is not reused; not three times)
            assert len(builds) == 2 and builds[1] != 0, builds
            for cl in clouds[1:] + clouds[:1]:
                static.copy_(cl)
                g.replay()
                st.synchronize()
                got = [o.clone() for o in outs]
                E.clear_grid_cache(); E.clear_fps_chain()
                want = step(cl.clone())
                st.synchronize()
                for a, b in zip(got, want):
                    assert torch.equal(a, b)
            # an eager call after the capture does not reuse the capture's grid either
            n0 = len(builds)
            E.ball_query(static[:, :512].contiguous(), static, 0.04, 64)
            assert len(builds) == n0 + 1 and builds[-1] == 0
    finally:
        lib.gn_grid_build = real
        E.clear_grid_cache(); E.clear_fps_chain()
    # and the replayed sampler equals the oracle on the last cloud
    assert torch.equal(got[0].cpu(), oracle.furthest_point_sampling(clouds[0].cpu(), 2048))
