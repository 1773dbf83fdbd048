"""GPU parity: every operator of the C ABI (through the pointnet2._ext / knn_pytorch drop-in
bindings) against the CPU oracle on the same seeded inputs.  Index outputs must be BIT-EXACT."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from graspnet_amd import pointnet2_ext
    return pointnet2_ext


def _cloud(seed, b, n, variant="uniform"):
    from graspnet_amd import synthetic as S
    return S.make_batch(seed, b, n, variant)


def _dev(t):
    return t.cuda()


@pytest.mark.parametrize("n,m,variant", [(20000, 2048, "uniform"), (20000, 2048, "surface"), (2048, 1024, "uniform"),
                                         (1024, 512, "uniform"), (512, 256, "uniform"), (300, 128, "uniform"),
                                         (77, 77, "uniform"), (5000, 700, "surface"), (9000, 64, "uniform"),
                                         (13000, 64, "uniform"), (17000, 64, "uniform"), (24000, 64, "uniform"),
                                         (13000, 900, "surface"), (16384, 2048, "surface"), (20480, 1500, "uniform"),
                                         (12289, 300, "surface")])
def test_fps_bit_exact(E, oracle, n, m, variant):
    x = _cloud(100 + n, 2, n, variant)
    want = oracle.furthest_point_sampling(x, m)
    got = E.furthest_point_sampling(_dev(x), m).cpu()
    assert torch.equal(got, want)


@pytest.mark.parametrize("n,variant", [(50000, "uniform"), (30000, "surface"), (60000, "uniform"), (70000, "uniform")])
def test_fps_large_n(E, oracle, n, variant):
    """tiered kernel (temps in registers, coordinates in registers / LDS / L2) up to 65 536 positions,
    any-n global-memory kernel beyond; duplicates in the surface variant force exact ties"""
    x = _cloud(7, 2, n, variant)
    want = oracle.furthest_point_sampling(x, 300)
    got = E.furthest_point_sampling(_dev(x), 300).cpu()
    assert torch.equal(got, want)


def test_fps_adversarial_golden(E, golden_dir):
    """duplicates (exact ties), near-origin skipped points, index 0 skipped."""
    z = np.load(golden_dir + "/ops_small.npz")
    for name in ("n1000", "n300", "n4096dup"):
        x = torch.from_numpy(z[name + "/pts"])
        m = z[name + "/fps"].shape[1]
        got = E.furthest_point_sampling(_dev(x), m).cpu().numpy()
        assert np.array_equal(got, z[name + "/fps"]), name


def test_fps_all_ties_tree_order(E, oracle):
    n = 1024
    x = torch.zeros(2, n, 3)
    x[:, :, 0] = 1.0
    x[:, 0] = torch.tensor([0.0, 0.0, 1.0])
    x[1, 5::7, 1] = 0.25           # second scene: two tied groups
    want = oracle.furthest_point_sampling(x, 40)
    got = E.furthest_point_sampling(_dev(x), 40).cpu()
    assert torch.equal(got, want)
    assert got[0, 1] == 512


def test_fps_all_points_skipped(E, oracle):
    x = torch.full((1, 600, 3), 0.01)
    want = oracle.furthest_point_sampling(x, 16)
    got = E.furthest_point_sampling(_dev(x), 16).cpu()
    assert torch.equal(got, want) and not got.any()


def _first_tie_between_distinct_points(c, m):
    """first FPS step at which the maximal running distance is shared by DIFFERENT locations (copies of one location are
    interchangeable: either gives the same sample) -- numpy restatement of sampling_gpu.cu:105-120's update in fp32,
    products and sums rounded one by one like the kernels (-ffp-contract=off).  Before that step the sample sequence does
    not depend on any tie order, which is all this walk needs."""
    c = c.astype(np.float32)
    mag = (c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]) + c[:, 2] * c[:, 2]
    live = mag.astype(np.float64) > 1e-3
    tmp = np.where(live, np.float32(1e10), np.float32(-1)).astype(np.float32)
    cur = 0
    for j in range(1, m):
        d = c - c[cur]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        tmp = np.where(live, np.minimum(tmp, d2), tmp)
        w = np.flatnonzero(tmp == tmp.max())
        if len(w) > 1 and len(np.unique(c[w], axis=0)) > 1:
            return j
        cur = int(w[0])
    return 0x7fffffff


@pytest.mark.parametrize("variant,seed", [("uniform", 1000), ("uniform", 1024), ("surface", 1000), ("surface", 1016)])
def test_fps_tie_step_is_exact(variant, seed):
    """the SA1 sampler's tie report (first step with a tie between distinct locations; lets SA2-4 return the identity
    prefix when it lies beyond their sample count) against the walk above: fp32 ties between distinct points do occur
    (about one uniform cloud in five within 2048 steps), while the 1000 duplicated points of a `surface` cloud must NOT be
    reported, wherever the region cuts put the copies"""
    from graspnet_amd import _lib as L
    lib = L.lib()
    x = _cloud(seed, 8, 20000, variant)
    xd = x.cuda()
    ws = torch.empty(max(int(lib.gn_fps_workspace_bytes(8, 20000)), 16), dtype=torch.uint8, device="cuda")
    idx = torch.empty(8, 2048, dtype=torch.int32, device="cuda")
    want = [_first_tie_between_distinct_points(x[i].numpy(), 2048) for i in range(8)]
    for _ in range(3):       # the slot a point lands in depends on atomic ticket order: the report must not
        tie = torch.full((8,), -7, dtype=torch.int32, device="cuda")
        L.check(lib.gn_furthest_point_sampling(xd.data_ptr(), 8, 20000, 2048, ws.data_ptr(), idx.data_ptr(), None,
                                               tie.data_ptr(), None, torch.cuda.current_stream().cuda_stream), "fps")
        torch.cuda.synchronize()
        assert tie.cpu().tolist() == want
    if (variant, seed) == ("surface", 1000):      # no tie between distinct points in these 8 scenes: the SA2-4 shortcut holds
        assert min(want) == 0x7fffffff


@pytest.mark.parametrize("n,m,variant,cell", [(20000, 2048, "uniform", "32"), (20000, 700, "surface", "16"), (5000, 500, "surface", "16"),
                                              (16384, 300, "uniform", "16"), (4100, 2048, "uniform", "32"), (33000, 300, "surface", None),
                                              (50000, 2048, "uniform", None), (70000, 300, "uniform", None),
                                              (140000, 200, "uniform", None), (262144, 64, "uniform", None)])
def test_fps_cell_kernel_bit_exact(E, oracle, monkeypatch, n, m, variant, cell):
    """fps_cell.hip (regions of 64-slot rows in an L2-resident workspace, exact work skipping): the default above 24 576
    positions, forced through GN_FPS_CELL for the sizes the register kernels serve; 16 x 16 and 32 x 32 regions, 1 / 2 / 4
    rows per region, duplicated points (surface), a few never-sampled points near the origin"""
    if cell is not None:
        monkeypatch.setenv("GN_FPS_CELL", cell)
    x = _cloud(900 + n, 2, n, variant)
    x[0, 5:40] *= 0.01                  # ||p||^2 <= 1e-3: skipped by the reference (sampling_gpu.cu:105)
    x[1, 0] = 0.001                     # the first sample itself is a skipped point
    want = oracle.furthest_point_sampling(x, m)
    got = E.furthest_point_sampling(_dev(x), m).cpu()
    assert torch.equal(got, want)


@pytest.mark.parametrize("n,m,cell", [(4097, 1, "16"), (5003, 2, "32"), (7919, 33, "16"), (16385, 600, "16"), (24577, 600, None),
                                      (33333, 257, None), (65536, 100, None), (65537, 100, None), (131073, 64, None)])
def test_fps_cell_kernel_odd_sizes(E, oracle, monkeypatch, n, m, cell):
    """sizes around the cell kernel's layout boundaries (16 x 16 / 32 x 32 regions, 1 / 2 / 4 rows, the 65 536-point switch),
    primes, one and two samples; sampled-with-replacement clouds (exact duplicates) on every other case"""
    if cell is not None:
        monkeypatch.setenv("GN_FPS_CELL", cell)
    x = _cloud(1300 + n, 2, n, "surface" if n % 2 else "uniform")
    want = oracle.furthest_point_sampling(x, m)
    got = E.furthest_point_sampling(_dev(x), m).cpu()
    assert torch.equal(got, want)


def test_fps_cell_kernel_ties_and_skips(E, oracle, monkeypatch):
    """the adversarial cases of the register kernels on the cell kernel: every point at one of two tied locations (the
    reference's tree order decides), every point skipped (index 0 throughout), exact duplicates of the whole cloud"""
    monkeypatch.setenv("GN_FPS_CELL", "16")
    n = 4096
    x = torch.zeros(2, n, 3)
    x[:, :, 0] = 1.0
    x[:, 0] = torch.tensor([0.0, 0.0, 1.0])
    x[1, 5::7, 1] = 0.25
    want = oracle.furthest_point_sampling(x, 40)
    assert torch.equal(E.furthest_point_sampling(_dev(x), 40).cpu(), want)
    y = torch.full((1, 5000, 3), 0.01)
    got = E.furthest_point_sampling(_dev(y), 16).cpu()
    assert torch.equal(got, oracle.furthest_point_sampling(y, 16)) and not got.any()
    z = _cloud(31, 1, 3000, "uniform").repeat(1, 3, 1)       # every location three times
    assert torch.equal(E.furthest_point_sampling(_dev(z), 900).cpu(), oracle.furthest_point_sampling(z, 900))
    # more samples than distinct locations: once every location is taken the maximum is 0.0 and the reference's tree order
    # alone decides (a winner at distance zero is still a winner, not "no candidate")
    few = _cloud(32, 2, 50, "uniform").repeat(1, 100, 1)      # 5000 points at 50 locations
    assert torch.equal(E.furthest_point_sampling(_dev(few), 200).cpu(), oracle.furthest_point_sampling(few, 200))
    monkeypatch.setenv("GN_FPS_CELL", "32")
    assert torch.equal(E.furthest_point_sampling(_dev(few), 200).cpu(), oracle.furthest_point_sampling(few, 200))


@pytest.mark.parametrize("n,variant,seed,cell", [(20000, "uniform", 1024, "32"), (20000, "surface", 1000, "16"),
                                                 (50000, "uniform", 1000, None), (30000, "surface", 1003, None)])
def test_fps_cell_tie_step_is_exact(monkeypatch, n, variant, seed, cell):
    """the cell kernel's tie report against the numpy walk (as test_fps_tie_step_is_exact for the region kernel), at
    config[3]'s 50 000 points too: behind it SA2-4 take the identity-prefix shortcut there as well"""
    from graspnet_amd import _lib as L
    lib = L.lib()
    if cell is not None:
        monkeypatch.setenv("GN_FPS_CELL", cell)
    b = 4 if n <= 20000 else 2
    x = _cloud(seed, b, n, variant)
    xd = x.cuda()
    ws = torch.empty(max(int(lib.gn_fps_workspace_bytes(b, n)), 16), dtype=torch.uint8, device="cuda")
    idx = torch.empty(b, 2048, dtype=torch.int32, device="cuda")
    want = [_first_tie_between_distinct_points(x[i].numpy(), 2048) for i in range(b)]
    for _ in range(2):
        tie = torch.full((b,), -7, dtype=torch.int32, device="cuda")
        L.check(lib.gn_furthest_point_sampling(xd.data_ptr(), b, n, 2048, ws.data_ptr(), idx.data_ptr(), None,
                                               tie.data_ptr(), None, torch.cuda.current_stream().cuda_stream), "fps")
        torch.cuda.synchronize()
        assert tie.cpu().tolist() == want


def test_fps_warm_start_behind_a_tie(oracle):
    """the deeper samplers behind SA1's tie report: steps below the reported step are written as the identity prefix, the
    running distances rebuilt without the arg-max chain, the chain run from there.  Uniform scenes 1024.. hold ties at
    steps 195 / 1262 / 1779 (scenes 0, 3, 5); any smaller report must give the same samples too."""
    from graspnet_amd import _lib as L
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    x = _cloud(1024, 8, 20000, "uniform")
    xd = x.cuda()
    ws = torch.empty(max(int(lib.gn_fps_workspace_bytes(8, 20000)), 16), dtype=torch.uint8, device="cuda")
    idx1 = torch.empty(8, 2048, dtype=torch.int32, device="cuda")
    xyz1 = torch.empty(8, 2048, 3, device="cuda")
    tie = torch.empty(8, dtype=torch.int32, device="cuda")
    L.check(lib.gn_furthest_point_sampling(xd.data_ptr(), 8, 20000, 2048, ws.data_ptr(), idx1.data_ptr(), xyz1.data_ptr(),
                                           tie.data_ptr(), None, st), "fps1")
    torch.cuda.synchronize()
    assert tie.cpu().tolist()[0] == 195
    level = xyz1
    for n, m in ((2048, 1024), (1024, 512), (512, 256)):
        want = oracle.furthest_point_sampling(level.cpu(), m)
        outs = []
        for flag in (tie, torch.clamp(tie, max=100), torch.clamp(tie, max=2), torch.zeros_like(tie)):
            idx = torch.full((8, m), -1, dtype=torch.int32, device="cuda")
            nx = torch.full((8, m, 3), float("nan"), device="cuda")
            L.check(lib.gn_furthest_point_sampling(level.data_ptr(), 8, n, m, None, idx.data_ptr(), nx.data_ptr(), None,
                                                   flag.data_ptr(), st), "fps deeper")
            torch.cuda.synchronize()
            assert torch.equal(idx.cpu(), want)
            assert torch.equal(nx, torch.gather(level, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)))
        level = nx


@pytest.mark.parametrize("n,variant", [(95000, "uniform"), (85000, "surface"), (95000, "surface")])
def test_operators_at_the_forks_own_cloud_sizes(E, oracle, model_oracle, n, variant):
    """grasp_service.py:289 samples 95 000 points, demo_pointcloud.py:31 / foundationstereo.py:55 85 000 (B = 1): the FULL-length
    SA1 sampler (2048 samples: fps_cell.hip's 32 x 32 regions), SA1's ball query and CloudCrop's four cylinder queries at
    that size, bit-exact against the oracle through the drop-in operators (grid path)"""
    x = _cloud(8000 + n, 1, n, variant)
    xd = _dev(x)
    inds = E.furthest_point_sampling(xd, 2048).cpu()
    assert torch.equal(inds, oracle.furthest_point_sampling(x, 2048))
    c = torch.gather(x, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    assert torch.equal(E.ball_query(_dev(c), xd, 0.04, 64).cpu(), oracle.ball_query(c, x, 0.04, 64))
    seeds = c[:, :1024].contiguous()
    rng = np.random.default_rng(n)
    rot = model_oracle.viewpoint_to_matrix(torch.from_numpy(rng.standard_normal((1024, 3)).astype(np.float32)),
                                           torch.zeros(1024)).view(1, 1024, 9).contiguous()
    for hmax in (0.01, 0.02, 0.03, 0.04):
        got = E.cylinder_query(_dev(seeds), xd, _dev(rot), 0.05, -0.02, hmax, 64).cpu()
        assert torch.equal(got, oracle.cylinder_query(seeds, x, rot, 0.05, -0.02, hmax, 64)), hmax
    f = x.transpose(1, 2).contiguous()
    idx = oracle.ball_query(c, x, 0.04, 64)
    assert torch.equal(E.group_points(_dev(f), _dev(idx)).cpu(), oracle.group_points(f, idx))


SA = [(20000, 2048, 0.04, 64), (2048, 1024, 0.1, 32), (1024, 512, 0.2, 16), (512, 256, 0.3, 16)]


@pytest.mark.parametrize("n,m,r,ns", SA + [(3000, 333, 0.05, 100), (500, 7, 0.001, 8), (64, 64, 10.0, 64)])
def test_ball_query_bit_exact(E, oracle, n, m, r, ns):
    x = _cloud(200 + n, 2, n, "surface" if n == 3000 else "uniform")
    c = x[:, torch.randperm(n, generator=torch.Generator().manual_seed(n))[:m]].contiguous()
    if r == 0.001:
        c = c + 0.5      # empty balls -> zero rows
    want = oracle.ball_query(c, x, r, ns)
    got = E.ball_query(_dev(c), _dev(x), r, ns).cpu()
    assert torch.equal(got, want)


@pytest.mark.parametrize("n,m,r,ns,variant", [(20000, 2048, 0.04, 64, "uniform"), (20000, 2048, 0.04, 64, "surface"),
                                              (50000, 2048, 0.04, 64, "uniform"), (5000, 300, 0.1, 32, "surface"),
                                              (3000, 100, 0.3, 16, "uniform"), (2048, 64, 5.0, 64, "uniform"),
                                              (9000, 50, 0.0005, 8, "uniform")])
def test_ball_query_grid_bit_exact(oracle, n, m, r, ns, variant):
    """grid-accelerated ball query == reference semantics (first ns hits in index order, padding, zero rows),
    incl. dense balls (hundreds of hits), a ball that swallows the whole cloud, empty balls and far-away centres"""
    from graspnet_amd import _lib as L
    lib = L.lib()
    x = _cloud(400 + n, 2, n, variant)
    c = x[:, torch.randperm(n, generator=torch.Generator().manual_seed(n))[:m]].contiguous()
    c[0, :5] += 3.0            # centres far outside the cloud: no hit
    c[1, 5:9] -= 0.03          # slightly outside
    want = oracle.ball_query(c, x, r, ns)
    xd, cd = x.cuda(), c.cuda()
    ws = torch.empty((lib.gn_grid_workspace_bytes(2, n),), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.gn_grid_build(xd.data_ptr(), 2, n, r, ws.data_ptr(), st), "grid_build")
    got = torch.empty((2, m, ns), dtype=torch.int32, device="cuda")
    L.check(lib.gn_ball_query_grid(cd.data_ptr(), ws.data_ptr(), 2, n, m, r, ns, got.data_ptr(), st), "ball_query_grid")
    assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("n,m,hmax", [(20000, 1024, 0.01), (20000, 1024, 0.04), (4096, 100, 0.02), (50000, 256, 0.03)])
def test_cylinder_query_bit_exact(E, oracle, model_oracle, n, m, hmax):
    x = _cloud(300 + n, 2, n)
    c = x[:, :m].contiguous()
    rng = np.random.default_rng(n + m)
    rot = model_oracle.viewpoint_to_matrix(torch.from_numpy(rng.standard_normal((2 * m, 3)).astype(np.float32)),
                                           torch.zeros(2 * m)).view(2, m, 9).contiguous()
    want = oracle.cylinder_query(c, x, rot, 0.05, -0.02, hmax, 64)
    got = E.cylinder_query(_dev(c), _dev(x), _dev(rot), 0.05, -0.02, hmax, 64).cpu()
    assert torch.equal(got, want)


def test_drop_in_queries_grid_equals_scan(E, model_oracle, monkeypatch):
    """E.ball_query / E.cylinder_query bin the cloud for N >= 4096 (grid.hip); GN_QUERY_PATH=brute keeps the index-order
    scan (query.hip): identical index lists, incl. centres far outside the cloud (zero rows) and nsample < 64"""
    x = _cloud(77, 2, 9000, "surface")
    c = x[:, ::9][:, :700].contiguous()
    c[0, :5] += 3.0
    rng = np.random.default_rng(9)
    rot = model_oracle.viewpoint_to_matrix(torch.from_numpy(rng.standard_normal((2 * 700, 3)).astype(np.float32)),
                                           torch.zeros(2 * 700)).view(2, 700, 9).contiguous()
    xd, cd, rd = _dev(x), _dev(c), _dev(rot)
    grid_b = [E.ball_query(cd, xd, r, ns) for r, ns in ((0.04, 64), (0.1, 32), (0.011, 7))]
    grid_c = [E.cylinder_query(cd, xd, rd, 0.05, -0.02, h, ns) for h, ns in ((0.01, 64), (0.04, 64), (0.02, 10))]
    monkeypatch.setenv("GN_QUERY_PATH", "brute")
    scan_b = [E.ball_query(cd, xd, r, ns) for r, ns in ((0.04, 64), (0.1, 32), (0.011, 7))]
    scan_c = [E.cylinder_query(cd, xd, rd, 0.05, -0.02, h, ns) for h, ns in ((0.01, 64), (0.04, 64), (0.02, 10))]
    for a, b in zip(grid_b + grid_c, scan_b + scan_c):
        assert torch.equal(a, b)
    assert not grid_b[0][0, :5].any() and not grid_c[1][0, :5].any()


def test_cylinder_query_any_3x3_matrix(E, oracle, model_oracle):
    """cylinder_query_gpu.cu:39-47 takes ANY 3x3 `rot`; the grid kernels' cell windows assume a rotation, so each seed
    checks R^T R = I on the device and walks every cell when it fails: sheared, scaled (shrinking AND stretching),
    reflected, singular and NaN matrices give the oracle's lists at a size (N >= 4096) that takes the grid path"""
    n, m = 9000, 256
    x = _cloud(611, 2, n, "surface")
    c = x[:, ::7][:, :m].contiguous()
    rng = np.random.default_rng(5)
    rot = model_oracle.viewpoint_to_matrix(torch.from_numpy(rng.standard_normal((2 * m, 3)).astype(np.float32)),
                                           torch.zeros(2 * m)).view(2, m, 3, 3).clone()
    shear = torch.eye(3)
    shear[0, 1] = 0.7
    shear[2, 0] = -0.4
    rot[:, 0:40] = rot[:, 0:40] @ shear                      # sheared
    rot[:, 40:80] *= 0.25                                    # shrinks distances: hits far outside the nominal reach
    rot[:, 80:120] *= 3.0                                    # stretches
    rot[:, 120:140, :, 0] *= -1.0                            # reflection (still orthonormal: takes the window)
    rot[:, 140:150] = 0.0                                    # singular: x_rot = 0 for every point
    rot[0, 150:155] = float("nan")
    rot[:, 155:175] = rot[:, 155:175] * (1.0 + 1e-4)         # just outside the orthonormality tolerance
    rot[:, 175:195] = rot[:, 175:195] * (1.0 + 4e-6)         # inside it
    rot = rot.reshape(2, m, 9).contiguous()
    for hmin, hmax, ns in ((-0.02, 0.04, 64), (-0.02, 0.01, 16)):
        want = oracle.cylinder_query(c, x, rot, 0.05, hmin, hmax, ns)
        got = E.cylinder_query(_dev(c), _dev(x), _dev(rot), 0.05, hmin, hmax, ns).cpu()
        assert torch.equal(got, want)
        assert want[:, 40:80].ne(want[:, 40:80, :1]).any()      # the shrunk cylinders really do collect hits


def test_cylinder_unique_grid_any_3x3_matrix(model_oracle):
    """the same for CloudCrop's four nested cylinders on the grid (gn_cylinder_query_unique_grid) against the scan form"""
    from graspnet_amd import _lib as L
    lib = L.lib()
    import ctypes
    n, m = 8000, 128
    x = _cloud(612, 2, n, "uniform").cuda()
    c = x[:, :m].contiguous()
    rng = np.random.default_rng(6)
    rot = model_oracle.viewpoint_to_matrix(torch.from_numpy(rng.standard_normal((2 * m, 3)).astype(np.float32)),
                                           torch.zeros(2 * m)).view(2, m, 3, 3).clone()
    rot[:, 0:30] *= 0.3
    rot[:, 30:60, 0, 1] += 0.5
    rot[1, 60:64] = float("nan")
    rot = rot.reshape(2, m, 9).contiguous().cuda()
    st = torch.cuda.current_stream().cuda_stream
    hmax = (ctypes.c_float * 4)(0.01, 0.02, 0.03, 0.04)
    ws = torch.empty((lib.gn_grid_workspace_bytes(2, n),), dtype=torch.uint8, device="cuda")
    L.check(lib.gn_grid_build(x.data_ptr(), 2, n, 0.04, ws.data_ptr(), st), "grid_build")
    ug = torch.zeros((2, m, 256), dtype=torch.int32, device="cuda")
    cg = torch.zeros((2, m), dtype=torch.int32, device="cuda")
    us, cs = torch.zeros_like(ug), torch.zeros_like(cg)
    L.check(lib.gn_cylinder_query_unique_grid(c.data_ptr(), rot.data_ptr(), ws.data_ptr(), 2, n, m, 0.05, -0.02, hmax, 4, 64,
                                              ug.data_ptr(), cg.data_ptr(), st), "unique_grid")
    L.check(lib.gn_cylinder_query_unique(c.data_ptr(), x.data_ptr(), rot.data_ptr(), 2, n, m, 0.05, -0.02, hmax, 4, 64,
                                         us.data_ptr(), cs.data_ptr(), st), "unique")
    assert torch.equal(cg, cs)
    ug, us, cnt = ug.cpu(), us.cpu(), cs.cpu()
    for b in range(2):
        for j in range(m):
            k = int(cnt[b, j])
            assert sorted(ug[b, j, :k].tolist()) == sorted(us[b, j, :k].tolist()), (b, j)   # grid: grouped by depth class


@pytest.mark.parametrize("bad", [float("inf"), -float("inf"), float("nan"), 1e30, -3e38])
def test_queries_with_non_finite_and_far_points(E, oracle, model_oracle, monkeypatch, bad):
    """an Inf / NaN / 1e30 coordinate in a cloud that takes the grid path (N >= 4096): the reference's scan simply never
    hits such a point; the grid is built over the finite box (cells counted in float, no int overflow) and returns the
    same lists as the oracle and as the index-order scan kernels"""
    n, m = 6000, 300
    x = _cloud(613, 2, n, "uniform")
    x[0, 17, 0] = bad
    x[0, 4000, 2] = bad
    x[1, 0] = bad
    x[1, 5999, 1] = -bad
    c = x[:, 100:100 + m].contiguous()
    c[0, 3] = bad                                            # a centre out there as well: zero row
    rng = np.random.default_rng(8)
    rot = model_oracle.viewpoint_to_matrix(torch.from_numpy(rng.standard_normal((2 * m, 3)).astype(np.float32)),
                                           torch.zeros(2 * m)).view(2, m, 9).contiguous()
    wb = oracle.ball_query(c, x, 0.06, 32)
    wc = oracle.cylinder_query(c, x, rot, 0.05, -0.02, 0.04, 64)
    gb = E.ball_query(_dev(c), _dev(x), 0.06, 32)
    gc = E.cylinder_query(_dev(c), _dev(x), _dev(rot), 0.05, -0.02, 0.04, 64)
    assert torch.equal(gb.cpu(), wb) and torch.equal(gc.cpu(), wc)
    monkeypatch.setenv("GN_QUERY_PATH", "brute")
    assert torch.equal(E.ball_query(_dev(c), _dev(x), 0.06, 32), gb)
    assert torch.equal(E.cylinder_query(_dev(c), _dev(x), _dev(rot), 0.05, -0.02, 0.04, 64), gc)


def test_group_gather_bit_exact(E, oracle):
    g = torch.Generator().manual_seed(3)
    # vectorised global gathers (C=3, long rows) / LDS-staged rows with 8, 16, 16 (odd n, partial last group) and 2 channels
    # per workgroup / the scalar kernel (element count not a multiple of 4)
    for (b, c, n, npnt, ns) in [(2, 3, 20000, 2048, 64), (2, 128, 2048, 1024, 32), (1, 256, 512, 256, 16), (2, 20, 1001, 50, 8),
                                (1, 9, 8192, 64, 4), (3, 5, 17, 3, 2)]:
        f = torch.rand(b, c, n, generator=g)
        idx = torch.randint(0, n, (b, npnt, ns), generator=g, dtype=torch.int32)
        assert torch.equal(E.group_points(_dev(f), _dev(idx)).cpu(), oracle.group_points(f, idx))
        gi = torch.randint(0, n, (b, npnt), generator=g, dtype=torch.int32)
        assert torch.equal(E.gather_points(_dev(f), _dev(gi)).cpu(), oracle.gather_points(f, gi))


def test_three_nn_interpolate_bit_exact(E, oracle):
    for (n, m) in [(512, 256), (1024, 512), (100, 3), (70, 2500)]:
        u, k = _cloud(1, 2, n), _cloud(2, 2, m)
        k[:, m // 2:] = k[:, :m - m // 2]        # duplicates -> exact distance ties
        d2w, iw = oracle.three_nn(u, k)
        d2, i = E.three_nn(_dev(u), _dev(k))
        assert torch.equal(i.cpu(), iw) and torch.equal(d2.cpu(), d2w)
        w = torch.rand(2, n, 3)
        f = torch.rand(2, 9, m)
        assert torch.equal(E.three_interpolate(_dev(f), _dev(iw), _dev(w)).cpu(), oracle.three_interpolate(f, iw, w))


def test_grad_ops(E, oracle, monkeypatch):
    """the three gradient operators as sorted-segment sums: BIT-IDENTICAL to the oracle's sequential ascending-entry loop
    (and so identical from run to run), incl. destinations hit by hundreds of entries (wave rebuild path), by none, and
    SA-sized shapes; GN_GRAD=atomic keeps the reference's float-atomic scatter (equal only to rounding)"""
    g = torch.Generator().manual_seed(5)
    for (b, c, f_n, npnt, ns, hot) in [(2, 6, 40, 12, 4, False), (2, 9, 2048, 1024, 32, False), (1, 3, 20000, 2048, 64, False),
                                       (2, 5, 300, 64, 16, True)]:
        idx = torch.randint(0, f_n, (b, npnt, ns), generator=g, dtype=torch.int32)
        if hot:
            idx[:, ::2, :8] = 7            # one destination collects 256 entries per cloud: > 64 -> the ordered wave rebuild
        go = torch.rand(b, c, npnt, ns, generator=g)
        want = oracle.group_points_grad(go, idx, f_n)
        got = E.group_points_grad(_dev(go), _dev(idx), f_n)
        assert torch.equal(got.cpu(), want)
        assert torch.equal(E.group_points_grad(_dev(go), _dev(idx), f_n), got)          # run-to-run identical
        gi = torch.randint(0, f_n, (b, npnt), generator=g, dtype=torch.int32)
        if hot:
            gi[:, :40] = 3
        go2 = torch.rand(b, c, npnt, generator=g)
        assert torch.equal(E.gather_points_grad(_dev(go2), _dev(gi), f_n).cpu(), oracle.gather_points_grad(go2, gi, f_n))
        i3 = torch.randint(0, f_n, (b, npnt, 3), generator=g, dtype=torch.int32)
        if hot:
            i3[:, :, 0] = 5
        w3 = torch.rand(b, npnt, 3, generator=g)
        assert torch.equal(E.three_interpolate_grad(_dev(go2), _dev(i3), _dev(w3), f_n).cpu(),
                           oracle.three_interpolate_grad(go2, i3, w3, f_n))
    monkeypatch.setenv("GN_GRAD", "atomic")
    torch.testing.assert_close(E.group_points_grad(_dev(go), _dev(idx), f_n).cpu(), want, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(E.three_interpolate_grad(_dev(go2), _dev(i3), _dev(w3), f_n).cpu(),
                               oracle.three_interpolate_grad(go2, i3, w3, f_n), rtol=1e-5, atol=1e-4)


def test_group_points_grad_on_a_real_padded_ball_query(E, oracle):
    """ball_query pads short neighbour lists by repeating their first hit up to nsample times, so a real SA1 index tensor holds
    many destinations with hundreds of entries (sparse corners of the cloud): the wave sorts such segments in LDS (<= 4096
    entries; the ordered ballot rebuild beyond) -- bit-identical to the oracle's sequential loop, and no ms-scale cliff"""
    x = _cloud(71, 2, 20000, "surface")
    x[:, 15000:] *= 3.0                                    # a sparse halo: balls there hold 1-3 points -> 60+ repeats each
    c = x[:, ::9][:, :2048].contiguous()
    idx = oracle.ball_query(c, x, 0.04, 64)
    idx[0, :100] = 123                                     # and one destination with 6400 entries (> 4096: ballot rebuild)
    counts = torch.bincount(idx[0].reshape(-1).long(), minlength=20000)
    assert int((counts > 64).sum()) > 50 and int(counts.max()) > 4096
    go = torch.rand(2, 4, 2048, 64, generator=torch.Generator().manual_seed(9))
    want = oracle.group_points_grad(go, idx, 20000)
    god, idd = _dev(go), _dev(idx)
    got = E.group_points_grad(god, idd, 20000)
    assert torch.equal(got.cpu(), want)
    times = []
    for _ in range(3):                                      # device time of the call (HIP events), best of three: no host jitter
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        got2 = E.group_points_grad(god, idd, 20000)
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
        assert torch.equal(got2, got)
    # "no ms-scale cliff" as a RATIO inside this run (not an absolute time: other processes may share the card): the same
    # call on an index tensor without long segments is the yardstick
    idx_flat = torch.randint(0, 20000, idx.shape, dtype=torch.int32, generator=torch.Generator().manual_seed(3))
    idf = _dev(idx_flat)
    E.group_points_grad(god, idf, 20000)
    base = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        E.group_points_grad(god, idf, 20000)
        b.record()
        torch.cuda.synchronize()
        base.append(a.elapsed_time(b))
    assert min(times) < 60.0 * max(min(base), 0.05), \
        "group_points_grad on a padded ball_query index took %.2f ms against %.2f ms without long segments" % (min(times), min(base))


def test_knn_bit_exact(oracle):
    from graspnet_amd import knn_ext
    rng = np.random.default_rng(0)
    for (B, D, Nr, Nq, k) in [(1, 3, 300, 300, 1), (2, 3, 5000, 1024, 1), (2, 5, 64, 33, 7), (1, 3, 40, 10, 40), (1, 3, 700, 300, 3)]:
        r = torch.from_numpy(rng.standard_normal((B, D, Nr)).astype(np.float32))
        q = torch.from_numpy(rng.standard_normal((B, D, Nq)).astype(np.float32))
        r[:, :, Nr // 2:] = r[:, :, :Nr - Nr // 2]
        want = torch.zeros(B, k, Nq, dtype=torch.int64)
        oracle.knn(r, q, want)
        got = torch.zeros(B, k, Nq, dtype=torch.int64, device="cuda")
        assert knn_ext.knn(r.cuda(), q.cuda(), got) == 1
        assert torch.equal(got.cpu(), want)


def test_knn_wave_kernel_shapes_and_the_scan_kernel_agree(oracle, monkeypatch):
    """the wave-per-query kernel at the reference's call-site shapes (utils/label_generation.py:58-60: 300 x 300, :84-86:
    Np' x 1024, k = 1), with four queries per wave (many queries), ragged tails, k in {1,2,3,4,5,8}, fewer references than k,
    exact duplicate references (ties -> lower index first); == the oracle (== the reference's knn_cpu.cpp,
    tests/test_oracle_properties.py) and == the one-lane-per-query scan kernel"""
    from graspnet_amd import knn_ext
    rng = np.random.default_rng(4)
    for (B, D, Nr, Nq, k) in [(1, 3, 300, 300, 1), (1, 3, 7919, 1024, 1), (3, 3, 1000, 8197, 2), (2, 3, 257, 4099, 3),
                              (1, 4, 63, 65, 4), (1, 3, 129, 17, 5), (2, 2, 100, 70, 8), (1, 3, 5, 20, 8), (1, 3, 1, 3, 1)]:
        r = torch.from_numpy(rng.standard_normal((B, D, Nr)).astype(np.float32))
        q = torch.from_numpy(rng.standard_normal((B, D, Nq)).astype(np.float32))
        r[:, :, Nr // 2:] = r[:, :, :Nr - Nr // 2]              # every reference point twice: exact distance ties
        q[:, :, :min(Nq, Nr)] = r[:, :, :min(Nq, Nr)]           # and queries that coincide with references (distance 0)
        want = torch.zeros(B, k, Nq, dtype=torch.int64)
        oracle.knn(r, q, want)
        got = torch.full((B, k, Nq), -7, dtype=torch.int64, device="cuda")
        assert knn_ext.knn(r.cuda(), q.cuda(), got) == 1
        assert torch.equal(got.cpu(), want), (B, D, Nr, Nq, k)
        monkeypatch.setenv("GN_KNN", "scan")
        got2 = torch.full((B, k, Nq), -7, dtype=torch.int64, device="cuda")
        knn_ext.knn(r.cuda(), q.cuda(), got2)
        monkeypatch.delenv("GN_KNN")
        assert torch.equal(got2, got), (B, D, Nr, Nq, k)


def test_error_behaviour_matches_reference(E):
    x = torch.rand(1, 10, 3)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        E.furthest_point_sampling(x, 4)
    with pytest.raises(RuntimeError, match="must be a contiguous tensor"):
        E.furthest_point_sampling(torch.rand(1, 3, 10).cuda().transpose(1, 2), 4)
    with pytest.raises(RuntimeError, match="must be a float tensor"):
        E.ball_query(x.cuda().double(), x.cuda(), 0.1, 4)
    with pytest.raises(RuntimeError, match="must be an int tensor"):
        E.group_points(torch.rand(1, 3, 10).cuda(), torch.zeros(1, 2, 2, dtype=torch.int64).cuda())
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        E.ball_query(x.cuda(), x, 0.1, 4)


@pytest.mark.parametrize("policy", ["contracted", "contracted_b"])
def test_operators_under_each_evaluation_policy(E, oracle, model_oracle, policy):
    """Round-4 VERDICT item 3.  The reference's own binary is built with nvcc's default --fmad=true (pointnet2/setup.py:24-27),
    so it CONTRACTS the index-deciding expressions (sampling_gpu.cu:108-109, ball_query_gpu.cu:36-37,
    cylinder_query_gpu.cu:39-45, interpolate_gpu.cu:37, knn.cu:89-90); which order it fuses them in cannot be observed here.
    Both plausible orders are therefore switchable on both sides (gn_set_fp_policy / oracle.set_fp_policy; every other test
    runs the default, as written): HIP == oracle BIT-EXACT under each -- every sampler kernel family (register, region, cell),
    scan and grid queries, the de-duplicated cylinder lists, three_nn's returned squared distances, knn."""
    import ctypes
    from graspnet_amd import _lib as L, knn_ext, set_fp_policy
    prev_h, prev_o = set_fp_policy(policy), oracle.set_fp_policy(policy)
    try:
        assert L.lib().gn_get_fp_policy() == L.FP_POLICIES[policy]
        moved = 0
        for n, m, variant in ((20000, 600, "uniform"), (2048, 1024, "uniform"), (5000, 300, "surface"), (50000, 200, "uniform"),
                              (13000, 100, "surface")):
            x = _cloud(100 + n, 2, n, variant)
            want = oracle.furthest_point_sampling(x, m)
            assert torch.equal(E.furthest_point_sampling(_dev(x), m).cpu(), want), ("fps", n)
        x = _cloud(220, 2, 20000, "uniform")
        c = x[:, :2048].contiguous()
        rng = np.random.default_rng(3)
        rot = model_oracle.viewpoint_to_matrix(torch.from_numpy(rng.standard_normal((2 * 1024, 3)).astype(np.float32)),
                                               torch.zeros(2 * 1024)).view(2, 1024, 9).contiguous()
        seeds = c[:, :1024].contiguous()
        for grid in ("grid", "brute"):
            os.environ["GN_QUERY_PATH"] = grid
            assert torch.equal(E.ball_query(_dev(c), _dev(x), 0.04, 64).cpu(), oracle.ball_query(c, x, 0.04, 64)), grid
            for hmax in (0.01, 0.04):
                got = E.cylinder_query(_dev(seeds), _dev(x), _dev(rot), 0.05, -0.02, hmax, 64).cpu()
                assert torch.equal(got, oracle.cylinder_query(seeds, x, rot, 0.05, -0.02, hmax, 64)), (grid, hmax)
        del os.environ["GN_QUERY_PATH"]
        # the four nested cylinders as one de-duplicated list (fused path), grid and scan, against the oracle's four lists
        lib, st = L.lib(), torch.cuda.current_stream().cuda_stream
        hm = (ctypes.c_float * 4)(0.01, 0.02, 0.03, 0.04)
        xd, sd, rd = x.cuda(), seeds.cuda(), rot.cuda()
        ws = torch.empty((lib.gn_grid_workspace_bytes(2, 20000),), dtype=torch.uint8, device="cuda")
        L.check(lib.gn_grid_build(xd.data_ptr(), 2, 20000, 0.04, ws.data_ptr(), st), "grid")
        want4 = [oracle.cylinder_query(seeds, x, rot, 0.05, -0.02, h, 64) for h in (0.01, 0.02, 0.03, 0.04)]
        for form in ("grid", "scan"):
            u = torch.zeros((2, 1024, 256), dtype=torch.int32, device="cuda")
            cnt = torch.zeros((2, 1024), dtype=torch.int32, device="cuda")
            if form == "grid":
                L.check(lib.gn_cylinder_query_unique_grid(sd.data_ptr(), rd.data_ptr(), ws.data_ptr(), 2, 20000, 1024, 0.05, -0.02, hm,
                                                          4, 64, u.data_ptr(), cnt.data_ptr(), st), "ug")
            else:
                L.check(lib.gn_cylinder_query_unique(sd.data_ptr(), xd.data_ptr(), rd.data_ptr(), 2, 20000, 1024, 0.05, -0.02, hm, 4,
                                                     64, u.data_ptr(), cnt.data_ptr(), st), "us")
            u, cnt = u.cpu(), cnt.cpu()
            for b in range(2):
                for j in range(0, 1024, 37):
                    ent = u[b, j, :int(cnt[b, j])]
                    for d in range(4):
                        mine = sorted((ent[(ent >> 28) & (1 << d) != 0] & 0x0fffffff).tolist())
                        ref = want4[d][b, j].tolist()
                        assert mine == sorted(set(ref)), (form, b, j, d)
        un, kn = _cloud(1, 2, 1024), _cloud(2, 2, 512)
        d2w, iw = oracle.three_nn(un, kn)
        d2, i3 = E.three_nn(_dev(un), _dev(kn))
        assert torch.equal(i3.cpu(), iw) and torch.equal(d2.cpu(), d2w)
        for nr, nq, k in ((300, 300, 1), (5000, 1024, 3), (700, 70, 12)):
            r, q = torch.randn(2, 3, nr), torch.randn(2, 3, nq)
            want = torch.zeros(2, k, nq, dtype=torch.int64)
            oracle.knn(r, q, want)
            got = torch.zeros(2, k, nq, dtype=torch.int64, device="cuda")
            knn_ext.knn(r.cuda(), q.cuda(), got)
            assert torch.equal(got.cpu(), want), (nr, nq, k)
        # and the policy is not a no-op: three_nn's distances differ from the as-written ones somewhere
        set_fp_policy("as_written")
        d2_plain, _ = E.three_nn(_dev(un), _dev(kn))
        moved = int((d2_plain.cpu() != d2w).sum())
        assert moved > 0
    finally:
        set_fp_policy(prev_h)
        oracle.set_fp_policy(prev_o)
