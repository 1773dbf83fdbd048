"""GPU: randomised shape sweeps of the round-4 kernels against their CPU restatements -- the cases the hand-picked tests do not
think of (ragged tails, tiny and odd sizes, many duplicates, counts of zero).  Seeds are fixed: a failure reproduces."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_knn_random_shapes(oracle):
    from graspnet_amd import knn_ext
    rng = np.random.default_rng(2024)
    for it in range(40):
        B = int(rng.integers(1, 4)); D = int(rng.integers(1, 6)); Nr = int(rng.integers(1, 700)); Nq = int(rng.integers(1, 900))
        k = int(rng.integers(1, 13))
        r = rng.standard_normal((B, D, Nr)).astype(np.float32)
        q = rng.standard_normal((B, D, Nq)).astype(np.float32)
        if it % 3 == 0:                     # coarse coordinates: many exactly equal distances
            r, q = np.round(r * 2) / 2, np.round(q * 2) / 2
        r, q = torch.from_numpy(r), torch.from_numpy(q)
        want = torch.zeros(B, k, Nq, dtype=torch.int64)
        oracle.knn(r, q, want)
        got = torch.full((B, k, Nq), -3, dtype=torch.int64, device="cuda")
        knn_ext.knn(r.cuda(), q.cuda(), got)
        assert torch.equal(got.cpu(), want), (it, B, D, Nr, Nq, k)


def test_voxel_down_sample_random_clouds():
    from graspnet_amd import postprocess as P
    from oracle import post_oracle as PO
    rng = np.random.default_rng(77)
    for it in range(25):
        B = int(rng.integers(1, 5)); n = int(rng.integers(1, 3000))
        voxel = float(rng.choice([0.002, 0.005, 0.01, 0.05]))
        scale = float(rng.choice([0.02, 0.3, 3.0]))
        pts = rng.uniform(-scale, scale, size=(B, n, 3)).astype(np.float32)
        if it % 4 == 0:
            pts = (np.round(pts / 0.004) * 0.004).astype(np.float32)          # many exact duplicates
        if it % 5 == 0:
            pts[:, rng.integers(0, n, size=max(1, n // 50))] = np.nan
        nv = rng.integers(0, n + 1, size=B).astype(np.int32)
        out, cnt, status = P.voxel_down_sample_batched(torch.from_numpy(pts).cuda(), voxel, n_valid=torch.from_numpy(nv).cuda())
        assert not status.any()
        o, c = out.cpu().numpy(), cnt.cpu().numpy()
        for b in range(B):
            want = PO.voxel_down_sample_ordered(pts[b], voxel, n_valid=nv[b])
            assert c[b] == len(want) and np.array_equal(o[b, :c[b]], want), (it, b, n, voxel)
            assert not o[b, c[b]:].any()


def test_collision_detect_random_grasps():
    """random poses and gripper sizes incl. degenerate ones, scene points at random scales and offsets: the float32 pre-test
    must never lose a pair the float64 evaluation counts"""
    from graspnet_amd import _lib as L
    from oracle import post_oracle as PO
    rng = np.random.default_rng(5)
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    for it in range(20):
        B = int(rng.integers(1, 4)); npmax = int(rng.integers(1, 4000)); M = int(rng.integers(1, 60))
        off = rng.uniform(-5, 5, size=(B, 1, 3)) if it % 2 else np.zeros((B, 1, 3))
        scene = rng.uniform(-0.15, 0.15, size=(B, npmax, 3)) + off
        nps = rng.integers(0, npmax + 1, size=B).astype(np.int32)
        g = np.zeros((B, M, 17), np.float32)
        g[:, :, 0] = rng.random((B, M))
        g[:, :, 1] = rng.uniform(0.0, 0.12, size=(B, M))            # width (0 included)
        g[:, :, 2] = rng.choice([0.02, 0.0, 0.05], size=(B, M))       # height
        g[:, :, 3] = rng.choice([0.01, 0.02, 0.03, 0.04], size=(B, M))
        A = rng.standard_normal((B, M, 3, 3))
        Q, _ = np.linalg.qr(A)
        g[:, :, 4:13] = Q.reshape(B, M, 9).astype(np.float32)
        ctr = rng.integers(0, npmax, size=(B, M))
        g[:, :, 13:16] = (np.take_along_axis(scene, ctr[:, :, None], 1) + rng.normal(0, 0.01, size=(B, M, 3))).astype(np.float32)
        counts = rng.integers(0, M + 1, size=B).astype(np.int32)
        # adversarial scene points: on / just inside / just outside the box faces of the first grasps (float64 placement in the
        # grasp frame): the conservative float32 pre-test may not lose the ones the float64 evaluation counts
        for b in range(B):
            for gi in range(min(M, 6)):
                w_, h_, d_ = float(g[b, gi, 1]), float(g[b, gi, 2]), float(g[b, gi, 3])
                R = g[b, gi, 4:13].astype(np.float64).reshape(3, 3)
                T = g[b, gi, 13:16].astype(np.float64)
                for face, t in (("xd", [d_, 0, 0]), ("xs", [d_ - 0.06 - 0.01 - max(ad if False else 0.05, 0.01), 0, 0]),
                                ("yl", [d_ - 0.03, -(w_ / 2 + 0.01), 0]), ("yr", [d_ - 0.03, w_ / 2 + 0.01, 0]), ("z", [d_ - 0.03, 0, h_ / 2])):
                    for delta in (0.0, 1e-9, -1e-9, 1e-7, -1e-7, 3e-6, -3e-6):
                        tt = np.array(t, np.float64)
                        tt[np.argmax(np.abs(tt)) if face[0] != "z" else 2] += delta
                        slot = int(rng.integers(0, npmax))
                        scene[b, slot] = T + R @ tt          # targets = (p - T) @ R  ->  p = T + R t
        sd, gd = torch.from_numpy(scene).cuda(), torch.from_numpy(g).cuda()
        npd, cd = torch.from_numpy(nps).cuda(), torch.from_numpy(counts).cuda()
        cm = torch.empty((B, M), dtype=torch.uint8, device="cuda"); em = torch.empty_like(cm)
        io = torch.empty((B, M, 5), device="cuda")
        ad, ct = float(rng.choice([0.05, 0.01, 0.005])), float(rng.choice([0.01, 0.05]))
        L.check(lib.gn_collision_detect_batched(sd.data_ptr(), npmax, npd.data_ptr(), gd.data_ptr(), cd.data_ptr(), B, M, ad, ct, 0.01,
                                                0.01, 0.01, 0.06, cm.data_ptr(), em.data_ptr(), io.data_ptr(), st), "collision")
        cm, em, io = cm.cpu().numpy().astype(bool), em.cpu().numpy().astype(bool), io.cpu().numpy()
        for b in range(B):
            k = int(counts[b])
            with np.errstate(all="ignore"):
                wcm, wem, wio = PO.collision_detect(scene[b, :nps[b]], g[b, :k], approach_dist=ad, collision_thresh=ct, voxel_size=0.01)
                assert np.array_equal(cm[b, :k], wcm) and np.array_equal(em[b, :k], wem), (it, b)
                for j in range(5):
                    assert np.array_equal(io[b, :k, j], wio[j].astype(np.float32), equal_nan=True), (it, b, j)
            assert not cm[b, k:].any() and not em[b, k:].any()


def test_front_end_random_images():
    from graspnet_amd import postprocess as P
    rng = np.random.default_rng(31)
    for it in range(12):
        B = int(rng.integers(1, 4)); h = int(rng.integers(1, 90)); w = int(rng.integers(1, 130))
        depth = rng.integers(0, 4000, size=(B, h, w), dtype=np.uint16)
        depth[rng.random((B, h, w)) < rng.uniform(0, 0.9)] = 0
        mask = (rng.random((B, h, w)) < rng.uniform(0.05, 1.0)).astype(np.uint8)
        dmin, dmax = (0.0, np.inf) if it % 2 else (0.4, 2.5)
        vp, cnt = P.valid_pixels(torch.from_numpy(depth.view(np.int16)).cuda(), torch.from_numpy(mask).cuda(), 1000.0, dmin, dmax)
        for b in range(B):
            z = depth[b] / 1000.0
            m = (mask[b] != 0) & (depth[b] > 0) & (z > dmin) & (z < dmax)
            valid = np.nonzero(m.reshape(-1))[0]
            assert int(cnt[b]) == len(valid) and np.array_equal(vp[b, :len(valid)].cpu().numpy(), valid), (it, b, h, w)
