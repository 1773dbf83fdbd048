"""CPU: definitional properties of the oracle operators (independent brute-force numpy checks),
and the oracle's knn against the reference's own knn CPU build when oracle/_ref is present."""
import ctypes
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pts(seed, b, n):
    rng = np.random.default_rng(seed)
    p = rng.uniform(-0.3, 0.3, size=(b, n, 3)).astype(np.float32)
    p[..., 2] += 0.5
    return torch.from_numpy(p)


def test_fps_prefix_and_greedy(oracle):
    x = _pts(0, 2, 700)
    f64 = oracle.furthest_point_sampling(x, 64).numpy()
    f32 = oracle.furthest_point_sampling(x, 32).numpy()
    assert np.array_equal(f64[:, :32], f32)            # prefix property
    assert np.all(f64[:, 0] == 0)
    # greedy check in float64 (ties are measure-zero for random data)
    for b in range(2):
        p = x[b].numpy().astype(np.float64)
        d = np.full(700, 1e10)
        cur = 0
        for j in range(1, 64):
            d = np.minimum(d, ((p - p[cur]) ** 2).sum(1))
            cur = int(np.argmax(d))
            assert cur == f64[b, j]


def test_fps_tree_tie_order(oracle):
    """All candidate distances equal -> the winner is decided purely by the reference's tree:
    among exact ties the thread with the smallest LSB-first (bit-reversed) id wins."""
    n = 1024
    x = torch.zeros(1, n, 3)
    x[0, :, 0] = 1.0          # everything at (1,0,0) except ...
    x[0, 0] = torch.tensor([0.0, 0.0, 1.0])   # ... the start point; all d2 equal (=2)
    got = oracle.furthest_point_sampling(x, 3).numpy()[0]
    # candidates k=1..1023 all tie; tid = k % 512; bitrev9 minimal non-excluded: tid 0 holds k=0 (d=0)
    # and k=512 (d=2) -> thread 0's best is k=512, and slot 0 wins every tree level.
    assert got[1] == 512
    # after picking 512, all other points have d2 = min(2, 0) = 0 -> all tie at 0 -> thread 0: k=0 first
    # (strict '>' from best=-1 takes the first k of the thread).
    assert got[2] == 0


def test_fps_skips_near_origin_points(oracle):
    x = _pts(1, 1, 600)
    x[0, 17] = 0.0
    x[0, 300] = torch.tensor([0.02, 0.02, 0.01])   # |p|^2 = 9e-4 <= 1e-3
    sel = oracle.furthest_point_sampling(x, 598).numpy()[0]
    assert 17 not in sel[1:] and 300 not in sel[1:]


def test_ball_query_definition(oracle):
    x = _pts(2, 2, 500)
    c = x[:, :40].contiguous()
    r, ns = 0.08, 12
    idx = oracle.ball_query(c, x, r, ns).numpy()
    r2 = np.float32(r) * np.float32(r)
    for b in range(2):
        for j in range(40):
            d = x[b].numpy() - c[b, j].numpy()
            # float32 evaluation in the reference's order
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            hits = np.nonzero(d2 < r2)[0][:ns]
            want = np.zeros(ns, np.int32)
            if len(hits):
                want[:] = hits[0]
                want[:len(hits)] = hits
            assert np.array_equal(idx[b, j], want)


def test_ball_query_empty_rows_are_zero(oracle):
    x = _pts(3, 1, 100)
    c = (x[:, :5] + 10.0).contiguous()
    assert not oracle.ball_query(c, x, 0.01, 8).numpy().any()


def test_cylinder_query_definition(oracle, model_oracle):
    x = _pts(4, 1, 800)
    c = x[:, :30].contiguous()
    rng = np.random.default_rng(4)
    rot = model_oracle.viewpoint_to_matrix(torch.from_numpy(rng.standard_normal((30, 3)).astype(np.float32)),
                                           torch.zeros(30)).view(1, 30, 9).contiguous()
    idx = oracle.cylinder_query(c, x, rot, 0.05, -0.02, 0.04, 16).numpy()[0]
    for j in range(30):
        R = rot[0, j].view(3, 3).numpy().astype(np.float64)
        d = (x[0].numpy() - c[0, j].numpy()).astype(np.float64) @ R
        ok = (d[:, 1] ** 2 + d[:, 2] ** 2 < 0.05 ** 2 - 1e-9) & (d[:, 0] > -0.02 + 1e-7) & (d[:, 0] < 0.04 - 1e-7)
        for k in np.nonzero(ok)[0][:4]:       # clear-cut members that come early must be present
            if k in idx[j] or len(np.nonzero(ok)[0]) > 16:
                continue
            raise AssertionError("cylinder member missing")


def test_three_nn_and_interpolate(oracle):
    u, k = _pts(5, 2, 64), _pts(6, 2, 33)
    d2, idx = oracle.three_nn(u, k)
    full = ((u[:, :, None, :] - k[:, None, :, :]) ** 2).sum(-1)
    order = torch.argsort(full, dim=2, stable=True)[:, :, :3]
    assert torch.equal(order.int(), idx)
    w = torch.rand(2, 64, 3)
    feats = torch.rand(2, 7, 33)
    out = oracle.three_interpolate(feats, idx, w)
    want = sum(torch.gather(feats, 2, idx[:, :, i].long().unsqueeze(1).expand(-1, 7, -1)) * w[:, :, i].unsqueeze(1)
               for i in range(3))
    assert torch.allclose(out, want, atol=1e-6)


def test_group_and_gather_are_copies(oracle):
    f = torch.rand(2, 5, 50)
    idx = torch.randint(0, 50, (2, 9, 4), dtype=torch.int32)
    g = oracle.group_points(f, idx)
    assert torch.equal(g, torch.gather(f.unsqueeze(2).expand(-1, -1, 9, -1), 3,
                                       idx.long().unsqueeze(1).expand(-1, 5, -1, -1)))
    gi = torch.randint(0, 50, (2, 11), dtype=torch.int32)
    assert torch.equal(oracle.gather_points(f, gi), torch.gather(f, 2, gi.long().unsqueeze(1).expand(-1, 5, -1)))


def test_grad_ops_match_autograd(oracle):
    f = torch.rand(2, 3, 20, requires_grad=True)
    idx = torch.randint(0, 20, (2, 6, 4), dtype=torch.int32)
    go = torch.rand(2, 3, 6, 4)
    torch.gather(f.unsqueeze(2).expand(-1, -1, 6, -1), 3, idx.long().unsqueeze(1).expand(-1, 3, -1, -1)).backward(go)
    assert torch.allclose(oracle.group_points_grad(go, idx, 20), f.grad, atol=1e-6)


def test_opt_n_threads_integer_form(oracle):
    """The HIP host computes floor(log2 n) with integers; the reference uses log()/log(2.0)
    (cuda_utils.h:18-24).  They must agree for every n."""
    lib = oracle.lib()
    for n in list(range(1, 5000)) + [2 ** k for k in range(1, 24)] + [2 ** k - 1 for k in range(2, 24)] + [20000, 50000]:
        p = n.bit_length() - 1
        assert lib.gno_opt_n_threads(n) == max(min(1 << p, 512), 1), n


def test_knn_against_reference_build(oracle):
    so = os.path.join(ROOT, "oracle", "_ref", "knn_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/knn_ref.so not built (reference tree absent)")
    ref = ctypes.CDLL(so)
    rng = np.random.default_rng(0)
    L = ctypes.c_long
    for (B, D, Nr, Nq, k) in [(2, 3, 300, 300, 1), (1, 3, 500, 1024, 3), (2, 5, 64, 33, 7), (1, 3, 40, 10, 40)]:
        r = torch.from_numpy(rng.standard_normal((B, D, Nr)).astype(np.float32))
        q = torch.from_numpy(rng.standard_normal((B, D, Nq)).astype(np.float32))
        r[:, :, Nr // 2:] = r[:, :, :Nr - Nr // 2]      # duplicates -> exact ties
        i1 = torch.zeros(B, k, Nq, dtype=torch.int64)
        i2 = torch.zeros(B, k, Nq, dtype=torch.int64)
        oracle.knn(r, q, i1)
        ref.knn_ref(ctypes.c_void_p(r.data_ptr()), ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(i2.data_ptr()),
                    L(B), L(D), L(k), L(Nr), L(Nq))
        assert torch.equal(i1, i2)
        assert i1.min() >= 1        # 1-based


def test_knn_topk_property(oracle):
    r, q = torch.rand(1, 3, 200), torch.rand(1, 3, 50)
    idx = torch.zeros(1, 4, 50, dtype=torch.int64)
    oracle.knn(r, q, idx)
    d = ((r[0].T[None] - q[0].T[:, None]) ** 2).sum(-1)
    want = torch.argsort(d, dim=1, stable=True)[:, :4].T + 1
    assert torch.equal(idx[0], want)


def _round_f32(fr):
    """round-to-nearest-even of an exact Fraction to float32 (no double rounding)"""
    from fractions import Fraction
    import math
    if fr == 0:
        return np.float32(0.0)
    sign = -1 if fr < 0 else 1
    a = abs(fr)
    e = math.floor(math.log2(float(a)))
    while Fraction(2) ** e > a:
        e -= 1
    while Fraction(2) ** (e + 1) <= a:
        e += 1
    e = max(e, -126)                               # subnormals share the exponent -126
    q = a / (Fraction(2) ** (e - 23))              # in units of the last place
    n = q.numerator // q.denominator
    rem = q - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
        n += 1
    return np.float32(sign * float(Fraction(n) * Fraction(2) ** (e - 23)))


def test_evaluation_policies_are_what_they_say(oracle):
    """the oracle's three evaluation policies of a*a + b*b + c*c (gn_oracle_fp.inc) against EXACT rational arithmetic with one
    explicit round-to-nearest-even per operation: policy 0 rounds every product and sum, policy 1 = fma(c,c,fma(a,a,b*b)),
    policy 2 = fma(c,c,fma(b,b,a*a)) -- through three_nn's squared distances (the one operator that RETURNS the value) and
    knn's accumulation s = fma(t,t,s); and the policies do differ on some of these inputs (else the switch tests nothing)"""
    from fractions import Fraction as F
    rng = np.random.default_rng(11)
    u = torch.from_numpy(rng.uniform(-0.3, 0.3, (1, 40, 3)).astype(np.float32))
    k = torch.from_numpy(rng.uniform(-0.3, 0.3, (1, 1, 3)).astype(np.float32))      # one known point: dist2[:, :, 0] is its distance

    def exact(a, b, c, pol):
        a, b, c = F(float(a)), F(float(b)), F(float(c))
        r = _round_f32
        if pol == 0:
            return r(F(float(r(F(float(r(a * a))) + F(float(r(b * b)))))) + F(float(r(c * c))))
        if pol == 1:
            return r(c * c + F(float(r(a * a + F(float(r(b * b)))))))
        return r(c * c + F(float(r(b * b + F(float(r(a * a)))))))
    got = {}
    try:
        for name, pol in (("as_written", 0), ("contracted", 1), ("contracted_b", 2)):
            oracle.set_fp_policy(name)
            d2, _ = oracle.three_nn(u, k)
            got[pol] = d2[0, :, 0].numpy().copy()
            for j in range(40):
                dx, dy, dz = (np.float32(u[0, j, i]) - np.float32(k[0, 0, i]) for i in range(3))
                assert got[pol][j] == exact(dx, dy, dz, pol), (name, j)
    finally:
        oracle.set_fp_policy("as_written")
    assert (got[0] != got[1]).any() and (got[1] != got[2]).any()
