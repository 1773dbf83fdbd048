"""Shared arithmetic of the model-level parity tests and of profiles/parity_report.py.

THE RULE (BASELINE.json north_star: "within 1e-4 on grasp scores"), applied to every float key of `end_points`
and to the decoded rows:

    |hip - oracle|  <=  1e-4 * max(1, ||oracle key||_inf)

i.e. absolute 1e-4 for keys whose values stay within [-1, 1], and 1e-4 of the key's largest magnitude where they do
not: an fp32 result of magnitude S carries an absolute noise floor of a few S * 2^-24 per operation, and the network
is 15-18 layers deep, so an absolute 1e-4 on a key whose entries reach the hundreds (SA / FP features, view scores) is below
what ANY fp32 evaluation order can promise -- the torch-CPU oracle itself sits that far from a float64 evaluation
(measured, see profiles/parity_r02.json: "oracle_vs_f64").  Seeds whose discrete top-view choice flipped on a near-tie
see a different rotation downstream and are accounted separately (counted, bounded, checked to be near-ties).
"""
import numpy as np
import torch

TOL = 1e-4          # north_star
FLOAT_KEYS = ("sa1_features", "sa2_features", "sa3_features", "sa4_features", "fp2_features", "objectness_score",
              "view_score", "grasp_top_view_score", "grasp_score_pred", "grasp_angle_cls_pred", "grasp_width_pred",
              "grasp_tolerance_pred")
HEAD_KEYS = ("grasp_score_pred", "grasp_angle_cls_pred", "grasp_width_pred", "grasp_tolerance_pred")
INDEX_KEYS = ("sa1_inds", "sa2_inds", "fp2_inds", "sa1_xyz", "sa2_xyz", "sa3_xyz", "sa4_xyz", "fp2_xyz")


def limit(oracle_t):
    """The tolerance of THE RULE for one key."""
    return TOL * max(1.0, float(oracle_t.detach().abs().max()))


def seed_mask_for(key, same_view):
    """Boolean mask broadcastable to `key`'s tensor selecting the seeds whose top view did not flip
    (only keys downstream of the view choice depend on it)."""
    if key in HEAD_KEYS:                   # (B,12,M,D)
        return same_view[:, None, :, None]
    return None


def key_errors(got, want, mask=None):
    """max / rms of |got - want| (float64), the key's scale, and where the maximum sits."""
    g, w = got.detach().double().cpu(), want.detach().double().cpu()
    assert g.shape == w.shape, (g.shape, w.shape)
    e = (g - w).abs()
    if mask is not None:
        m = mask.expand_as(e)
        e = torch.where(m, e, torch.zeros_like(e))
        n = int(m.sum())
    else:
        n = e.numel()
    # the per-element picture behind the norm-relative form of THE RULE: the error at elements that are themselves small
    # (|oracle| < 1, where "1e-4 absolute" and "1e-4 of the element" coincide), and the largest error relative to the
    # element's own magnitude max(1, |oracle element|)
    small = w.abs() < 1.0
    if mask is not None:
        small = small & mask.expand_as(e)
    return {"max": float(e.max()), "rms": float((e.pow(2).sum() / max(n, 1)).sqrt()), "scale": float(w.abs().max()),
            "n": n, "small_n": int(small.sum()), "small_max": float(e[small].max()) if small.any() else 0.0,
            "elementwise_rel_max": float((e / w.abs().clamp(min=1.0)).max())}


def compare_end_points(ep, rep, tru=None):
    """-> {key: {"vs_oracle": {...}, "limit": L, "ok": bool [, "vs_f64", "oracle_vs_f64"]}}, flips info."""
    gi, ri = ep["grasp_top_view_inds"].cpu(), rep["grasp_top_view_inds"]
    same = gi == ri
    out = {}
    for k in FLOAT_KEYS:
        m = seed_mask_for(k, same)
        e = key_errors(ep[k], rep[k], m)
        lim = limit(rep[k])
        rec = {"vs_oracle": e, "limit": lim, "ok": e["max"] <= lim}
        if tru is not None:
            rec["vs_f64"] = key_errors(ep[k], tru[k], m)
            rec["oracle_vs_f64"] = key_errors(rep[k], tru[k], m)
        out[k] = rec
    vs = rep["view_score"]
    gap = (torch.gather(vs, 2, ri.unsqueeze(-1)) - torch.gather(vs, 2, gi.unsqueeze(-1))).squeeze(-1)
    flips = {"top_view": int((~same).sum()), "seeds": int(same.numel()),
             "max_gap_in_oracle_scores": float(gap[~same].abs().max()) if (~same).any() else 0.0}
    return out, flips, same


def rows_by_seed(pred):
    return {tuple(r[13:16].tolist()): r for r in pred}


def compare_rows(preds, rpreds, tpreds=None):
    """pred_decode rows matched by seed centre.  A row whose rotation / depth differs is a discrete flip (top view,
    angle class or depth class decided the other way on a near-tie); a centre present on one side only is an
    objectness flip.  Scores and widths are compared on the rest."""
    st = {"rows": 0, "class_flips": 0, "objectness_flips": 0, "score_max": 0.0, "score_sq": 0.0, "width_max": 0.0,
          "score_scale": 0.0, "hip_vs_f64_max": 0.0, "oracle_vs_f64_max": 0.0, "hip_vs_f64_sq": 0.0,
          "oracle_vs_f64_sq": 0.0, "rows_f64": 0}
    for b in range(len(rpreds)):
        a, r = preds[b].detach().cpu(), rpreds[b]
        ka, kr = rows_by_seed(a), rows_by_seed(r)
        kt = {tuple(t[13:16].float().tolist()): t for t in tpreds[b]} if tpreds is not None else {}   # rows stay float64
        st["objectness_flips"] += len(set(ka) ^ set(kr))
        for key in set(ka) & set(kr):
            x, y = ka[key], kr[key]
            if not torch.allclose(x[4:13], y[4:13], atol=1e-5) or float(x[3]) != float(y[3]):
                st["class_flips"] += 1
                continue
            st["rows"] += 1
            d = abs(float(x[0]) - float(y[0]))
            st["score_max"] = max(st["score_max"], d)
            st["score_sq"] += d * d
            st["score_scale"] = max(st["score_scale"], abs(float(y[0])))
            st["width_max"] = max(st["width_max"], abs(float(x[1]) - float(y[1])))
            z = kt.get(key)
            if z is not None and torch.allclose(z[4:13].float(), y[4:13], atol=1e-5) and float(z[3].float()) == float(y[3]):
                zt = float(z[0])
                st["rows_f64"] += 1
                eh, eo = abs(float(x[0]) - zt), abs(float(y[0]) - zt)
                st["hip_vs_f64_max"] = max(st["hip_vs_f64_max"], eh)
                st["oracle_vs_f64_max"] = max(st["oracle_vs_f64_max"], eo)
                st["hip_vs_f64_sq"] += eh * eh
                st["oracle_vs_f64_sq"] += eo * eo
    n = max(st["rows"], 1)
    st["score_rms"] = (st.pop("score_sq") / n) ** 0.5
    nf = max(st["rows_f64"], 1)
    st["hip_vs_f64_rms"] = (st.pop("hip_vs_f64_sq") / nf) ** 0.5
    st["oracle_vs_f64_rms"] = (st.pop("oracle_vs_f64_sq") / nf) ** 0.5
    st["score_limit"] = TOL * max(1.0, st["score_scale"])
    return st


def golden_head_sample_check(ep, g, same_view, key, stride=97):
    """Reference-python golden sample of a head output (tests/golden/make_golden.py: contiguous (B,12,M,D) tensor,
    every 97th element) against the HIP tensor; elements of seeds whose top view flipped are skipped.
    -> (max err, n compared, n skipped, scale)"""
    want = np.asarray(g["model/%s/sample" % key])
    shape = tuple(int(v) for v in g["model/%s/shape" % key])
    got_full = ep[key].contiguous().reshape(-1).cpu().numpy()
    got = got_full[::stride]
    assert got.shape == want.shape, (got.shape, want.shape)
    flat = np.arange(0, got_full.size, stride)
    b, _, s, _ = np.unravel_index(flat, shape)
    keep = same_view.numpy()[b, s]
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    return (float(err[keep].max()) if keep.any() else 0.0), int(keep.sum()), int((~keep).sum()), float(np.abs(want).max())


# ---- the assertions every model-level case runs (tests/test_model_gpu.py) ------------------------------------------
# float64 yardstick: the HIP path must be as close to exact arithmetic as the reference's own fp32 (torch-CPU) evaluation.
# Measured (profiles/parity_r02.json; golden batch, config[1] at B=8, 8 surface clouds), error vs float64 relative to the
# oracle's own error vs float64, per key and on the decoded score:
#   GN_MATH=split (default): rms 0.68-0.92x, max 0.58-1.11x   -> must not exceed the oracle's rms at all, max within 1.5x
#                            (a maximum over ~10^6 entries of two independent noise fields differs that much by chance);
#   GN_MATH=f32 (A/B path, k-ordered fp32 FMA chains = K/2 roundings per output): rms 0.86-1.44x, max up to 1.6x.
# Round 1 sat at 1.2-1.6x on the default path: every one of the six split products rounded at the accumulator's full
# magnitude; mlp_split.h mfma6_2acc (large products in one accumulator, small ones in another) removed that.
F64_RMS_FACTOR = {"split": 1.0, "f32": 1.6}
F64_MAX_FACTOR = {"split": 1.5, "f32": 2.0}
# regression guard on the distance to the oracle itself: two fp32 evaluations that are each within e of float64 are within
# ~2e of each other; measured max|hip - oracle| = 1.0-1.6 x max|oracle - float64| per key (profiles/parity_r02.json), so
# 3x binds an order of magnitude below THE RULE's norm-relative limit on the large keys
VS_ORACLE_OVER_ORACLE_F64 = 3.0
HEAD_ABS_TOL = 1e-4      # raw network outputs (grasp_score_pred, ...): plain absolute 1e-4, measured 1.5e-5
MAX_VIEW_FLIP_GAP = 2e-4  # a flipped top view must be a near-tie in the oracle's own scores (scale ~10)

REPORT = {}              # case name -> record; tests/conftest.py writes it to gpurun_out/ at session end


def assert_parity(name, ep, preds, rep, rpreds, tru=None, tpreds=None, math="split"):
    """Index keys bit-exact; THE RULE on every float key (+ absolute 1e-4 on the raw head outputs); flips bounded and
    near-ties; decoded rows under THE RULE; float64 yardstick when `tru` is given.  Records what it measured."""
    for k in INDEX_KEYS:
        assert torch.equal(ep[k].cpu(), rep[k]), "%s: %s differs from the oracle" % (name, k)
    keys, flips, same = compare_end_points(ep, rep, tru)
    rows = compare_rows(preds, rpreds, tpreds)
    REPORT[name] = {"math": math, "keys": keys, "flips": flips, "rows": rows}
    seeds = flips["seeds"]
    assert flips["top_view"] <= max(2, seeds // 1000), "%s: %d top-view flips of %d" % (name, flips["top_view"], seeds)
    assert flips["max_gap_in_oracle_scores"] <= MAX_VIEW_FLIP_GAP, (name, flips)
    for k, v in keys.items():
        assert v["ok"], "%s: %s max |hip-oracle| %.3e > limit %.3e (scale %.3f)" % (name, k, v["vs_oracle"]["max"], v["limit"],
                                                                                   v["vs_oracle"]["scale"])
        if k in HEAD_KEYS:
            assert v["vs_oracle"]["max"] <= HEAD_ABS_TOL, "%s: %s max |hip-oracle| %.3e > 1e-4 absolute" % (name, k, v["vs_oracle"]["max"])
        if tru is not None:
            assert v["vs_oracle"]["max"] <= VS_ORACLE_OVER_ORACLE_F64 * v["oracle_vs_f64"]["max"] + 1e-8, \
                "%s: %s max |hip-oracle| %.3e > 3 x the oracle's own distance from float64 %.3e" % (
                    name, k, v["vs_oracle"]["max"], v["oracle_vs_f64"]["max"])
            fr, fm = F64_RMS_FACTOR[math], F64_MAX_FACTOR[math]
            assert v["vs_f64"]["rms"] <= fr * v["oracle_vs_f64"]["rms"] + 1e-9, \
                "%s: %s rms error vs float64 %.3e, the oracle's own %.3e" % (name, k, v["vs_f64"]["rms"], v["oracle_vs_f64"]["rms"])
            assert v["vs_f64"]["max"] <= fm * v["oracle_vs_f64"]["max"] + 1e-8, \
                "%s: %s max error vs float64 %.3e, the oracle's own %.3e" % (name, k, v["vs_f64"]["max"], v["oracle_vs_f64"]["max"])
    total = rows["rows"] + rows["class_flips"]
    assert rows["rows"] >= 0.95 * sum(int(r.shape[0]) for r in rpreds) - 2, (name, rows)
    assert rows["class_flips"] <= max(2, total // 500), "%s: %d angle/depth/view class flips of %d rows" % (name, rows["class_flips"], total)
    assert rows["objectness_flips"] <= max(2, seeds // 1000), (name, rows)
    assert rows["score_max"] <= rows["score_limit"], "%s: decoded score max |hip-oracle| %.3e > %.3e" % (name, rows["score_max"], rows["score_limit"])
    assert rows["width_max"] <= 1e-5, (name, rows["width_max"])
    if tpreds is not None:
        assert rows["hip_vs_f64_rms"] <= F64_RMS_FACTOR[math] * rows["oracle_vs_f64_rms"] + 1e-9, \
            "%s: decoded score rms vs float64 %.3e, the oracle's own %.3e" % (name, rows["hip_vs_f64_rms"], rows["oracle_vs_f64_rms"])
        assert rows["hip_vs_f64_max"] <= F64_MAX_FACTOR[math] * rows["oracle_vs_f64_max"] + 1e-8, \
            "%s: decoded score max vs float64 %.3e, the oracle's own %.3e" % (name, rows["hip_vs_f64_max"], rows["oracle_vs_f64_max"])
    return REPORT[name]
