"""Machine-readable parity report: run on the GPU box, writes gpurun_out/parity_<tag>.json which is
then copied to profiles/.

For GN_MATH = split and f32, on config[1] at its own size (B=8 x 20000 uniform clouds, seeds 1000..1007) and on the golden
batch (B=2: uniform + surface):
  * per float key of end_points: max / rms |hip - oracle|, |hip - float64|, |oracle - float64|, the key's scale and THE
    RULE's limit (tests/parity_util.py);
  * decoded rows matched by seed: score / width error vs oracle and vs float64, discrete flip counts;
  * stage-isolated ("teacher forced") errors: every stage fed with the float64 truth's inputs rounded to fp32 and
    compared with the truth's output -- for the HIP kernels (both maths) and for the oracle's torch-CPU stage, so the
    stage that loses accuracy, if any, is named.
"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graspnet-baseline_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import parity_util as PU  # noqa: E402
from graspnet_amd import GraspNet, pred_decode, synthetic as S, _lib as L  # noqa: E402
from graspnet_amd.graspnet import SA_CFG, _ci  # noqa: E402
from oracle import model_oracle as MO, oracle as O  # noqa: E402


def pm(t):      # (B,C,n) truth/oracle tensor -> position-major fp32 on the GPU
    return t.float().transpose(1, 2).contiguous().cuda()


def errs(got, want_cm, posmajor=False):
    """got (B,n,C) position-major (posmajor=True) or (B,C,n) vs float64 truth (B,C,n)."""
    g = got.detach().double().cpu()
    if posmajor:
        g = g.transpose(1, 2)
    e = (g - want_cm.double()).abs()
    return {"max": float(e.max()), "rms": float(e.pow(2).mean().sqrt()), "scale": float(want_cm.abs().max())}


def teacher_forced(net, sd, rep, tru, split):
    """Every stage on the truth's inputs (fp32-rounded): HIP kernels vs truth output; the oracle's stage likewise."""
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    f32, i32 = torch.float32, torch.int32
    B, N = rep["input_xyz"].shape[0], rep["input_xyz"].shape[1]
    out = {}
    pkd = net._packed
    xyz_lv = [rep["input_xyz"]] + [rep["sa%d_xyz" % l] for l in (1, 2, 3, 4)]
    bb = "view_estimator.backbone"
    for li, (npoint, radius, nsample, c, mlp) in enumerate(SA_CFG):
        lv_xyz, new_xyz = xyz_lv[li].cuda(), xyz_lv[li + 1].cuda()
        lv_n = lv_xyz.shape[1]
        idx = torch.empty((B, npoint, nsample), dtype=i32, device="cuda")
        L.check(lib.gn_ball_query(new_xyz.data_ptr(), lv_xyz.data_ptr(), B, lv_n, npoint, radius, nsample, idx.data_ptr(), st), "bq")
        o = torch.empty((B, npoint, mlp[2]), dtype=f32, device="cuda")
        key = "sa%d_features" % (li + 1)
        if c == 0:
            if split:
                pk = pkd["sa1_split"]
                L.check(lib.gn_sa_group_mlp_pool_split(lv_xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), B, lv_n, npoint, nsample,
                                                       radius, _ci(*mlp), pk.wptr, pk.bptr, o.data_ptr(), st), key)
            else:
                pk = pkd["sa1"]
                L.check(lib.gn_sa_group_mlp_pool(lv_xyz.data_ptr(), None, new_xyz.data_ptr(), idx.data_ptr(), B, lv_n, npoint, nsample, 0,
                                                 radius, _ci(*mlp), pk.wptr, pk.bptr, o.data_ptr(), st), key)
            feat_in = None
        else:
            feat_in = tru["sa%d_features" % li].float()
            x = pm(feat_in)
            pfeat = torch.empty((B, lv_n, mlp[0]), dtype=f32, device="cuda")
            if split:
                pre, pk = pkd["sa%d_pre_split" % (li + 1)], pkd["sa%d_hoist_split" % (li + 1)]
                L.check(lib.gn_point_linear_split(x.data_ptr(), B * lv_n, c, mlp[0], pre.w[0].data_ptr(), pre.b[0].data_ptr(),
                                                  pfeat.data_ptr(), st), "lin")
                L.check(lib.gn_sa_hoisted_mlp_pool_split(lv_xyz.data_ptr(), pfeat.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), B, lv_n,
                                                         npoint, nsample, radius, _ci(*mlp), pk.wptr, pk.bptr, o.data_ptr(), st), key)
            else:
                pre, pk = pkd["sa%d_pre" % (li + 1)], pkd["sa%d_hoist" % (li + 1)]
                L.check(lib.gn_point_linear(x.data_ptr(), B * lv_n, c, mlp[0], pre.w[0].data_ptr(), pre.b[0].data_ptr(),
                                            pfeat.data_ptr(), st), "lin")
                L.check(lib.gn_sa_hoisted_mlp_pool(lv_xyz.data_ptr(), pfeat.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), B, lv_n,
                                                   npoint, nsample, radius, _ci(*mlp), pk.wptr, pk.bptr, o.data_ptr(), st), key)
        torch.cuda.synchronize()
        _, of, _ = MO.sa_module(xyz_lv[li], feat_in, sd, "%s.sa%d" % (bb, li + 1), npoint, radius, nsample)
        out[key] = {"hip": errs(o, tru[key], True), "oracle": errs(of, tru[key])}

    def fp(unknown, known, skip_t, known_t, key, want, prefix):
        n, m = unknown.shape[1], known.shape[1]
        u, k = unknown.cuda(), known.cuda()
        d2 = torch.empty((B, n, 3), dtype=f32, device="cuda")
        i3 = torch.empty((B, n, 3), dtype=i32, device="cuda")
        L.check(lib.gn_three_nn(u.data_ptr(), k.data_ptr(), B, n, m, d2.data_ptr(), i3.data_ptr(), st), "3nn")
        o = torch.empty((B, n, 256), dtype=f32, device="cuda")
        pk = pkd[key + "_split" if split else key]
        fn = lib.gn_fp_interp_mlp_split if split else lib.gn_fp_interp_mlp
        kx, sx = pm(known_t), pm(skip_t)
        L.check(fn(kx.data_ptr(), sx.data_ptr(), d2.data_ptr(), i3.data_ptr(), B, n, m, 256, 256, _ci(256, 256), pk.wptr, pk.bptr,
                   o.data_ptr(), st), key)
        torch.cuda.synchronize()
        of = MO.fp_module(unknown, known, skip_t.float(), known_t.float(), sd, prefix)
        out[key] = {"hip": errs(o, want, True), "oracle": errs(of, want)}

    fp(rep["sa3_xyz"], rep["sa4_xyz"], tru["sa3_features"], tru["sa4_features"], "fp1", tru["_fp1_features"], bb + ".fp1")
    fp(rep["sa2_xyz"], rep["sa3_xyz"], tru["sa2_features"], tru["_fp1_features"], "fp2", tru["fp2_features"], bb + ".fp2")

    # ApproachNet
    M = 1024
    x = pm(tru["fp2_features"])
    o = torch.empty((B, M, 302), dtype=f32, device="cuda")
    pk = pkd["approach_split" if split else "approach"]
    fn = lib.gn_approach_mlp_split if split else lib.gn_approach_mlp
    L.check(fn(x.data_ptr(), B * M, 256, 302, pk.wptr, pk.bptr, o.data_ptr(), st), "approach")
    torch.cuda.synchronize()
    oep = MO.approach_net(rep["fp2_xyz"], tru["fp2_features"].float(), sd, {})
    want_v = tru["view_score"]
    ev = (o[:, :, 2:].double().cpu() - want_v).abs()
    eo = (oep["view_score"].double() - want_v).abs()
    out["view_score"] = {"hip": {"max": float(ev.max()), "rms": float(ev.pow(2).mean().sqrt()), "scale": float(want_v.abs().max())},
                         "oracle": {"max": float(eo.max()), "rms": float(eo.pow(2).mean().sqrt()), "scale": float(want_v.abs().max())}}

    # CloudCrop on the oracle's rotations (the truth was forced onto the same discrete views)
    xyz = rep["input_xyz"].cuda()
    seed = rep["fp2_xyz"].cuda()
    rot = rep["grasp_top_view_rot"].reshape(B, M, 9).contiguous().cuda()
    hm = (ctypes.c_float * 4)(0.01, 0.02, 0.03, 0.04)
    uidx = torch.empty((B, M, 256), dtype=i32, device="cuda")
    ucnt = torch.empty((B, M), dtype=i32, device="cuda")
    L.check(lib.gn_cylinder_query_unique(seed.data_ptr(), xyz.data_ptr(), rot.data_ptr(), B, N, M, 0.05, -0.02, hm, 4, 64,
                                         uidx.data_ptr(), ucnt.data_ptr(), st), "cyl")
    vp = torch.empty((B, M * 4, 256), dtype=f32, device="cuda")
    pk = pkd["crop_split" if split else "crop"]
    fn = lib.gn_crop_unique_mlp_pool_split if split else lib.gn_crop_unique_mlp_pool
    L.check(fn(xyz.data_ptr(), seed.data_ptr(), rot.data_ptr(), uidx.data_ptr(), ucnt.data_ptr(), B, N, M, 4, 64, _ci(64, 128, 256),
               pk.wptr, pk.bptr, vp.data_ptr(), st), "crop")
    torch.cuda.synchronize()
    want = tru["_vp_features"].reshape(B, 256, M * 4)
    out["vp_features"] = {"hip": errs(vp, want, True), "oracle": errs(rep["_vp_features"].reshape(B, 256, M * 4), want)}

    # heads on the truth's vp_features
    x = pm(want)
    head = torch.empty((B, M * 4, 48), dtype=f32, device="cuda")
    pk = pkd["heads_split" if split else "heads"]
    fn = lib.gn_heads_mlp_split if split else lib.gn_heads_mlp
    L.check(fn(x.data_ptr(), B * M * 4, 256, 48, pk.wptr, pk.bptr, head.data_ptr(), st), "heads")
    torch.cuda.synchronize()
    hv = head.reshape(B, M, 4, 48).permute(0, 3, 1, 2).double().cpu()
    vpf = tru["_vp_features"].float()
    oo = MO.head_net(vpf, sd, "grasp_generator.operation")
    ot = MO.head_net(vpf, sd, "grasp_generator.tolerance")
    ohead = torch.cat([oo, ot], 1).double()
    for lo, key in ((0, "grasp_score_pred"), (12, "grasp_angle_cls_pred"), (24, "grasp_width_pred"), (36, "grasp_tolerance_pred")):
        w = tru[key]
        eh, eo = (hv[:, lo:lo + 12] - w).abs(), (ohead[:, lo:lo + 12] - w).abs()
        out[key] = {"hip": {"max": float(eh.max()), "rms": float(eh.pow(2).mean().sqrt()), "scale": float(w.abs().max())},
                    "oracle": {"max": float(eo.max()), "rms": float(eo.pow(2).mean().sqrt()), "scale": float(w.abs().max())}}
    return out


def run_case(name, clouds, sd, stages=True):
    t0 = time.time()
    rep = MO.forward({"point_clouds": clouds.clone()}, sd, keep_internal=True)
    rp = MO.pred_decode(rep)
    tru = MO.forward({"point_clouds": clouds.clone()}, sd, f64=True, keep_internal=True,
                     force_top_view_inds=rep["grasp_top_view_inds"])
    tp = MO.pred_decode(tru)
    rec = {"clouds": name, "B": int(clouds.shape[0]), "N": int(clouds.shape[1]), "cpu_s": time.time() - t0, "math": {}}
    for math in ("split", "f32"):
        net = GraspNet().load_state_dict(sd)
        net.math_split = math == "split"
        ep = net({"point_clouds": clouds.cuda()})
        preds = pred_decode(ep)
        torch.cuda.synchronize()
        keys, flips, same = PU.compare_end_points(ep, rep, tru)
        idx_ok = all(torch.equal(ep[k].cpu(), rep[k]) for k in PU.INDEX_KEYS)
        m = {"index_keys_bit_exact": idx_ok, "keys": keys, "flips": flips, "rows": PU.compare_rows(preds, rp, tp)}
        if stages:
            m["stage_isolated_vs_f64"] = teacher_forced(net, sd, rep, tru, math == "split")
        rec["math"][math] = m
        print("== %s  GN_MATH=%s  index keys bit-exact: %s  top-view flips %d/%d" % (name, math, idx_ok, flips["top_view"], flips["seeds"]))
        for k, v in keys.items():
            print("  %-22s scale %9.3f  |hip-or| max %.2e rms %.2e (limit %.2e %s) | hip-f64 max %.2e rms %.2e | or-f64 max %.2e rms %.2e"
                  % (k, v["vs_oracle"]["scale"], v["vs_oracle"]["max"], v["vs_oracle"]["rms"], v["limit"], "ok" if v["ok"] else "EXCEEDED",
                     v["vs_f64"]["max"], v["vs_f64"]["rms"], v["oracle_vs_f64"]["max"], v["oracle_vs_f64"]["rms"]))
        r = m["rows"]
        print("  decoded rows %d  class flips %d  objectness flips %d | score |hip-or| max %.2e rms %.2e (scale %.2f) | vs f64: hip max %.2e rms %.2e, "
              "oracle max %.2e rms %.2e" % (r["rows"], r["class_flips"], r["objectness_flips"], r["score_max"], r["score_rms"], r["score_scale"],
                                            r["hip_vs_f64_max"], r["hip_vs_f64_rms"], r["oracle_vs_f64_max"], r["oracle_vs_f64_rms"]))
        if stages:
            for k, v in m["stage_isolated_vs_f64"].items():
                print("  stage %-22s scale %9.3f  hip max %.2e rms %.2e | oracle max %.2e rms %.2e | rms ratio %.2f"
                      % (k, v["hip"]["scale"], v["hip"]["max"], v["hip"]["rms"], v["oracle"]["max"], v["oracle"]["rms"],
                         v["hip"]["rms"] / max(v["oracle"]["rms"], 1e-30)))
    return rec


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    O.build()
    O.set_num_threads(16)
    torch.set_num_threads(16)
    sd = S.make_state_dict(0)
    cases = [("golden batch: uniform seed 1000 + surface seed 1001",
              torch.stack([torch.from_numpy(S.make_cloud(1000, 20000, "uniform")), torch.from_numpy(S.make_cloud(1001, 20000, "surface"))], 0)),
             ("config[1]: 8 uniform clouds, seeds 1000..1007", S.make_batch(1000, 8, 20000, "uniform")),
             ("8 surface clouds, seeds 1000..1007", S.make_batch(1000, 8, 20000, "surface"))]
    if os.environ.get("PARITY_QUICK"):
        cases = cases[:1]
    report = {"rule": "|hip - oracle| <= 1e-4 * max(1, ||oracle key||_inf) (tests/parity_util.py)", "cases": []}
    for name, cl in cases:
        report["cases"].append(run_case(name, cl, sd))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "parity_%s.json" % tag)
    with open(path, "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
