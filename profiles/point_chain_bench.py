"""Isolated timings of the per-point chains (heads, ApproachNet, FP1, FP2, the three hoisted linears) at the batch-8 shapes
of config[1], A/B over GN_POINT_WPC (1 = 64-row tiles and one workgroup per CU, 2 = 32-row tiles and two), with a bit-wise
comparison of the two outputs (the k order of every dot product is the same in both).
   python profiles/point_chain_bench.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
import torch
from graspnet_amd import GraspNet, synthetic as S, _lib as L

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = 8
net = GraspNet().load_state_dict(S.make_state_dict(0))
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
dev = "cuda"
g = torch.Generator(device="cpu").manual_seed(5)


def rnd(*shape):
    return torch.randn(*shape, generator=g).to(dev)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cases = []
# heads: 1024 seeds x 4 depths per scene
x_h = rnd(B * 4096, 256); o_h = torch.empty(B * 4096, 48, device=dev)
pk = net._packed["heads_split"]
cases.append(("heads 256-128|128-128|128-36|12 (32768 rows)", o_h,
              lambda: L.check(lib.gn_heads_mlp_split(x_h.data_ptr(), B * 4096, 256, 48, pk.wptr, pk.bptr, o_h.data_ptr(), st), "heads")))
x_a = rnd(B * 1024, 256); o_a = torch.empty(B * 1024, 302, device=dev)
pa = net._packed["approach_split"]
cases.append(("approach 256-256-302-302 (8192 rows)", o_a,
              lambda: L.check(lib.gn_approach_mlp_split(x_a.data_ptr(), B * 1024, 256, 302, pa.wptr, pa.bptr, o_a.data_ptr(), st), "approach")))


def fp_case(name, key, n, m):
    known, skip = rnd(B, m, 256), rnd(B, n, 256)
    d2 = torch.rand(B, n, 3, generator=g).to(dev) * 0.01
    i3 = torch.randint(0, m, (B, n, 3), generator=g, dtype=torch.int32).to(dev)
    o = torch.empty(B, n, 256, device=dev)
    p = net._packed[key]
    import ctypes
    mlp = (ctypes.c_int * 2)(256, 256)
    cases.append((name, o, lambda: L.check(lib.gn_fp_interp_mlp_split(known.data_ptr(), skip.data_ptr(), d2.data_ptr(), i3.data_ptr(),
                                                                     B, n, m, 256, 256, mlp, p.wptr, p.bptr, o.data_ptr(), st), "fp")))


fp_case("fp1 512-256-256 (4096 rows)", "fp1_split", 512, 128)
fp_case("fp2 512-256-256 (8192 rows)", "fp2_split", 1024, 512)
for li, (rows, cin) in ((2, (B * 2048, 128)), (3, (B * 1024, 256)), (4, (B * 512, 256))):
    x = rnd(rows, cin); o = torch.empty(rows, 128, device=dev)
    p = net._packed["sa%d_pre_split" % li]
    cases.append(("linear sa%d %d-128 (%d rows)" % (li, cin, rows), o,
                  (lambda x=x, o=o, p=p, rows=rows, cin=cin: L.check(lib.gn_point_linear_split(
                      x.data_ptr(), rows, cin, 128, p.w[0].data_ptr(), p.b[0].data_ptr(), o.data_ptr(), st), "lin"))))

tot = {}
for name, out, fn in cases:
    res = {}
    for wpc in ("1", "2"):
        os.environ["GN_POINT_WPC"] = wpc
        os.environ["GN_LINEAR_WPC"] = wpc
        os.environ["GN_HEADS_DENSE"] = os.environ.get("PCB_DENSE", "0")
        out.zero_()
        fn()
        torch.cuda.synchronize()
        res[wpc] = (out.clone(), timed(fn))
        tot[wpc] = tot.get(wpc, 0.0) + res[wpc][1]
    same = torch.equal(res["1"][0], res["2"][0])
    print("%-50s wpc1 %7.1f us   wpc2 %7.1f us   outputs %s" % (name, res["1"][1], res["2"][1], "identical" if same else
          "DIFFER max %.3e" % float((res["1"][0] - res["2"][0]).abs().max())), flush=True)
print("sum: wpc1 %.1f us, wpc2 %.1f us" % (tot["1"], tot["2"]))
# every chain in its plane form (latency shapes) against its fp32-tile form (mlp_point_wide.hip: heads 128-row, ApproachNet /
# FP 64-row tiles): launch time and launch time x CUs the grid occupies (what a launch costs other batches in flight)
del os.environ["GN_POINT_WPC"]; del os.environ["GN_LINEAR_WPC"]
tiles = {"heads": (64, 128), "approach": (32, 64), "fp1": (32, 64), "fp2": (32, 64)}
rows_of = {"heads": B * 4096, "approach": B * 1024, "fp1": B * 512, "fp2": B * 1024}
tw = {"0": 0.0, "1": 0.0}
cw = {"0": 0.0, "1": 0.0}
for name, out, fn in cases[:4]:
    key = name.split()[0]
    res = {}
    for wide in ("0", "1"):
        os.environ["GN_POINT_WIDE"] = wide
        out.zero_()
        fn()
        torch.cuda.synchronize()
        res[wide] = (out.clone(), timed(fn))
        wgs = min(256, -(-rows_of[key] // tiles[key][int(wide)]))
        tw[wide] += res[wide][1]
        cw[wide] += res[wide][1] * wgs
    del os.environ["GN_POINT_WIDE"]
    print("%-50s plane tiles %7.1f us   fp32 tiles %7.1f us   outputs %s" % (name, res["0"][1], res["1"][1],
          "identical" if torch.equal(res["0"][0], res["1"][0]) else "DIFFER max %.3e" % float((res["0"][0] - res["1"][0]).abs().max())))
print("sum of the four chains: plane %.1f us (%.0f CU-us), fp32 tiles %.1f us (%.0f CU-us)" % (tw["0"], cw["0"], tw["1"], cw["1"]))
