"""Phase shares of the EIGHT-wave CloudCrop kernel alone (instrumented build, profiles/build_probe.sh), per wave:
   GN_LIB_PATH=profiles/probe/libgraspnet_hip_probe.so GN_CROP_WAVES=8 python profiles/crop8_phase_probe.py
group A (waves 0-3) marks: 0 slot top (row loads issued) | 1 wait at the slot barrier | 2 layer 3 + pooling | 3 wait at the half barrier |
                           4 layer 1 of chunk c+2 | 5 staging of chunk c+3
group B (waves 4-7) marks: 0 slot top (layer-2 fragments requested) | 1 wait at the slot barrier | 2 layer 2 | 3 wait at the half barrier |
                           6 layer 3 + pooling"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
os.environ.setdefault("GN_CROP_WAVES", "8")
import torch
from graspnet_amd import GraspNet, synthetic as S, _lib as L
from graspnet_amd.graspnet import _ci
net = GraspNet().load_state_dict(S.make_state_dict(0))
x = S.make_batch(1000, 8, 20000).cuda()
lib = L.lib()
ep = net({"point_clouds": x}); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
lib.gn_phase_probe_read_split(buf, 1)
g = ep["_gn"]; B, N, M = 8, 20000, 1024
rot = ep["grasp_top_view_rot"].reshape(B, M, 9).contiguous(); seed = ep["fp2_xyz"].contiguous()
vp = torch.empty((B, M * 4, 256), device="cuda"); pk = net._packed["crop_split"]
st = torch.cuda.current_stream().cuda_stream
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(4):
    L.check(lib.gn_crop_unique_mlp_pool_split(x.data_ptr(), seed.data_ptr(), rot.data_ptr(), g["uidx"].data_ptr(), g["ucount"].data_ptr(), B, N, M, 4, 64,
                                              _ci(64, 128, 256), pk.wptr, pk.bptr, vp.data_ptr(), st), "crop")
b.record(); torch.cuda.synchronize()
print("crop kernel, GN_CROP_WAVES=%s (instrumented): %.3f ms per launch" % (os.environ["GN_CROP_WAVES"], a.elapsed_time(b) / 4))
lib.gn_phase_probe_read_split(buf, 1)
namesA = ["top", "wait slot bar", "L3 pool+close", "-", "L1", "put", "-", "-", "-"]
namesB = ["top", "wait slot bar", "L2 rest", "-", "-", "-", "L3 pool+close", "-", "-"]
for w in range(8):
    v = [buf[w * 16 + i] for i in range(16)]
    tot = sum(v[:9])
    if tot == 0:
        continue
    tot += sum(v[11:16])
    sub = {11: "L2 mfma+wait", 12: "L2 epilogue", 13: "L3 mfma", 14: "-", 15: "L3 setup"}
    print("        sub-marks (taken out of their phase): " + "  ".join("%s %.1f%%" % (sub[i], 100.0 * v[i] / tot) for i in range(11, 16) if v[i]))
    print("wave %d: ticks/slot %.0f  slots %d | " % (w, tot / max(v[9], 1), v[9]) +
          "  ".join("%s %.1f%%" % ((namesA if w < 4 else namesB)[i], 100.0 * v[i] / tot) for i in range(9) if v[i]))
