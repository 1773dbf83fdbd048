"""Phase shares of the CloudCrop split kernel ALONE (instrumented build, profiles/build_probe.sh):
   GN_LIB_PATH=profiles/probe/libgraspnet_hip_probe.so python profiles/crop_phase_probe2.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
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
print("crop kernel (instrumented): %.3f ms per launch" % (a.elapsed_time(b) / 4))
lib.gn_phase_probe_read_split(buf, 1)
names = ["fetch-issue", "L1", "bar1", "L2", "bar2", "L3 between units (close/finish/2nd pass/tail)", "L3 units without pool", "put", "L3 units with pool", "chunks", "mtiles", "mixed tiles"]
for w in range(4):
    v = [buf[w * 16 + i] for i in range(16)]
    tot = sum(v[:9])
    print("wave %d: total %.3e  chunks %d mtiles %d | " % (w, tot, v[9], v[10]) + "  ".join("%s %.1f%%" % (names[i], 100.0 * v[i] / max(tot, 1)) for i in range(9)))
v = [sum(buf[w * 16 + i] for w in range(4)) for i in range(16)]
print("extra slots per M tile: standalone pool passes %.0f  finishes %.0f  classification %.0f" % (v[12] / max(v[10], 1), v[13] / max(v[10], 1), v[14] / max(v[10], 1)))
print("per M tile (avg over waves, probe clock ticks): " + "  ".join("%s %.0f" % (names[i], v[i] / max(v[10], 1)) for i in range(9)), " mtiles/chunk %.2f  mixed tiles %.2f of tiles  total mtiles %d" % (v[10] / max(v[9], 1), v[11] / max(v[10], 1), v[10] / 4))
