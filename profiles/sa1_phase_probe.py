"""Phase breakdown of sa_xyz_split_kernel (SA1 shape, two workgroups per CU) -- instrumented build:
   GN_LIB_PATH=profiles/probe/libgraspnet_hip_probe.so python profiles/sa1_phase_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
import torch
from graspnet_amd import GraspNet, synthetic as S, _lib as L
net = GraspNet().load_state_dict(S.make_state_dict(0))
lib = L.lib()
B, N, NP, NS = 8, 20000, 2048, 64
torch.manual_seed(0)
xyz = torch.rand(B, N, 3, device="cuda")
new_xyz = xyz[:, :NP].contiguous()
idx = torch.randint(0, N, (B, NP, NS), device="cuda", dtype=torch.int32)
out = torch.empty(B, NP, 128, device="cuda")
pk = net._packed["sa1_split"]
mlp = (ctypes.c_int * 3)(64, 64, 128)
buf = (ctypes.c_ulonglong * 128)()
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    lib.gn_phase_probe_read_split(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    L.check(lib.gn_sa_group_mlp_pool_split(xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), B, N, NP, NS, 0.04, mlp, pk.wptr, pk.bptr,
                                           out.data_ptr(), st), "sa1")
    b.record(); torch.cuda.synchronize()
lib.gn_phase_probe_read_split(buf, 1)
names = ["fetch", "L1 (f32 MFMA + split)", "bar1", "L2 + put", "bar2", "L3 + pool"]
v = [sum(buf[w * 16 + i] for w in range(4)) for i in range(16)]
print("kernel %.3f ms; ticks per 64-row tile (avg over waves): " % a.elapsed_time(b) + "  ".join("%s %.0f" % (names[i], v[i] / max(v[9], 1)) for i in range(6)),
      " total %.0f  (MFMAs per wave and tile: L2 24, L3 48 -> %d ticks at the bare rate)" % (sum(v[:6]) / max(v[9], 1), 72 * 34.3))
