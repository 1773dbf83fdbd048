"""Phase breakdown of sa_hoist_split_kernel (SA2 shape) -- needs the instrumented build:
   GN_LIB_PATH=profiles/probe/libgraspnet_hip_probe.so python profiles/sa_phase_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
import torch
from graspnet_amd import GraspNet, synthetic as S, _lib as L

net = GraspNet().load_state_dict(S.make_state_dict(0))
lib = L.lib()
B, N, NP, NS = 8, 2048, 1024, 32
torch.manual_seed(0)
xyz = torch.rand(B, N, 3, device="cuda")
pfeat = torch.randn(B, N, 128, device="cuda")
new_xyz = xyz[:, :NP].contiguous()
idx = torch.randint(0, N, (B, NP, NS), device="cuda", dtype=torch.int32)
out = torch.empty(B, NP, 256, device="cuda")
pk = net._packed["sa2_hoist_split"]
mlp = (ctypes.c_int * 3)(128, 128, 256)
buf = (ctypes.c_ulonglong * 128)()
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    lib.gn_phase_probe_read_split(buf, 1)
    L.check(lib.gn_sa_hoisted_mlp_pool_split(xyz.data_ptr(), pfeat.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), B, N, NP, NS,
                                             0.1, mlp, pk.wptr, pk.bptr, out.data_ptr(), st), "sa")
    torch.cuda.synchronize()
lib.gn_phase_probe_read_split(buf, 1)
names = ["loop-top", "L1+fetch", "bar1", "L2+put", "bar2", "L3+pool"]
v = [sum(buf[w * 16 + i] for w in range(4)) for i in range(16)]
tiles = v[9] / 4.0
print("tiles %d; cycles per 64-row tile (avg over waves): " % tiles + "  ".join("%s %.0f" % (names[i], v[i] / max(v[9], 1)) for i in range(6)),
      " total %.0f" % (sum(v[:6]) / max(v[9], 1)))
for w in range(4):
    vw = [buf[w * 16 + i] for i in range(16)]
    print("  wave %d: " % w + "  ".join("%s %.0f" % (names[i], vw[i] / max(vw[9], 1)) for i in range(6)))
