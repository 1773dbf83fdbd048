"""Phase breakdown of point_mlp_split_kernel for the heads chain (256->256->256->48 on 32768 rows) -- instrumented build:
   GN_LIB_PATH=profiles/probe/libgraspnet_hip_probe.so python profiles/point_phase_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
import torch
from graspnet_amd import GraspNet, synthetic as S, _lib as L

net = GraspNet().load_state_dict(S.make_state_dict(0))
lib = L.lib()
rows = 8 * 4096
x = torch.randn(rows, 256, device="cuda")
out = torch.empty(rows, 48, device="cuda")
pk = net._packed["heads_split"]
buf = (ctypes.c_ulonglong * 128)()
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    lib.gn_phase_probe_read_point(buf, 1)
    L.check(lib.gn_heads_mlp_split(x.data_ptr(), rows, 256, 48, pk.wptr, pk.bptr, out.data_ptr(), st), "heads")
    torch.cuda.synchronize()
lib.gn_phase_probe_read_point(buf, 1)
names = ["load+split", "bar", "layer1 (mfma|bar|store)", "bar", "layer2", "bar", "layer3 -> global", "tile top"]
v = [sum(buf[w * 16 + i] for w in range(4)) for i in range(16)]
print("cycles per 64-row tile (avg over waves): " + "  ".join("%s %.0f" % (names[i], v[i] / max(v[9], 1)) for i in range(8)),
      " total %.0f" % (sum(v[:8]) / max(v[9], 1)))
for w in range(4):
    vw = [buf[w * 16 + i] for i in range(16)]
    print("  wave %d: " % w + "  ".join("%s %.0f" % (names[i], vw[i] / max(vw[9], 1)) for i in range(8)))
