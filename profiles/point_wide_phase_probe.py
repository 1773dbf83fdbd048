"""Phase breakdown of point_mlp_wide_kernel (heads chain on 128-row fp32 tiles, 32768 rows) -- instrumented build:
   bash profiles/build_probe.sh && GN_LIB_PATH=profiles/probe/libgraspnet_hip_probe.so python profiles/point_wide_phase_probe.py"""
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
names = ["load", "bar", "layer1 bar+store", "bar", "layer2 bar+store", "bar", "layer3 store", "tile top", "-", "-", "L1 mfma loop", "L2 mfma loop", "L3 mfma loop"]
for wide, reader, tp in (("1", lib.gn_phase_probe_read_point_wide, 128), ("0", lib.gn_phase_probe_read_point, 64)):
    os.environ["GN_POINT_WIDE"] = wide
    for it in range(3):
        reader(buf, 1)
        L.check(lib.gn_heads_mlp_split(x.data_ptr(), rows, 256, 48, pk.wptr, pk.bptr, out.data_ptr(), st), "heads")
        torch.cuda.synchronize()
    reader(buf, 1)
    v = [sum(buf[w * 16 + i] for w in range(4)) for i in range(16)]
    print("%d-row tiles: ticks per tile (avg over waves): " % tp + "  ".join("%s %.0f" % (names[i], v[i] / max(v[9], 1)) for i in range(8)),
          " total %.0f" % ((sum(v[:8]) + sum(v[10:13])) / max(v[9], 1)))
    for w in range(4):
        vw = [buf[w * 16 + i] for i in range(16)]
        print("  wave %d: " % w + "  ".join("%s %.0f" % (names[i], vw[i] / max(vw[9], 1)) for i in (0, 1, 10, 2, 3, 11, 4, 5, 12, 6, 7)))
