"""Where do the CloudCrop kernel's cycles go?  Needs the instrumented build (profiles/build_probe.sh):
   GN_LIB_PATH=profiles/probe/libgraspnet_hip_probe.so python profiles/crop_phase_probe.py
Prints, per wave, the share of cycles between the marks in crop_unique8_mlp_kernel."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
import torch  # noqa: E402
from graspnet_amd import GraspNet, synthetic as S, _lib as L  # noqa: E402

net = GraspNet().load_state_dict(S.make_state_dict(0))
x = S.make_batch(1000, 8, 20000).cuda()
lib = L.lib()
buf = (ctypes.c_ulonglong * 128)()
for it in range(3):
    net({"point_clouds": x})
    torch.cuda.synchronize()
    (lib.gn_phase_probe_read_split if net.math_split else lib.gn_phase_probe_read)(buf, 1)
# note: with the split path the table sums the CloudCrop kernel AND the hoisted SA2-4 kernels (same translation unit);
# GN_PROBE_ONLY=sa / crop restricts the forward to one of them by running the other on the fp32 kernels
names = ["fetch-issue", "L1", "bar1", "L2", "bar2", "L3+pool", "finish", "put", "bar3", "chunks", "mtiles"]
tot_all = 0
for w in range(8):
    v = [buf[w * 16 + i] for i in range(16)]
    tot = sum(v[:9])
    tot_all += tot
    print("wave %d: total %.3e cyc  chunks %d  mtiles %d | " % (w, tot, v[9], v[10]) +
          "  ".join("%s %.1f%%" % (names[i], 100.0 * v[i] / max(tot, 1)) for i in range(9)))
v = [sum(buf[w * 16 + i] for w in range(8)) for i in range(16)]
print("per chunk (avg over waves): " + "  ".join("%s %.0f" % (names[i], v[i] / max(v[9], 1)) for i in range(9)),
      " mtiles/chunk %.2f" % (v[10] / max(v[9], 1)))
