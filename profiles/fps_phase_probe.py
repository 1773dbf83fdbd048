"""Phase breakdown of fps_region_kernel (20000 -> 2048, batch 8; the probe reports cloud 0) -- instrumented build:
   GN_LIB_PATH=profiles/probe/libgraspnet_hip_probe.so python profiles/fps_phase_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
import torch
from graspnet_amd import synthetic as S, _lib as L
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
for variant in ("uniform", "surface"):
    x = S.make_batch(1000, 8, 20000, variant).cuda()
    idx = torch.empty(8, 2048, dtype=torch.int32, device="cuda")
    tie = torch.empty(8, dtype=torch.int32, device="cuda")
    for _ in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        L.check(lib.gn_furthest_point_sampling(x.data_ptr(), 8, 20000, 2048, None, idx.data_ptr(), None, tie.data_ptr(), None, st), "fps")
        b.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    lib.gn_phase_probe_read_fps(buf)
    names = ["skip test", "active groups", "wave cand + atomic", "barrier", "read-back + fetch", "tie check etc"]
    print("%s: kernel %.3f ms (instrumented)" % (variant, a.elapsed_time(b)))
    for w in range(8):
        v = [buf[w * 8 + i] for i in range(8)]
        tot = sum(v[:6])
        print("  wave %d: " % w + "  ".join("%s %.0f" % (names[i], v[i] / 2047.0) for i in range(6)) +
              "  | total %.0f ticks/iter, %.2f active groups/iter" % (tot / 2047.0, v[6] / 2047.0))
