"""Phase breakdown of fps_cell_kernel (batch 8; the probe reports cloud 0) -- instrumented build:
   GN_LIB_PATH=profiles/probe/libgraspnet_hip_probe.so python profiles/fps_cell_phase_probe.py [n] [GN_FPS_CELL]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
if len(sys.argv) > 2:
    os.environ["GN_FPS_CELL"] = sys.argv[2]
import torch
from graspnet_amd import synthetic as S, _lib as L
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
M = 2048
for variant in ("uniform", "surface"):
    x = S.make_batch(1000, 8, n, variant).cuda()
    idx = torch.empty(8, M, dtype=torch.int32, device="cuda")
    tie = torch.empty(8, dtype=torch.int32, device="cuda")
    ws = torch.empty(max(int(lib.gn_fps_workspace_bytes(8, n)), 16), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        L.check(lib.gn_furthest_point_sampling(x.data_ptr(), 8, n, M, ws.data_ptr(), idx.data_ptr(), None, tie.data_ptr(), None, st), "fps")
        b.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 128)()
    lib.gn_phase_probe_read_fps_cell.argtypes = [ctypes.c_void_p]
    lib.gn_phase_probe_read_fps_cell(buf)
    names = ["skip", "fetch", "regions", "wave cand", "fold+barrier", "read-back etc"]
    print("n=%d %s: kernel %.3f ms (instrumented)" % (n, variant, a.elapsed_time(b)))
    for w in range(16):
        v = [buf[w * 8 + i] for i in range(8)]
        tot = sum(v[:6])
        print("  wave %2d: " % w + "  ".join("%s %.0f" % (names[i], v[i] / (M - 1.0)) for i in range(6)) +
              "  | total %.0f ticks/step, %.2f regions/step in %.2f batches" % (tot / (M - 1.0), v[6] / (M - 1.0), v[7] / (M - 1.0)))
