import os, sys, torch
ROOT = os.getcwd(); sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd")); sys.path.insert(0, ROOT)
from graspnet_amd import _lib as L, synthetic as S
lib = L.lib()
for n, b in ((50000, 8), (20000, 8)):
    x = S.make_batch(1000, b, n, "uniform").cuda()
    ws = torch.empty(max(int(lib.gn_fps_workspace_bytes(b, n)), 16), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for m in (2, 3, 66, 2048):
        idx = torch.empty(b, m, dtype=torch.int32, device="cuda"); tie = torch.zeros(b, dtype=torch.int32, device="cuda")
        f = lambda: L.check(lib.gn_furthest_point_sampling(x.data_ptr(), b, n, m, ws.data_ptr(), idx.data_ptr(), None, tie.data_ptr(), None, st), "fps")
        f(); torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): f()
        e.record(); torch.cuda.synchronize()
        print("n=%d m=%d: %.4f ms" % (n, m, a.elapsed_time(e) / 5), flush=True)
