"""Does SA1's sampler report a tie on the bench's synthetic clouds?  (tie step >= 1024 lets SA2-4 return the identity prefix.)
   python profiles/tie_probe2.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
import torch
from graspnet_amd import synthetic as S, _lib as L
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
for variant in ("uniform", "surface"):
    for seed in (1000, 1008, 1016, 1024):
        x = S.make_batch(seed, 8, 20000, variant).cuda()
        ws = torch.empty(max(int(lib.gn_fps_workspace_bytes(8, 20000)), 16), dtype=torch.uint8, device="cuda")
        idx = torch.empty(8, 2048, dtype=torch.int32, device="cuda")
        nx = torch.empty(8, 2048, 3, device="cuda")
        tie = torch.full((8,), -7, dtype=torch.int32, device="cuda")
        L.check(lib.gn_furthest_point_sampling(x.data_ptr(), 8, 20000, 2048, ws.data_ptr(), idx.data_ptr(), nx.data_ptr(),
                                               tie.data_ptr(), None, st), "fps")
        torch.cuda.synchronize()
        print(variant, seed, "first tie step per cloud:", tie.cpu().tolist(), flush=True)

# time the SA2 sampler behind that flag
x = S.make_batch(1008, 8, 20000, "uniform").cuda()
ws = torch.empty(max(int(lib.gn_fps_workspace_bytes(8, 20000)), 16), dtype=torch.uint8, device="cuda")
idx = torch.empty(8, 2048, dtype=torch.int32, device="cuda"); nx = torch.empty(8, 2048, 3, device="cuda")
tie = torch.full((8,), -7, dtype=torch.int32, device="cuda")
idx2 = torch.empty(8, 1024, dtype=torch.int32, device="cuda"); nx2 = torch.empty(8, 1024, 3, device="cuda")
L.check(lib.gn_furthest_point_sampling(x.data_ptr(), 8, 20000, 2048, ws.data_ptr(), idx.data_ptr(), nx.data_ptr(), tie.data_ptr(), None, st), "fps")
for label, flag in (("with SA1's flag", tie), ("flag = 0 (tie unknown)", torch.zeros_like(tie))):
    for _ in range(3):
        L.check(lib.gn_furthest_point_sampling(nx.data_ptr(), 8, 2048, 1024, None, idx2.data_ptr(), nx2.data_ptr(), None, flag.data_ptr(), st), "fps2")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.check(lib.gn_furthest_point_sampling(nx.data_ptr(), 8, 2048, 1024, None, idx2.data_ptr(), nx2.data_ptr(), None, flag.data_ptr(), st), "fps2")
    e1.record(); torch.cuda.synchronize()
    print("SA2 sampler %s: %.1f us; identity prefix: %s" % (label, e0.elapsed_time(e1) / 20 * 1e3,
          bool((idx2 == torch.arange(1024, device="cuda", dtype=torch.int32)).all())))
