import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "graspnet-baseline_amd"))
import torch
from graspnet_amd import synthetic as S, pointnet2_ext as E, _lib as L
lib = L.lib()
x = S.make_batch(1000, 8, 20000).cuda(); c = x[:, :2048].contiguous()
st = torch.cuda.current_stream().cuda_stream
ws = torch.empty((lib.gn_grid_workspace_bytes(8, 20000),), dtype=torch.uint8, device="cuda")
out = torch.empty((8, 2048, 64), dtype=torch.int32, device="cuda")
def t(fn, reps=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("brute ball_query SA1: %.3f ms" % t(lambda: E.ball_query(c, x, 0.04, 64)))
print("grid build: %.3f ms" % t(lambda: L.check(lib.gn_grid_build(x.data_ptr(), 8, 20000, 0.04, ws.data_ptr(), st), "b")))
print("grid ball_query SA1: %.3f ms" % t(lambda: L.check(lib.gn_ball_query_grid(c.data_ptr(), ws.data_ptr(), 8, 20000, 2048, 0.04, 64, out.data_ptr(), st), "q")))
print("equal:", torch.equal(out, E.ball_query(c, x, 0.04, 64)))
