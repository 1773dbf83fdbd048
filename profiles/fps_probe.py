import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "graspnet-baseline_amd"))
import torch
from graspnet_amd import synthetic as S, pointnet2_ext as E
x = S.make_batch(1000, 8, 20000).cuda()
ref = None
for _ in range(2): out = E.furthest_point_sampling(x, 2048)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): out = E.furthest_point_sampling(x, 2048)
torch.cuda.synchronize()
print("variant", os.environ.get("GN_FPS_VARIANT", "0"), "fps 20000->2048: %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3), "checksum", int(out.long().sum()))
