import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "graspnet-baseline_amd"))
import torch, ctypes
from graspnet_amd import synthetic as S, _lib as L
lib = L.lib()
for variant in ("uniform", "surface"):
    x = S.make_batch(1000, 4, 20000, variant).cuda()
    idx = torch.empty((4, 2048), dtype=torch.int32, device="cuda")
    nx = torch.empty((4, 2048, 3), device="cuda")
    flag = torch.full((4,), -7, dtype=torch.int32, device="cuda")
    L.check(lib.gn_furthest_point_sampling(x.data_ptr(), 4, 20000, 2048, None, idx.data_ptr(), nx.data_ptr(), flag.data_ptr(), None,
                                           torch.cuda.current_stream().cuda_stream), "fps")
    torch.cuda.synchronize()
    print(variant, "tie flags:", flag.tolist())
