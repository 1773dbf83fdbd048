"""rocprofv3 target: the batched post-processing pipeline (voxel down-sample + collision detect + sort/NMS) of config[4] on the
8 decoded scenes of one config[1] batch, 50 times on one stream.  `rocprofv3 --kernel-trace --stats -- python3 profiles/post_probe.py`"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
import torch  # noqa: E402
from graspnet_amd import GraspNet, postprocess as P, synthetic as S  # noqa: E402

net = GraspNet().load_state_dict(S.make_state_dict(0))
clouds = S.make_batch(1000, 8, 20000, "uniform").cuda()
g, c = net.decode_padded(net({"point_clouds": clouds}))
torch.cuda.synchronize()
for _ in range(50):
    P.grasp_pipeline(clouds, g, c, voxel_size=0.01, approach_dist=0.05, collision_thresh=0.01)
torch.cuda.synchronize()
print("ok")
