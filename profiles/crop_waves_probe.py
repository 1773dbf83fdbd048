"""CloudCrop kernel alone, 4-wave (default) against the 8-wave form (GN_CROP_WAVES=8): HIP-event time per launch at the
headline shape (batch 8 x 1024 seeds on 20 000-pt uniform clouds), outputs compared bit for bit."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
import torch  # noqa: E402
from graspnet_amd import GraspNet, _lib as L, synthetic as S  # noqa: E402

net = GraspNet().load_state_dict(S.make_state_dict(0))
clouds = S.make_batch(1000, 8, 20000, "uniform").cuda()
ep = net({"point_clouds": clouds})
priv = ep["_gn"]
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
xyz, seed, rot = ep["input_xyz"], ep["fp2_xyz"], ep["grasp_top_view_rot"].reshape(8, 1024, 9).contiguous()
uidx, ucount = priv["uidx"], priv["ucount"]
mlp = (ctypes.c_int * 3)(64, 128, 256)
pks = net._packed["crop_split"]
outs = {}
for waves in sys.argv[1:] or ["4", "8", "4", "8"]:
    os.environ["GN_CROP_WAVES"] = waves
    v = torch.empty((8, 4096, 256), device="cuda")

    def run():
        L.check(lib.gn_crop_unique_mlp_pool_split(xyz.data_ptr(), seed.data_ptr(), rot.data_ptr(), uidx.data_ptr(), ucount.data_ptr(),
                                                  8, 20000, 1024, 4, 64, mlp, pks.wptr, pks.bptr, v.data_ptr(), st), "crop")
    for _ in range(3):
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        run()
    b.record()
    torch.cuda.synchronize()
    print("GN_CROP_WAVES=%s: %.4f ms per launch" % (waves, a.elapsed_time(b) / 20), flush=True)
    outs.setdefault(waves, v.clone())
ks = sorted(outs)
print("bit-identical:", all(torch.equal(outs[ks[0]], outs[k]) for k in ks))
