"""Probe other BASELINE.json configs on the GPU box: B=1 latency, N=50000 (config 4) forward + op stress."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "graspnet-baseline_amd"))
import torch
from graspnet_amd import GraspNet, synthetic as S, pointnet2_ext as E
net = GraspNet().load_state_dict(S.make_state_dict(0))

def run(B, N, reps=5):
    clouds = S.make_batch(1000, B, N).cuda()
    for _ in range(2):
        ep = net({"point_clouds": clouds}); net.decode_padded(ep)
    torch.cuda.synchronize()
    net.timing = {}
    t0 = time.perf_counter()
    for _ in range(reps):
        ep = net({"point_clouds": clouds}); g, c = net.decode_padded(ep)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    st = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in net.timing.items()}
    net.timing = None
    top = sorted(st.items(), key=lambda kv: -kv[1])[:6]
    print("B=%d N=%d: %.3f ms/forward (%.1f scenes/s single stream); grasps %s; top kernels %s" % (
        B, N, dt * 1e3, B / dt, c.tolist()[:4], [(k, round(v, 3)) for k, v in top]))

run(1, 20000)
run(8, 20000)
from graspnet_amd import GraphedGraspNet
for B in (1, 8):
    cl = S.make_batch(1000, B, 20000).cuda()
    gn = GraphedGraspNet(net, B, 20000)
    gn(cl); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): gn(cl)
    torch.cuda.synchronize(); print("hipGraph replay B=%d N=20000: %.3f ms/forward" % (B, (time.perf_counter() - t0) / 10 * 1e3))
run(8, 50000, reps=3)
run(1, 50000, reps=3)
# config 4 op-level stress: ball_query M=2048 ns=64 r=0.04 on 50k points
x = S.make_batch(7, 8, 50000).cuda(); c = x[:, :2048].contiguous()
for name, fn in (("ball_query 50k", lambda: E.ball_query(c, x, 0.04, 64)), ("fps 50k->2048", lambda: E.furthest_point_sampling(x, 2048))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    print("%s: %.3f ms" % (name, (time.perf_counter() - t0) * 1e3))
