"""Decoded grasp-score error of the HIP paths (GN_MATH=f32 / split) and of the fp32 CPU oracle against a float64
evaluation (same neighbour indices, same discrete view choice) on the golden batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
import torch
from oracle import model_oracle as MO, oracle as O
from graspnet_amd import GraspNet, pred_decode, synthetic as S
O.build(); O.set_num_threads(16); torch.set_num_threads(16)
sd = S.make_state_dict(0)
cl = S.make_batch(1000, 2, 20000, "uniform")
rep = MO.forward({"point_clouds": cl.clone()}, sd)
rp = MO.pred_decode(rep)
tep = MO.forward({"point_clouds": cl.clone()}, sd, f64=True, force_top_view_inds=rep["grasp_top_view_inds"])
tp = MO.pred_decode(tep)


def rows(p):
    return {tuple(x[13:16].tolist()): x for x in p}


def cmp(name, a, t):
    worst, n, flips = 0.0, 0, 0
    for b in range(2):
        ka, kt = rows(a[b].cpu().double()), rows(t[b].double())
        for k in ka:
            if k in kt:
                x, y = ka[k], kt[k]
                if not torch.allclose(x[4:13], y[4:13], atol=1e-5) or abs(float(x[3] - y[3])) > 1e-6:
                    flips += 1
                    continue
                n += 1
                worst = max(worst, abs(float(x[0] - y[0])))
    print("%-18s rows %d flips %d  max |score - f64| %.3e" % (name, n, flips, worst))


cmp("oracle fp32 (CPU)", rp, tp)
for math in ("f32", "split"):
    net = GraspNet().load_state_dict(sd)
    net.math_split = math == "split"
    cmp("HIP " + math, pred_decode(net({"point_clouds": cl.cuda()})), tp)
