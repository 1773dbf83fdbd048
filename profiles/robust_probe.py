"""Accuracy of the split path on clouds other than the golden batch: decoded-score and feature error against float64
for the CPU oracle, the fp32 MFMA kernels and the split kernels (surface clouds with duplicates, other seeds)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
import torch
from oracle import model_oracle as MO, oracle as O
from graspnet_amd import GraspNet, pred_decode, synthetic as S
O.build(); O.set_num_threads(16); torch.set_num_threads(16)
sd = S.make_state_dict(0)
nets = {}
for math in ("f32", "split"):
    n = GraspNet().load_state_dict(sd); n.math_split = math == "split"; nets[math] = n


def rows(p):
    return {tuple(x[13:16].tolist()): x for x in p}


for seed, variant in ((7, "surface"), (21, "uniform"), (33, "surface")):
    cl = S.make_batch(seed, 1, 20000, variant)
    rep = MO.forward({"point_clouds": cl.clone()}, sd)
    tep = MO.forward({"point_clouds": cl.clone()}, sd, f64=True, force_top_view_inds=rep["grasp_top_view_inds"])
    rp, tp = MO.pred_decode(rep), MO.pred_decode(tep)
    line = "%s seed %d:" % (variant, seed)
    for name, pr, ep in [("oracle", rp, rep)] + [(m, pred_decode(nets[m]({"point_clouds": cl.cuda()})), None) for m in ("f32", "split")]:
        ka, kt = rows(pr[0].cpu().double()), rows(tp[0].double())
        worst, n, flips = 0.0, 0, 0
        for k in ka:
            if k in kt:
                x, y = ka[k], kt[k]
                if not torch.allclose(x[4:13], y[4:13], atol=1e-5) or abs(float(x[3] - y[3])) > 1e-6:
                    flips += 1
                    continue
                n += 1
                worst = max(worst, abs(float(x[0] - y[0])) / (1.0 + abs(float(y[0]))))
        line += "  %s rows %d flips %d max rel score err %.2e" % (name, n, flips, worst)
    for m in ("f32", "split"):
        ep = nets[m]({"point_clouds": cl.cuda()})
        e = (ep["view_score"].cpu().double() - tep["view_score"].double()).abs().max().item()
        eo = (rep["view_score"].double() - tep["view_score"].double()).abs().max().item()
        line += "  | view_score err %s %.2e (oracle %.2e)" % (m, e, eo)
    print(line)
