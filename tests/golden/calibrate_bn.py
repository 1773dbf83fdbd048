"""Calibrate the BN running statistics of the synthetic weights.

Random conv weights with random BN statistics give a degenerate network (every
seed picks the same approach view).  This script runs the CPU
oracle model once over two synthetic scenes with BN statistics taken from the
batch (plus a seeded perturbation) and stores ONLY those statistics
(~8.7k floats) in graspnet_amd/data/bn_calib_s<seed>.npz;
graspnet_amd.synthetic.make_state_dict() overlays them on its numpy-generated
weights.  Deterministic inputs, run once in the build container; the file is a
data fixture of ours (no reference content).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
from oracle import model_oracle as MO     # noqa: E402
from graspnet_amd import synthetic as S   # noqa: E402


def main(seed=0):
    sd = S.make_state_dict(seed, calibrated=False)
    clouds = torch.stack([torch.from_numpy(S.make_cloud(900, 20000, "uniform")),
                          torch.from_numpy(S.make_cloud(901, 20000, "surface"))], 0)
    MO.CALIBRATE_RNG = np.random.default_rng(seed + 12345)
    try:
        MO.forward({"point_clouds": clouds}, sd)
    finally:
        MO.CALIBRATE_RNG = None
    out = {k: v.numpy() for k, v in sd.items() if k.endswith("running_mean") or k.endswith("running_var")}
    path = os.path.join(ROOT, "graspnet-baseline_amd", "graspnet_amd", "data", "bn_calib_s%d.npz" % seed)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "tensors")


if __name__ == "__main__":
    for sd_seed in ([int(a) for a in sys.argv[1:]] or [0]):
        main(sd_seed)
