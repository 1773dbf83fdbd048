"""config[3]'s operator stress shapes (N = 50 000, M = 2048 centres, nsample 64; batch 8) on their own, for the PMC passes of
profiles/run_profile_ops.sh: the same calls bench.py's `config3_n50000.op_rooflines` times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
import torch
import bench
print(bench.op_rooflines(torch.device("cuda", 0), reps=5, n_points=50000))
