"""Per-iteration time of the small FPS levels (SA2: 2048 -> 1024, SA3: 1024 -> 512, SA4: 512 -> 256), full chains (no
prefix shortcut), batch 8.  The GN_FPS_NT switch this script sweeps existed only while the thread counts were chosen
(results in profiles/README.md); without it every line times the default.   python profiles/fps_small_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
import torch
from graspnet_amd import _lib as L
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(3)
for n, m in ((2048, 1024), (1024, 512), (512, 256)):
    x = torch.rand(8, n, 3, generator=g).cuda() * 0.5 + 0.2
    ref = None
    for nt in ("0", "64", "128", "256", "512"):
        if nt == "0": os.environ.pop("GN_FPS_NT", None)
        else: os.environ["GN_FPS_NT"] = nt
        idx = torch.empty(8, m, dtype=torch.int32, device="cuda")
        def run():
            L.check(lib.gn_furthest_point_sampling(x.data_ptr(), 8, n, m, None, idx.data_ptr(), None, None, None, st), "fps")
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        if ref is None: ref = idx.clone()
        print("n=%d m=%d NT=%s: %.1f us = %.0f ns/iteration  same indices: %s" % (n, m, nt, us, us * 1e3 / (m - 1), bool(torch.equal(ref, idx))), flush=True)
