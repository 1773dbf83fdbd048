"""group_points drop-in at the SA2 shape: sweep of the LDS-staged kernel's channel group / grid size (GN_GROUP_CG, GN_GROUP_WGS
are read once per process, so each setting runs in its own process)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, os.path.join(ROOT, "graspnet-baseline_amd"))
    import torch
    from graspnet_amd import pointnet2_ext as E
    res = []
    for (b, c, n, npnt, ns) in ((8, 128, 2048, 1024, 32), (8, 256, 1024, 512, 16), (8, 256, 512, 256, 16)):
        f = torch.rand(b, c, n, device="cuda"); idx = torch.randint(0, n, (b, npnt, ns), device="cuda", dtype=torch.int32)
        E.group_points(f, idx); torch.cuda.synchronize()
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): E.group_points(f, idx)
        z.record(); torch.cuda.synchronize()
        t = a.elapsed_time(z) / 50 * 1e-3
        byts = (b * c * n + b * npnt * ns + b * c * npnt * ns) * 4
        res.append("%dx%dx%d: %.1f us %.2f TB/s" % (c, n, npnt * ns, t * 1e6, byts / t / 1e12))
    print("CG=%s WGS=%s  " % (os.environ.get("GN_GROUP_CG", "-"), os.environ.get("GN_GROUP_WGS", "-")) + " | ".join(res))
else:
    for cg in ("16", "8", "4", "2"):
        for wgs in ("512", "768", "1024", "2048"):
            env = dict(os.environ, GN_GROUP_CG=cg, GN_GROUP_WGS=wgs)
            subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, check=False)
