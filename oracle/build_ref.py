"""Build oracle/_ref/knn_ref.so from the reference's own knn CPU source.

The reference file is compiled WHERE IT LIES (never copied): g++ on
<ref>/knn/src/cpu/knn_cpu.cpp plus our C-ABI trampoline knn_ref_shim.cpp.
knn_cpu.cpp pulls in <torch/extension.h> via cpu/vision.h, so the include
paths of the installed torch + CPython are passed; nothing from torch is
linked (the function is plain C++).  Output only into oracle/_ref/ (git-ignored).
"""
import argparse
import os
import subprocess
import sys
import sysconfig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref"))
    a = ap.parse_args()
    src = os.path.join(a.ref, "knn", "src", "cpu", "knn_cpu.cpp")
    if not os.path.exists(src):
        print("reference knn source not found at", src, "- skipping")
        return 0
    os.makedirs(a.out, exist_ok=True)
    out = os.path.join(a.out, "knn_ref.so")
    here = os.path.dirname(os.path.abspath(__file__))
    shim = os.path.join(here, "knn_ref_shim.cpp")
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(shim)):
        return 0
    from torch.utils.cpp_extension import include_paths
    inc = include_paths() + [sysconfig.get_paths()["include"], os.path.join(a.ref, "knn", "src")]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-w"]
    for i in inc:
        cmd += ["-I", i]
    cmd += ["-o", out, src, shim]
    print(" ".join(cmd))
    subprocess.check_call(cmd)
    return 0


if __name__ == "__main__":
    sys.exit(main())
