"""CPU: the C-ABI library builds/loads and exports every symbol include/graspnet_hip.h declares."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "graspnet_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gn_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from graspnet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = _lib.lib()
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "missing export " + n
        assert n in _lib.SIGNATURES, "binding lacks a signature for " + n
    assert set(_lib.SIGNATURES) == set(names)
    assert b"gfx950" in lib.gn_version()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "graspnet-baseline_amd", "graspnet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libgn_oracle" not in src, f


def test_cpu_tensors_are_refused():
    import pytest
    import torch
    from graspnet_amd import pointnet2_ext as E
    with pytest.raises(RuntimeError, match="CPU not supported"):
        E.ball_query(torch.zeros(1, 2, 3), torch.zeros(1, 5, 3), 0.1, 4)


def test_header_is_plain_c_and_links_against_the_library(tmp_path):
    """include/graspnet_hip.h is the drop-in boundary for C / C++ hosts: it must compile as C99 (no torch / C++ types in
    the signatures) and a C program must link against the shared library and call it without a GPU (version string,
    workspace sizes, argument refusal with a message)"""
    import shutil
    import subprocess
    from graspnet_amd import _lib as L
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "host.c"
    src.write_text('#include <stdio.h>\n#include <string.h>\n#include "graspnet_hip.h"\n'
                   'int main(void) {\n'
                   '  if (!gn_version() || !strlen(gn_version())) return 1;\n'
                   '  if (gn_graspnet_workspace_bytes(8, 20000) == 0 || gn_grid_workspace_bytes(8, 20000) == 0) return 2;\n'
                   '  if (gn_grad_workspace_bytes(2, 100, 640) != 4u * (2u * 101u + 2u * 100u + 2u * 640u)) return 3;\n'
                   '  if (gn_ball_query(NULL, NULL, 1, 10, 4, 0.1f, 8, NULL, NULL) == 0) return 4;   /* null pointers are refused */\n'
                   '  if (!strstr(gn_last_error(), "ball_query")) return 5;\n'
                   '  printf("%s\\n", gn_version());\n  return 0;\n}\n')
    exe = tmp_path / "host"
    libdir = os.path.dirname(L.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src),
                           "-o", str(exe), "-L", libdir, "-lgraspnet_hip", "-Wl,-rpath," + libdir])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert out.stdout.strip()


def test_argument_checks_answer_before_any_launch():
    """the status / gn_last_error convention on arguments the host can get wrong (no GPU needed: every check below fails
    before the first launch): a view count beyond the packed kernel shape, a null layer pointer, a misaligned workspace,
    nsample beyond the grid kernels' 64-lane register, grasps without counts"""
    import ctypes
    from graspnet_amd import _lib
    lib = _lib.lib()
    dummy = (ctypes.c_char * 4096)()
    addr = (ctypes.addressof(dummy) + 255) & ~255                    # a 256-byte-aligned non-null pointer that is never read
    w, o = _lib.Weights(), _lib.Outputs()
    for f, _ in _lib.Outputs._fields_:
        if f not in ("grasps", "counts", "fork_streams", "crop_uidx", "crop_ucount"):
            v = getattr(o, f)
            if hasattr(v, "__len__"):
                for i in range(len(v)):
                    v[i] = addr
            else:
                setattr(o, f, addr)

    def forward(ws=addr):
        return lib.gn_graspnet_forward(ctypes.byref(w), addr, 1, 4096, ws, ctypes.byref(o), _lib.STAGE_HOOK(), None, None)

    w.num_view = 400
    assert forward() == -1 and b"num_view" in lib.gn_last_error()
    w.num_view = 0
    assert forward() == -1 and b"weight or bias pointer" in lib.gn_last_error()          # all layer pointers are null
    assert forward(addr + 4) == -1 and b"256-byte aligned" in lib.gn_last_error()
    o.grasps = addr
    assert forward() == -1 and b"grasps and counts" in lib.gn_last_error()
    assert lib.gn_ball_query_grid(addr, addr, 1, 5000, 8, 0.04, 65, addr, None) == -1 and b"nsample <= 64" in lib.gn_last_error()
    assert lib.gn_furthest_point_sampling(None, 1, 100, 10, None, addr, None, None, None, None) == -1
    assert lib.gn_fps_workspace_bytes(8, 50000) >= 8 * 65536 * 20 and lib.gn_fps_workspace_bytes(8, 2048) == 0


def test_argument_checks_of_the_round4_entry_points():
    """the batched front end / post-processing entry points answer a bad call before any launch (no GPU needed)"""
    import ctypes
    from graspnet_amd import _lib
    lib = _lib.lib()
    dummy = (ctypes.c_char * 4096)()
    addr = (ctypes.addressof(dummy) + 255) & ~255
    # voxel down-sample: null output / misaligned workspace / non-positive voxel size
    assert lib.gn_voxel_down_sample_batched(addr, None, 2, 100, 0.01, addr, None, addr, addr, None) == -1
    assert b"null pointer" in lib.gn_last_error()
    assert lib.gn_voxel_down_sample_batched(addr, None, 2, 100, 0.01, addr + 8, addr, addr, addr, None) == -1
    assert b"256-byte aligned" in lib.gn_last_error()
    assert lib.gn_voxel_down_sample_batched(addr, None, 2, 100, 0.0, addr, addr, addr, addr, None) == -1
    assert lib.gn_voxel_batched_workspace_bytes(8, 20000) > 8 * 65536 * 12 and lib.gn_voxel_batched_workspace_bytes(0, 5) == 0
    assert lib.gn_voxel_workspace_bytes(20000) == 256 + lib.gn_voxel_batched_workspace_bytes(1, 20000)
    # collision detect: null scene / too many scene points for the queue's packed index
    assert lib.gn_collision_detect_batched(None, 100, None, addr, None, 1, 8, 0.05, 0.01, 0.01, 0.01, 0.01, 0.06, addr, None, None,
                                           None) == -1 and b"null pointer" in lib.gn_last_error()
    assert lib.gn_collision_detect_batched(addr, 1 << 28, None, addr, None, 1, 8, 0.05, 0.01, 0.01, 0.01, 0.01, 0.06, addr, None,
                                           None, None) == -1 and b"2^28" in lib.gn_last_error()
    # front end: zero depth factor, neither positions nor random words
    assert lib.gn_depth_valid_compact(addr, None, 1, 4, 4, 0.0, 0.0, 1.0, addr, addr, addr, None) == -1
    assert lib.gn_depth_sample_cloud(addr, 1, 4, 4, 1.0, 1.0, 0.0, 0.0, 1000.0, addr, addr, None, None, 8, addr, None, None) == -1
    assert b"positions or random words" in lib.gn_last_error()
    assert lib.gn_frontend_workspace_bytes(2, 720, 1280) == 2 * (900 + 1) * 4
    # knn: k > 8 needs the scratch plane
    assert lib.gn_knn(addr, addr, 1, 3, 10, 10, 9, addr, None, None) == -1 and b"scratch" in lib.gn_last_error()
