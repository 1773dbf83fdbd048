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
