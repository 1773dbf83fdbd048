"""CPU, build container only: dropin.install() under the REFERENCE's own python.

The reference's wrappers (`/root/reference/pointnet2/pointnet2_utils.py:26` `import pointnet2._ext as _ext`,
`/root/reference/knn/knn_modules.py:7` `from knn_pytorch import knn_pytorch`) are imported from where they lie, over the
modules dropin.install() registers; every `_ext.<fn>(...)` / `knn_pytorch.knn(...)` call site in those two files is
found by parsing their source, and must resolve to a shim function taking exactly that many positional arguments
(pointnet2_utils.py:71,105,111,138,175,197,230,250,280,451; knn_modules.py:16).  No compute runs here (no GPU): the
reference's autograd Functions are driven with CPU tensors, which must fail with the reference extension's own message.
Skipped where /root/reference does not exist (the GPU box)."""
import ast
import inspect
import os
import sys

import pytest
import torch

REF = os.environ.get("GRASPNET_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "pointnet2")), reason="reference checkout not present")


def _ext_calls(path, owner):
    """[(function name, number of positional args, line)] for every `<owner>.<fn>(...)` call in a source file."""
    tree = ast.parse(open(path).read())
    out = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and isinstance(node.func.value, ast.Name) \
                and node.func.value.id == owner:
            assert not node.keywords, "keyword call at %s:%d" % (path, node.lineno)
            out.append((node.func.attr, len(node.args), node.lineno))
    return out


@pytest.fixture(scope="module")
def installed():
    from graspnet_amd import dropin
    saved = {k: sys.modules.get(k) for k in ("pointnet2", "pointnet2._ext", "knn_pytorch", "knn_pytorch.knn_pytorch",
                                             "pointnet2_utils", "pytorch_utils", "knn_modules")}
    saved_path = list(sys.path)
    ext, knn = dropin.install()
    sys.path[:0] = [os.path.join(REF, "pointnet2"), os.path.join(REF, "knn")]
    import pointnet2_utils as ref_utils      # the reference's file, unchanged
    import knn_modules as ref_knn
    yield ext, knn, ref_utils, ref_knn
    sys.path[:] = saved_path
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def test_reference_wrappers_import_over_the_dropin(installed):
    ext, knn, ref_utils, ref_knn = installed
    assert ref_utils._ext is ext and ref_knn.knn_pytorch is knn
    assert ref_utils.__file__.startswith(REF) and ref_knn.__file__.startswith(REF)


def test_every_ext_call_site_resolves_with_the_right_arity(installed):
    ext, knn, ref_utils, ref_knn = installed
    calls = _ext_calls(ref_utils.__file__, "_ext")
    names = {c[0] for c in calls}
    assert names == {"furthest_point_sampling", "gather_points", "gather_points_grad", "three_nn", "three_interpolate",
                     "three_interpolate_grad", "group_points", "group_points_grad", "ball_query", "cylinder_query"}
    for fn, nargs, line in calls:
        f = getattr(ext, fn, None)
        assert callable(f), "pointnet2_utils.py:%d calls _ext.%s, which the drop-in does not export" % (line, fn)
        params = [p for p in inspect.signature(f).parameters.values()
                  if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) and p.default is p.empty]
        assert len(params) == nargs, "pointnet2_utils.py:%d passes %d args to _ext.%s, the drop-in takes %d" % (line, nargs, fn, len(params))
    kcalls = _ext_calls(ref_knn.__file__, "knn_pytorch")
    assert [(c[0], c[1]) for c in kcalls] == [("knn", 3)]
    assert len(inspect.signature(knn.knn).parameters) == 3


def test_reference_functions_reach_the_shim_and_fail_like_the_reference_on_cpu(installed):
    """the reference's autograd Functions call straight into the shim; a CPU tensor gets the reference extension's own
    refusal (e.g. sampling.cpp:86 / ball_query.cpp:33 `TORCH_CHECK(false, "CPU not supported")`)"""
    _, _, U, K = installed
    xyz = torch.rand(1, 32, 3)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        U.furthest_point_sample(xyz, 8)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        U.ball_query(0.1, 4, xyz, xyz[:, :4].contiguous())
    with pytest.raises(RuntimeError, match="CPU not supported"):
        U.gather_operation(xyz.transpose(1, 2).contiguous(), torch.zeros(1, 4, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="CPU not supported"):
        U.three_nn(xyz, xyz)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        U.cylinder_query(0.05, -0.02, 0.04, 8, xyz, xyz[:, :4].contiguous(), torch.eye(3).reshape(1, 1, 9).repeat(1, 4, 1))
    # the reference checks contiguity / dtype in C++ before dispatch (utils.h:10-30); the shim raises the same messages
    with pytest.raises(RuntimeError, match="must be a contiguous tensor"):
        U.grouping_operation(torch.rand(1, 32, 3).transpose(1, 2), torch.zeros(1, 4, 2, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="CPU not supported"):
        K.knn(torch.rand(1, 3, 16), torch.rand(1, 3, 4), 1)
