"""Register the MI355X operators under the reference's module names so the
unchanged reference python (`import pointnet2._ext as _ext`,
pointnet2_utils.py:26; `from knn_pytorch import knn_pytorch`, knn_modules.py:7)
runs on them.  See INTEGRATION.md."""
import sys
import types

from . import knn_ext, pointnet2_ext


def install():
    pointnet2_ext.clear_grid_cache()       # a fresh installation starts without remembered clouds / sampling calls
    pointnet2_ext.clear_fps_chain()
    pkg = sys.modules.get("pointnet2")
    if pkg is None:
        pkg = types.ModuleType("pointnet2")
        pkg.__path__ = []
        sys.modules["pointnet2"] = pkg
    pkg._ext = pointnet2_ext
    sys.modules["pointnet2._ext"] = pointnet2_ext
    kp = types.ModuleType("knn_pytorch")
    kp.__path__ = []
    kp.knn_pytorch = knn_ext
    sys.modules["knn_pytorch"] = kp
    sys.modules["knn_pytorch.knn_pytorch"] = knn_ext
    return pointnet2_ext, knn_ext
