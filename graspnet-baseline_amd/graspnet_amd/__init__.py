"""graspnet_amd -- MI355X-native GraspNet inference path.

  GraspNet, pred_decode      drop-in for models/graspnet.py (inference)
  pointnet2_ext, knn_ext     drop-in for the native modules pointnet2._ext / knn_pytorch.knn_pytorch
  dropin.install()           registers the two above under the reference's module names
"""
from .graspnet import GraphedGraspNet, GraspNet, pred_decode  # noqa: F401
