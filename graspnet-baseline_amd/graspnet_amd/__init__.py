"""graspnet_amd -- MI355X-native GraspNet inference path.

  GraspNet, pred_decode      drop-in for models/graspnet.py (inference)
  load_checkpoint            build the model from a checkpoint file as the reference's callers do (weights-only loader)
  PipelinedGraspNet          successive batches on several HIP streams (the throughput a plain loop does not get)
  GraphedGraspNet            one batch shape captured in a hipGraph (latency)
  pointnet2_ext, knn_ext     drop-in for the native modules pointnet2._ext / knn_pytorch.knn_pytorch
  dropin.install()           registers the two above under the reference's module names
  set_fp_policy              how the index-deciding float expressions are evaluated: "as_written" (default) | "contracted" |
                             "contracted_b" (the reference's nvcc build contracts them into FMAs: include/graspnet_hip.h)
"""
from .graspnet import GraphedGraspNet, GraspNet, PipelinedGraspNet, load_checkpoint, pred_decode  # noqa: F401
from ._lib import set_fp_policy  # noqa: F401,E402
