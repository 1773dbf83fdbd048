"""Drop-in for `knn_pytorch.knn_pytorch` (knn/src/vision.cpp:3-5, knn/src/knn.h:12-65).

knn(ref (B,D,Nr) f32, query (B,D,Nq) f32, idx (B,k,Nq) int64) -> 1, idx filled in
place with 1-BASED reference indices.  Like the reference it performs no shape
checks ("TODO check dimensions", knn.h:15) beyond what is needed not to fault.
GPU tensors only in this build: the reference's CPU branch (knn_cpu.cpp) is the
oracle's job, not the product's.
"""
import torch

from . import _lib as L


def knn(ref, query, idx):
    if not ref.is_cuda:
        raise RuntimeError("CPU not supported")
    if not (query.is_cuda and idx.is_cuda):
        raise RuntimeError("query and idx must be CUDA tensors")
    if ref.dtype != torch.float32 or query.dtype != torch.float32 or idx.dtype != torch.int64:
        raise RuntimeError("knn: ref/query must be float tensors and idx a long tensor")
    if not (ref.is_contiguous() and query.is_contiguous() and idx.is_contiguous()):
        raise RuntimeError("knn: tensors must be contiguous")
    lib = L.lib()
    b, d, nr = ref.shape
    nq = query.size(2)
    k = idx.size(1)
    scratch = torch.empty((b, k, nq), dtype=torch.float32, device=ref.device)
    with torch.cuda.device(ref.device):
        L.check(lib.gn_knn(ref.data_ptr(), query.data_ptr(), b, d, nr, nq, k, idx.data_ptr(),
                           scratch.data_ptr(), L.stream_of(ref)), "knn")
    return 1
