"""tensor_truth_amd -- MI355X (gfx950) implementation of tensor-truth's retrieval
hot path: batch chunk embedding, exact query x corpus top-k scan, cross-encoder
rerank, behind the reference's embedding-model / retriever / rerank-postprocessor
plugin surface.

All arithmetic runs in hand-written HIP kernels inside ``libtt_hip.so`` (C ABI in
``include/tt_hip.h``), called through ctypes.  There is no CPU fallback: importing
the compute modules without the built library, or calling them without a GPU,
raises.  PyTorch is used for device memory, streams and torch.distributed only.
"""
from ._lib import LibraryNotBuiltError, lib_path, load_library  # noqa: F401

__version__ = "0.1.0"
