"""ctypes binding of libtt_hip.so (include/tt_hip.h).  Fails loudly when the HIP
library has not been built -- there is deliberately no fallback path."""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libtt_hip.so"
_lock = threading.Lock()
_lib = None


class LibraryNotBuiltError(RuntimeError):
    pass


class TTError(RuntimeError):
    """A libtt_hip.so call returned a non-zero status."""


def lib_path() -> str:
    # TT_LIB_NAME=libtt_hip_diag.so: the diagnostic build (csrc `make DIAG=1`), for tools/probes only
    return os.path.join(_HERE, os.environ.get("TT_LIB_NAME") or _LIB_NAME)


def is_diag() -> bool:
    """True when the diagnostic build is the one loaded (TT_LIB_NAME=libtt_hip_diag.so, `make DIAG=1`) -- the only build that
    reads the kernels' A/B environment switches; the product library reads no environment variable at all."""
    return "diag" in (os.environ.get("TT_LIB_NAME") or "")


_isa_ok = None


def isa_checked() -> bool:
    """True when csrc/check_isa.py has disassembled THIS build of the library and found no packed-f32 instruction of the form
    that misbehaves beside another stream's MFMAs (its stamp, ``libtt_hip.so.isa_ok``, holds the library's sha256).  Code that
    runs kernels on two streams at once (the retriever's own stream, vector_index.py) asks first and stays on one stream
    otherwise."""
    global _isa_ok
    if _isa_ok is None:
        import hashlib

        ok = False
        try:
            with open(lib_path() + ".isa_ok") as fh:
                want = fh.read().strip()
            h = hashlib.sha256()
            with open(lib_path(), "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 20), b""):
                    h.update(blk)
            ok = h.hexdigest() == want
        except OSError:
            ok = False
        _isa_ok = ok
    return _isa_ok


_F32P = ctypes.POINTER(c_float)
_I32P = ctypes.POINTER(c_int32)

# name -> (restype, argtypes); must list every symbol include/tt_hip.h declares
SIGNATURES = {
    "tt_version": (c_int, []),
    "tt_arch": (c_char_p, []),
    "tt_last_error": (c_char_p, []),
    "tt_device_cu_count": (c_int, []),
    "tt_scan_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "tt_scan_topk": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_int, c_int32, c_void_p, c_void_p,
                             c_void_p, c_size_t, c_void_p, c_void_p]),
    "tt_scan_exact_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "tt_scan_topk_exact": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_int, c_int32, c_void_p, c_void_p,
                                   c_void_p, c_size_t, c_void_p]),
    "tt_scan_segmented_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "tt_scan_topk_segmented": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                       c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_scan_shadow_bytes": (c_size_t, [c_int64, c_int]),
    "tt_scan_shadow_build": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "tt_scan_shadow_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "tt_scan_topk_shadow": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_int, c_int, c_int32, c_void_p, c_void_p,
                                    c_void_p, c_size_t, c_void_p, c_void_p]),
    "tt_topk_merge": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "tt_encoder_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "tt_encoder_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_encoder_cls_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "tt_encoder_forward_cls": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                       c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_embed_pool": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "tt_embed_pool_mean": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "tt_embed_pool_mean_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    # fp16 twins of the 16-bit path (same signatures)
    "tt_encoder_workspace_bytes_f16": (c_size_t, [c_void_p, c_int]),
    "tt_encoder_cls_workspace_bytes_f16": (c_size_t, [c_void_p, c_int, c_int]),
    "tt_encoder_forward_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_encoder_forward_cls_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                       c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_embed_pool_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "tt_embed_pool_mean_f16": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "tt_rerank_head_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_gemm_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "tt_layernorm_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "tt_attention_varlen_f16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                    c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "tt_rerank_head": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_adjacent_cosine": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "tt_gemm_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "tt_layernorm_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "tt_attention_varlen": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                    c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "tt_quantize_rows_fp8": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "tt_layernorm_bf16_fp8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p,
                                      c_void_p]),
    "tt_gemm_fp8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                            c_void_p]),
    "tt_gemm_fp8_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                               c_int, c_int, c_int, c_int, c_void_p]),
    "tt_encoder_f32_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "tt_encoder_forward_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                       c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_embed_pool_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "tt_rerank_head_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_gemm_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "tt_encoder_x3_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "tt_encoder_x3_cls_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "tt_encoder_forward_x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                      c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_encoder_forward_x3_cls": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                      c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_rerank_head_x3": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_split_planes": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "tt_gemm_x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "tt_attention_x3": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    # the split-plane forward with fp16 planes ("f16x3": x3_path.hip's second instantiation)
    "tt_encoder_x3_workspace_bytes_f16": (c_size_t, [c_void_p, c_int]),
    "tt_encoder_x3_cls_workspace_bytes_f16": (c_size_t, [c_void_p, c_int, c_int]),
    "tt_encoder_forward_x3_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                          c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_encoder_forward_x3_cls_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                              c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_rerank_head_x3_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_split_planes_f16": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "tt_gemm_x3_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "tt_attention_x3_f16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                    c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    # reference precision on two matrix-time units (csrc/f16c_path.hip)
    "tt_encoder_f16c_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "tt_encoder_f16c_cls_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "tt_encoder_forward_f16c": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_encoder_forward_f16c_cls": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                            c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_rerank_head_f16c": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tt_f16c_scale_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "tt_f16c_quantize": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "tt_gemm_f16c": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                             c_int, c_void_p]),
    "tt_attention_f16c": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                  c_int, c_int, c_void_p]),
    "tt_prof_enable": (c_int, [c_int]),
    "tt_prof_read": (c_int, [c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int)]),
}


def load_library():
    """Load (once) and return the ctypes handle; raises LibraryNotBuiltError."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not os.path.exists(path):
            raise LibraryNotBuiltError(
                f"{path} not found. Build it first: `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C {os.path.join(_HERE, 'csrc')}` (needs hipcc, --offload-arch=gfx950). "
                "tensor_truth_amd has no CPU fallback."
            )
        # torch FIRST: its wheel carries its own copy of the HIP runtime, and whichever copy initialises second in a process sees
        # "no ROCm-capable device" -- the library's launches must land on the runtime torch's tensors and streams live in
        import torch  # noqa: F401

        lib = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load_library().tt_last_error().decode("utf-8", "replace")
        raise TTError(f"{what} failed with status {status}: {msg}")
