"""Which arithmetic a HIP embedder / reranker runs in, and who decides.

The reference never passes a dtype to its reranker (``SentenceTransformerRerank(model=, top_n=, device=)``,
``src/tensortruth/services/model_manager.py:333-337``) and passes one to its embedder only when the per-model config
holds ``torch_dtype`` (``services/model_manager.py:218-229``; ``app_utils/config_schema.py:66-76``: default None), so the
reference's own precision is fp32 -- and so is this package's DEFAULT: an unchanged reference call (no dtype anywhere)
resolves to mode "reference", whose scores are within north_star's 1e-3 relative of the reference's CPU path.
BASELINE.json's configurations name bf16 (and fp8 for config 5): those are the modes a caller NAMES (``torch_dtype:
"bfloat16"`` in the per-model config, as the reference's own config schema allows; ``TT_PRECISION=bf16``;
``ModelManager.set_precision("bf16")``), and the ones ``bench.py`` names for its headline.

    mode "reference"  fp32 semantics: scores within north_star's 1e-3 relative of the CPU path.  Implementations
                      (``TT_REFERENCE_IMPL``); "f16x3" / "bf16x3" for hidden a multiple of 128 with 64- or 32-wide heads (round 6:
                      bge-small-en-v1.5 and ms-marco-MiniLM-L-6-v2 included), "f16c" for multiples of 256 with 64-wide heads,
                      anything else "fp32":
                        "f16x3"  (default, round 4) every operand as two fp16 planes, three fp16 MFMA products per product
                                 (``encoder_x3`` on fp16 planes): a third of the bf16 matrix rate, 22 significand bits -- holds
                                 1e-3 with a margin even on the stress fixture (tests/stress_weights.py)
                        "f16c"   fp16 main products + block-scaled e4m3 correction terms (``encoder_f16c``): HALF the bf16
                                 rate, 1e-4 on the standard fixture, but 7e-3 relative on the stress fixture's smallest scores
                                 (ranking intact): the fast variant, for callers who accept that
                        "bf16x3" round 3's two bf16 planes (16 bits): 2e-5 / 1.0e-3 (standard / stress)
                        "fp32"   the fp32 MFMA (``encoder_f32``, 1/16 of the bf16 rate)
    mode "bf16"       bf16 weights / activations, fp32 accumulate (BASELINE configs 2-4; the reference's ``torch_dtype: bfloat16``)
    mode "fp16"       IEEE fp16 weights / activations, fp32 accumulate: the bf16 mode's rate (v_mfma_*_f16) with three more
                      mantissa bits at every rounding point -- scores about ten times closer to the reference than bf16;
                      what the reference's ``torch_dtype: "float16"`` asks for, and FlagEmbedding's own default for the BGE
                      models.  Values beyond +-65504 saturate at the GEMM / LayerNorm outputs.
    mode "fp8"        the layer projections in OCP e4m3 (BASELINE config 5's "fp8 MFMA reranker")

Resolution order (first that says something): ``model_kwargs["precision"]``; ``model_kwargs["torch_dtype"]`` (float32 ->
reference; float16 -> fp16; bfloat16 -> bf16) and ``model_kwargs["gemm_dtype"]``; ``ModelManager.precision`` (a config key the
application sets once; ModelManager puts it into the model_kwargs it builds); the process environment ``TT_PRECISION``;
"reference" (round 4; rounds 1-3 defaulted to bf16, which put the unchanged calls 10x outside the tolerance).  The active
mode is logged when a model is loaded.
"""
from __future__ import annotations

import logging
import os
from typing import Any, Dict, Optional, Tuple

logger = logging.getLogger(__name__)

MODES = ("bf16", "fp16", "fp8", "reference")
# what a call that names no dtype gets: the reference's own arithmetic (services/model_manager.py:333-337 passes none;
# app_utils/config_schema.py:66-76 defaults torch_dtype to None) -- fp32 semantics
DEFAULT_MODE = "reference"
_ALIASES = {"bf16": "bf16", "bfloat16": "bf16", "fp16": "fp16", "float16": "fp16", "half": "fp16", "fp8": "fp8", "e4m3": "fp8",
            "reference": "reference", "fp32": "reference", "float32": "reference", "float": "reference", "bf16x3": "reference"}


def canonical(value) -> str:
    v = str(value).replace("torch.", "").strip().lower()
    if v not in _ALIASES:
        raise ValueError(f"precision {value!r}: expected one of {MODES}")
    return _ALIASES[v]


def resolve(model_kwargs: Optional[Dict[str, Any]] = None, environ=None) -> str:
    mk = model_kwargs or {}
    if mk.get("precision") is not None:
        return canonical(mk["precision"])
    td = mk.get("torch_dtype")
    if td is not None:
        t = str(td).replace("torch.", "")
        if t in ("float32", "fp32", "float"):
            return "reference"
        if t in ("float16", "fp16", "half") and mk.get("gemm_dtype") is None:
            return "fp16"
        if mk.get("gemm_dtype") is None:
            return "bf16"          # bfloat16, and anything else the HIP path maps to bf16 (logged by the callers)
    if mk.get("gemm_dtype") is not None:
        return canonical(mk["gemm_dtype"])
    env = (os.environ if environ is None else environ).get("TT_PRECISION")
    if env:
        return canonical(env)
    return DEFAULT_MODE


def reference_impl(cfg, environ=None) -> str:
    """"f16x3" (two fp16 planes per operand: three matrix-time units) where the model shape fits, else "fp32" (fp32 MFMA);
    ``TT_REFERENCE_IMPL`` = "f16c" / "bf16x3" / "fp32" selects another implementation."""
    from . import encoder_f16c, encoder_x3

    forced = (os.environ if environ is None else environ).get("TT_REFERENCE_IMPL", "").strip().lower()
    if forced in ("fp32", "float32", "f32"):
        return "fp32"
    if forced in ("bf16x3", "x3", "split-bf16"):
        return "bf16x3" if encoder_x3.supports(cfg) else "fp32"
    if forced in ("f16c", "fast"):
        return "f16c" if encoder_f16c.supports(cfg) else "fp32"
    return "f16x3" if encoder_x3.supports(cfg) else "fp32"


def build_encoder(cfg, state, device, model_kwargs: Optional[Dict[str, Any]], what: str) -> Tuple[Any, Any, str]:
    """-> (weights, encoder, description) for the resolved precision; logs it."""
    from .encoder import Encoder, EncoderWeights

    mode = resolve(model_kwargs)
    if mode == "reference":
        impl = reference_impl(cfg)
        if impl == "f16c":
            from .encoder_f16c import EncoderF16C, EncoderWeightsF16C

            w = EncoderWeightsF16C(cfg, state, device)
            enc, desc = EncoderF16C(w), ("reference (fp32 semantics as fp16 products + block-scaled e4m3 correction terms on the "
                                         "matrix cores, fp32 residual stream)")
        elif impl == "f16x3":
            import torch

            from .encoder_x3 import EncoderWeightsX3, EncoderX3

            w = EncoderWeightsX3(cfg, state, device, dtype=torch.float16)
            enc, desc = EncoderX3(w), "reference (fp32 semantics as split-fp16 on the fp16 matrix cores, fp32 residual stream)"
        elif impl == "bf16x3":
            from .encoder_x3 import EncoderWeightsX3, EncoderX3

            w = EncoderWeightsX3(cfg, state, device)
            enc, desc = EncoderX3(w), "reference (fp32 semantics as split-bf16 on the bf16 matrix cores, fp32 residual stream)"
        else:
            from .encoder_f32 import EncoderF32, EncoderWeightsF32

            w = EncoderWeightsF32(cfg, state, device)
            enc, desc = EncoderF32(w), "reference (fp32 weights, activations and MFMA)"
            if os.environ.get("TT_REFERENCE_IMPL", "").strip().lower() not in ("fp32", "float32", "f32"):
                # the rate cliff of the default mode: the split-plane kernels take hidden sizes that are multiples of 128
                # with 64- or 32-wide heads (bge-m3, the bge rerankers; round 6: bge-small 384 and MiniLM's 32-wide heads too); any
                # other shape runs on the fp32 MFMA -- ~1/16 of the bf16 matrix rate instead of 1/3
                logger.warning("%s: hidden=%d heads=%d has no split-plane kernel: the reference precision falls back to the fp32 MFMA "
                               "path (about 1/16 of the bf16 rate; name torch_dtype=bfloat16 / float16 or TT_PRECISION for the "
                               "16-bit modes)", what, getattr(cfg, "hidden", -1), getattr(cfg, "heads", -1))
    elif mode == "fp16":
        import torch

        w = EncoderWeights(cfg, state, device, dtype=torch.float16)
        enc, desc = Encoder(w), "fp16 (fp32 accumulate)"
    else:
        w = EncoderWeights(cfg, state, device)
        w.set_gemm_dtype(mode)
        enc, desc = Encoder(w), ("bf16 (fp32 accumulate)" if mode == "bf16" else "fp8 e4m3 layer projections, bf16 elsewhere")
    logger.info("%s: precision = %s", what, desc)
    return w, enc, desc
