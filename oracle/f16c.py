"""CPU restatement of the "c-planes" operand format of the f16c reference-precision path (csrc/f16c.h, f16c_path.hip).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  The format is this repository's own (the reference computes in
plain fp32: ``services/model_manager.py:333-337``, ``app_utils/config_schema.py:66-76``); what this module pins is that
the HIP producers (quantiser, GEMM / attention / LayerNorm epilogues) write exactly these planes and scale bytes, and
that a product assembled from them is the fp32 product to ~2^-16 -- so that the path's END result can be held to the
fp32 oracle (``oracle/encoder.py``) at north_star's 1e-3 relative.

    hi  = fp16(x)                        x8  = e4m3(x * 2^(127 - s))          lo8 = e4m3((x - hi) * 2^(127 - s + 11))
    s   = max(biased_exponent(absmax of the 32-element block) - 7, 0)         (an E8M0 byte: the block scale is 2^(s - 127))
    weight flavour: lo8 carries its own block exponent (part 0), x8's is stored 11 lower (part 1)
"""
from __future__ import annotations

import numpy as np
import torch

E4M3_MAX = 448.0


def _biased_exp(amax: torch.Tensor) -> torch.Tensor:
    e = (amax.contiguous().view(torch.int32) >> 23) & 0xFF
    return e.clamp(max=254)


def _e4m3(x: torch.Tensor) -> torch.Tensor:
    return x.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float32).to(torch.float8_e4m3fn)


def _ldexp(x: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
    """x * 2^e exactly (fp64: torch.ldexp forms 2^e in the tensor's own type, and 0 * 2^138 is NaN in fp32)."""
    return torch.ldexp(x.to(torch.float64), e.to(torch.int32))


def quantize(x: torch.Tensor, weight: bool = False):
    """fp32 [rows][K] -> dict(hi fp16 [rows][K], x8 / lo8 float8 [rows][K], s / s_lo uint8-valued int32 [rows][K/32]) as the
    HIP quantiser computes them (``s_lo`` only for the weight flavour; for activations lo8's scale is s - 11 by definition)."""
    x = x.to(torch.float32)
    rows, K = x.shape
    xb = x.view(rows, K // 32, 32)
    eb = _biased_exp(xb.abs().amax(-1))
    s = (eb - 7).clamp(min=0)
    hi = x.clamp(-65504.0, 65504.0).to(torch.float16)
    lo = (x - hi.to(torch.float32)).view(rows, K // 32, 32)
    sh = (127 - s).to(torch.float32).unsqueeze(-1)
    x8 = _e4m3(_ldexp(xb, sh)).view(rows, K)
    out = {"hi": hi, "s": s}
    if not weight:
        out["x8"] = x8
        out["lo8"] = _e4m3(_ldexp(lo, sh + 11)).view(rows, K)
        return out
    el = _biased_exp(lo.abs().amax(-1))
    s_lo = (el - 7).clamp(min=0)
    tiny = s < 11
    out["x8"] = torch.where(tiny.unsqueeze(-1).expand(-1, -1, 32).reshape(rows, K), torch.zeros_like(x8.to(torch.float32)),
                            x8.to(torch.float32)).to(torch.float8_e4m3fn)
    out["lo8"] = _e4m3(_ldexp(lo, (127 - s_lo).unsqueeze(-1))).view(rows, K)
    out["s_lo"] = s_lo
    out["s_x"] = torch.where(tiny, torch.zeros_like(s), s - 11)
    return out


def planes_bytes(q: dict, weight: bool = False) -> np.ndarray:
    """The row layout the kernels read: [hi: 2K | x8: K | lo8: K] (activations) or [hi | lo8 | x8] (weights), uint8 [rows][4K]."""
    hi = q["hi"].contiguous().view(torch.uint8).numpy().reshape(q["hi"].shape[0], -1)
    x8 = q["x8"].contiguous().view(torch.uint8).numpy()
    lo8 = q["lo8"].contiguous().view(torch.uint8).numpy()
    return np.concatenate([hi, lo8, x8] if weight else [hi, x8, lo8], axis=1)


def a_scale_at(row, blk, nks):
    r, g = row & 255, blk & 3
    img = ((((r >> 7) * 2 + ((r >> 6) & 1)) * 16 + (r & 15)) * 16) + g * 4 + ((r >> 4) & 3)
    return ((row >> 8) * nks + (blk >> 2)) * 1024 + img


def w_scale_at(n, part, blk, nks):
    c, g = n & 255, blk & 3
    img = ((((c >> 6) * 16 + (c & 15)) * 4 + g) * 4) + ((c >> 5) & 1) * 2 + ((c >> 4) & 1)
    return (((n >> 8) * 2 + part) * nks + (blk >> 2)) * 1024 + img


def tiled_scales(q: dict, weight: bool = False) -> np.ndarray:
    """The tiled scale array (csrc/f16c.h xc_a_scale_at / xc_w_scale_at) of a quantised operand; rows padded to 256."""
    rows, nblk = q["s"].shape
    nks = nblk // 4
    r256 = (rows + 255) // 256
    out = np.zeros(r256 * nks * 1024 * (2 if weight else 1), dtype=np.uint8)
    rr, bb = np.meshgrid(np.arange(rows), np.arange(nblk), indexing="ij")
    if weight:
        out[w_scale_at(rr, 0, bb, nks)] = q["s_lo"].numpy().astype(np.uint8)
        out[w_scale_at(rr, 1, bb, nks)] = q["s_x"].numpy().astype(np.uint8)
    else:
        out[a_scale_at(rr, bb, nks)] = q["s"].numpy().astype(np.uint8)
    return out


def dequant(q: dict, weight: bool = False):
    """-> (hi, x8, lo8) as fp32 values at their true magnitudes."""
    rows, K = q["hi"].shape
    hi = q["hi"].to(torch.float32)
    if weight:
        sx = (q["s_x"] + 11 - 127).to(torch.int32)
        sl = (q["s_lo"] - 127).to(torch.int32)
    else:
        sx = (q["s"] - 127).to(torch.int32)
        sl = sx - 11
    x8 = _ldexp(q["x8"].to(torch.float32).view(rows, K // 32, 32), sx.unsqueeze(-1)).to(torch.float32).view(rows, K)
    lo8 = _ldexp(q["lo8"].to(torch.float32).view(rows, K // 32, 32), sl.unsqueeze(-1)).to(torch.float32).view(rows, K)
    return hi, x8, lo8


def matmul(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """a [M][K] . w [N][K]^T the f16c way, accumulated in fp64: hi.hi + x8(a).lo8(w) + lo8(a).x8(w)."""
    ah, a8, al = dequant(quantize(a, False), False)
    wh, w8, wl = dequant(quantize(w, True), True)
    d = torch.float64
    return (ah.to(d) @ wh.to(d).T + a8.to(d) @ wl.to(d).T + al.to(d) @ w8.to(d).T).to(torch.float32)
