"""GPU parity of the 256x256-tile GEMM kernels (`gemm_kernel_v3`, the persistent `gemm_kernel_p`) -- the kernels the
headline's `roofline` is quoted on -- DIRECTLY through `tt_gemm_bf16`, at shapes large enough that the launcher picks
them (>= 128 tiles; smaller grids take the 128x128 kernel, tests/test_encoder_gpu.py), including odd tile counts
(the XCD remap and the super-tile walk must stay bijective) and the bench's own M = 473 600.

Reference: fp32 matmul of the same bf16-rounded operands on the CPU (oracle arithmetic: fp32 products, fp32 sums), bias /
exact-erf GELU / residual in fp32, ONE rounding to bf16 at the end -> |err| <= 2^-7 |ref| + 2e-3 per element.
At full size the CPU reference is taken on a random sample of the rows (every column), and the whole output is checked for
being finite and for row-permutation equivariance (rows of A permuted -> rows of C permuted, bit for bit)."""
import numpy as np
import pytest
import torch

from oracle import encoder as oe

pytestmark = pytest.mark.gpu


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _run(lib, _lib, a, w, bias, res, epi):
    m, k = a.shape
    n = w.shape[0]
    c = torch.empty(m, n, dtype=torch.bfloat16, device=a.device)
    rc = lib.tt_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), res.data_ptr() if epi == 2 else None,
                          c.data_ptr(), m, n, k, epi, _stream())
    _lib.check(rc, "tt_gemm_bf16")
    return c


def _ref_rows(a_rows, w, bias, res_rows, epi):
    ref = a_rows.float() @ w.float().T + bias
    if epi == 1:
        ref = oe.gelu_erf(ref)
    elif epi == 2:
        ref = ref + res_rows.float()
    return ref


# (m, n, k, epilogue): the four projections of a layer (QK / o-proj + residual / FFN-up + GELU / FFN-down + residual)
# with even and odd numbers of row tiles, tile counts that are not multiples of 8, and a K of 6 tiles (bge-small's 384)
SHAPES = [
    (8192, 2048, 1024, 0), (8192, 1024, 1024, 2), (8192, 4096, 1024, 1), (8192, 1024, 4096, 2),
    (256 * 37, 1024, 1024, 0), (256 * 37, 1024, 1024, 2), (256 * 37, 4096, 1024, 1), (256 * 45, 3072, 1024, 0),
    (256 * 33, 1024, 4096, 2), (256 * 129, 256, 384, 0), (256 * 131, 512, 1536, 1), (256 * 43, 768, 3072, 2),
]


@pytest.mark.parametrize("m,n,k,epi", SHAPES)
def test_256_tile_kernels_against_the_cpu_reference(dev, built_lib, m, n, k, epi):
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    assert (m // 256) * (n // 256) >= 128, "shape would take the 128x128 kernel"
    g = torch.Generator().manual_seed(m + n + k + epi)
    a = torch.randn(m, k, generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(n, generator=g) * 0.1
    res = torch.randn(m, n, generator=g).to(torch.bfloat16)
    got = _run(lib, _lib, a.to(dev), w.to(dev), bias.to(dev), res.to(dev), epi)
    torch.cuda.synchronize()
    ref = _ref_rows(a, w, bias, res, epi)
    err = (got.float().cpu() - ref).abs()
    bad = err > 2 ** -7 * ref.abs() + 2e-3
    assert not bad.any(), f"{int(bad.sum())} elements off, max err {err.max().item()}, first at {torch.nonzero(bad)[0].tolist()}"


@pytest.mark.parametrize("n,k,epi", [(2048, 1024, 0), (1024, 1024, 2), (4096, 1024, 1), (1024, 4096, 2)])
def test_bench_sized_gemms_sampled_rows_and_row_equivariance(dev, built_lib, n, k, epi):
    """M = 473 600 (32 queries x 50 pairs x 292 tokens + the padding of the packing: the bench's GEMM launches)."""
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    m = 473_600
    g = torch.Generator(device=dev).manual_seed(n + k + epi)
    a = torch.randn(m, k, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(n, device=dev, generator=g) * 0.1
    res = torch.randn(m, n, device=dev, generator=g).to(torch.bfloat16) if epi == 2 else torch.empty(1, device=dev)
    got = _run(lib, _lib, a, w, bias, res, epi)
    torch.cuda.synchronize()
    assert bool(got.view(torch.int16).bitwise_and(0x7F80).ne(0x7F80).all()), "non-finite output"
    # 1. a random sample of rows (every tile row has the same chance; first and last rows always) against the CPU
    rows = torch.from_numpy(np.unique(np.concatenate([np.random.default_rng(n + k).integers(0, m, 1536),
                                                      np.arange(0, 256), np.arange(m - 256, m)])))
    ref = _ref_rows(a[rows.to(dev)].cpu(), w.cpu(), bias.cpu(), res[rows.to(dev)].cpu() if epi == 2 else None, epi)
    err = (got[rows.to(dev)].float().cpu() - ref).abs()
    bad = err > 2 ** -7 * ref.abs() + 2e-3
    assert not bad.any(), f"{int(bad.sum())} sampled elements off, max err {err.max().item()}"
    # 2. size-independent property: a row of C depends on its own row of A (and of the residual) only -> permuting the
    # rows permutes the output, bit for bit, whichever tile, CU or round a row lands in
    perm = torch.randperm(m, device=dev, generator=g)
    got_p = _run(lib, _lib, a[perm].contiguous(), w, bias, res[perm].contiguous() if epi == 2 else res, epi)
    torch.cuda.synchronize()
    assert torch.equal(got_p, got[perm])
