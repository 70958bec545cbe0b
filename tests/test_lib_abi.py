"""CPU: libtt_hip.so loads and exports every symbol include/tt_hip.h declares.
No compute call is made (no GPU here)."""
import ctypes
import os
import re

import pytest


def _declared_symbols():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "tt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tt_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound(built_lib):
    from tensor_truth_amd import _lib

    syms = _declared_symbols()
    assert len(syms) >= 8
    raw = ctypes.CDLL(built_lib)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in tt_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in _lib.SIGNATURES"
    for s in _lib.SIGNATURES:
        assert s in syms, f"{s} bound in _lib but not declared in tt_hip.h"
    lib = _lib.load_library()
    assert lib.tt_version() == 1
    assert lib.tt_arch() == b"gfx950"
    assert isinstance(lib.tt_last_error(), bytes)


def test_product_library_exports_exactly_the_header_and_reads_no_environment(built_lib):
    """`nm -D libtt_hip.so` = the names include/tt_hip.h declares -- no debug entry point rides along -- and the
    binary holds no TT_* string: every A/B switch of a measured-and-rejected variant is a compile-time constant in the product
    (csrc/common.h TT_DIAG_ENV_INT) and exists only in the diagnostic build (`make DIAG=1`, libtt_hip_diag.so)."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", built_lib], check=True, capture_output=True, text=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("tt_")})
    assert exported == _declared_symbols(), sorted(set(exported) ^ set(_declared_symbols()))
    # nothing ELSE either -- the library is built with -fvisibility=hidden and the header opens the C names; no mangled
    # C++ internal (_Z14tt_gemm_launch..., _Z16tt_select_launch...) or template instantiation rides along as a dynamic symbol.
    # (Left: the toolchain's own data symbols, __hip_cuid_* / __hip_fatbin*, which hipcc emits per translation unit.)
    stray = sorted(ln.split()[-1] for ln in out.splitlines()
                   if ln.split()[-1] not in exported and not ln.split()[-1].startswith(("__hip_", "_init", "_fini", "__bss", "_edata", "_end")))
    assert not stray, f"dynamic symbols beside the header's: {stray[:8]}"
    assert not [x for x in stray if "tt_" in x]
    data = open(built_lib, "rb").read()
    # (TT_EPI_* are the epilogue enumerators of tt_hip.h quoted in the text of a HIP error message, not switches)
    env_names = sorted({m.group(0).decode() for m in re.finditer(rb"TT_[A-Z0-9_]{3,}", data)} - {"TT_EPI_SCAN"})
    assert not env_names, f"the product library names environment switches: {env_names}"
    assert b"getenv" not in data, "the product library imports getenv"


def test_workspace_sizing_is_pure_host(built_lib):
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    small = lib.tt_scan_workspace_bytes(1000, 384, 16, 10)
    big = lib.tt_scan_workspace_bytes(1_000_000, 1024, 64, 50)
    assert 0 < small < big < (1 << 31)
    assert lib.tt_scan_exact_workspace_bytes(1000, 384, 16, 10) >= 64 * 1024 * 4
    assert lib.tt_scan_workspace_bytes(-1, 1024, 1, 1) == 0


def test_product_path_refuses_cpu_tensors(built_lib):
    import torch

    from tensor_truth_amd import scan

    c = torch.zeros(8, 128, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        scan.scan_topk(c, c[:2], 2)


def _check_isa():
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("tt_check_isa", os.path.join(root, "tensor-truth_amd", "csrc", "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_no_packed_f32_form_that_breaks_beside_mfma(built_lib, tmp_path):
    """gfx950, measured (tools/probes/pk_mfma_hazard.cpp, profiles/r03_pk_mfma_hazard.log): a v_pk_mul/add/fma_f32 whose op_sel bit
    for src1 is set (the LOW result reads src1's HIGH dword) returns a wrong low result in lanes 48-63 whenever another kernel's MFMA
    loop shares the SIMD; every other packed form is sound.  The compiler's SLP vectoriser emitted that form in embed_ln_kernel, which
    made forwards differ when a scan ran on a second stream.  No kernel of the library may hold it.
    The same check is a post-link step of the Makefile (csrc/check_isa.py), whose stamp gates the retriever's second stream."""
    ck = _check_isa()
    if not os.path.exists(f"{ck.LLVM}/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    dis = ck.device_disassembly(built_lib, str(tmp_path))
    fragile, other, n_packed, n_mfma = ck.fragile_packed(dis)
    assert n_mfma > 1000                                        # it is the device code we are looking at
    assert n_packed > 1000                                      # (the sound forms are in use: GEMM epilogues, softmax)
    assert not fragile, f"{len(fragile)} packed-f32 instructions read src1's high dword for the low result, e.g. {fragile[0]}"
    # the other packed families (f16 / bf16 / integer) were not probed: none of them may carry that selector either
    assert not other, f"{len(other)} packed instructions with op_sel set for src1 (unprobed beside MFMAs), e.g. {other[0]}"


def test_isa_stamp_matches_the_built_library(built_lib):
    """The Makefile's post-link check stamped THIS binary (sha256): the retriever may use its own stream.  A stale or missing stamp
    (a build without llvm-objdump, a library swapped in by hand) turns the second stream off instead of trusting it."""
    from tensor_truth_amd import _lib

    ck = _check_isa()
    if not os.path.exists(f"{ck.LLVM}/llvm-objdump") and not os.path.exists(ck.stamp_path(built_lib)):
        pytest.skip("built without llvm-objdump: no stamp, second stream off")
    assert open(ck.stamp_path(built_lib)).read().strip() == ck.sha256_of(built_lib)
    assert _lib.isa_checked()
    # the fragile form is recognised when it is there
    assert ck.fragile_packed("v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]\nv_mfma_f32_16x16x32_bf16 a, b, c\n")[0]
    assert not ck.fragile_packed("v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[1,0]\n")[0]
