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


def _device_disassembly(lib_path, tmp_path):
    """Disassembly of every gfx950 code object bundled in the shared library (one offload bundle per translation unit)."""
    import subprocess

    llvm = "/opt/rocm/lib/llvm/bin"
    fat = str(tmp_path / "fat.bin")
    subprocess.run([f"{llvm}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]
    assert starts, "no offload bundle in the library"
    out = []
    for i, s in enumerate(starts):
        chunk, co = str(tmp_path / f"b{i}.bin"), str(tmp_path / f"b{i}.co")
        with open(chunk, "wb") as fh:
            fh.write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={chunk}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        out.append(subprocess.run([f"{llvm}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout)
    return "\n".join(out)


def test_no_packed_f32_form_that_breaks_beside_mfma(built_lib, tmp_path):
    """gfx950, measured (tools/probes/pk_mfma_hazard.cpp, profiles/r03_pk_mfma_hazard.log): a v_pk_mul/add/fma_f32 whose op_sel bit
    for src1 is set (the LOW result reads src1's HIGH dword) returns a wrong low result in lanes 48-63 whenever another kernel's MFMA
    loop shares the SIMD; every other packed form is sound.  The compiler's SLP vectoriser emitted that form in embed_ln_kernel, which
    made forwards differ when a scan ran on a second stream.  No kernel of the library may hold it."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    dis = _device_disassembly(built_lib, tmp_path)
    assert dis.count("v_mfma_") > 1000                         # it is the device code we are looking at
    packed = re.findall(r"v_pk_(?:mul|add|fma)_f32[^\n]*", dis)
    assert len(packed) > 1000                                   # (the sound forms are in use: GEMM epilogues, softmax)
    fragile = [p for p in packed if re.search(r"op_sel:\[[01],1", p)]
    assert not fragile, f"{len(fragile)} packed-f32 instructions read src1's high dword for the low result, e.g. {fragile[0]}"
    # the other packed families (f16 / bf16 / integer) were not probed: none of them may carry that selector either
    other = [p for p in re.findall(r"v_pk_\w+[^\n]*", dis) if re.search(r"op_sel:\[[01],1", p)]
    assert not other, f"{len(other)} packed instructions with op_sel set for src1 (unprobed beside MFMAs), e.g. {other[0]}"
