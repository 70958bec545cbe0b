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
