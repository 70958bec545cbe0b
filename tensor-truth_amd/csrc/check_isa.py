#!/usr/bin/env python3
"""Post-link guard of libtt_hip.so: no kernel may hold the packed-f32 form that misbehaves beside another kernel's MFMAs.

Measured on gfx950 (tools/probes/pk_mfma_hazard.cpp, profiles/r03_pk_mfma_hazard.log -- an empirical finding of this build,
not a documented erratum): a ``v_pk_mul/add/fma_f32`` whose op_sel bit for src1 is set (the LOW result reads src1's HIGH
dword) returned a wrong low result in lanes 48-63 whenever another stream's MFMA loop shared the SIMD.  hipcc's SLP
vectoriser emits that form; ``rowops.hip`` is therefore built with ``-fno-slp-vectorize`` (Makefile: ``override``, so a
command-line CXXFLAGS cannot drop it).  Results only go wrong under two-stream load, silently -- so the build itself checks:

    python3 check_isa.py ../libtt_hip.so        (run by the Makefile after every link, and by __graft_entry__.build())

disassembles every gfx950 code object of the library and exits non-zero on a hit.  On success it writes
``<library>.isa_ok`` = sha256 of the library; ``tensor_truth_amd._lib.isa_checked()`` compares it at load time, and the
retriever keeps its scans on the caller's stream (one stream: the form is harmless there) when the stamp is missing or
stale -- e.g. a build on a machine without llvm-objdump.
"""
from __future__ import annotations

import hashlib
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("TT_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
FRAGILE = re.compile(r"op_sel:\[[01],1")


def device_disassembly(lib_path: str, tmp_dir: str) -> str:
    """Disassembly of every gfx950 code object bundled in the shared library (one offload bundle per translation unit)."""
    fat = os.path.join(tmp_dir, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]
    if not starts:
        raise RuntimeError("no offload bundle in the library")
    out = []
    for i, s in enumerate(starts):
        chunk, co = os.path.join(tmp_dir, f"b{i}.bin"), os.path.join(tmp_dir, f"b{i}.co")
        with open(chunk, "wb") as fh:
            fh.write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={chunk}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        out.append(subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout)
    return "\n".join(out)


def fragile_packed(dis: str):
    """-> (fragile packed-f32 instructions, other packed instructions carrying the same selector, #packed-f32, #mfma)."""
    packed = re.findall(r"v_pk_(?:mul|add|fma)_f32[^\n]*", dis)
    fragile = [p for p in packed if FRAGILE.search(p)]
    other = [p for p in re.findall(r"v_pk_\w+[^\n]*", dis) if FRAGILE.search(p) and p not in fragile]
    return fragile, other, len(packed), dis.count("v_mfma_")


def sha256_of(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def stamp_path(lib_path: str) -> str:
    return lib_path + ".isa_ok"


def main(argv) -> int:
    lib_path = os.path.abspath(argv[1] if len(argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "libtt_hip.so"))
    stamp = stamp_path(lib_path)
    if os.path.exists(stamp):
        os.remove(stamp)
    if not os.path.exists(f"{LLVM}/llvm-objdump"):
        print(f"check_isa: {LLVM}/llvm-objdump not found -- NOT checked; no stamp written (the retriever's second stream stays off)")
        return 0
    with tempfile.TemporaryDirectory() as tmp:
        dis = device_disassembly(lib_path, tmp)
    fragile, other, n_packed, n_mfma = fragile_packed(dis)
    if n_mfma < 1000:
        print(f"check_isa: only {n_mfma} MFMA instructions found -- this is not the device code of libtt_hip.so")
        return 2
    if fragile or other:
        bad = (fragile + other)[0]
        print(f"check_isa: FAILED -- {len(fragile)} packed-f32 (+ {len(other)} other packed) instructions read src1's high dword for the "
              f"low result, e.g.\n    {bad.strip()}\n(profiles/r03_pk_mfma_hazard.log: wrong lanes 48-63 beside another stream's MFMAs)")
        return 1
    with open(stamp, "w") as fh:
        fh.write(sha256_of(lib_path) + "\n")
    print(f"check_isa: ok -- {n_packed} packed-f32 instructions, none with op_sel set for src1; stamped {os.path.basename(stamp)}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
