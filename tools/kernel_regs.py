"""Register / scratch / LDS use of the kernels in libtt_hip.so whose (mangled) name matches a pattern:
python tools/kernel_regs.py gemm_kernel_v3 [lib].  Reads the code objects' metadata notes (no GPU)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("TT_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tensor-truth_amd", "libtt_hip.so")
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob)]
        for i, s in enumerate(starts):
            chunk, co = os.path.join(tmp, f"b{i}.bin"), os.path.join(tmp, f"b{i}.co")
            open(chunk, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={chunk}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                if not name or pat not in name.group(1):
                    continue
                get = lambda k: (re.search(rf"\.{k}:\s+(\d+)", blk) or [None, "?"])[1]
                demangled = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
                print(f"bundle {i}: vgpr {get('vgpr_count')} sgpr {get('sgpr_count')} spill {get('vgpr_spill_count')} scratch "
                      f"{get('private_segment_fixed_size')} lds {get('group_segment_fixed_size')}  {demangled[:150]}")


if __name__ == "__main__":
    main()
