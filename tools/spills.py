"""Register use of the built fused kernels: max VGPRs and every kernel with spilled registers (the three-workgroups-per-CU builds sit
on 168 VGPRs: a spill there is reloaded behind a wait for the whole LDS-DMA ring).   python3 tools/spills.py [objects...]"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")
ROOT = Path(__file__).resolve().parent.parent


def kernels(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, dev = Path(td) / "fat.bin", Path(td) / "dev.o"
        subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(obj)], check=True)
        subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={fat}", f"--output={dev}"], check=True, capture_output=True)
        txt = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(dev)], capture_output=True, text=True).stdout
    return list(zip(re.findall(r"\.name:\s+(\S+)", txt), map(int, re.findall(r"\.vgpr_count:\s+(\d+)", txt)),
                    map(int, re.findall(r"\.vgpr_spill_count:\s+(\d+)", txt)), map(int, re.findall(r"\.sgpr_spill_count:\s+(\d+)", txt))))


def main(argv):
    objs = [Path(a) for a in argv] or sorted((ROOT / "kiez_amd" / "csrc").glob("kz_knn_h*.o"))
    bad = 0
    for o in objs:
        ks = kernels(o)
        sp = [k for k in ks if k[2] or k[3]]
        bad += len(sp)
        print(f"{o.name}: {len(ks)} kernels, max {max(k[1] for k in ks)} VGPRs, {len(sp)} with spills")
        for k in sp:
            print("   ", k)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
