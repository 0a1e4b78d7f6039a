"""M0 discipline of the hand-issued LDS-DMA copies (kz_knn_device.h: kz_glds16_s / kz_glds4_s set M0 in inline asm).

hipcc warns that a clobber of the reserved register m0 "may lead to undefined behaviour": it does not save and restore M0 around
the asm statement.  That is harmless exactly when no compiler-managed M0 value is ever live ACROSS such a statement -- i.e. when
every instruction that reads M0 takes it from the `s_mov_b32 m0, ...` of ITS OWN asm statement.  This script checks that on the
device code of the built objects: for every kernel, every M0 reader (LDS-DMA `global_load_lds_*` / `buffer_load ... lds`, `ds_gws_*`,
`s_movrel*` / `v_movrel*`, `s_sendmsg*`, LDS-direct reads) must be the next instruction (`s_nop` apart) after a write of M0 -- the
shape of kz_glds16_s / kz_glds4_s.  (Round 4's rule -- "some M0 write earlier in the basic block" -- would have passed a compiler-set
M0, then an inline-asm clobber, then a compiler-issued reader of the OLD value: a reader that is not glued to its own write is a
violation now, whoever issued it.)
With that shown, the one diagnostic is switched off for the fp16 kernel units (Makefile: -Wno-inline-asm), so that any NEW warning
is visible.     python3 tools/check_m0.py [objects...]     (default: the fp16 kernel objects of kiez_amd/csrc)"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")
ROOT = Path(__file__).resolve().parent.parent
READS_M0 = re.compile(r"^\s*(global_load_lds_|buffer_load_.*\blds\b|ds_gws_|s_movrel|v_movrel|s_sendmsg|ds_read_.*\bgds\b|ds_write_.*\bgds\b|v_interp_|lds_direct)")
WRITES_M0 = re.compile(r"^\s*s_(mov_b32|movk_i32|add_[ui]32|lshl_b32|or_b32|and_b32)\s+m0\b")
LABEL = re.compile(r"^[0-9a-f]+ <[^>]+>:")
BRANCH = re.compile(r"^\s*(s_cbranch|s_branch|s_endpgm|s_setpc|s_swappc)")


def device_disassembly(obj: Path) -> str:
    with tempfile.TemporaryDirectory() as td:
        fat, dev = Path(td) / "fat.bin", Path(td) / "dev.o"
        subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(obj)], check=True)
        subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={fat}", f"--output={dev}"], check=True, capture_output=True)
        return subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--no-show-raw-insn", str(dev)], check=True, capture_output=True, text=True).stdout


NOP = re.compile(r"^\s*s_nop\b")


def check(text: str):
    """-> (readers, violations): every M0 reader must directly follow (s_nop apart) an M0 write of its own basic block."""
    readers, bad = 0, []
    fresh_m0, kernel = False, "?"     # fresh_m0: the last instruction that was not an s_nop wrote M0
    for line in text.splitlines():
        if LABEL.match(line):
            name = line.split("<", 1)[1].rsplit(">", 1)[0]
            if not name.startswith("L") and "BB" not in name:
                kernel = name
            fresh_m0 = False
            continue
        body = line.split("//")[0]
        if not body.strip():
            continue
        if WRITES_M0.match(body):
            fresh_m0 = True
        elif NOP.match(body):
            pass
        else:
            if READS_M0.match(body):
                readers += 1
                if not fresh_m0:
                    bad.append((kernel, body.strip()))
            fresh_m0 = False
    return readers, bad


def main(argv):
    objs = [Path(a) for a in argv] or sorted((ROOT / "kiez_amd" / "csrc").glob("kz_knn_h*.o"))
    total, failures = 0, []
    for o in objs:
        readers, bad = check(device_disassembly(o))
        total += readers
        failures += [(o.name,) + b for b in bad]
        print(f"{o.name}: {readers} M0 readers, {len(bad)} not directly behind their own M0 write")
    for f in failures[:20]:
        print("  VIOLATION", f)
    print("M0 readers checked:", total)
    return 1 if failures or total == 0 else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
