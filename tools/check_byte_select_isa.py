"""Build-time ISA sentinel for the hipcc 7.2 byte-select fault (profiles/r03_augment_stage.txt note (h)).

What went wrong in round 3: with the per-image op records kept loop-invariant (scalar registers), one instantiation of the fused
augmentation kernels counted CutOut's fill value for the wrong pixels - the per-byte selects `inside ? value : b[i]` whose CONSTANT arm
sat in a scalar register were generated wrongly; moving the constant into a vector register (vgpr_byte(), csrc/augment.hip) made the
same build bit-exact.  The shipped kernels re-read the records (nothing is hoisted) AND take every such constant through vgpr_byte().

This script disassembles the device code of chambers_amd/csrc/augment.o and fails if, in any fused_* kernel, a select / byte-merge
instruction (v_cndmask_b32 in any encoding, v_bfi_b32, v_perm_b32) names a bare SGPR as one of its DATA operands - the form the
fault needs.  The shipped object has none (nor has the object built with -DCHB_NO_VGPR_BYTE from the un-hoisted source: there the
constants arrive by vector loads); a later change that hoists the records, or a new select site fed from a scalar, shows up here
on the CPU tier instead of as a wrong histogram on one shape.

    python tools/check_byte_select_isa.py [path/to/augment.o]      exit code 0 = clean; prints a line per kernel checked group
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELECTS = ("v_cndmask_b32", "v_bfi_b32", "v_perm_b32")


def device_disassembly(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, dev = os.path.join(td, "fat.bin"), os.path.join(td, "dev.o")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--input=" + fat, "--output=" + dev, "--unbundle"])
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", dev]).decode()


def check(text):
    """-> (kernels checked, select instructions seen, [(kernel, instruction)] with a bare SGPR data operand)."""
    kernels, n_sel, bad, cur = 0, 0, [], None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = m.group(1) if "fused_" in m.group(1) else None
            kernels += cur is not None
            continue
        if cur is None:
            continue
        ins = line.split("//")[0].strip()
        if not ins.startswith(SELECTS):
            continue
        if re.search(r"//\s*[0-9A-Fa-f]+:\s*00000000\s*$", line):     # alignment padding behind s_endpgm decodes as `v_cndmask_b32_e32 v0, s0, v0, vcc`
            continue
        n_sel += 1
        ops = [o.strip().split(" ")[0] for o in ins.split(None, 1)[1].split(",")]
        # data operands only: not cndmask's condition (vcc / an SGPR pair), not bfi's bit mask (its first source: a constant like
        # 0x00ff00ff lives in a scalar register by right), not perm's byte selector (its last source)
        data = ops[2:4] if ins.startswith("v_bfi_b32") else ops[1:3]
        if any(re.fullmatch(r"s\d+", o) for o in data):
            bad.append((cur, ins))
    return kernels, n_sel, bad


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "chambers_amd", "csrc", "augment.o")
    kernels, n_sel, bad = check(device_disassembly(obj))
    print("%d fused kernels, %d select / byte-merge instructions, %d with a scalar data operand" % (kernels, n_sel, len(bad)))
    for k, ins in bad[:20]:
        print("  %s: %s" % (k[:100], ins))
    return 1 if bad or not kernels else 0


if __name__ == "__main__":
    sys.exit(main())
