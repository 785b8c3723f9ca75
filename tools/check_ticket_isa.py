"""Build-time ISA check of the persistent NT GEMMs' tile-queue ticket (csrc/gemm.hip).

The ticket is drawn by an inline-asm returning atomic (`global_atomic_add vT, ..., sc0`) whose destination VGPR is written by the
hardware LATER, while hipcc believes it defined right behind the asm statement.  The kernels consume it only behind the counted
`s_waitcnt vmcnt` of the following K-step (through a second asm statement that marks the point), but nothing in the language stops
the register allocator from copying, spilling or re-using vT in between - this script disassembles the device code of
chambers_amd/csrc/gemm.o and fails if, in any kernel, an instruction names vT textually behind the atomic without an `s_waitcnt
vmcnt` in between (the loop body is laid out wait -> consume -> ... -> draw; a copy or spill right behind the draw would show).

    python tools/check_ticket_isa.py [path/to/gemm.o]      exit code 0 = clean; prints one line per atomic found
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VMEM = re.compile(r"^\s*(global_|buffer_|flat_|scratch_)")


def device_disassembly(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, dev = os.path.join(td, "fat.bin"), os.path.join(td, "dev.o")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--input=" + fat, "--output=" + dev, "--unbundle"])
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", dev]).decode()


def uses_reg(line, reg):
    """True if the instruction text names VGPR `reg` (an int) alone or inside a range v[a:b]."""
    body = line.split("//")[0]
    if ("v%d" % reg) not in body and "v[" not in body:
        return False
    for m in re.finditer(r"\bv(\d+)\b", body):
        if int(m.group(1)) == reg:
            return True
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", body):
        if int(m.group(1)) <= reg <= int(m.group(2)):
            return True
    return False


def check(text):
    """Per kernel holding the returning atomic into vT: every OTHER instruction that names vT (besides the `v_mov vT, 1` that feeds the
    atomic) must have an `s_waitcnt ... vmcnt(` between itself and the atomic on the backward scan - i.e. sit in the part of the loop
    body that follows a vmcnt wait and precedes the atomic (the kernels consume the ticket behind the step-end wait and draw the next
    one later in the same body).  Anything that names vT textually behind the atomic with no wait in between is reported."""
    problems, found = [], []
    kernels, cur = [], None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = (m.group(1), [])
            kernels.append(cur)
        elif cur is not None and line.strip():
            cur[1].append(line.split("//")[0].strip())
    for name, body in kernels:
        atoms = [k for k, ins in enumerate(body) if ins.startswith("global_atomic_add ") and "sc0" in ins]
        for a in atoms:
            reg = int(re.search(r"global_atomic_add\s+v(\d+),", body[a]).group(1))
            touching = [k for k, ins in enumerate(body) if k != a and uses_reg(ins, reg)]
            bad = []
            for k in touching:
                if k < a and re.match(r"v_mov_b32(_e32)?\s+v%d,\s*1$" % reg, body[k]) and not any(uses_reg(body[j], reg) for j in range(k + 1, a)):
                    continue                                  # the constant operand of the atomic
                j, waited = k - 1, False
                while j >= 0:
                    if j == a:
                        break
                    if re.search(r"s_waitcnt.*vmcnt\(", body[j]):
                        waited = True
                        break
                    j -= 1
                if j < 0 and not waited:
                    bad.append(body[k] + "   [no vmcnt wait above it in the kernel]")
                elif not waited:
                    bad.append(body[k])
            found.append((name, reg, len(touching), not bad))
            for b in bad:
                problems.append("%s: v%d named behind the atomic with no vmcnt wait in between: %s" % (name, reg, b))
    return found, problems


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "chambers_amd", "csrc", "gemm.o")
    found, problems = check(device_disassembly(obj))
    for kernel, reg, n, ok in found:
        print("ticket atomic -> v%d, %d other instructions name it, %s   %s" % (reg, n, "all behind a vmcnt wait" if ok else "SOME NOT", kernel[:100]))
    if not found:
        print("no returning global_atomic_add found in", obj)
        return 1
    for p in problems:
        print("PROBLEM:", p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
