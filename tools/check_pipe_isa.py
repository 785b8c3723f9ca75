"""Build-time ISA check of the persistent pipelined attention kernels (csrc/attention.hip: attn_bwd_pipe_kernel, attn_fwd_pipe_kernel).

Their correctness leans on three things the compiler is not told and could undo without a word (all three happened during
development, DESIGN section 4 "Round 4"):
  1. m0 belongs to the inline-asm LDS-DMA pieces (`s_mov_b32 m0, sN` + `s_nop 0` + `global_load_lds_dword[x4]`): every m0 write in
     these kernels must be such a triple, and every LDS-DMA must sit behind one;
  2. no FLAT access (a pointer into LDS that lost its address space becomes `flat_load` + `s_waitcnt vmcnt(0) lgkmcnt(0)`: it
     would drain the producer's pieces at every use);
  3. the only `s_waitcnt vmcnt(...)` instructions are the hand-written ones - a compiler-inserted `vmcnt(0)` in a role loop makes a
     wave wait for its own fire-and-forget stores or pieces one step later.  Expected: vmcnt(0) only in the prologue (before the
     first `s_barrier` pair), behind the head-counter atomics, and in the forward's loader waves; `vmcnt(14)` / `vmcnt(13)`
     (producer) and `vmcnt(4)` (phase-B waves) in the backward.

    python tools/check_pipe_isa.py [path/to/attention.o]        exit code 0 = clean
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_disassembly(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, dev = os.path.join(td, "fat.bin"), os.path.join(td, "dev.o")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--input=" + fat, "--output=" + dev, "--unbundle"])
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", dev]).decode()


def kernels_of(text, needle):
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = m.group(1) if needle in m.group(1) else None
            if cur:
                out[cur] = []
            continue
        ins = line.split("//")[0].strip()
        if cur and ins:
            out[cur].append(ins)
    return out


def check_kernel(name, ins):
    problems = []
    n_dma = 0
    for k, i in enumerate(ins):
        if re.search(r"\bm0\b", i):
            ok = re.fullmatch(r"s_mov_b32 m0, s\d+", i) and k + 2 < len(ins) and ins[k + 1].startswith("s_nop") and ins[k + 2].startswith("global_load_lds_dword")
            if not ok:
                problems.append("m0 touched outside an LDS-DMA triple: %s" % i)
        if i.startswith("global_load_lds") or ("buffer_load" in i and " lds" in i):
            n_dma += 1
            if not (k >= 2 and re.fullmatch(r"s_mov_b32 m0, s\d+", ins[k - 2]) and ins[k - 1].startswith("s_nop")):
                problems.append("LDS-DMA without its m0 triple: %s" % i)
        if i.startswith(("flat_", "scratch_")):        # scratch = register spills: their loads are VMEM operations with waits of their own
            problems.append("flat / scratch access: %s" % i)
    waits = [(k, i) for k, i in enumerate(ins) if i.startswith("s_waitcnt") and "vmcnt" in i]
    barriers = [k for k, i in enumerate(ins) if i.startswith("s_barrier")]
    atomics = [k for k, i in enumerate(ins) if i.startswith("global_atomic_add")]
    bwd = "attn_bwd_pipe" in name
    allowed_counted = {"vmcnt(14)", "vmcnt(13)", "vmcnt(4)"} if bwd else set()
    n_zero_in_loops = 0
    for k, i in waits:
        cnt = re.search(r"vmcnt\(\d+\)", i).group(0)
        if cnt != "vmcnt(0)":
            if cnt not in allowed_counted:
                problems.append("unexpected counted wait %s" % i)
            continue
        in_prologue = len(barriers) >= 2 and k < barriers[1] + 8 if bwd else (len(barriers) >= 1 and k < barriers[1] + 8 if len(barriers) > 1 else True)
        behind_atomic = any(0 < k - a <= 6 for a in atomics)
        if in_prologue or behind_atomic:
            continue
        n_zero_in_loops += 1
    # the forward's three loader waves share ONE loop: exactly one hand-written vmcnt(0) there; the backward has none in its loops
    if n_zero_in_loops != (0 if bwd else 1):
        problems.append("%d s_waitcnt vmcnt(0) inside the role loops (expected %d)" % (n_zero_in_loops, 0 if bwd else 1))
    if n_dma < (30 if bwd else 3):
        problems.append("only %d LDS-DMA instructions found" % n_dma)
    return problems, n_dma, len(waits)


def check(text):
    res = {}
    for needle in ("attn_bwd_pipe_kernel", "attn_fwd_pipe_kernel"):
        for name, ins in kernels_of(text, needle).items():
            res[name] = check_kernel(name, ins)
    return res


def main():
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "chambers_amd", "csrc", "attention.o")
    res = check(device_disassembly(obj))
    bad = 0
    for name, (problems, n_dma, n_waits) in sorted(res.items()):
        print("%s: %d LDS-DMA pieces, %d vmcnt waits, %d problems" % (name[:70], n_dma, n_waits, len(problems)))
        for p in problems[:10]:
            print("   " + p)
        bad += len(problems)
    return 1 if bad or len(res) != 4 else 0


if __name__ == "__main__":
    sys.exit(main())
