"""Build recipe for libchambers_hip.so (hipcc, gfx950 only, in-tree).

`python -m chambers_amd._build` or `__graft_entry__.build()` run it.  The shared object is
written next to the sources (chambers_amd/csrc/libchambers_hip.so): it is git-ignored but
travels with the tree to the GPU box.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libchambers_hip.so")
STAMP_PATH = os.path.join(CSRC, ".build_stamp")

# per-file extra flags: augment.hip must round after every float op like the TF CPU kernels
SOURCES = {
    "augment.hip": ["-ffp-contract=off"],
    "imageio.hip": ["-ffp-contract=off"],
    "gemm.hip": [],
    "layernorm.hip": [],
    "attention.hip": [],
    "attention_general.hip": [],
    "elementwise.hip": ["-ffp-contract=off"],
    "metric.hip": [],
    "vit_block.hip": [],
}
COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
          "-munsafe-fp-atomics"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _digest():
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for name in sorted(os.listdir(root)):
            if name.endswith((".hip", ".hpp", ".h")):
                with open(os.path.join(root, name), "rb") as f:
                    h.update(name.encode())
                    h.update(f.read())
    h.update(" ".join(COMMON).encode())
    h.update(repr(sorted(SOURCES.items())).encode())
    return h.hexdigest()


def is_current():
    if not (os.path.exists(LIB_PATH) and os.path.exists(STAMP_PATH)):
        return False
    with open(STAMP_PATH) as f:
        return f.read().strip() == _digest()


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link the C-ABI shared library."""
    if not force and is_current():
        return LIB_PATH
    hipcc = _hipcc()
    objs = []
    procs = []
    for src, extra in SOURCES.items():
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc] + COMMON + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[chambers_amd build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError("hipcc failed on %s" % src)
        if verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print("[chambers_amd build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP_PATH, "w") as f:
        f.write(_digest())
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
