"""CPU tier: the C-ABI shared library loads without a GPU and exports every symbol include/chambers_hip.h declares
(no compute call is made here); the product path refuses to run without it / without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "chambers_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(chb_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = _declared()
    assert len(names) >= 25
    for must in ("chb_aug_pointwise", "chb_aug_affine", "chb_aug_equalize", "chb_gemm_nt", "chb_gemm_tn", "chb_attention_fwd",
                 "chb_attention_bwd", "chb_layernorm_fwd", "chb_adamw", "chb_normalize_u8"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from chambers_amd import _lib
    lib = ctypes.CDLL(_lib.lib_path())
    for name in _declared():
        assert hasattr(lib, name), "libchambers_hip.so lacks %s" % name
    typed = _lib.load()
    assert typed.chb_version() == 4 and typed.chb_build_arch() == b"gfx950"
    # every declared entry has a ctypes prototype (and vice versa)
    assert set(_declared()) == set(_lib.PROTOTYPES) | set(_lib.INFO_SYMBOLS)


def test_every_entry_cites_the_reference():
    text = open(os.path.join(ROOT, "include", "chambers_hip.h")).read()
    assert text.count(".py:") >= 15          # file:line citations of the reference interface each entry replaces


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_product_path_fails_loudly_without_gpu():
    from chambers_amd import _lib, augmentations as aug
    from chambers_amd.engine import ViTConfig, ViTEngine
    x = torch.zeros((1, 8, 8, 3), dtype=torch.uint8)
    with pytest.raises(_lib.ChambersHipError):
        aug.Invert()(x)
    with pytest.raises(RuntimeError):
        ViTEngine(ViTConfig(16, 128, 1, 2, 256, image_size=(32, 32)), 1)


def test_missing_library_is_an_error(monkeypatch):
    from chambers_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "lib_path", lambda: "/nonexistent/libchambers_hip.so")
    with pytest.raises(_lib.ChambersHipError):
        _lib.load()


def test_product_never_imports_the_oracle():
    bad = []
    for root, _d, files in os.walk(os.path.join(ROOT, "chambers_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(f)
    assert not bad, bad


def test_tile_queue_ticket_register_is_untouched_until_its_wait():
    """the tile queue's returning atomic lands in a VGPR hipcc believes defined at once.  On the BUILT code object no
    instruction may name that register between the atomic and a vmcnt wait (tools/check_ticket_isa.py), and the queue is opt-in."""
    import importlib.util
    import shutil
    obj = os.path.join(ROOT, "chambers_amd", "csrc", "gemm.o")
    if not os.path.exists(obj) or not shutil.which("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("needs the built gemm.o and the ROCm llvm tools")
    spec = importlib.util.spec_from_file_location("check_ticket_isa", os.path.join(ROOT, "tools", "check_ticket_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    found, problems = mod.check(mod.device_disassembly(obj))
    assert len(found) >= 20 and not problems, problems[:5]
    src = open(os.path.join(ROOT, "chambers_amd", "engine.py")).read()
    assert 'set_option("GEMM_TILE_QUEUE"' not in src          # the engine never forces the queue on


def test_fused_augment_kernels_hold_no_select_with_a_scalar_data_operand():
    """the hipcc 7.2 fault behind vgpr_byte() (csrc/augment.hip) needs a per-byte select whose constant arm sits in
    a scalar register; on the BUILT augment.o no select / byte-merge instruction of a fused_* kernel may name one
    (tools/check_byte_select_isa.py) - a hoisted record or an unguarded new select site fails here, on the CPU tier."""
    import importlib.util
    import shutil
    obj = os.path.join(ROOT, "chambers_amd", "csrc", "augment.o")
    if not os.path.exists(obj) or not shutil.which("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("needs the built augment.o and the ROCm llvm tools")
    spec = importlib.util.spec_from_file_location("check_byte_select_isa", os.path.join(ROOT, "tools", "check_byte_select_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    kernels, n_sel, bad = mod.check(mod.device_disassembly(obj))
    assert kernels >= 50 and n_sel >= 1000 and not bad, bad[:5]
    # the checker does see the form it is looking for
    fake = "0000 <fused_x>:\n\tv_cndmask_b32_e32 v1, s5, v2, vcc // 0\n\tv_cndmask_b32_e64 v1, v3, v2, s[4:5] // 0\n"
    assert len(mod.check(fake)[2]) == 1


def test_pipelined_attention_kernels_keep_their_waits_and_m0():
    """The persistent attention kernels (csrc/attention.hip) hand-manage m0 and every vmcnt wait; on the BUILT attention.o:
    no m0 write outside an LDS-DMA triple, no flat / scratch access, no compiler-inserted vmcnt(0) inside a role loop
    (tools/check_pipe_isa.py: each of these showed up during development and cost either correctness or the pipelining)."""
    import importlib.util
    import shutil
    obj = os.path.join(ROOT, "chambers_amd", "csrc", "attention.o")
    if not os.path.exists(obj) or not shutil.which("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("needs the built attention.o and the ROCm llvm tools")
    spec = importlib.util.spec_from_file_location("check_pipe_isa", os.path.join(ROOT, "tools", "check_pipe_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.check(mod.device_disassembly(obj))
    assert len(res) == 4
    for name, (problems, n_dma, _n_waits) in res.items():
        assert not problems, (name, problems[:5])
