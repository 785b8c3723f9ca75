"""CPU tier: the hand-written ctypes prototypes (chambers_amd/_lib.py) against the declarations of include/chambers_hip.h - argument
COUNT and TYPES, not just names (an int / int64_t drift would otherwise show up as a corrupted stride on the GPU), and the byte
layout of the two structs the ABI passes by pointer against what a C compiler makes of the header."""
import ctypes
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "chambers_hip.h")


def _strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _ctype_of(param):
    """C parameter declaration -> the ctypes type _lib.PROTOTYPES must hold for it."""
    p = " ".join(param.split())
    if "*" in p:
        return ctypes.c_char_p if re.match(r"const char\s*\*", p) else ctypes.c_void_p
    base = re.sub(r"\b[a-zA-Z_][a-zA-Z0-9_]*$", "", p).strip() if " " in p else p       # drop the parameter name
    table = {"int": ctypes.c_int, "int32_t": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "uint32_t": ctypes.c_uint32}
    assert base in table, "unmapped C type %r in %r" % (base, param)
    return table[base]


def _declarations():
    text = _strip_comments(open(HEADER).read())
    text = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", text, flags=re.S)
    out = {}
    for ret, name, params in re.findall(r"\b(int|int64_t|const char\*)\s+(chb_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        params = params.strip()
        plist = [] if params in ("", "void") else [q.strip() for q in params.split(",")]
        out[name] = (ret, [_ctype_of(q) for q in plist])
    return out


def test_prototypes_match_the_header_types():
    from chambers_amd import _lib
    decl = _declarations()
    assert set(decl) == set(_lib.PROTOTYPES) | set(_lib.INFO_SYMBOLS)
    bad = []
    for name, argtypes in _lib.PROTOTYPES.items():
        want = decl[name][1]
        same = len(want) == len(argtypes) and all(ctypes.sizeof(a) == ctypes.sizeof(b) and
                                                  (a in (ctypes.c_void_p, ctypes.c_char_p)) == (b in (ctypes.c_void_p, ctypes.c_char_p)) and
                                                  (a is ctypes.c_float) == (b is ctypes.c_float) for a, b in zip(want, argtypes))
        if not same:
            bad.append((name, [t.__name__ for t in want], [t.__name__ for t in argtypes]))
    assert not bad, bad
    assert decl["chb_aug_fused_workspace_ints"][0] == "int64_t" and decl["chb_build_arch"][0] == "const char*"


def _struct_fields(name):
    text = _strip_comments(open(HEADER).read())
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
    fields = []
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        if not stmt:
            continue
        m = re.match(r"((?:const )?[a-z0-9_]+)\s*(.*)", stmt)
        base, rest = m.group(1), m.group(2)
        for item in rest.split(","):
            item = item.strip()
            fields.append((item.lstrip("* "), base + ("*" if item.startswith("*") or base.endswith("*") else "")))
    return fields


@pytest.mark.parametrize("cname,pyname", [("chb_vit_block", "VitBlock"), ("chb_profile_record", "ProfileRecord"), ("chb_tn_fold_item", "TnFoldItem")])
def test_struct_mirrors_match_the_header(cname, pyname, tmp_path):
    from chambers_amd import _lib
    py = getattr(_lib, pyname)
    fields = _struct_fields(cname)
    assert [n for n, _t in fields] == [n for n, _t in py._fields_]
    for (n, ct), (_n, pt) in zip(fields, py._fields_):
        if ct.endswith("*"):
            assert pt is ctypes.c_void_p, n
        else:
            want = {"int32_t": 4, "uint32_t": 4, "float": 4, "int64_t": 8}[ct]
            assert ctypes.sizeof(pt) == want and (pt is ctypes.c_float) == (ct == "float"), n
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        pytest.skip("no C compiler")
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(void) {\n  printf("%%zu\\n", sizeof(%s));\n%s  return 0;\n}\n'
                   % (HEADER, cname, "".join('  printf("%%zu\\n", offsetof(%s, %s));\n' % (cname, n) for n, _t in fields)))
    exe = tmp_path / "layout"
    subprocess.check_call([cc, "-o", str(exe), str(src)])
    nums = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert nums[0] == ctypes.sizeof(py)
    assert nums[1:] == [getattr(py, n).offset for n, _t in py._fields_]
