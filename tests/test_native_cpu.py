"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/ctvae_hip.h declares
(no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ctvae_amd.build import build
    build()
    from ctvae_amd import native
    return native


def test_exports_match_header(lib):
    hdr = open(os.path.join(ROOT, "include", "ctvae_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(ctvae_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == lib.EXPORTS, (set(declared) ^ set(lib.EXPORTS))
    handle = lib.load()
    for name in declared:
        assert hasattr(handle, name), name
    assert handle.ctvae_arch() == b"gfx950"
    assert handle.ctvae_workspace_bytes() >= 64 << 20
    assert b"bad argument" in handle.ctvae_error_string(-22)


def test_ctypes_signatures_match_header_prototypes(lib):
    """Every prototype of include/ctvae_hip.h against the ctypes argtypes of native.py: same number of parameters, and
    pointer / int / float / size_t / long in the same positions (a shifted argument would otherwise only show up as a
    wrong result on the GPU)."""
    import ctypes
    hdr = open(os.path.join(ROOT, "include", "ctvae_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = dict((m.group(1), m.group(2)) for m in re.finditer(r"\b(ctvae_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", hdr, re.S))
    def kind(param):
        param = " ".join(param.split())
        if "*" in param:
            return "ptr"
        t = param.rsplit(" ", 1)[0] if " " in param else param
        return {"int": "int", "float": "float", "size_t": "size", "long": "long", "double": "double"}[t.replace("const ", "")]
    ck = {ctypes.c_void_p: "ptr", ctypes.c_int: "int", ctypes.c_float: "float", ctypes.c_size_t: "size", ctypes.c_long: "long",
          ctypes.c_char_p: "ptr"}
    checked = 0
    for name, argtypes in lib.SIGNATURES.items():
        params = [p for p in protos[name].split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(argtypes), f"{name}: header has {len(params)} parameters, native.py {len(argtypes)}"
        for i, (p_, a) in enumerate(zip(params, argtypes)):
            hk, nk = kind(p_), ck[a]
            if {hk, nk} == {"size", "long"}:          # c_size_t and c_long are the same ctypes class on LP64
                continue
            assert hk == nk, f"{name} parameter {i} ({' '.join(p_.split())}): header {hk}, native.py {nk}"
        checked += 1
    assert checked == len(lib.SIGNATURES) >= 50


def test_product_path_has_no_cpu_fallback():
    import torch
    from ctvae_amd import kernels as K
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        K.to_nhwc(torch.zeros(1, 3, 4, 4))
    with pytest.raises(RuntimeError):
        K.ConvAct.apply(torch.zeros(1, 4, 4, 32), torch.zeros(32, 32, 1, 1), None, None, K.ConvSpec(K.CONV, 32, 32, 1))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "ct-vae_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "/root/reference" not in src, f
