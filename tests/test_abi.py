"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/v2a_cfm.h
declares; argument validation answers without touching a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from v2a_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "v2a_cfm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(v2a_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(lib):
    syms = _declared_symbols()
    assert len(syms) >= 13, syms
    L = lib.lib()
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/v2a_cfm.h but not exported"
    assert sorted(lib.EXPORTS) == syms


def test_abi_version_and_error_string(lib):
    L = lib.lib()
    assert L.v2a_abi_version() == lib.ABI_VERSION == 8
    g = lib.GemmArgs()
    g.nseg = 5
    assert L.v2a_gemm(ctypes.byref(g), None) == -1                 # V2A_ERR_ARG, before any HIP call
    assert b"nseg" in L.v2a_last_error()
    assert L.v2a_rmsnorm(None, 0, None, 0, 0, 1, 64, None, None, 0, 0, 0, None) == -1
    assert b"null" in L.v2a_last_error()
    assert L.v2a_dwconv_silu_residual(1, 2, 3, 4, 1, 8, 64, 7, None, None) == -1
    assert b"kernel_size" in L.v2a_last_error()


def test_struct_layout_matches_header(lib):
    """ctypes mirrors of v2a_gemm_args / v2a_attn_args: field order and sizes follow the header."""
    src = open(os.path.join(ROOT, "include", "v2a_cfm.h")).read()
    body = re.search(r"typedef struct v2a_gemm_args \{(.*?)\} v2a_gemm_args;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[\d+\])?\s*$", part.strip())[0])
    assert names == [f[0] for f in lib.GemmArgs._fields_]
    assert ctypes.sizeof(lib.GemmArgs) % 8 == 0
    # the library reports the struct size it was compiled with: the ctypes mirror must agree byte for byte
    assert lib.lib().v2a_gemm_args_size() == ctypes.sizeof(lib.GemmArgs)


def _struct_fields(src, name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        for part in decl.strip().split(","):
            part = part.strip()
            if part:
                names.append(re.findall(r"\*?\s*([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[\d+\])?\s*$", part)[0])
    return names


def test_integration_md_binding_matches_header(lib):
    """The ctypes stub a maintainer would copy out of INTEGRATION.md section 1 mirrors v2a_gemm_args field for field
    (a short struct makes the kernel read garbage pointers), and the tuning struct mirrors v2a_tuning."""
    src = open(os.path.join(ROOT, "include", "v2a_cfm.h")).read()
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"class GemmArgs\(ctypes\.Structure\):.*?_fields_ = \[(.*?)\]\n\n", md, re.S).group(1)
    md_fields = re.findall(r'\("([a-z_0-9A-Z]+)",', block)
    assert md_fields == _struct_fields(src, "v2a_gemm_args")
    # the snippet is executable as printed: build the class from it and compare sizes with the library
    ns = {}
    code = re.search(r"(class GemmArgs\(ctypes\.Structure\):.*?\]\n)\n", md, re.S).group(1)
    exec("import ctypes\n" + code, ns)
    assert ctypes.sizeof(ns["GemmArgs"]) == lib.lib().v2a_gemm_args_size()
    assert [f[0] for f in lib.Tuning._fields_] == _struct_fields(src, "v2a_tuning")


def test_set_tuning_validates_and_resets(lib):
    L = lib.lib()
    t = lib.Tuning(9, 0, 0, 0)
    assert L.v2a_set_tuning(ctypes.byref(t)) == -1 and b"gemm_force_tile" in L.v2a_last_error()
    t = lib.Tuning(4, 0, 0, 0)                                     # tile configuration 4 is not defined by the header
    assert L.v2a_set_tuning(ctypes.byref(t)) == -1 and b"gemm_force_tile" in L.v2a_last_error()
    t = lib.Tuning(3, 0, 1, 0, 5)                                  # valid tile, invalid dwconv rows: nothing may be applied
    assert L.v2a_set_tuning(ctypes.byref(t)) == -1 and b"dwconv_rows_per_wave" in L.v2a_last_error()
    lib.set_tuning(force_tile=3, eight_phase=1)
    lib.set_tuning()
    assert L.v2a_set_tuning(None) == 0


def test_roll_head_struct_and_new_entry_points_validate_on_cpu(lib):
    """N1 / N2 entry points: the ctypes mirror of v2a_roll_head_args follows the header; bad arguments are refused before
    any HIP call (no GPU here)."""
    src = open(os.path.join(ROOT, "include", "v2a_cfm.h")).read()
    body = re.search(r"typedef struct v2a_roll_head_args \{(.*?)\} v2a_roll_head_args;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        for part in decl.strip().split(","):
            part = part.strip()
            if part:
                names.append(re.findall(r"\*?\s*([A-Za-z_][A-Za-z0-9_]*)\s*$", part)[0])
    assert names == [f[0] for f in lib.RollHeadArgs._fields_]
    L = lib.lib()
    assert L.v2a_im2col(16, 1, 4, 4, 8, 3, 3, 1, 1, 3, 4, 16, 128, 0, 0, 0, None) == -1
    assert b"Ho/Wo" in L.v2a_last_error()
    assert L.v2a_pool2d(16, 32, None, 1, 4, 4, 6, 2, 2, 0, 1, 2, 2, 0, 0, None) == -1
    assert b"geometry" in L.v2a_last_error()
    assert L.v2a_lstm_layer(16, 16, 16, None, None, 10, 256, 16, None) == -1
    assert b"hidden size" in L.v2a_last_error()
    assert L.v2a_elu_pad(16, 32, 4, 6, 0, 0, 1, None) == -1
    assert b"v2a_elu_pad" in L.v2a_last_error()
    a = lib.RollHeadArgs()
    assert L.v2a_roll_head(ctypes.byref(a), None) == -1
    assert b"null" in L.v2a_last_error()
    g = lib.GemmArgs()
    g.nseg, g.M, g.N, g.compute_dtype, g.a_dtype = 1, 64, 64, 0, 0
    g.a[0], g.lda[0], g.ka[0], g.w, g.ldw, g.out, g.ldo = 4096, 64, 64, 8192, 64, 12288, 64
    g.a_row_offset, g.a_ktile_offset = 16, 16
    assert L.v2a_gemm(ctypes.byref(g), None) == -1
    assert b"offset tables" in L.v2a_last_error()


def test_missing_library_fails_loudly(lib, monkeypatch):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libv2a_cfm.so")
    with pytest.raises(lib.V2AError, match="no CPU fallback"):
        lib.lib()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "video-to-audio-and-piano-rp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("the oracle", "").replace("CPU oracle", "").replace("like the oracle", ""), f
