"""CPU: the C-ABI library builds/loads and exports exactly what include/openclip_hip.h declares."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "openclip_hip.h")


DEBUG_HEADER = os.path.join(ROOT, "include", "openclip_hip_debug.h")


def _declared(header=HEADER):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ocn_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    from open_clip_amd import build
    return build.build()


def test_header_declares_the_python_table(lib_path):
    from open_clip_amd import _lib
    declared = _declared()
    table = sorted(list(_lib.SIGNATURES) + list(_lib._SPECIAL))
    assert declared == table, (set(declared) ^ set(table))
    # the developer knobs live in their own header and table: none of them is part of the boundary
    assert _declared(DEBUG_HEADER) == sorted(_lib.DEBUG_SIGNATURES), set(_declared(DEBUG_HEADER)) ^ set(_lib.DEBUG_SIGNATURES)
    assert not set(declared) & set(_lib.DEBUG_SIGNATURES)


def test_library_exports_every_declared_symbol(lib_path):
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (ocn_[a-z0-9_]+)", out))
    missing = (set(_declared()) | set(_declared(DEBUG_HEADER))) - exported
    assert not missing, missing


def test_library_loads_and_reports_errors(lib_path):
    from open_clip_amd import _lib
    lib = _lib.load()
    hdr = int(re.search(r"#define OCN_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert lib.ocn_version() == hdr == _lib.ABI_VERSION  # header == library == ctypes table: load() refuses any other library
    # argument validation happens on the host before any launch: safe without a GPU
    with pytest.raises(RuntimeError, match="K=48 must be a multiple of 32"):
        _lib.call("ocn_gemm_nt", 0, 16, 48, 16, 48, 16, 64, 8, 64, 48, 0, 0, 0, 1.0, 0)
    with pytest.raises(RuntimeError, match="null operand"):
        _lib.call("ocn_layernorm_fwd", 0, 0, 0, 0, 0, 0, 0, 0, 4, 64, 1e-5, 0)


def test_library_contains_gfx950_code_object(lib_path):
    data = open(lib_path, "rb").read()
    assert b"gfx950" in data
