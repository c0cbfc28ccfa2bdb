"""CPU: libssrhip.so builds for gfx950, loads, and exports every symbol include/ssrhip.h declares.
No compute call is made (there is no GPU here)."""
import os
import re

import pytest

import ssr_speech_amd  # noqa: F401
from ssr_speech_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.LIB_PATH


def test_header_symbols_are_exported(built):
    hdr = open(os.path.join(ROOT, "include", "ssrhip.h")).read()
    declared = set(re.findall(r"\b(ssrhip_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_version_and_error_string(built):
    L = _lib.lib()
    assert L.ssrhip_version() == 104
    assert isinstance(L.ssrhip_last_error(), bytes)


def test_argument_validation_needs_no_gpu(built):
    """Contract errors are reported through the return code + ssrhip_last_error, never by crashing."""
    L = _lib.lib()
    a = _lib.GemvArgs()
    assert L.ssrhip_gemv(a, None) != 0
    assert b"null" in L.ssrhip_last_error()
    a.W, a.y, a.x, a.B, a.N, a.K, a.groups = 8, 8, 8, 3, 16, 64, 1
    assert L.ssrhip_gemv(a, None) != 0
    assert b"B=3" in L.ssrhip_last_error()


def test_struct_sizes_match_header(built):
    import ctypes as C
    L = _lib.lib()
    for i, st in enumerate(_lib.ABI_STRUCTS):
        assert L.ssrhip_sizeof(i) == C.sizeof(st), st.__name__
    assert C.sizeof(_lib.SamplerState) == 60
