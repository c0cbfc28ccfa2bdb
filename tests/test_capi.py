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
    assert L.ssrhip_version() == 107 == _lib.ABI_VERSION
    assert isinstance(L.ssrhip_last_error(), bytes)


def test_pair_launches_never_share_a_granule_buffer_with_their_neighbour(built):
    """The paired GEMV launches of the 2-row decode step (csrc/gemv.hip gemv_pair_kernel) publish their outputs as tagged granules; a
    launch resets the buffer the NEXT pair launch will use, so two launches that follow each other — cyclically: the last of a step is
    followed by the first of the next step, graph replay after graph replay — must never use the same one of the three buffers. Host
    logic, no GPU: the rule engine.hip applies (ssrhip_pair_buffer) for every count of pairs per step."""
    L = _lib.lib()
    for n in range(2, 201):
        bufs = [L.ssrhip_pair_buffer(i, n) for i in range(n)]
        assert all(b in (0, 1, 2) for b in bufs), (n, bufs)
        assert all(bufs[i] != bufs[(i + 1) % n] for i in range(n)), (n, bufs)
    assert L.ssrhip_pair_buffer(0, 1) == -1 and L.ssrhip_pair_buffer(3, 3) == -1 and L.ssrhip_pair_buffer(-1, 5) == -1


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
