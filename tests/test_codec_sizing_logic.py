"""CPU: the bookkeeping of `WMEncodecModel._sized` (round 6, DESIGN.md Part I.4) without a GPU — which passes run dry, which shapes count as
covered, and what happens when the driver-allocation counter moves during a real pass. The GPU side of the same contract is
tests/test_gpu_codec.py::test_sized_codec_calls_never_reach_the_driver_for_memory_and_change_no_result."""
import types

import pytest
import torch

import ssr_speech_amd  # noqa: F401
from ssr_speech_amd.codec import wmencodec as WM


class _Stream:
    def __init__(self, handle):
        self.cuda_stream = handle


@pytest.fixture
def model(monkeypatch):
    m = object.__new__(WM.WMEncodecModel)
    m.presize, m.device = True, torch.device("cpu")
    m._envelopes, m._keep, m._small_reserved = {}, {}, set()
    m.sizing_passes = m.mallocs_in_flight = m.passes_repeated = 0
    m.lib = types.SimpleNamespace(name="real library")
    state = types.SimpleNamespace(stream=_Stream(11), mallocs=100, syncs=0, script=[])
    monkeypatch.setattr(WM.torch.cuda, "current_stream", lambda dev=None: state.stream)
    monkeypatch.setattr(WM.torch.cuda, "synchronize", lambda dev=None: setattr(state, "syncs", state.syncs + 1))
    monkeypatch.setattr(WM, "_device_mallocs", lambda dev: state.mallocs)
    return m, state


def _runner(m, state, log, grow_on_real=()):
    """a pass that records whether it ran dry (library swapped for the no-launch stand-in) and may move the allocation counter"""
    n_real = [0]

    def run():
        dry = isinstance(m.lib, WM._NoLaunch)
        log.append("dry" if dry else "real")
        if not dry:
            n_real[0] += 1
            if n_real[0] in grow_on_real:
                state.mallocs += 2
        return (torch.zeros(3), [torch.ones(2, 2), None])
    return run


def test_a_new_shape_runs_dry_first_and_covered_shapes_do_not(model):
    m, state = model
    log = []
    out = m._sized("decode", 4, 100, _runner(m, state, log))
    assert log == ["dry", "real"] and m.sizing_passes == 1 and state.syncs == 2          # idle before the dry pass, idle after it
    assert m.lib.name == "real library" and isinstance(out, tuple)
    for B, T in [(4, 100), (2, 100), (4, 60), (1, 1)]:                                    # that size or smaller: covered
        log.clear()
        m._sized("decode", B, T, _runner(m, state, log))
        assert log == ["real"], (B, T, log)
    for B, T in [(5, 100), (4, 101)]:                                                      # larger in either dimension: sized again
        log.clear()
        m._sized("decode", B, T, _runner(m, state, log))
        assert log == ["dry", "real"], (B, T, log)
    assert m.sizing_passes == 3 and m.mallocs_in_flight == 0 and m.passes_repeated == 0
    # envelopes are per entry point and per stream; dominated entries are pruned
    assert sorted(m._envelopes[("decode", 11)]) == [(4, 101), (5, 100)]
    log.clear()
    m._sized("encode", 1, 1, _runner(m, state, log))
    assert log == ["dry", "real"]
    state.stream = _Stream(12)
    log.clear()
    m._sized("decode", 1, 1, _runner(m, state, log))
    assert log == ["dry", "real"]


def test_the_library_is_restored_when_the_dry_pass_raises(model):
    m, state = model

    def boom():
        raise ValueError("shape error inside the pass")

    with pytest.raises(ValueError):
        m._sized("decode", 2, 10, boom)
    assert m.lib.name == "real library" and m.sizing_passes == 0 and ("decode", 11) in m._envelopes and m._envelopes[("decode", 11)] == []


def test_a_pass_that_saw_a_driver_allocation_is_repeated(model):
    m, state = model
    log = []
    m._sized("decode", 4, 100, _runner(m, state, log))
    log.clear()
    before = state.syncs
    out = m._sized("decode", 4, 100, _runner(m, state, log, grow_on_real=(1,)))           # the counter moves during the first real attempt
    assert log == ["real", "real"] and m.mallocs_in_flight == 2 and m.passes_repeated == 1 and state.syncs == before + 1
    assert torch.equal(out[0], torch.zeros(3))
    log.clear()
    m._sized("decode", 4, 100, _runner(m, state, log, grow_on_real=(1, 2, 3)))            # it keeps moving: three attempts, then the last result stands
    assert log == ["real"] * 4 and m.passes_repeated == 4


def test_presize_off_is_a_plain_call(model):
    m, state = model
    m.presize = False
    log = []
    m._sized("decode", 9, 999, _runner(m, state, log))
    assert log == ["real"] and state.syncs == 0 and not m._envelopes


def test_tensors_of_walks_nested_results():
    a, b = torch.zeros(1), torch.ones(2)
    assert WM._tensors_of((a, [b, None, (a,)], "x", 3)) == [a, b, a]
    assert WM._tensors_of(None) == []
    stand_in = WM._NoLaunch()
    assert stand_in.ssrhip_gemm(1, 2, 3) == 0
    with pytest.raises(AttributeError):
        stand_in.something_else
