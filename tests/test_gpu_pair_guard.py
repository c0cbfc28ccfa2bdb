"""GPU: the pair launches of the 2-row decode step (csrc/gemv.hip gemv_pair_kernel / gemv_pair_merge_kernel) need their 256 workgroups
resident together — a precondition the library ENFORCES since round 6 instead of documenting it:

  * one engine per device holds the pairing slot (csrc/engine.hip: a per-process table + an flock on /dev/shm/ssrhip_pair_<pci>.lock);
    a second live 2-row engine — same process or another one — steps with the ordinary launches and says so;
  * two engines stepping concurrently on two streams therefore produce exactly the tokens of each run alone;
  * a foreign kernel that squats on half the CUs for longer than the spin bound makes a pair launch give up: the engine RAISES at
    the next poll (never silent garbage, never a hang), falls back to the ordinary launches and is correct again after the next start;
    a squatter that leaves in time only delays the step.

The reference decodes sequentially (inference_v2.py:331-333: the loop a user will parallelise); none of this exists there.
"""
import gc
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

import ssr_speech_amd  # noqa: F401
from ssr_speech_amd import _lib, layout as LY, weights as W
from ssr_speech_amd.engine import DecodeEngine, DecodeKnobs, LMWeightsArena

pytestmark = pytest.mark.gpu

STEPS = 24


@pytest.fixture(scope="module")
def arena():
    if torch.cuda.get_device_properties(0).multi_processor_count < 256:
        pytest.skip("the pair launches need 256 CUs")
    gc.collect()                                     # engines of earlier test modules give their slot back when they are collected
    args = W.lm_args_830m()
    sd = W.lm_state_dict(args, seed=0, device="cuda")
    return args, LMWeightsArena(args, sd, "cuda")


def _inputs(args, seed, L=40, N=60):
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(0, 100, (1, L), generator=g)
    y = torch.randint(0, 2048, (1, N, 4), generator=g)
    unc = torch.randint(0, 101, (1, L), generator=g)
    cated, _, num_task, _ = LY.build_layout(y[0].T.numpy(), np.asarray([[N, N]]), args)
    kn = DecodeKnobs(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=2, cfg_coef=1.5, cfg_stride=5, use_cfg=True, text_len=L,
                     n_spans=num_task, seed=seed)
    return [x[0].numpy(), unc[0].numpy()], cated, kn


def _engine(arena_, pair_mode=0):
    return DecodeEngine(arena_, 1, True, 1024, 256, pair_mode=pair_mode)


def _run_alone(eng, inp, steps=STEPS):
    rows, cated, kn = inp
    eng.start(rows, [cated], [kn])
    eng.decode(steps)
    st = eng.states()[0]
    return eng.tokens(0, int(st.n_steps))


def test_the_first_two_row_engine_pairs_and_a_second_one_does_not(arena, capfd):
    args, ar = arena
    e1 = _engine(ar)
    rows, cated, kn = _inputs(args, 11)
    e1.start(rows, [cated], [kn])
    assert e1.pairing, e1.pairing_why
    e2 = _engine(ar)
    e2.start(rows, [cated], [kn])
    assert not e2.pairing and "another decode engine of this process" in e2.pairing_why
    assert "WITHOUT pair launches" in capfd.readouterr().err               # said once, on stderr
    e3 = _engine(ar, pair_mode=1)
    e3.start(rows, [cated], [kn])
    assert not e3.pairing and "pair_mode 1" in e3.pairing_why
    e1.close()                                                             # the slot goes back ...
    e4 = _engine(ar)
    e4.start(rows, [cated], [kn])
    assert e4.pairing, e4.pairing_why                                      # ... and the next engine takes it
    for e in (e2, e3, e4):
        e.close()


def test_another_process_holding_the_slot_turns_pairing_off_here(arena):
    """The cross-process half of the guard: a child takes the device's lock file the way the library does and holds it."""
    args, ar = arena
    e0 = _engine(ar)
    rows, cated, kn = _inputs(args, 12)
    e0.start(rows, [cated], [kn])
    assert e0.pairing, e0.pairing_why
    e0.close()
    locks = [f for f in os.listdir("/dev/shm") if f.startswith("ssrhip_pair_") and f.endswith(".lock")]
    assert locks, "the library keeps its lock file under /dev/shm"
    child = subprocess.Popen([sys.executable, "-c",
                              "import fcntl, sys, time\n"
                              "fds = [open('/dev/shm/' + n, 'r') for n in sys.argv[1:]]\n"
                              "[fcntl.flock(f, fcntl.LOCK_EX | fcntl.LOCK_NB) for f in fds]\n"
                              "print('held', flush=True)\ntime.sleep(60)\n"] + locks, stdout=subprocess.PIPE, text=True)
    try:
        assert child.stdout.readline().strip() == "held"
        e1 = _engine(ar)
        e1.start(rows, [cated], [kn])
        assert not e1.pairing and "another process holds the pair-launch lock" in e1.pairing_why, e1.pairing_why
        tok = _run_alone(e1, (rows, cated, kn))
        e1.close()
    finally:
        child.kill()
        child.wait()
    e2 = _engine(ar)                                                        # the kernel dropped the dead process's lock
    e2.start(rows, [cated], [kn])
    assert e2.pairing, e2.pairing_why
    assert np.array_equal(_run_alone(e2, (rows, cated, kn)), tok)           # paired and unpaired steps: the same tokens
    e2.close()


def test_two_engines_stepping_concurrently_on_two_streams_equal_their_runs_alone(arena):
    args, ar = arena
    inps = [_inputs(args, 21), _inputs(args, 22, L=33, N=75)]
    engs = [_engine(ar), _engine(ar)]
    alone = [_run_alone(e, i) for e, i in zip(engs, inps)]
    assert engs[0].pairing and not engs[1].pairing
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    t0 = time.time()
    for rnd in range(3):
        for e, i, st in zip(engs, inps, streams):
            with torch.cuda.stream(st):
                e.start(i[0], [i[1]], [i[2]])
        for chunk in range(STEPS // 4):                                     # interleave the enqueues: both chains are in flight together
            for e, st in zip(engs, streams):
                with torch.cuda.stream(st):
                    e.decode(4)
        for k, (e, st) in enumerate(zip(engs, streams)):
            with torch.cuda.stream(st):
                s_ = e.states()[0]
                assert np.array_equal(e.tokens(0, int(s_.n_steps)), alone[k]), (rnd, k)
    assert time.time() - t0 < 30
    for e in engs:
        e.close()


@pytest.mark.parametrize("squat_ms,must_raise", [(150.0, False), (2500.0, True)])
def test_a_kernel_squatting_on_most_cus_delays_or_raises_but_never_corrupts(arena, squat_ms, must_raise):
    """200 workgroups that hold 158 of a CU's 160 KB of LDS each spin on a side stream while a paired chain steps: a pair workgroup needs
    ~18 KB, so nothing of it fits on their CUs, and the 56 CUs left hold 56 pair workgroups — the launch cannot become resident as a
    whole. (Round 6's first runs squatted too politely: with 140 KB of LDS per squatter a pair workgroup simply became resident BESIDE
    it — a 9 s squat only slowed the step.)
    Leaves in time: the pair launch waits for its missing workgroups and the tokens are the usual ones. Stays longer than the spin
    bound (1 s of the constant 100 MHz clock since round 6; the sweep count of round 5 waited out a 9 s squatter): the launch gives up,
    `states()` raises within about a second, and the engine — now without pair launches — decodes correctly again."""
    args, ar = arena
    inp = _inputs(args, 31)
    eng = _engine(ar)
    want = _run_alone(eng, inp)
    assert eng.pairing, eng.pairing_why
    L = _lib.lib()
    # a HIGH-priority stream: HIP keeps its own pool of hardware queues per priority, so the squatter cannot land on the hardware queue the
    # chain is launched on (in the full suite the process has made dozens of streams and an ordinary one may share the chain's queue — the
    # squatter then simply runs in FRONT of the chain: nothing is starved, nothing raises; seen in three full runs of the round)
    side = torch.cuda.Stream(priority=-1)
    rows, cated, kn = inp
    eng.start(rows, [cated], [kn])
    torch.cuda.synchronize()
    started = torch.zeros(1, dtype=torch.int32).pin_memory()                # every squatter workgroup counts itself in when it is resident
    _lib.check(L.ssrhip_debug_occupy(200, 158 * 1024, squat_ms, started.data_ptr(), side.cuda_stream), "ssrhip_debug_occupy")
    t_wait = time.time()
    while int(started[0]) < 200:                                            # (a fresh stream's first launch can take tens of ms: two of five runs of the
        assert time.time() - t_wait < 5.0, f"only {int(started[0])} of 200 squatters became resident"      # round started the chain too early and starved nobody)
        time.sleep(0.001)
    t0 = time.time()
    eng.decode(STEPS)
    raised = False
    try:
        st = eng.states()[0]
        got = eng.tokens(0, int(st.n_steps))
    except RuntimeError as e:
        raised = True
        assert "gave up" in str(e)
    dt = time.time() - t0
    torch.cuda.synchronize()
    assert dt < 3.0 + (0.0 if (must_raise and raised) else squat_ms / 1000.0), f"{dt:.1f} s: the wait is bounded by the clock, not by the squatter"
    if must_raise and not raised and dt >= 0.9 * squat_ms / 1000.0:
        assert np.array_equal(got, want)
        pytest.skip(f"the chain waited {dt:.1f} s behind the squatter instead of beside it (the two streams share a hardware queue): nothing to starve")
    if must_raise:
        assert raised, f"a {squat_ms:.0f} ms squatter must trip the spin bound"
    if raised:
        assert not eng.pairing and eng.pair_mode == 1                       # recovered: new context, ordinary launches
    else:
        assert np.array_equal(got, want)
    assert np.array_equal(_run_alone(eng, inp), want)                       # and the engine is usable afterwards, either way
    eng.close()
