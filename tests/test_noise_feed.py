"""CPU: `TorchCpuNoiseFeed` — the chunked host-side stand-in for the Exp(1) draws `torch.multinomial` makes once per decode
step in the reference (models/ssr.py:85). It must (a) produce the same numbers as per-step draws, whatever the chunking,
(b) leave the generator where the reference leaves it after exactly n steps, (c) behave the same for the global generator
and for a private `torch.Generator` seeded alike (the contract of `inference_batch`: row i == a batch-1 run after
`torch.manual_seed(seed + i)`)."""
import pytest
import torch

import ssr_speech_amd  # noqa: F401
from ssr_speech_amd.engine import TorchCpuNoiseFeed

K, CARD = 4, 2056


def per_step(n, generator=None):
    return torch.stack([torch.empty(K, CARD).exponential_(1, generator=generator) for _ in range(n)]) if n else torch.empty(0, K, CARD)


@pytest.mark.parametrize("chunks", [[16, 16, 16], [1, 5, 64, 3], [37]])
def test_chunked_draws_equal_per_step_draws_global_generator(chunks):
    n = sum(chunks)
    torch.manual_seed(123)
    torch.randint(0, 101, (1, 30))                    # the uncond_x draw that precedes sampling (ssr.py:574)
    want = per_step(n)
    after = torch.get_rng_state()
    torch.manual_seed(123)
    torch.randint(0, 101, (1, 30))
    feed = TorchCpuNoiseFeed([None], K, CARD)
    got = []
    for c in chunks:
        buf = torch.empty(c, K, CARD)
        feed.draw(0, buf)
        got.append(buf)
    assert torch.equal(torch.cat(got), want)
    assert torch.equal(torch.get_rng_state(), after)


@pytest.mark.parametrize("n_taken", [0, 1, 16, 17, 40, 47, 48])
def test_finish_rewinds_to_the_state_after_exactly_n_steps(n_taken):
    torch.manual_seed(7)
    per_step(n_taken)
    want_state = torch.get_rng_state()
    want_next = torch.rand(3)
    torch.manual_seed(7)
    feed = TorchCpuNoiseFeed([None], K, CARD)
    for _ in range(3):                                 # 48 steps drawn ahead, only n_taken consumed
        feed.draw(0, torch.empty(16, K, CARD))
    feed.finish(0, n_taken)
    assert torch.equal(torch.get_rng_state(), want_state)
    assert torch.equal(torch.rand(3), want_next)


def test_private_generators_match_the_global_stream_and_do_not_interact():
    seeds = [1000, 1001]
    want = []
    for s in seeds:
        torch.manual_seed(s)
        u = torch.randint(0, 101, (1, 12))
        want.append((u, per_step(20)))
    gens = [torch.Generator().manual_seed(s) for s in seeds]
    unc = [torch.randint(0, 101, (1, 12), generator=g) for g in gens]
    feed = TorchCpuNoiseFeed(gens, K, CARD)
    got = [[], []]
    for _ in range(2):                                 # interleaved chunks, as the engine draws them
        for u in range(2):
            buf = torch.empty(10, K, CARD)
            feed.draw(u, buf)
            got[u].append(buf)
    for u in range(2):
        assert torch.equal(unc[u], want[u][0])
        assert torch.equal(torch.cat(got[u]), want[u][1])
    feed.finish(0, 13)
    ref = torch.Generator().manual_seed(seeds[0])
    torch.randint(0, 101, (1, 12), generator=ref)
    per_step(13, generator=ref)
    assert torch.equal(gens[0].get_state(), ref.get_state())


def test_self_check_and_the_per_step_fallback(monkeypatch):
    """the chunked draw is an assumption about this torch build — the feed checks it once per shape and, on a build where
    it fails, draws step by step into the same staging buffer (same numbers, same generator end state, `finish` included)."""
    assert TorchCpuNoiseFeed.chunked_draws_match(K, CARD) is True            # this image's build
    monkeypatch.setitem(TorchCpuNoiseFeed._chunk_ok, (K, CARD), False)      # pretend the build fails the check
    torch.manual_seed(11)
    want = per_step(40)
    torch.manual_seed(11)
    feed = TorchCpuNoiseFeed([None], K, CARD)
    assert feed.chunked is False
    got = []
    for c in (16, 16, 16):
        buf = torch.empty(c, K, CARD)
        feed.draw(0, buf)
        got.append(buf)
    assert torch.equal(torch.cat(got)[:40], want)
    feed.finish(0, 40)
    torch.manual_seed(11)
    per_step(40)
    assert torch.equal(torch.get_rng_state(), torch.get_rng_state())
    ref_next = torch.rand(2)
    torch.manual_seed(11)
    feed2 = TorchCpuNoiseFeed([None], K, CARD)
    for c in (16, 16, 16):
        feed2.draw(0, torch.empty(c, K, CARD))
    feed2.finish(0, 40)
    assert torch.equal(torch.rand(2), ref_next)
