"""CPU, 2 processes over gloo: the N>1 path of the DP decode — sharding is a partition, per-utterance seeds do not depend
on the world size, and the one all-gather returns every utterance's ragged token tensor to every rank in global order."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

import ssr_speech_amd  # noqa: F401
from ssr_speech_amd import dp


def test_shard_range_is_a_partition():
    for n in (0, 1, 5, 8, 64, 65):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                s, e = dp.shard_range(n, world, r)
                assert 0 <= s <= e <= n and (e - s) in (n // world, n // world + 1)
                cover += list(range(s, e))
            assert cover == list(range(n))
    assert dp.utterance_seed(7, 3) == 10


def test_balanced_shards_partition_and_balance():
    """`dp.balanced_shards`: a partition (ascending global indices per rank), a pure function of (costs, world), equal costs -> equal
    counts, and on the ragged bench queue (64 utterances, L uniform in 20..120, 150-frame prompts: 52..1052 steps each) eight ranks end
    within 10 % of the mean — the contiguous split by position does not."""
    g = torch.Generator().manual_seed(77)
    costs = []
    for _ in range(64):
        L = int(torch.randint(20, 121, (1,), generator=g))
        torch.randint(0, 100, (1, L), generator=g); torch.randint(0, 2048, (1, 150, 4), generator=g)      # the bench leg's draws
        costs.append(dp.utterance_cost(L, 150, 4))
    for world in (1, 2, 3, 8):
        owners = dp.balanced_shards(costs, world)
        assert sorted(i for o in owners for i in o) == list(range(64)) and all(o == sorted(o) for o in owners)
        assert owners == dp.balanced_shards(list(costs), world)
    bal = dp.plan_stats(costs, dp.balanced_shards(costs, 8))
    con = dp.plan_stats(costs, dp.contiguous_shards(64, 8))
    assert bal["max_over_mean"] <= 1.10, bal
    assert con["max_over_mean"] > bal["max_over_mean"] + 0.05, (con, bal)
    eq = dp.balanced_shards([5.0] * 64, 8)
    assert all(len(o) == 8 for o in eq) and eq[0] == list(range(0, 64, 8))
    few = dp.balanced_shards([3.0, 9.0, 1.0], 8)                     # fewer utterances than ranks: empty shards, the rest one each
    assert sorted(len(o) for o in few) == [0] * 5 + [1] * 3
    assert dp.utterance_cost(67, 150) > dp.utterance_cost(30, 150) > 0


def _count_collectives(dist):
    """wrap torch.distributed's collectives of THIS process so that a test can assert how many a call issued"""
    calls = {}
    for name in ("all_gather_into_tensor", "all_reduce", "all_gather", "broadcast", "all_to_all_single"):
        def wrap(fn, name=name):
            def inner(*a, **k):
                calls[name] = calls.get(name, 0) + 1
                return fn(*a, **k)
            return inner
        setattr(dist, name, wrap(getattr(dist, name)))
    return calls


def _tokens(i, K=4):
    g = torch.Generator().manual_seed(100 + i)
    T = 5 + (i * 7) % 11
    return torch.randint(0, 2048, (K, T), generator=g)


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, e = dp.shard_range(n_total, world, rank)
    local = [_tokens(i) for i in range(s, e)]
    calls = _count_collectives(dist)
    allt = dp.gather_tokens(local, n_total, 4, pad_token=2048)                       # no bound on the lengths: they travel first
    ok = len(allt) == n_total and all(torch.equal(allt[i], _tokens(i)) for i in range(n_total)) and calls == {"all_gather_into_tensor": 2}
    calls.clear()
    allt = dp.gather_tokens(local, n_total, 4, pad_token=2048, caps=[16] * n_total)  # bounded lengths: ONE fixed-layout block
    ok = ok and len(allt) == n_total and all(torch.equal(allt[i], _tokens(i)) for i in range(n_total)) and calls == {"all_gather_into_tensor": 1}
    # a length beyond its bound on ONE rank (rank 1's results do not fit caps = 4): no abort, no corrupted block — that rank flags it in
    # the block's header and EVERY rank falls back to the length exchange: the right tokens everywhere, three collectives in all
    calls.clear()
    mine = [t[:, :3] for t in local] if rank == 0 else local
    allt = dp.gather_tokens(mine, n_total, 4, pad_token=2048, caps=[4] * n_total)
    s0, e0 = dp.shard_range(n_total, world, 0)
    want = [(_tokens(i)[:, :3] if s0 <= i < e0 else _tokens(i)) for i in range(n_total)]
    ok = ok and all(torch.equal(allt[i], want[i]) for i in range(n_total)) and calls == {"all_gather_into_tensor": 3}
    # a rank whose decode FAILED still takes part, and every rank raises naming it — nobody hangs
    try:
        dp.gather_tokens(local if rank == 0 else [], n_total, 4, pad_token=2048, caps=[16] * n_total, ok=(rank == 0))
        ok = False
    except RuntimeError as ex:
        ok = ok and "rank(s) [1]" in str(ex)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [5, 8])
def test_gather_tokens_world2(n_total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


class _StubModel:
    """Stands in for SSR_Speech on the CPU: `inference_batch` returns tokens that depend only on (utterance, seed + global
    index), which is the contract `dp.generate` relies on."""

    class args:
        n_codebooks = 4
        empty_token = 2048

    def inference_batch(self, utterances, seed=0, first_index=0, indices=None, **kw):
        out = []
        for j, u in enumerate(utterances):
            g = torch.Generator().manual_seed(seed + (indices[j] if indices is not None else first_index + j))
            T = int(u["x"].shape[1]) + 3
            out.append((torch.randint(0, 2048, (1, 4, T), generator=g), None, None, None))
        return out


def _utts(n_total):
    """text lengths 4, 5, .. and prompts of 3 frames: costs differ, so the balanced plan is NOT the contiguous one"""
    return [dict(x=torch.zeros(1, 4 + i, dtype=torch.long), y=torch.zeros(1, 3, 4, dtype=torch.long)) for i in range(n_total)]


def _plan(n_total, world):
    return dp.balanced_shards([dp.utterance_cost(4 + i, 3, 4) for i in range(n_total)], world)


def _gen_worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    utts = _utts(n_total)
    ref = _StubModel().inference_batch(utts, seed=40, first_index=0)          # what ONE process would produce
    ok = True
    calls = _count_collectives(dist)
    for balance in (True, False):
        calls.clear()
        toks, (mine, outs) = dp.generate(_StubModel(), utts, seed=40, balance=balance)
        ok = ok and calls == {"all_gather_into_tensor": 1}               # north_star: a SINGLE all-gather (no length exchange, no agreement all-reduce)
        want = _plan(n_total, world)[rank] if balance else list(range(*dp.shard_range(n_total, world, rank)))
        ok = ok and len(toks) == n_total and all(torch.equal(toks[i], ref[i][0][0]) for i in range(n_total)) and mine == want and len(outs) == len(mine) \
            and all(torch.equal(o[0], ref[gi][0]) for gi, o in zip(mine, outs))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 1])
def test_generate_world2_equals_single_process(n_total):
    """`dp.generate` on 2 ranks == the same utterances decoded by one process (per-utterance seeds), incl. an empty shard."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gen_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


class _FailsOnRank1(_StubModel):
    def inference_batch(self, utterances, seed=0, first_index=0, indices=None, **kw):
        import torch.distributed as dist
        if dist.get_rank() == 1:
            raise ValueError("boom")
        return super().inference_batch(utterances, seed=seed, first_index=first_index, indices=indices, **kw)


def _fail_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    utts = _utts(6)
    calls = _count_collectives(dist)
    try:
        dp.generate(_FailsOnRank1(), utts, seed=1)
        q.put((rank, "returned"))
    except RuntimeError as e:
        q.put((rank, ("this" if "on this rank" in str(e) else "another") + ("" if calls == {"all_gather_into_tensor": 1} else f" {calls}")))
    dist.barrier()                       # both ranks are still in step: nobody is stuck in the all-gather
    dist.destroy_process_group()


def test_generate_failure_on_one_rank_raises_on_all_ranks():
    """A rank whose decode throws must not strand the others in the token all-gather: every rank raises instead."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fail_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, "another"), (1, "this")]


class _StubTokenizer:
    """Stands in for `AudioTokenizer` on the CPU: a waveform that depends only on the utterance's own codes."""

    def decode_batch(self, frames, scale=None):
        return [f.to(torch.float32).sum(1, keepdim=True).repeat_interleave(320, dim=-1) for f in frames]


class _StubModelMasks(_StubModel):
    def inference_batch(self, utterances, seed=0, first_index=0, indices=None, **kw):
        return [(r, torch.zeros(1, r.shape[-1], dtype=torch.long), [(0, 2)], [(0, 2)])
                for r, _, _, _ in super().inference_batch(utterances, seed, first_index, indices=indices, **kw)]


def _synth_worker(rank, world, port, n_total, outdir, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    utts = _utts(n_total)
    stats = {}
    waves, mine, toks = dp.synthesize(_StubModelMasks(), _StubTokenizer(), utts, seed=40, tts=True, output_dir=outdir, stats=stats)
    ref = _StubModelMasks().inference_batch(utts, seed=40, first_index=0)      # what ONE process would produce
    ref_w = [w[..., 2 * 320:] for w in _StubTokenizer().decode_batch([r[0] for r in ref])]
    ok = mine == _plan(n_total, world)[rank] and len(waves) == len(mine) and len(toks) == n_total \
        and all(torch.equal(w, ref_w[gi]) for gi, w in zip(mine, waves)) and "codec_s" in stats and stats["shard"] == mine \
        and all(os.path.exists(os.path.join(outdir, f"utt{i:05d}.wav")) for i in mine)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 1])
def test_synthesize_world2_each_rank_decodes_its_own_slice(n_total, tmp_path):
    """`dp.synthesize` on 2 ranks: tokens are all-gathered, then rank r renders the waveforms of ITS (cost-balanced) shard only; together
    the two ranks produce exactly the waveforms (and files) one process would (incl. an empty shard: n_total = 1)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_synth_worker, args=(r, 2, port, n_total, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
    assert sorted(os.listdir(tmp_path)) == [f"utt{i:05d}.wav" for i in range(n_total)]
