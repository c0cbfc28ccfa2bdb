"""Data-parallel sharding of independent utterances over the GPUs of one node and the ONE collective of the
path: an all-gather of the generated codec tokens before codec decode. The reference has no
multi-GPU inference (`--sample_batch_size` is a sequential loop, `inference_v2.py:331-333`); each utterance
is an independent AR chain (own text, prompt, cache, RNG stream, stop state), so sharding needs no
data-path collective during decode. Works with any torch.distributed backend (nccl == RCCL on ROCm; gloo in the CPU tests).
"""
from __future__ import annotations

from typing import List, NamedTuple, Sequence, Tuple

import torch


class LocalShard(NamedTuple):
    """second element of `generate`'s return: this rank's utterances (ascending GLOBAL indices, cost-balanced: not a range) and their 4-tuples"""
    indices: List[int]
    outs: list


class Synthesized(NamedTuple):
    """`synthesize`'s return: waves[j] = waveform of utterance indices[j] (this rank's shard); tokens = every utterance's codes"""
    waves: list
    indices: List[int]
    tokens: list


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of utterances for `rank`: sizes differ by at most one, order preserved."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def utterance_cost(text_len: int, prompt_frames: int, n_codebooks: int = 4) -> float:
    """What an utterance is expected to cost its rank, in decode steps: the reference stops a span once the audio position passes
    10 x the text length (models/ssr.py:739), so `10 L - T0` bounds — and for long texts tracks — the number of generated frames;
    the prefill of its L + T0 positions is worth about one step per 32 positions (13 ms for ~500 positions x 2 rows against 0.83 ms
    per step)."""
    steps = max(10 * int(text_len) + 2 - int(prompt_frames), 1) + n_codebooks + 1
    return float(steps) + (int(text_len) + int(prompt_frames)) / 32.0


def token_cap(text_len: int, y_frames: int, n_spans: int = 1, n_codebooks: int = 4) -> int:
    """Upper bound on the columns T' of an utterance's result `res [1, K, T']`, from sizes every rank knows before anything is decoded:
    the kept part of `y` (<= y_frames) plus what the spans can generate — a span ends at the latest when the audio position passes
    10 x the text length (models/ssr.py:739) and then emits its K eog steps (`SSR_Speech.inference`: cap = max(10 L + 2 - T0, 1) +
    n_spans (K + 1), T0 >= 0). This is what lets the all-gather use ONE fixed-layout block instead of exchanging lengths first."""
    return int(y_frames) + max(10 * int(text_len) + 2, 1) + int(n_spans) * (int(n_codebooks) + 1)


def balanced_shards(costs: Sequence[float], world: int) -> List[List[int]]:
    """Utterance -> rank by cost instead of by position ("length-sorted round-robin to balance steps"): longest processing
    time first — utterances in order of decreasing cost, each to the rank with the smallest load so far (ties: fewer utterances, then
    the lower rank). A rank runs its utterances through 8 slots with refill, so its time is ~ max(sum / 8, longest): both terms are
    what LPT evens out. Pure function of (costs, world): every rank computes the same plan without communicating. Returns owners[r] =
    the GLOBAL indices of rank r's utterances, ascending (utterance i keeps seed + i and its place in every result list)."""
    owners: List[List[int]] = [[] for _ in range(world)]
    load = [0.0] * world
    for i in sorted(range(len(costs)), key=lambda j: (-float(costs[j]), j)):
        r = min(range(world), key=lambda q: (load[q], len(owners[q]), q))
        owners[r].append(i)
        load[r] += float(costs[i])
    return [sorted(o) for o in owners]


def contiguous_shards(n_items: int, world: int) -> List[List[int]]:
    return [list(range(*shard_range(n_items, world, r))) for r in range(world)]


def plan_stats(costs: Sequence[float], owners: Sequence[Sequence[int]], slots: int = 8) -> dict:
    """max / mean of the ranks' estimated times under a plan (a rank's time ~ max(sum of its costs / slots, its longest utterance))."""
    t = [max(sum(costs[i] for i in o) / slots, max((costs[i] for i in o), default=0.0)) for o in owners]
    mean = sum(t) / max(len(t), 1)
    return {"rank_time": t, "max_over_mean": (max(t) / mean) if mean > 0 else 1.0}


def utterance_seed(seed: int, global_index: int) -> int:
    """Per-utterance RNG seed: independent of the world size (same convention as `inference_v2.py:332`, seed+num)."""
    return int(seed) + int(global_index)


def _collective_device(hint=None) -> torch.device:
    """Device the collective's buffers must live on. It is a property of the BACKEND, not of this rank's data: under nccl
    (RCCL) every rank must pass GPU tensors even when its own shard is empty (fewer utterances than ranks)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    if hint is not None and torch.device(hint).type == "cuda" and dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        return torch.device("cpu")        # gloo collectives run on host buffers
    return torch.device(hint) if hint is not None else torch.device("cpu")


def gather_tokens(local: Sequence[torch.Tensor], n_total: int, K: int, pad_token: int, device=None, force_collective: bool = False,
                  owners: Sequence[Sequence[int]] = None, caps: Sequence[int] = None, ok: bool = True) -> List[torch.Tensor]:
    """local[i]: int tensor [K, T_i] of this rank's utterances, in the order of owners[rank] (default: the rank-contiguous shard of `n_total`).
    Returns the list of all `n_total` token tensors on every rank.

    ONE collective (north_star: "a single RCCL all-gather ... to collect generated codec tokens"): every rank contributes one int32 block
    of a layout all ranks can compute without talking to each other,

        [ ok | n_local | lengths[n_max] | tokens[n_max][K][T_cap] ]          T_cap = max(caps), n_max = the largest shard

    `caps[i]` bounds T_i for EVERY utterance of the job (`token_cap`: known from text / prompt sizes before decoding); `ok = False` marks
    a rank whose decode failed — it still takes part (with no tokens) and every rank raises after the gather, so nobody is left waiting
    (this replaces round 4's MIN all-reduce + all-gather of lengths + all-gather of the block: three latency-bound collectives).
    Without `caps` the lengths have to travel first (two collectives; kept for callers that cannot bound their lengths).
    A world of one returns the local list without a collective unless `force_collective` (tests: a 1-rank nccl group then issues the
    very RCCL call an 8-GPU job issues)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
        if not ok:
            raise RuntimeError("dp.gather_tokens: the decode failed on this rank")
        return [t.clone() for t in local]
    world, rank = dist.get_world_size(), dist.get_rank()
    if device is None:
        device = _collective_device(local[0].device if len(local) else None)
    if owners is None:
        owners = contiguous_shards(n_total, world)
    assert len(owners) == world and sorted(i for o in owners for i in o) == list(range(n_total)), "owners must partition the utterances"
    if ok:
        assert len(local) == len(owners[rank]), (len(local), len(owners[rank]))
    n_max = max(max(len(o) for o in owners), 1)
    if caps is not None:
        assert len(caps) == n_total, (len(caps), n_total)
        t_cap = max(max((int(c) for c in caps), default=1), 1)
        if ok and any(int(t.shape[0]) != K for t in local):
            ok = False
        # a result longer than its bound (`token_cap` is derived from the reference's stop rules; an engine-side cap or a caller's own
        # `caps` may be tighter than what came out) must neither corrupt the block nor abort the job: the rank says so in
        # its header (2) and EVERY rank — they all read the same headers — then takes the two-collective path that exchanges lengths
        over = bool(ok) and any(int(t.shape[1]) > t_cap for t in local)
        head = 2 + n_max
        mine = torch.full((head + n_max * K * t_cap,), int(pad_token), dtype=torch.int32)
        mine[0], mine[1] = (2 if over else int(bool(ok))), len(local) if ok else 0
        mine[2:head] = 0
        if ok and not over:
            body = mine[head:].view(n_max, K, t_cap)
            for i, t in enumerate(local):
                mine[2 + i] = int(t.shape[1])
                body[i, :, : t.shape[1]] = t.detach().to("cpu", torch.int32)
        mine = mine.to(device)
        out = torch.empty(world * mine.numel(), dtype=torch.int32, device=device)
        dist.all_gather_into_tensor(out, mine)
        out = out.view(world, -1).cpu()
        bad = [r for r in range(world) if int(out[r, 0]) == 0]
        if bad:
            raise RuntimeError(f"dp.gather_tokens: the decode failed on rank(s) {bad}" + (" (this rank among them)" if rank in bad else ""))
        if any(int(out[r, 0]) == 2 for r in range(world)):
            return _gather_with_lengths(local, n_total, K, pad_token, device, owners, n_max)
        res: List[torch.Tensor] = [None] * n_total
        for r in range(world):
            assert int(out[r, 1]) == len(owners[r]), (r, int(out[r, 1]), len(owners[r]))
            body = out[r, head:].view(n_max, K, t_cap)
            for i, gi in enumerate(owners[r]):
                res[gi] = body[i, :, : int(out[r, 2 + i])].to(torch.int64).clone().to(device)
        return res
    if not ok:
        raise RuntimeError("dp.gather_tokens: the decode failed on this rank and no `caps` were given (the other ranks cannot be told)")
    return _gather_with_lengths(local, n_total, K, pad_token, device, owners, n_max)


def _gather_with_lengths(local, n_total: int, K: int, pad_token: int, device, owners, n_max: int) -> List[torch.Tensor]:
    """Two collectives: the lengths, then a block padded to the longest result of the job (callers without a bound on their lengths, and
    the fall-back of the one-collective path when a result came out longer than its bound)."""
    import torch.distributed as dist
    world = dist.get_world_size()
    lens = torch.zeros(n_max, dtype=torch.int32, device=device)
    for i, t in enumerate(local):
        lens[i] = t.shape[1]
    all_lens = torch.empty(world * n_max, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(all_lens, lens)
    t_max = int(all_lens.max().item())
    block = torch.full((n_max, K, max(t_max, 1)), pad_token, dtype=torch.int32, device=device)
    for i, t in enumerate(local):
        block[i, :, : t.shape[1]] = t.to(device, torch.int32)
    out = torch.empty((world,) + tuple(block.shape), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(out.view(-1), block.view(-1))
    all_lens = all_lens.view(world, n_max)
    res = [None] * n_total
    for r in range(world):
        for i, gi in enumerate(owners[r]):
            res[gi] = out[r, i, :, : int(all_lens[r, i])].to(torch.int64).clone()
    return res


def generate(model, utterances: Sequence[dict], seed: int = 0, pad_token: int = None, device=None, stats: dict = None,
             force_collective: bool = False, costs: Sequence[float] = None, balance: bool = True, caps: Sequence[int] = None, **decode_kw):
    """BASELINE config 4 in one call: shard `utterances` (dicts {x, y, mask_interval}, see `SSR_Speech.inference_batch`)
    over the ranks of the default process group, decode this rank's shard in lock-step (up to 8 utterances x CFG rows per
    engine pass, rows refilled as utterances finish), and all-gather the generated codec tokens — ONE collective, `gather_tokens` —
    so that every rank holds all of them before codec decode. Utterance i uses the RNG stream `seed + i` whatever the world size
    and whichever rank runs it. A decode that fails on one rank raises on every rank (the failure travels in the gathered block).

    The shard of a rank is chosen by COST (`balanced_shards` over `utterance_cost(L_i, T0_i)`; `costs` overrides the estimate —
    then only this rank's own utterances need their `y`, the others may be None, and `caps[i]` (`token_cap`: the bound on utterance
    i's result length every rank must agree on) has to come with it; `balance=False`: contiguous blocks by position).
    Returns (tokens, LocalShard(indices, outs)): tokens[i] = the int64 [K, T_i'] result of utterance i (all utterances, on every
    rank); indices = the ascending global indices of this rank's shard, outs = their 4-tuples."""
    import torch.distributed as dist
    in_group = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if in_group else 1
    rank = dist.get_rank() if world > 1 else 0
    collective = world > 1 or (force_collective and in_group)
    n_total = len(utterances)
    K = int(model.args.n_codebooks)
    if not balance or world == 1:
        owners = contiguous_shards(n_total, world)
    else:
        if costs is None:
            costs = [utterance_cost(u["x"].shape[-1], u["y"].shape[1], K) for u in utterances]
        assert len(costs) == n_total
        owners = balanced_shards(costs, world)
    if caps is None and collective:
        missing = [i for i, u in enumerate(utterances) if u is None or u.get("y") is None]
        if missing:
            raise ValueError(f"dp.generate: utterances {missing[:4]}... carry no `y` on this rank: pass `caps` (dp.token_cap per utterance) with `costs`")
        caps = [token_cap(u["x"].shape[-1], u["y"].shape[1], int(u["mask_interval"].shape[-2]) if u.get("mask_interval") is not None else 1, K)
                for u in utterances]
    mine = owners[rank]
    import time
    t0 = time.perf_counter()
    failure = None
    try:
        outs = model.inference_batch([utterances[i] for i in mine], seed=seed, indices=mine, **decode_kw) if mine else []
    except Exception as e:              # noqa: BLE001 — re-raised below, on EVERY rank
        failure, outs = e, []
    if stats is not None:           # wall time of this rank's lock-step decode (inference_batch ends on a device->host read)
        stats["decode_s"] = time.perf_counter() - t0
    if failure is not None and not collective:
        raise failure
    if pad_token is None:
        pad_token = int(model.args.empty_token)
    toks = [o[0][0] for o in outs]                                   # res [1, K, T'] -> [K, T']
    if device is None:
        device = _collective_device(getattr(model, "device", None))
    t1 = time.perf_counter()
    try:
        everyone = gather_tokens(toks, n_total, K, pad_token, device=device, force_collective=force_collective, owners=owners, caps=caps,
                                 ok=failure is None)
    except RuntimeError as e:
        if failure is not None:
            raise RuntimeError(f"dp.generate: the decode failed on this rank: {failure!r}") from failure
        raise RuntimeError(f"dp.generate: the decode failed on another rank ({e})") from e
    if stats is not None:
        stats["allgather_s"] = time.perf_counter() - t1
        stats["shard"] = list(mine)
    return everyone, LocalShard(list(mine), outs)


def synthesize(model, audio_tokenizer, utterances: Sequence[dict], seed: int = 0, use_watermark: bool = False, tts: bool = True,
               output_dir: str = None, names: Sequence[str] = None, sample_rate: int = 16000, stats: dict = None,
               force_collective: bool = False, **decode_kw):
    """The batched-TTS path end to end (BASELINE config 4; north_star: "a single RCCL all-gather ... to collect generated codec
    tokens before wmencodec decode"): `generate` (shard -> lock-step decode -> all-gather of the tokens), then the codec decode is
    again sharded by utterance: **rank r turns the utterances [lo, hi) of ITS OWN shard into waveforms** (one ragged pass of the
    codec per length bucket, `render_many`), using the all-gathered tokens — so every rank could decode any slice, and the
    waveforms of utterance i do not depend on the world size. No second collective: waveforms stay on the rank that made them
    (written to `output_dir/{names[i]}.wav` when given).

    utterances[i]: {x, y, mask_interval} as for `generate`, plus `wav` ([1, n] original 16 kHz audio, or a path) when
    `use_watermark` (the watermark decoder's skip input, inference_scale.py:67-78).
    Replaces the reference's per-sample loop inference_v2.py:331-358 + inference_scale.py:63-86.
    Returns Synthesized(waves, indices, tokens): waves[j] = waveform [1, 1, n] of utterance indices[j] (this rank's shard, ascending global
    indices — cost-balanced, see `generate`); tokens = all utterances' codes."""
    import time
    from .inference_scale import render_many
    tokens, (mine, outs) = generate(model, utterances, seed=seed, stats=stats, force_collective=force_collective, **decode_kw)   # costs / caps ride in decode_kw
    t0 = time.perf_counter()
    dev = getattr(model, "device", None)
    # the gathered tokens are the codec's input (what any rank could decode); marks / kept intervals are this rank's own
    results = [(tokens[gi].unsqueeze(0).to(dev if dev is not None else tokens[gi].device), o[1], o[2], o[3]) for gi, o in zip(mine, outs)]
    waves = []
    if results:
        audio = [utterances[gi].get("wav") for gi in mine] if use_watermark else None
        waves = render_many(audio_tokenizer, results, None, audio, bool(use_watermark), bool(tts))
    if output_dir is not None and waves:
        import os
        from .data.tokenizer import write_wav
        os.makedirs(output_dir, exist_ok=True)
        for gi, w in zip(mine, waves):
            name = names[gi] if names is not None else f"utt{gi:05d}"
            write_wav(os.path.join(output_dir, f"{name}.wav"), w[0].cpu(), sample_rate)
    if stats is not None:
        if waves and waves[0].is_cuda:
            torch.cuda.synchronize(waves[0].device)
        stats["codec_s"] = time.perf_counter() - t0
    return Synthesized(waves, list(mine), tokens)
