"""ORACLE (test infrastructure, not product code) — CPU restatement of the reference's
autoregressive codec-token decode path in plain PyTorch fp32 ops.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.
It mirrors the reference op-for-op (including its per-step `torch.cat` KV-cache growth and the full
[B*H,S,S] float-mask construction) so that, timed, it is a fair stand-in for the reference's CPU path.

Pinned against the reference itself: `oracle/make_golden.py` imports `/root/reference/models/ssr.py`
in the build container, runs both on identical weights/inputs/seeds and commits the reference's
outputs under `tests/golden/` (the reference's own repo holds no tests for this path).

Each function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def reference_params(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """The reference holds its weights as nn.Parameters (requires_grad=True) and runs under
    torch.no_grad(); ATen's CPU `linear` picks a different (1-ulp different) kernel path for
    weights that require grad. Flagging the oracle's tensors the same way makes every fp32
    intermediate BIT-identical to the reference (checked in tests/test_oracle_lm.py)."""
    return {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}


# ----------------------------------------------------------------------------- embeddings
def sine_pe(n: int, dim: int) -> Tensor:
    """models/modules/embedding.py:67-92 (SinePositionalEmbedding.extend_pe), fp32, [n, dim]."""
    pe = torch.zeros(n, dim)
    position = torch.arange(0, n, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(10000.0) / dim))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def pos_embed(x: Tensor, alpha: Tensor, pe: Tensor) -> Tensor:
    """embedding.py:94-97: x*1.0 + alpha*pe[:T] (x_scale=1 because scale=False, ssr.py:150,156)."""
    return x * 1.0 + alpha * pe[: x.size(1)].unsqueeze(0)


def embed_y(sd: Dict[str, Tensor], cated_y: Tensor, K: int) -> Tensor:
    """models/ssr.py:191-198: [K,T,B] int64 -> [B,T,D] (sum over codebooks)."""
    e = torch.stack([F.embedding(cated_y[k], sd[f"audio_embedding.{k}.word_embeddings.weight"]) for k in range(K)], dim=0)
    return e.sum(dim=0).transpose(1, 0)


# ----------------------------------------------------------------------------- layout (A3)
def rearrange(y: Tensor, non_mask_intervals, mask_intervals, args) -> List[Tensor]:
    """models/ssr.py:381-406."""
    K = args.n_codebooks
    sos = torch.full((K, 1), args.sos, dtype=torch.long)
    eos = torch.full((K, 1), args.eos, dtype=torch.long)
    eog = torch.full((K, 1), args.eog, dtype=torch.long)
    out = []
    for i, item in enumerate(non_mask_intervals):
        if i == 0:
            out.append(sos if item[0] == item[1] else torch.cat([sos, y[:, item[0]:item[1]]], dim=-1))
        elif i == len(non_mask_intervals) - 1:
            out.append(eos if item[0] == item[1] else torch.cat([y[:, item[0]:item[1]], eos], dim=-1))
        else:
            out.append(y[:, item[0]:item[1]])
    for item in mask_intervals:
        out.append(torch.cat([y[:, item[0]:item[1]], eog], dim=-1))
    return out


def pattern_sequence(tokens: Tensor, special: int) -> Tensor:
    """models/ssr.py:408-436 with delays=[0..K-1], empty_initial=0."""
    K, T = tokens.shape
    out = torch.full((K, T + K - 1), special, dtype=tokens.dtype)
    for t in range(T):
        for q in range(K):
            out[q, t + q] = tokens[q, t]
    return out


def revert_pattern_sequence(pattern: Tensor, special: int) -> Tensor:
    """models/ssr.py:438-464."""
    K, S = pattern.shape
    T = S - (K - 1)
    out = torch.full((K, T), special, dtype=pattern.dtype)
    for t in range(T):
        for q in range(K):
            if t + q < S:
                out[q, t] = pattern[q, t + q]
    return out


def insert_mask(shifted: List[Tensor], args) -> Tuple[List[Tensor], List[int]]:
    """models/ssr.py:472-494 (shuffle_mask_embedding=0)."""
    num_masks = (len(shifted) - 1) // 2
    assert num_masks == (len(shifted) - 1) / 2, len(shifted)
    emb_inds = list(range(args.mts, args.mts + args.max_n_spans))[:num_masks]
    mask_value = emb_inds + emb_inds
    inserted, mask_position = [], []
    for j in range(len(shifted) - 1):
        inserted.append(shifted[j])
        mask_position.append(sum(it.shape[1] for it in inserted))
        inserted.append(torch.full((args.n_codebooks, 1), mask_value[j], dtype=torch.long))
    inserted.append(shifted[-1])
    return inserted, mask_position


def build_layout(y: Tensor, mask_interval: Tensor, args):
    """models/ssr.py:604-625. y [K,T] int64; mask_interval [M,2]. Returns
    (cated_y [K,T0], mask_position, num_task, non_mask_intervals, mask_intervals)."""
    y_len = y.shape[1]
    starts = [int(it[0]) for it in mask_interval] + [y_len]
    ends = [0] + [int(it[1]) for it in mask_interval]
    mask_intervals = [(int(it[0]), int(it[1])) for it in mask_interval]
    non_mask_intervals = [(ns, ne) for ns, ne in zip(ends, starts)]
    rearranged = rearrange(y, non_mask_intervals, mask_intervals, args)
    shifted = [pattern_sequence(t, args.empty_token) for t in rearranged]           # :466-470
    inserted, mask_position = insert_mask(shifted, args)
    cated = torch.cat(inserted, dim=1)                                             # :496-502
    num_task = len(mask_position) // 2
    cated = cated[:, : mask_position[num_task]]
    return cated, mask_position, num_task, non_mask_intervals, mask_intervals


# ----------------------------------------------------------------------------- transformer
def mha(sd, pfx: str, x: Tensor, attn_mask: Tensor, nhead: int, past: Optional[Tensor]):
    """models/modules/activation.py:513-652 (KV-cache branch). x [B,T,D] batch-first;
    attn_mask float [B*H,T,S]. Returns (out [B,T,D], present [2,B,H,T,hd] | None)."""
    B, T, D = x.shape
    hd = D // nhead
    q_in = x.transpose(1, 0)                                                        # [T,B,D] (:495-500 batch_first)
    proj = F.linear(q_in, sd[pfx + "in_proj_weight"], sd[pfx + "in_proj_bias"])      # :86
    q, k, v = proj.chunk(3, dim=-1)                                                 # :88-89
    q = q.contiguous().view(T, B * nhead, hd).transpose(0, 1)                        # :543-545
    k = k.contiguous().view(T, B * nhead, hd).transpose(0, 1)
    v = v.contiguous().view(T, B * nhead, hd).transpose(0, 1)
    q = q.view(B, nhead, T, hd)                                                     # :622-624
    k = k.view(B, nhead, T, hd)
    v = v.view(B, nhead, T, hd)
    present = None
    if past is not None:
        present = torch.stack([k, v], dim=0)                                        # :627
        if past.ndim > 2:
            pk, pv = past
            k = torch.cat([pk, k], dim=-2)                                          # :630-631
            v = torch.cat([pv, v], dim=-2)
    S = k.shape[-2]
    m = attn_mask.view(B, nhead, -1, S)                                             # :620
    o = F.scaled_dot_product_attention(q, k, v, m, 0.0, is_causal=False)            # :634
    o = o.permute(2, 0, 1, 3).contiguous().view(B * T, D)                           # :635
    o = F.linear(o, sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"])          # :637
    o = o.view(T, B, D).transpose(1, 0)                                             # :638, :650
    return o, present


def layer_norm(sd, pfx: str, x: Tensor) -> Tensor:
    """models/modules/transformer.py:58-75, eps=1e-5."""
    return F.layer_norm(x, (x.shape[-1],), sd[pfx + "weight"], sd[pfx + "bias"], 1e-5)


def encoder_layer(sd, l: int, x: Tensor, attn_mask: Tensor, nhead: int, past):
    """models/modules/transformer.py:321-329 (norm_first), _ff_block :386-388 with F.relu (:188)."""
    p = f"decoder.layers.{l}."
    a, present = mha(sd, p + "self_attn.", layer_norm(sd, p + "norm1.", x), attn_mask, nhead, past)
    x = x + a
    h = F.linear(layer_norm(sd, p + "norm2.", x), sd[p + "linear1.weight"], sd[p + "linear1.bias"])
    h = F.linear(F.relu(h), sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    x = x + h
    return x, present


def decoder(sd, args, x: Tensor, attn_mask: Tensor, past: Optional[Tensor]):
    """models/modules/transformer.py:473-488."""
    presents = []
    for l in range(args.num_decoder_layers):
        x, pr = encoder_layer(sd, l, x, attn_mask, args.nhead, None if past is None else past[l])
        if pr is not None:
            presents.append(pr)
    x = layer_norm(sd, "decoder.norm.", x)
    present = torch.stack(presents, dim=0) if presents else None
    return x, present


def dec_forward(sd, args, x_input, x_len: int, x_attention_mask, x_padding_mask,
                y_input, y_len: int, y_attention_mask, y_padding_mask, past):
    """models/ssr.py:214-278 (last_3_tokens=False)."""
    x_attn_mask = F.pad(x_attention_mask, (0, y_len), value=True)
    y_attn_mask = F.pad(y_attention_mask, (x_len, 0), value=False)
    xy_attn_mask = torch.concat([x_attn_mask, y_attn_mask], dim=0)
    bsz, src_len = x_input.shape[0], x_len + y_len
    xy_padding_mask = torch.concat([x_padding_mask, y_padding_mask], dim=1)
    _pad = xy_padding_mask.view(bsz, 1, 1, src_len).expand(-1, args.nhead, -1, -1).reshape(bsz * args.nhead, 1, src_len)
    xy_attn_mask = xy_attn_mask.unsqueeze(0).repeat(_pad.shape[0], 1, 1)
    xy_attn_mask = xy_attn_mask.logical_or(_pad)
    new_attn_mask = torch.zeros(xy_attn_mask.shape, dtype=torch.float32)
    new_attn_mask.masked_fill_(xy_attn_mask, float("-inf"))
    xy_attn_mask = new_attn_mask
    xy_input = torch.cat([x_input, y_input], dim=1)
    if past is None:
        out, _ = decoder(sd, args, xy_input, xy_attn_mask, None)
        return out[:, x_len:], None
    if past.ndim > 3:
        xy_input = xy_input[:, -1:]
        xy_attn_mask = xy_attn_mask[:, -1:]
    out, present = decoder(sd, args, xy_input, xy_attn_mask, past)
    if out.shape[1] > x_len:
        return out[:, x_len:], present
    return out, present


def predict_heads(sd, args, y_out: Tensor) -> Tensor:
    """models/ssr.py:175-179, :688: Linear -> GELU(erf) -> Linear per codebook. [B,1,D]->[B,K,1,card]."""
    outs = []
    for k in range(args.n_codebooks):
        h = F.linear(y_out, sd[f"predict_layer.{k}.0.weight"], sd[f"predict_layer.{k}.0.bias"])
        h = F.gelu(h)
        outs.append(F.linear(h, sd[f"predict_layer.{k}.2.weight"], sd[f"predict_layer.{k}.2.bias"]))
    return torch.stack(outs, dim=1)


# ----------------------------------------------------------------------------- sampling (A11)
def top_k_top_p_filtering(logits, top_k=0, top_p=1.0, filter_value=-float("inf"), min_tokens_to_keep=1):
    """models/ssr.py:26-68 (in-place on `logits`)."""
    if top_k > 0:
        top_k = min(max(top_k, min_tokens_to_keep), logits.size(-1))
        remove = logits < torch.topk(logits, top_k)[0][..., -1, None]
        logits[remove] = filter_value
    if top_p < 1.0:
        sorted_logits, sorted_indices = torch.sort(logits, descending=True)
        cum = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        s_remove = cum > top_p
        if min_tokens_to_keep > 1:
            s_remove[..., :min_tokens_to_keep] = 0
        s_remove[..., 1:] = s_remove[..., :-1].clone()
        s_remove[..., 0] = 0
        remove = s_remove.scatter(1, sorted_indices, s_remove)
        logits[remove] = filter_value
    return logits


def topk_sampling(logits, top_k=10, top_p=1.0, temperature=1.0, noise: Optional[Tensor] = None):
    """models/ssr.py:71-86. `noise` (optional, same shape as logits, Exp(1) draws) replaces the
    generator draw inside torch.multinomial: PyTorch's CPU multinomial for one sample is
    argmax(probs / q), q ~ Exp(1) (aten/src/ATen/native/Distributions.cpp, multinomial fast path) —
    passing the recorded q reproduces it exactly."""
    if temperature != 1.0:
        logits = logits / temperature
    logits = top_k_top_p_filtering(logits, top_k=top_k, top_p=top_p)
    probs = F.softmax(logits, dim=-1)
    if noise is None:
        return torch.multinomial(probs, num_samples=1)
    return torch.argmax(probs / noise, dim=-1, keepdim=True)


# ----------------------------------------------------------------------------- logit state machine (A9, A10)
class SpanState:
    """Host-side counters of one generation span, models/ssr.py:647-652."""

    def __init__(self):
        self.prev_token = None
        self.consec_silence_count = 0
        self.num_gen = 0
        self.num_eog = 0
        self.num_cfg_tag = 1


def step_logits_to_samples(logits: Tensor, st: SpanState, args, y_len_now: int, x_len: int, *, top_k, top_p,
                           temperature, stop_repetition, silence_tokens, cfg_coef, cfg_stride, aug_text,
                           noise: Optional[Tensor] = None, rec: Optional[dict] = None) -> Tensor:
    """models/ssr.py:689-750: one iteration of the body after `predict_layer`. `logits` is
    [B,K,1,card]; returns samples [K,1] int64 and mutates `st`."""
    K = args.n_codebooks
    logits = logits.squeeze()
    if aug_text:
        if st.num_cfg_tag == cfg_stride:
            logits = cfg_coef * logits[0] + (1 - cfg_coef) * logits[1]
            st.num_cfg_tag = 1
        else:
            st.num_cfg_tag += 1
            logits = logits[0]
    assert logits.shape == (K, int(args.audio_vocab_size) + args.n_special + args.max_n_spans), logits.shape
    for jj in range(K):
        logits[jj][args.eos] = -10000.0
        logits[jj][args.sos] = -10000.0
        for m in range(args.mts, args.mts + args.max_n_spans):
            logits[jj][m] = -10000.0
    if st.num_gen < K - 1:
        for jj in range(st.num_gen + 1, K):
            logits[jj][args.empty_token] = 10000.0
    if st.num_eog > 0:
        for jj in range(st.num_eog + 1, K):
            logits[jj][args.eog] = -10000
            logits[jj][args.empty_token] = -10000
        if rec is not None:
            rec.setdefault("edited_logits", []).append(logits.clone())
        samples = topk_sampling(logits, top_k=top_k, top_p=top_p, temperature=temperature, noise=noise)
        if rec is not None:
            rec.setdefault("raw_samples", []).append(samples.clone())
        for jj in range(st.num_eog):
            samples[jj, 0] = args.empty_token
        samples[st.num_eog, 0] = args.eog
        st.num_eog += 1
    else:
        for jj in range(1, K):
            logits[jj][args.eog] = -10000
        if stop_repetition > 0 and st.prev_token in silence_tokens and st.consec_silence_count > stop_repetition:
            n = st.consec_silence_count - (stop_repetition - 1)
            if logits[0, st.prev_token] < 0:
                logits[0, st.prev_token] = logits[0, st.prev_token] * n
            else:
                logits[0, st.prev_token] = logits[0, st.prev_token] / n
        if rec is not None:
            rec.setdefault("edited_logits", []).append(logits.clone())
        samples = topk_sampling(logits, top_k=top_k, top_p=top_p, temperature=temperature, noise=noise)
        if rec is not None:
            rec.setdefault("raw_samples", []).append(samples.clone())
        if samples[0, 0] == args.eog or torch.argmax(logits[0], dim=-1) == args.eog or y_len_now > x_len * 10:
            samples[0, 0] = args.eog
            st.num_eog += 1
        if samples[0, 0] in silence_tokens and samples[0, 0] == st.prev_token:
            st.consec_silence_count += 1
        else:
            st.consec_silence_count = 0
        st.prev_token = samples[0, 0]
    st.num_gen += 1
    return samples


# ----------------------------------------------------------------------------- span re-assembly (A12)
def assemble(y: Tensor, generated: List[List[Tensor]], non_mask_intervals, args):
    """models/ssr.py:776-805 (aug_context=False). y [1,K,T]."""
    flatten_gen = []
    for span_list in generated:
        span = torch.stack(span_list, dim=0).transpose(1, 0)
        unshifted = revert_pattern_sequence(span, args.empty_token)
        assert unshifted.shape[1] == span.shape[1] - args.n_codebooks + 1
        flatten_gen.append(unshifted[:, :-1])
    res, marks, masks, tmp = [], [], [], 0
    for orig, gen in zip(non_mask_intervals, flatten_gen):
        res.append(y[0, :, orig[0]:orig[1]])
        masks.append((tmp, tmp + orig[1] - orig[0]))
        marks += [0] * (orig[1] - orig[0])
        res.append(gen)
        tmp += orig[1] - orig[0] + gen.shape[-1]
        marks += [1] * gen.shape[-1]
    if y.shape[-1] != non_mask_intervals[-1][1] + 1:
        last = non_mask_intervals[-1]
        res.append(y[0, :, last[0]:last[1]])
        masks.append((tmp, tmp + last[1] - last[0]))
        marks += [0] * (last[1] - last[0])
    res = torch.cat(res, dim=1).unsqueeze(0)
    marks = torch.LongTensor(marks).unsqueeze(0)
    return res, marks, masks, list(non_mask_intervals)


# ----------------------------------------------------------------------------- inference()
@torch.no_grad()
def inference(sd: Dict[str, Tensor], args, x: Tensor, y: Tensor, mask_interval: Tensor, *, top_k=-100, top_p=1.0,
              temperature=1.0, stop_repetition=-1, kvcache=1, silence_tokens=(1388, 1898, 131), cfg_coef=1.5,
              cfg_stride=1, aug_text=False, max_steps: Optional[int] = None, noise_fn=None, trace: Optional[dict] = None,
              uncond_x: Optional[Tensor] = None, prompt_x: Optional[Tensor] = None, prompt: Optional[Tensor] = None,
              aug_context=False, cfg_pretrained=False):
    """models/ssr.py:504-812. x [1,L] int64, y [1,T,K] int64, mask_interval [1,M,2]; `prompt_x` [1,Lp] / `prompt` [1,Tp,K]
    are only read when `aug_context` is on (:578-594; inference_scale.py:43-59 never sets it, nor `cfg_pretrained`).

    `max_steps` (test/bench only) stops the while-loop early; `trace` collects per-step logits/timings;
    `noise_fn(step)->Tensor[K,card]` supplies recorded Exp(1) noise; `uncond_x` overrides the CFG random text
    (ssr.py:574) — when None it is drawn from the global torch generator exactly like the reference."""
    K = args.n_codebooks
    n_text_tokens = args.text_vocab_size + 1
    silence_tokens = list(silence_tokens)
    assert cfg_coef >= 1.0
    assert x.ndim == 2 and y.ndim == 3
    y = y.transpose(2, 1)
    assert y.shape[0] == 1 and y.shape[1] == K
    assert mask_interval.shape == torch.Size((1, mask_interval.shape[1], 2))
    # :563-568 — the context is only prepended when the masked spans are short (< 2 s at 50 Hz)
    context_len = sum(int(item[1] - item[0]) for item in mask_interval[0])
    aug_context = bool(aug_context and context_len < 2 * 50)
    out_len = 0
    if aug_context:                                                                   # :578-594
        assert prompt is not None and prompt_x is not None
        prompt = prompt.transpose(2, 1)
        assert prompt.shape[0] == 1 and prompt.shape[1] == K
        out_len = prompt.shape[2]
        y = torch.cat([prompt, y], dim=-1)
        x = torch.cat([prompt_x, x], dim=1)
        mask_interval = mask_interval + out_len                                       # :607-608
    if aug_text:                                                                      # :571-577 / :582-588
        y = y.repeat(2, 1, 1)
        if cfg_pretrained:
            uncond_x = torch.full((1, x.shape[1]), args.text_vocab_size - 1, dtype=torch.long)
        elif uncond_x is None:
            uncond_x = torch.randint(0, n_text_tokens, (1, x.shape[1]))               # :574 / :585
        x = torch.cat([x, uncond_x], dim=0)
    B = x.shape[0]
    x_len = x.shape[-1]
    # table length is irrelevant to the values (embedding.py:67-92 regrows on demand); size it for the cap :739
    pe = sine_pe(max(4000, 11 * x_len + y.shape[2] + 64), args.d_model)
    x_attention_mask = torch.triu(torch.ones(x_len, x_len), diagonal=1).bool()
    x_input = pos_embed(F.embedding(x, sd["text_embedding.word_embeddings.weight"]),
                        sd["text_positional_embedding.alpha"], pe)
    cated_y, mask_position, num_task, non_mask_intervals, _ = build_layout(y[0], mask_interval[0], args)
    cated_y = cated_y.unsqueeze(0).permute(1, 2, 0)
    if aug_text:
        cated_y = cated_y.repeat(1, 1, 2)
    embedded_y = embed_y(sd, cated_y, K)
    x_padding_mask = torch.full((B, x_len), False)
    if aug_text and cfg_pretrained:
        x_padding_mask[1:, 1:] = True                                                 # :631-634
    past = torch.ones([args.num_decoder_layers, 2, B], dtype=torch.float32) if kvcache else None
    emb_inds = list(range(args.mts, args.mts + args.max_n_spans))
    generated = []
    total_steps = 0
    for idx in range(num_task):
        cur = []
        st = SpanState()
        mts = torch.full((K, 1), emb_inds[idx], dtype=torch.long)
        mts_emb = torch.stack([F.embedding(mts[k], sd[f"audio_embedding.{k}.word_embeddings.weight"]) for k in range(K)], dim=0)
        mts_emb = mts_emb.sum(dim=0, keepdim=True)
        if aug_text:
            mts_emb = mts_emb.repeat(2, 1, 1)
        embedded_y = torch.cat([embedded_y, mts_emb], dim=1)
        while True:
            y_input = pos_embed(embedded_y, sd["audio_positional_embedding.alpha"], pe)
            T = y_input.shape[1]
            y_attention_mask = torch.triu(torch.ones(T, T), diagonal=1).bool()
            y_padding_mask = torch.full((B, T), False)
            y_out, present = dec_forward(sd, args, x_input, x_len, x_attention_mask, x_padding_mask,
                                         y_input, T, y_attention_mask, y_padding_mask, past)
            if past is not None:
                past = torch.cat([past, present.to(past.dtype)], dim=-2) if past.ndim > 3 else present.to(past.dtype)
            y_out = y_out[:, -1:]
            logits = predict_heads(sd, args, y_out)
            if trace is not None:
                trace.setdefault("logits", []).append(logits.squeeze(2).clone())
            noise = noise_fn(total_steps) if noise_fn is not None else None
            samples = step_logits_to_samples(
                logits, st, args, T, x_len, top_k=top_k, top_p=top_p, temperature=temperature,
                stop_repetition=stop_repetition, silence_tokens=silence_tokens, cfg_coef=cfg_coef,
                cfg_stride=cfg_stride, aug_text=aug_text, noise=noise, rec=trace)
            cur.append(samples.squeeze(-1))
            total_steps += 1
            if trace is not None:
                trace.setdefault("samples", []).append(samples.squeeze(-1).clone())
            if st.num_eog == K:
                break
            if max_steps is not None and total_steps >= max_steps:
                if trace is not None:
                    trace["truncated"] = True
                return None
            s_emb = torch.stack([F.embedding(samples[k], sd[f"audio_embedding.{k}.word_embeddings.weight"]) for k in range(K)], dim=0)
            s_emb = s_emb.sum(dim=0, keepdim=True)
            if aug_text:
                s_emb = s_emb.repeat(2, 1, 1)
            embedded_y = torch.cat([embedded_y, s_emb], dim=1)
        generated.append(cur)
    res, marks, masks, nmi_out = assemble(y, generated, non_mask_intervals, args)
    if aug_context:                                                                   # :806-810
        res = res[:, :, out_len:]
        marks = marks[:, out_len:]
        masks = [(a - out_len, b - out_len) for a, b in masks]
        nmi_out = [(a - out_len, b - out_len) for a, b in nmi_out]
    return res, marks, masks, nmi_out
