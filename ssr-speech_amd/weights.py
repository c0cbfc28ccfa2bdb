"""Deterministic, re-creatable synthetic weights (no checkpoint ships offline: the reference's
`pretrained_models/test` is a 1-byte placeholder).

Every tensor element is a pure function of (tensor name, flat index, seed) through a 32-bit
integer hash, converted to fp32 with exactly one rounding, so the same values are produced on the
CPU (oracle / cpu_baseline) and on the GPU box without shipping 3.3 GB of fixtures.

Parameter names and shapes are the reference's `state_dict` keys:
  LM     `models/ssr.py:132-179`  (SSR_Speech.__init__)
  codec  `audiocraft/audiocraft/modules/seanet.py:113-150,209-254,503-553`, `quantization/core_vq.py:124-127`
"""
from __future__ import annotations

import math
import zlib
from argparse import Namespace
from collections import OrderedDict

import torch

_M32 = 0xFFFFFFFF


def _hash_u32(idx: torch.Tensor, seed: int) -> torch.Tensor:
    """lowbias32-style avalanche hash on int64 tensors holding u32 values (wraps mod 2^32)."""
    x = (idx + seed) & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


def uniform_pm1(name: str, numel: int, seed: int, device="cpu", chunk: int = 1 << 24) -> torch.Tensor:
    """fp32 values on the grid k*2^-23 in [-1, 1): exact in fp32 on any IEEE device."""
    tseed = (zlib.crc32(name.encode()) * 0x9E3779B1 + seed * 0x85EBCA6B + 0x1234567) & _M32
    out = torch.empty(numel, dtype=torch.float32, device=device)
    for s in range(0, numel, chunk):
        e = min(numel, s + chunk)
        idx = torch.arange(s, e, dtype=torch.int64, device=device)
        h = _hash_u32(idx * 0x9E3779B1, tseed)
        out[s:e] = (h >> 8).to(torch.float32) * (2.0 ** -23) - 1.0
    return out


def make_tensor(name: str, shape, kind: str, seed: int, device="cpu") -> torch.Tensor:
    """kind: 'lin:<fan_in>' U(-1/sqrt(fan_in), ..) | 'emb' U(-1,1) | 'ln_w' 1+0.1u | 'ln_b' 0.1u |
    'one' ones | 'zero' zeros | 'scale:<s>' U(-s,s)"""
    n = 1
    for d in shape:
        n *= int(d)
    if kind == "one":
        return torch.ones(shape, dtype=torch.float32, device=device)
    if kind == "zero":
        return torch.zeros(shape, dtype=torch.float32, device=device)
    u = uniform_pm1(name, n, seed, device).view(*shape)
    if kind.startswith("lin:"):
        b = torch.tensor(1.0 / math.sqrt(float(kind[4:])), dtype=torch.float32).item()
        return u * b
    if kind.startswith("scale:"):
        return u * torch.tensor(float(kind[6:]), dtype=torch.float32).item()
    if kind == "emb":
        return u
    if kind == "ln_w":
        return 1.0 + 0.1 * u
    if kind == "ln_b":
        return 0.1 * u
    raise ValueError(kind)


# ---------------------------------------------------------------------------------------------
# LM (models/ssr.py) parameter inventory
# ---------------------------------------------------------------------------------------------

def lm_args_830m() -> Namespace:
    """English 830M constants: reference `z_scripts/e830M.sh:37-65`, `config.py:57-90`."""
    return Namespace(
        n_special=5, eos=2051, sos=2052, mts=2053, eog=2049, empty_token=2048, audio_pad_token=2050,
        audio_vocab_size=2048, text_vocab_size=100, text_pad_token=100, max_n_spans=3, n_codebooks=4,
        d_model=2048, audio_embedding_dim=2048, nhead=16, num_decoder_layers=16,
        text_embedding_dropout=0.0, audio_embedding_dropout=0.0, text_positional_embedding_dropout=0.0,
        audio_positional_embedding_dropout=0.0, trm_dropout=0.0,
        shuffle_mask_embedding=0, predict_mask_token=1, predict_all=0, codebook_weight=None,
    )


def lm_args_tiny(d_model=128, nhead=2, layers=2, vocab=64) -> Namespace:
    """A small config with the same special-token structure, for fast parity tests."""
    a = lm_args_830m()
    a.audio_vocab_size = vocab
    a.empty_token, a.eog, a.audio_pad_token = vocab, vocab + 1, vocab + 2
    a.eos, a.sos, a.mts = vocab + 3, vocab + 4, vocab + 5
    a.d_model = a.audio_embedding_dim = d_model
    a.nhead, a.num_decoder_layers = nhead, layers
    a.text_vocab_size, a.text_pad_token = 30, 30
    return a


def lm_param_specs(args) -> "OrderedDict[str, tuple]":
    """name -> (shape, kind), in the reference's state_dict order (`models/ssr.py:132-179`)."""
    D = int(args.d_model)
    V = int(args.audio_vocab_size) if not isinstance(args.audio_vocab_size, str) else int(eval(args.audio_vocab_size))
    card = V + int(args.n_special) + int(args.max_n_spans)
    K = int(args.n_codebooks)
    F = 4 * D
    sp: "OrderedDict[str, tuple]" = OrderedDict()
    sp["text_embedding.word_embeddings.weight"] = ((int(args.text_vocab_size) + 1, D), "emb")
    for k in range(K):
        sp[f"audio_embedding.{k}.word_embeddings.weight"] = ((card, D), "emb")
    sp["text_positional_embedding.alpha"] = ((1,), "one")
    sp["audio_positional_embedding.alpha"] = ((1,), "one")
    for l in range(int(args.num_decoder_layers)):
        p = f"decoder.layers.{l}."
        sp[p + "self_attn.in_proj_weight"] = ((3 * D, D), f"lin:{D}")
        sp[p + "self_attn.in_proj_bias"] = ((3 * D,), f"lin:{D}")
        sp[p + "self_attn.out_proj.weight"] = ((D, D), f"lin:{D}")
        sp[p + "self_attn.out_proj.bias"] = ((D,), f"lin:{D}")
        sp[p + "linear1.weight"] = ((F, D), f"lin:{D}")
        sp[p + "linear1.bias"] = ((F,), f"lin:{D}")
        sp[p + "linear2.weight"] = ((D, F), f"lin:{F}")
        sp[p + "linear2.bias"] = ((D,), f"lin:{F}")
        sp[p + "norm1.weight"] = ((D,), "ln_w")
        sp[p + "norm1.bias"] = ((D,), "ln_b")
        sp[p + "norm2.weight"] = ((D,), "ln_w")
        sp[p + "norm2.bias"] = ((D,), "ln_b")
    sp["decoder.norm.weight"] = ((D,), "ln_w")
    sp["decoder.norm.bias"] = ((D,), "ln_b")
    Hh = V // 2
    for k in range(K):
        sp[f"predict_layer.{k}.0.weight"] = ((Hh, D), f"lin:{D}")
        sp[f"predict_layer.{k}.0.bias"] = ((Hh,), f"lin:{D}")
        # the output layer is scaled up so synthetic logits have O(1) spread (top-1 margins well
        # above fp32 summation-order noise; SURVEY §7 "hard parts")
        sp[f"predict_layer.{k}.2.weight"] = ((card, Hh), f"scale:{8.0 / math.sqrt(Hh):.9g}")
        sp[f"predict_layer.{k}.2.bias"] = ((card,), "scale:0.5")
    return sp


def lm_state_dict(args, seed: int = 0, device="cpu") -> "OrderedDict[str, torch.Tensor]":
    sd = OrderedDict()
    for name, (shape, kind) in lm_param_specs(args).items():
        sd[name] = make_tensor(name, shape, kind, seed, device)
    return sd


def lm_num_params(args) -> int:
    n = 0
    for shape, _ in lm_param_specs(args).values():
        m = 1
        for d in shape:
            m *= d
        n += m
    return n
