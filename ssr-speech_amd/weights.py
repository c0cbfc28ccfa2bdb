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


# ---------------------------------------------------------------------------------------------
# Codec (watermarked Encodec: SEANet encoder/decoder + LSTM + RVQ + watermark decoder) parameter inventory
# ---------------------------------------------------------------------------------------------
from dataclasses import dataclass, field


@dataclass
class CodecConfig:
    """Hyper-parameters the reference keeps in the checkpoint's `xp.cfg` (`audiocraft/config/model/encodec/
    default.yaml`, `encodec_large_nq4_s320.yaml`, SURVEY §8 header)."""
    channels: int = 1
    dimension: int = 128
    n_filters: int = 64
    n_residual_layers: int = 1
    ratios: tuple = (8, 5, 4, 2)
    kernel_size: int = 7
    residual_kernel_size: int = 3
    last_kernel_size: int = 7
    compress: int = 2
    lstm: int = 2
    pad_mode: str = "constant"
    n_q: int = 4
    bins: int = 2048
    sample_rate: int = 16000

    @property
    def hop(self) -> int:
        h = 1
        for r in self.ratios:
            h *= r
        return h

    @property
    def frame_rate(self) -> int:
        return self.sample_rate // self.hop


def codec_config_full() -> CodecConfig:
    return CodecConfig()


def codec_config_tiny() -> CodecConfig:
    return CodecConfig(dimension=64, n_filters=8, ratios=(4, 3, 2, 2), bins=64)   # hop 48; the wm decoder needs 4 ratios; D/16 = 4 label channels


def _conv_specs(sp, pfx, cout, cin, k, wn=True):
    if wn:
        sp[pfx + "conv.conv.bias"] = ((cout,), f"lin:{cin * k}")
        sp[pfx + "conv.conv.weight_g"] = ((cout, 1, 1), "wn_g")
        sp[pfx + "conv.conv.weight_v"] = ((cout, cin, k), f"lin:{cin * k}")
    else:
        sp[pfx + "conv.conv.weight"] = ((cout, cin, k), f"lin:{cin * k}")
        sp[pfx + "conv.conv.bias"] = ((cout,), f"lin:{cin * k}")


def _lstm_specs(sp, pfx, dim, layers):
    for l in range(layers):
        sp[pfx + f"lstm.weight_ih_l{l}"] = ((4 * dim, dim), f"lin:{dim}")
        sp[pfx + f"lstm.weight_hh_l{l}"] = ((4 * dim, dim), f"lin:{dim}")
        sp[pfx + f"lstm.bias_ih_l{l}"] = ((4 * dim,), f"lin:{dim}")
        sp[pfx + f"lstm.bias_hh_l{l}"] = ((4 * dim,), f"lin:{dim}")


def _encoder_specs(sp, pfx, c: CodecConfig):
    """SEANetEncoder.model layout (`audiocraft/modules/seanet.py:113-150`)."""
    assert c.n_residual_layers == 1, "only n_residual_layers=1 (dilation 1) is supported"
    i, mult = 0, 1
    _conv_specs(sp, f"{pfx}model.{i}.", c.n_filters, c.channels, c.kernel_size)
    i += 1
    for r in reversed(c.ratios):
        dim = mult * c.n_filters
        _conv_specs(sp, f"{pfx}model.{i}.block.1.", dim // c.compress, dim, c.residual_kernel_size)
        _conv_specs(sp, f"{pfx}model.{i}.block.3.", dim, dim // c.compress, 1)
        i += 2   # resblock, ELU
        _conv_specs(sp, f"{pfx}model.{i}.", dim * 2, dim, 2 * r)
        i += 1
        mult *= 2
    if c.lstm:
        _lstm_specs(sp, f"{pfx}model.{i}.", mult * c.n_filters, c.lstm)
        i += 1
    i += 1       # ELU
    _conv_specs(sp, f"{pfx}model.{i}.", c.dimension, mult * c.n_filters, c.last_kernel_size)


def _decoder_specs(sp, pfx, c: CodecConfig):
    """SEANetDecoder.model layout (`seanet.py:209-254`)."""
    i, mult = 0, 2 ** len(c.ratios)
    _conv_specs(sp, f"{pfx}model.{i}.", mult * c.n_filters, c.dimension, c.kernel_size)
    i += 1
    if c.lstm:
        _lstm_specs(sp, f"{pfx}model.{i}.", mult * c.n_filters, c.lstm)
        i += 1
    for r in c.ratios:
        dim = mult * c.n_filters
        i += 1   # ELU
        sp[f"{pfx}model.{i}.convtr.convtr.bias"] = ((dim // 2,), f"lin:{dim * 2 * r}")
        sp[f"{pfx}model.{i}.convtr.convtr.weight_g"] = ((dim, 1, 1), "wn_g")           # norm over dim 0 = IN channels
        sp[f"{pfx}model.{i}.convtr.convtr.weight_v"] = ((dim, dim // 2, 2 * r), f"lin:{dim * 2}")
        i += 1
        _conv_specs(sp, f"{pfx}model.{i}.block.1.", dim // 2 // c.compress, dim // 2, c.residual_kernel_size)
        _conv_specs(sp, f"{pfx}model.{i}.block.3.", dim // 2, dim // 2 // c.compress, 1)
        i += 1
        mult //= 2
    i += 1       # ELU
    _conv_specs(sp, f"{pfx}model.{i}.", c.channels, c.n_filters, c.last_kernel_size)


def codec_param_specs(c: CodecConfig) -> "OrderedDict[str, tuple]":
    """name -> (shape, kind) for `WMEncodecModel.state_dict()` (encoder, decoder, wmdecoder, quantizer)."""
    sp: "OrderedDict[str, tuple]" = OrderedDict()
    _encoder_specs(sp, "encoder.", c)
    _decoder_specs(sp, "decoder.", c)
    _decoder_specs(sp, "wmdecoder.", c)
    _encoder_specs(sp, "wmdecoder.skip_encoder.", c)
    sp["wmdecoder.wm_embed.weight"] = ((2, c.dimension // 16), "emb")
    _encoder_specs(sp, "wmdecoder.wm_encoder.", c)
    e = c.dimension // 16
    mult = 2 ** len(c.ratios)
    _conv_specs(sp, "wmdecoder.wm_proj0.1.", c.dimension, c.dimension + e, 1, wn=False)
    for j in (1, 2, 3):
        mult //= 2
        _conv_specs(sp, f"wmdecoder.wm_proj{j}.1.", mult * c.n_filters, mult * c.n_filters + e, 1, wn=False)
    _conv_specs(sp, "wmdecoder.wm_predictor.1.", 2, c.dimension, 1, wn=False)
    for q in range(c.n_q):
        p = f"quantizer.vq.layers.{q}._codebook."
        sp[p + "inited"] = ((1,), "one")
        sp[p + "cluster_size"] = ((c.bins,), "zero")
        sp[p + "embed"] = ((c.bins, c.dimension), f"scale:{1.0 / (q + 1):.6g}")
        sp[p + "embed_avg"] = ((c.bins, c.dimension), f"scale:{1.0 / (q + 1):.6g}")
    return sp


def codec_state_dict(c: CodecConfig, seed: int = 0, device="cpu") -> "OrderedDict[str, torch.Tensor]":
    sd = OrderedDict()
    for name, (shape, kind) in codec_param_specs(c).items():
        if kind == "wn_g":
            sd[name] = 1.0 + 0.25 * uniform_pm1(name, shape[0], seed, device).view(*shape)
        else:
            sd[name] = make_tensor(name, shape, kind, seed, device)
    return sd
