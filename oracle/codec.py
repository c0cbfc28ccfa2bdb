"""ORACLE (test infrastructure, not product code) — CPU restatement of the reference's watermarked-Encodec
codec (SEANet encoder/decoder with weight-norm convs, ELU, 2-layer LSTM, RVQ, watermark decoder) in plain
PyTorch fp32 ops, driven by a state-dict with the reference's key names.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this file.
Pinned against the reference: `oracle/make_golden_codec.py` imports the reference's `modules/{conv,lstm,seanet}.py`,
`quantization/`, `models/wmencodec.py` (build container only) and checks/commits golden vectors.

Each function cites the reference lines it restates (paths relative to /root/reference/audiocraft/audiocraft).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------- conv building blocks
def wn_weight(sd, pfx: str, which: str = "conv") -> Tensor:
    """Old-style torch.nn.utils.weight_norm (modules/conv.py:21-30): w = g * v / ||v||, the norm taken over
    every dim except 0 (Conv1d: per OUT channel; ConvTranspose1d weight is [C_in, C_out, k] so per IN channel)."""
    k = f"{pfx}{which}.{which}."
    if k + "weight" in sd:
        return sd[k + "weight"]
    return torch._weight_norm(sd[k + "weight_v"], sd[k + "weight_g"], 0)


def extra_padding(length: int, kernel_size: int, stride: int, padding_total: int) -> int:
    """modules/conv.py:47-53."""
    n_frames = (length - kernel_size + padding_total) / stride + 1
    ideal_length = (math.ceil(n_frames) - 1) * stride + (kernel_size - padding_total)
    return ideal_length - length


def pad1d(x: Tensor, paddings: Tuple[int, int], mode: str) -> Tensor:
    """modules/conv.py:71-88 (reflect on short inputs inserts extra zero padding first)."""
    length = x.shape[-1]
    pl, pr = paddings
    if mode == "reflect":
        max_pad = max(pl, pr)
        extra = 0
        if length <= max_pad:
            extra = max_pad - length + 1
            x = F.pad(x, (0, extra))
        padded = F.pad(x, paddings, mode)
        return padded[..., : padded.shape[-1] - extra]
    return F.pad(x, paddings, mode, 0.0)


def sconv1d(sd, pfx: str, x: Tensor, stride: int = 1, pad_mode: str = "constant") -> Tensor:
    """StreamableConv1d.forward, non-causal, dilation 1 (modules/conv.py:185-201)."""
    w = wn_weight(sd, pfx, "conv")
    b = sd[pfx + "conv.conv.bias"]
    k = w.shape[-1]
    padding_total = k - stride
    extra = extra_padding(x.shape[-1], k, stride, padding_total)
    pr = padding_total // 2
    pl = padding_total - pr
    x = pad1d(x, (pl, pr + extra), pad_mode)
    return F.conv1d(x, w, b, stride=stride)


def sconvtr1d(sd, pfx: str, x: Tensor, stride: int) -> Tensor:
    """StreamableConvTranspose1d.forward, non-causal (modules/conv.py:221-243)."""
    w = wn_weight(sd, pfx, "convtr")
    b = sd[pfx + "convtr.convtr.bias"]
    k = w.shape[-1]
    padding_total = k - stride
    y = F.conv_transpose1d(x, w, b, stride=stride)
    pr = padding_total // 2
    pl = padding_total - pr
    return y[..., pl: y.shape[-1] - pr]


def resblock(sd, pfx: str, x: Tensor, pad_mode: str) -> Tensor:
    """SEANetResnetBlock with true_skip (modules/seanet.py:16-60): x + conv1(ELU(conv3(ELU(x))))."""
    h = sconv1d(sd, pfx + "block.1.", F.elu(x), 1, pad_mode)
    h = sconv1d(sd, pfx + "block.3.", F.elu(h), 1, pad_mode)
    return x + h


def lstm(sd, pfx: str, x: Tensor, layers: int) -> Tensor:
    """StreamableLSTM (modules/lstm.py:10-25): y = LSTM(x) + x on [T,B,C]; gate order i,f,g,o (torch.nn.LSTM)."""
    xs = x.permute(2, 0, 1)
    C = xs.shape[-1]
    m = torch.nn.LSTM(C, C, layers)           # the reference's own ATen op (modules/lstm.py:17), fed the same tensors
    m.load_state_dict({k[len(pfx) + 5:]: v for k, v in sd.items() if k.startswith(pfx + "lstm.")})
    with torch.no_grad():
        y, _ = m(xs)
    return (y + xs).permute(1, 2, 0)


# ----------------------------------------------------------------------------- SEANet
def encoder_layout(cfg) -> List[tuple]:
    """The nn.Sequential of SEANetEncoder as (index, kind, stride) (modules/seanet.py:113-150)."""
    out, i = [(0, "conv", 1)], 1
    for r in reversed(cfg.ratios):
        out += [(i, "res", 1), (i + 1, "elu", 1), (i + 2, "conv", r)]
        i += 3
    if cfg.lstm:
        out.append((i, "lstm", 1))
        i += 1
    out += [(i, "elu", 1), (i + 1, "conv", 1)]
    return out


def decoder_layout(cfg) -> List[tuple]:
    """SEANetDecoder (modules/seanet.py:209-254)."""
    out, i = [(0, "conv", 1)], 1
    if cfg.lstm:
        out.append((i, "lstm", 1))
        i += 1
    for r in cfg.ratios:
        out += [(i, "elu", 1), (i + 1, "convtr", r), (i + 2, "res", 1)]
        i += 3
    out += [(i, "elu", 1), (i + 1, "conv", 1)]
    return out


def run_layers(sd, pfx: str, layout, x: Tensor, cfg, lo: int = 0, hi: Optional[int] = None) -> Tensor:
    """Apply model[lo:hi] of a SEANet Sequential."""
    for (i, kind, s) in layout:
        if i < lo or (hi is not None and i >= hi):
            continue
        p = f"{pfx}model.{i}."
        if kind == "conv":
            x = sconv1d(sd, p, x, s, cfg.pad_mode)
        elif kind == "convtr":
            x = sconvtr1d(sd, p, x, s)
        elif kind == "res":
            x = resblock(sd, p, x, cfg.pad_mode)
        elif kind == "elu":
            x = F.elu(x)
        elif kind == "lstm":
            x = lstm(sd, p, x, cfg.lstm)
    return x


def seanet_encoder(sd, pfx: str, x: Tensor, cfg) -> Tensor:
    return run_layers(sd, pfx, encoder_layout(cfg), x, cfg)


def seanet_decoder(sd, pfx: str, z: Tensor, cfg) -> Tensor:
    return run_layers(sd, pfx, decoder_layout(cfg), z, cfg)


# ----------------------------------------------------------------------------- RVQ
def rvq_encode(sd, emb: Tensor, cfg) -> Tensor:
    """ResidualVectorQuantization.encode + EuclideanCodebook.quantize (quantization/core_vq.py:164-172, 382-392;
    vq.py:87-95). emb [B,D,T] -> codes [B,n_q,T] int64."""
    B, D, T = emb.shape
    residual = emb
    all_idx = []
    for q in range(cfg.n_q):
        E = sd[f"quantizer.vq.layers.{q}._codebook.embed"]
        x = residual.permute(0, 2, 1).reshape(-1, D)                  # "b d n -> b n d" then "... d -> (...) d"
        embed = E.t()
        dist = -(x.pow(2).sum(1, keepdim=True) - 2 * x @ embed + embed.pow(2).sum(0, keepdim=True))
        idx = dist.max(dim=-1).indices.view(B, T)
        quant = F.embedding(idx, E).permute(0, 2, 1)
        residual = residual - quant
        all_idx.append(idx)
    return torch.stack(all_idx).transpose(0, 1)


def rvq_decode(sd, codes: Tensor, cfg) -> Tensor:
    """ResidualVectorQuantization.decode (core_vq.py:394-400; vq.py:97-103). codes [B,n_q,T] -> [B,D,T].
    F.embedding raises on out-of-range ids exactly like the reference."""
    out = torch.tensor(0.0)
    for q in range(codes.shape[1]):
        E = sd[f"quantizer.vq.layers.{q}._codebook.embed"]
        out = out + F.embedding(codes[:, q], E).permute(0, 2, 1)
    return out


# ----------------------------------------------------------------------------- WMEncodecModel
def encode(sd, x: Tensor, cfg):
    """WMEncodecModel.encode (models/wmencodec.py:324-339), renormalize=False -> scale None."""
    assert x.dim() == 3
    emb = seanet_encoder(sd, "encoder.", x, cfg)
    codes = rvq_encode(sd, emb, cfg)
    return codes, None, emb


def decode(sd, codes: Tensor, cfg) -> Tensor:
    """WMEncodecModel.decode (wmencodec.py:341-356)."""
    return seanet_decoder(sd, "decoder.", rvq_decode(sd, codes, cfg), cfg)


def wm_embed(sd, labels: Tensor) -> Tensor:
    """nn.Embedding(2, D/16, max_norm=True) (modules/seanet.py:503): rows with L2 norm > 1 are renormalised to
    norm 1 (scale = 1/(norm+1e-7)) at lookup time."""
    w = sd["wmdecoder.wm_embed.weight"]
    norm = w.norm(p=2, dim=1, keepdim=True)
    w = torch.where(norm > 1.0, w * (1.0 / (norm + 1e-7)), w)
    return F.embedding(labels, w)


def wmdecode(sd, codes: Tensor, labels: Tensor, wav: Tensor, cfg):
    """WMEncodecModel.wmdecode -> WMSEANetDecoder.forward (wmencodec.py:358-375; modules/seanet.py:555-600).
    Returns (out wav, mark [B,T',2])."""
    x = rvq_decode(sd, codes, cfg)
    enc = encoder_layout(cfg)
    dec = decoder_layout(cfg)
    sp = "wmdecoder.skip_encoder."
    ratios = list(cfg.ratios)
    assert len(ratios) == 4, "the watermark decoder's slicing (seanet.py:560-591) is written for 4 ratios"
    z = run_layers(sd, sp, enc, wav, cfg, 0, 2)
    z = run_layers(sd, sp, enc, z, cfg, 2, 5)
    skips, labs = [z], [torch.repeat_interleave(labels, ratios[0] * ratios[1] * ratios[2], dim=-1)]
    z = run_layers(sd, sp, enc, z, cfg, 5, 8)
    skips.append(z)
    labs.append(torch.repeat_interleave(labels, ratios[0] * ratios[1], dim=-1))
    z = run_layers(sd, sp, enc, z, cfg, 8, 11)
    skips.append(z)
    labs.append(torch.repeat_interleave(labels, ratios[0], dim=-1))
    z = run_layers(sd, sp, enc, z, cfg, 11, None)
    skips.append(z)
    labs.append(labels)
    cuts = [(0, 4), (4, 7), (7, 10), (10, None)]
    for j, (lo, hi) in enumerate(cuts):
        cat = torch.cat([skips.pop(), wm_embed(sd, labs.pop()).transpose(2, 1)], dim=1)
        out = sconv1d(sd, f"wmdecoder.wm_proj{j}.1.", F.elu(cat), 1, cfg.pad_mode) + x
        x = run_layers(sd, "wmdecoder.", dec, out, cfg, lo, hi)
    m = seanet_encoder(sd, "wmdecoder.wm_encoder.", x, cfg)
    m = sconv1d(sd, "wmdecoder.wm_predictor.1.", F.elu(m), 1, cfg.pad_mode)
    return x, m.transpose(2, 1)


def detect_watermark(sd, x: Tensor, cfg) -> Tensor:
    """WMEncodecModel.detect_watermark (wmencodec.py:377-382) — including its argmax over TIME (dim=-1 of [B,2,T'])."""
    m = seanet_encoder(sd, "wmdecoder.wm_encoder.", x, cfg)
    m = sconv1d(sd, "wmdecoder.wm_predictor.1.", F.elu(m), 1, cfg.pad_mode).squeeze(-1)
    return torch.argmax(m, dim=-1)
