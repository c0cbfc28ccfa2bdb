"""Sample-rate conversion for the front-end (`convert_audio`, reference `data/tokenizer.py:96` and `data/encode.py:84-85`:
`torchaudio.transforms.Resample(sr, target_sr)` with its defaults — windowed-sinc interpolation, Hann window,
`lowpass_filter_width=6`, `rolloff=0.99`).

torchaudio is not part of this image, so the published algorithm is implemented here: a polyphase FIR whose `new/gcd` phases
are the rows of one filter bank, applied with stride `orig/gcd` — which is exactly a C_in = 1 strided convolution, the kernel the
codec's first layer already uses (`ssrhip_conv_cin1`, include/ssrhip.h): phases = output channels, taps = kernel width, and the
time-major output [frames][phases] IS the resampled signal in order. The filter bank is computed once per rate pair on the host in
float64 and rounded to float32, as torchaudio does for `dtype=None`. There is no CPU path (like the rest of the package).
"""
from __future__ import annotations

import ctypes as C
import math
from functools import lru_cache
from typing import Tuple

import numpy as np
import torch

from .. import _lib

LOWPASS_FILTER_WIDTH = 6
ROLLOFF = 0.99


@lru_cache(maxsize=32)
def sinc_filter_bank(orig_freq: int, new_freq: int, lowpass_filter_width: int = LOWPASS_FILTER_WIDTH, rolloff: float = ROLLOFF) -> Tuple[np.ndarray, int, int, int]:
    """-> (bank float32 [new, 2*width + orig], width, orig, new) with orig / new already divided by their gcd.

    bank[p][j] = scale * sinc(t) * cos^2(pi t / (2 * lowpass_filter_width)),  t = clamp((-p / new + (j - width) / orig) * base, +-lowpass_filter_width),
    base = min(orig, new) * rolloff, scale = base / orig, sinc(t) = sin(pi t) / (pi t)."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        sinc = np.where(t == 0, 1.0, np.sin(t) / t)
    bank = (sinc * window * (base / orig)).astype(np.float32)
    return bank, width, orig, new


_device_banks = {}


def resample(wav: torch.Tensor, sr: int, target_sr: int) -> torch.Tensor:
    """wav [..., n] at `sr` Hz -> [..., ceil(n * target_sr / sr)] at `target_sr` Hz, float32, on the device `wav` came from."""
    if int(sr) == int(target_sr):
        return wav
    if not torch.cuda.is_available():
        raise RuntimeError("ssr_speech_amd.data.resample needs the ROCm GPU (libssrhip.so); there is no CPU path in this package")
    bank, width, orig, new = sinc_filter_bank(int(sr), int(target_sr))
    dev = torch.device("cuda", torch.cuda.current_device())
    key = (int(sr), int(target_sr), dev.index)
    if key not in _device_banks:
        _device_banks[key] = (torch.from_numpy(bank).to(dev).contiguous(), torch.zeros(new, dtype=torch.float32, device=dev))
    dbank, zero_bias = _device_banks[key]
    lead, n = wav.shape[:-1], wav.shape[-1]
    target_len = int(math.ceil(new * n / orig))
    if wav.numel() == 0:
        return torch.zeros(*lead, target_len, dtype=torch.float32, device=wav.device)
    x = wav.reshape(-1, n).to(dev, torch.float32)
    B = x.shape[0]
    padded = torch.nn.functional.pad(x, (width, width + orig)).contiguous()            # [B, n + 2*width + orig]
    frames = n // orig + 1                                                          # conv1d(stride = orig) output length
    out = torch.empty(B, frames, new, dtype=torch.float32, device=dev)
    for b0 in range(0, B, 65535):                                                   # grid.y limit of the kernel
        b1 = min(B, b0 + 65535)
        _lib.check(_lib.lib().ssrhip_conv_cin1(padded[b0:b1].data_ptr(), dbank.data_ptr(), zero_bias.data_ptr(), out[b0:b1].data_ptr(), b1 - b0, frames,
                                               bank.shape[1], orig, new, padded.shape[1], frames * new, _lib.stream_ptr()), "ssrhip_conv_cin1 (resample)")
    y = out.reshape(B, frames * new)[:, :target_len]
    return y.reshape(*lead, target_len).to(wav.device)
