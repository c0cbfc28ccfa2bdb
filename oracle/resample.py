"""ORACLE (test infrastructure only; never imported by the product) — CPU restatement of `torchaudio.transforms.Resample(orig, new)`
with its defaults (`resampling_method="sinc_interp_hann"`, `lowpass_filter_width=6`, `rolloff=0.99`, `dtype=None`), which the
reference calls in `data/tokenizer.py:96` and `data/encode.py:84-85`.

PARITY UNPINNED: torchaudio is a dependency of the reference that is absent from /root/reference and from this image (the
reference pins no version; `torchaudio.functional.resample` has had this form since 0.9: `_get_sinc_resample_kernel` +
`_apply_sinc_resample_kernel`), so this file restates the published algorithm and is checked by properties only
(tests/test_resample.py: unit DC gain, tone preservation below the cut-off, rejection above it, output length, linearity) —
there is no torchaudio output to compare with.

Algorithm (per channel): with orig, new reduced by their gcd, base = min(orig, new) * rolloff, width = ceil(6 * orig / base):
  kernel[p][j] = (base / orig) * sinc(t) * cos^2(pi t / 12),  t = clamp((-p / new + (j - width) / orig) * base, -6, 6),  p < new, j < 2 width + orig
  y[f * new + p] = sum_j kernel[p][j] * xpad[f * orig + j],   xpad = x with `width` zeros in front and `width + orig` behind
  output = the first ceil(new * n / orig) samples.
The kernel is evaluated in float64 and rounded to float32 (torchaudio's `dtype=None` path); the convolution here accumulates in float64.
"""
import math

import numpy as np


def sinc_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base))
    kern = np.zeros((new, 2 * width + orig), dtype=np.float64)
    for p in range(new):
        for j in range(2 * width + orig):
            t = (-p / new + (j - width) / orig) * base
            t = max(-lowpass_filter_width, min(lowpass_filter_width, t))
            win = math.cos(t * math.pi / lowpass_filter_width / 2) ** 2
            tp = t * math.pi
            kern[p, j] = (1.0 if tp == 0 else math.sin(tp) / tp) * win * (base / orig)
    return kern.astype(np.float32), width, orig, new


def resample(x: np.ndarray, orig_freq: int, new_freq: int) -> np.ndarray:
    """x [..., n] -> [..., ceil(n * new / orig)] float32."""
    x = np.asarray(x, dtype=np.float32)
    if int(orig_freq) == int(new_freq):
        return x
    kern, width, orig, new = sinc_kernel(orig_freq, new_freq)
    lead, n = x.shape[:-1], x.shape[-1]
    flat = x.reshape(-1, n)
    pad = np.concatenate([np.zeros((flat.shape[0], width), np.float32), flat, np.zeros((flat.shape[0], width + orig), np.float32)], axis=1)
    frames = n // orig + 1
    K = kern.shape[1]
    win = np.lib.stride_tricks.sliding_window_view(pad, K, axis=1)[:, ::orig][:, :frames]       # [B, frames, K]
    y = np.einsum("bfk,pk->bfp", win.astype(np.float64), kern.astype(np.float64)).reshape(flat.shape[0], frames * new)
    target = int(math.ceil(new * n / orig))
    return y[:, :target].astype(np.float32).reshape(*lead, target)
