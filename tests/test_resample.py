"""CPU: the resampling oracle (oracle/resample.py, a restatement of torchaudio.transforms.Resample's defaults — PARITY UNPINNED, torchaudio
is not in this image) checked by the properties a band-limited resampler must have, and the product's host-side filter bank against it."""
import math

import numpy as np
import pytest

import ssr_speech_amd  # noqa: F401
from ssr_speech_amd.data.resample import sinc_filter_bank
from oracle import resample as O

PAIRS = [(44100, 16000), (48000, 16000), (22050, 16000), (24000, 16000), (8000, 16000), (16000, 24000), (32000, 16000)]


@pytest.mark.parametrize("a,b", PAIRS)
def test_filter_bank_of_the_product_is_the_oracles(a, b):
    bank, width, orig, new = sinc_filter_bank(a, b)
    kern, w2, o2, n2 = O.sinc_kernel(a, b)
    assert (width, orig, new) == (w2, o2, n2) and bank.dtype == np.float32
    assert orig == a // math.gcd(a, b) and new == b // math.gcd(a, b)
    assert bank.shape == (new, 2 * width + orig) and width == math.ceil(6 * orig / (min(orig, new) * 0.99))
    np.testing.assert_array_equal(bank, kern)
    np.testing.assert_allclose(bank.sum(1), 1.0, atol=2e-3)          # every phase passes DC (the Hann-windowed sinc overshoots by ~5e-4)


@pytest.mark.parametrize("a,b", PAIRS)
@pytest.mark.parametrize("n", [1, 7, 1000, 44100 // 4 + 13])
def test_output_length_and_dc(a, b, n):
    y = O.resample(np.ones((2, n), np.float32), a, b)
    g = math.gcd(a, b)
    assert y.shape == (2, math.ceil((b // g) * n / (a // g))) and y.dtype == np.float32
    if n >= 1000:
        mid = y[:, y.shape[1] // 4: -y.shape[1] // 4]
        np.testing.assert_allclose(mid, 1.0, atol=2e-3)             # away from the zero-padded ends


@pytest.mark.parametrize("a,b", PAIRS)
def test_tone_below_the_cutoff_is_preserved_and_above_is_rejected(a, b):
    n = a // 2
    t_in = np.arange(n) / a
    f_lo = 0.2 * min(a, b) / 2                                       # well inside the pass band of both rates
    y = O.resample(np.sin(2 * np.pi * f_lo * t_in).astype(np.float32), a, b)
    t_out = np.arange(y.shape[0]) / b
    want = np.sin(2 * np.pi * f_lo * t_out)
    k = y.shape[0] // 8
    assert np.abs(y[k:-k] - want[k:-k]).max() < 3e-3
    if b < a:                                                        # a tone above the new Nyquist must not alias in
        f_hi = 0.5 * (b / 2 + a / 2)
        z = O.resample(np.sin(2 * np.pi * f_hi * t_in).astype(np.float32), a, b)
        assert np.sqrt(np.mean(z[k:-k] ** 2)) < 6e-2                  # 6 zero crossings, Hann: a wide transition band (-29 dB at 1.19 x Nyquist for 22.05 -> 16 kHz)


def test_linearity_identity_and_batch_shape():
    g = np.random.default_rng(0)
    x1, x2 = g.standard_normal((3, 2, 999)).astype(np.float32), g.standard_normal((3, 2, 999)).astype(np.float32)
    y = O.resample(2 * x1 - 3 * x2, 22050, 16000)
    assert y.shape[:2] == (3, 2)
    np.testing.assert_allclose(y, 2 * O.resample(x1, 22050, 16000) - 3 * O.resample(x2, 22050, 16000), atol=1e-5)
    assert O.resample(x1, 16000, 16000) is not None and np.array_equal(O.resample(x1, 16000, 16000), x1)
