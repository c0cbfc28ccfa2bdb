"""GPU: `data/resample.py` (torchaudio's windowed-sinc Resample as a polyphase filter on `ssrhip_conv_cin1`) against the oracle, and the two
front-end paths that call it: `convert_audio` / `tokenize_audio` (reference data/tokenizer.py:87-97, :141-160) on a 44.1 kHz stereo file."""
import numpy as np
import pytest
import torch

import ssr_speech_amd  # noqa: F401
from ssr_speech_amd import weights as W
from ssr_speech_amd.data.resample import resample
from ssr_speech_amd.data.tokenizer import AudioTokenizer, convert_audio, tokenize_audio, write_wav
from oracle import resample as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("a,b", [(44100, 16000), (48000, 16000), (22050, 16000), (24000, 16000), (8000, 16000), (16000, 24000), (32000, 16000)])
@pytest.mark.parametrize("n", [1, 5, 440, 441, 9999, 44100 + 17])
def test_resample_matches_oracle(a, b, n):
    g = torch.Generator().manual_seed(a + n)
    x = torch.randn(2, n, generator=g) * 0.3
    want = O.resample(x.numpy(), a, b)
    got_cpu_in = resample(x, a, b)                                  # CPU tensor in -> CPU tensor out
    assert got_cpu_in.device.type == "cpu" and got_cpu_in.dtype == torch.float32 and tuple(got_cpu_in.shape) == want.shape
    np.testing.assert_allclose(got_cpu_in.numpy(), want, rtol=0, atol=2e-6)     # fp32 FMA chain of <= 475 taps vs float64 accumulation
    got_dev = resample(x.cuda().view(2, 1, n), a, b)                # leading dims are kept, device tensor stays on the device
    assert got_dev.device.type == "cuda" and tuple(got_dev.shape) == (2, 1, want.shape[-1])
    assert torch.equal(got_dev.view(2, -1).cpu(), got_cpu_in)


def test_equal_rates_return_the_input_and_empty_input():
    x = torch.randn(1, 100)
    assert resample(x, 16000, 16000) is x
    assert tuple(resample(torch.zeros(1, 0), 44100, 16000).shape) == (1, 0)


def test_tokenize_audio_resamples_a_44k_stereo_file(tmp_path):
    cfg = W.codec_config_tiny()
    sd = W.codec_state_dict(cfg, seed=3)
    tok = AudioTokenizer(device="cuda", config=cfg, state_dict=sd)
    g = torch.Generator().manual_seed(0)
    n = 44100 // 2 + 5
    wav = torch.randn(2, n, generator=g) * 0.1
    fn = str(tmp_path / "stereo44.wav")
    write_wav(fn, wav, 44100)
    from ssr_speech_amd.data.tokenizer import read_wav
    q, sr = read_wav(fn)
    assert sr == 44100
    mono16 = convert_audio(q, sr, tok.sample_rate, tok.channels)
    want = O.resample(q.mean(0, keepdim=True).numpy(), 44100, tok.sample_rate)
    np.testing.assert_allclose(mono16.numpy(), want, rtol=0, atol=2e-6)
    codes, scale, emb = tokenize_audio(tok, fn)
    qp = torch.nn.functional.pad(q, (0, -q.shape[-1] % 320))       # the reference pads at the FILE's rate, then converts (data/tokenizer.py:147-153)
    want = O.resample(qp.mean(0, keepdim=True).numpy(), 44100, tok.sample_rate)
    ref_codes, _, ref_emb = tok.encode(torch.from_numpy(want).unsqueeze(0))
    assert codes.shape == ref_codes.shape and codes.shape[-1] == -(-want.shape[-1] // cfg.hop)
    torch.testing.assert_close(emb, ref_emb, rtol=0, atol=1e-4)
