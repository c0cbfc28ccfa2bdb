"""CPU: oracle/codec.py against the REFERENCE's golden vectors (tests/golden/codec_*.npz from oracle/make_golden_codec.py):
codes identical, fp32 tensors within 5e-6 (bit-exact in 4 of the 5 cases: same ATen ops)."""
import glob
import os

import numpy as np
import pytest
import torch

import ssr_speech_amd  # noqa: F401
from ssr_speech_amd import weights as W
from oracle import codec as OC

CASES = sorted(os.path.basename(p)[6:-4] for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "codec_*.npz")))


def load_case(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"codec_{name}.npz"))
    c = [int(v) for v in g["cfg"]]
    cfg = W.CodecConfig(dimension=c[0], n_filters=c[1], bins=c[2], n_q=c[3], ratios=tuple(c[4:]), pad_mode=str(g["pad_mode"]))
    sd = W.codec_state_dict(cfg, seed=int(g["weight_seed"]))
    return g, cfg, sd


@pytest.mark.parametrize("name", CASES)
def test_codec_oracle_matches_reference(golden_dir, name):
    g, cfg, sd = load_case(golden_dir, name)
    wav = torch.from_numpy(g["wav"])
    codes, scale, emb = OC.encode(sd, wav, cfg)
    assert scale is None
    assert np.array_equal(codes.numpy(), g["codes"])
    np.testing.assert_allclose(emb.numpy(), g["emb"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(OC.decode(sd, torch.from_numpy(g["codes"]), cfg).numpy(), g["decoded"], rtol=0, atol=5e-6)
    out, mark = OC.wmdecode(sd, torch.from_numpy(g["codes"]), torch.from_numpy(g["labels"]), torch.from_numpy(g["wav_pad"]), cfg)
    np.testing.assert_allclose(out.numpy(), g["wmdecoded"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(mark.numpy(), g["mark"], rtol=0, atol=5e-6)
    assert np.array_equal(OC.detect_watermark(sd, torch.from_numpy(g["wmdecoded"]), cfg).numpy(), g["detect"])


def test_output_length_rules():
    """The reference's own shape tests (audiocraft/tests/modules/test_conv.py:151-203, test_seanet.py:18-56) restated:
    encoder T -> ceil(T/hop) frames, decoder frames -> frames*hop samples."""
    cfg = W.codec_config_tiny()
    sd = W.codec_state_dict(cfg, seed=1)
    for n in (cfg.hop * 3, cfg.hop * 3 + 1, cfg.hop * 4 - 1):
        x = torch.randn(1, 1, n)
        codes, _, emb = OC.encode(sd, x, cfg)
        frames = -(-n // cfg.hop)
        assert emb.shape == (1, cfg.dimension, frames) and codes.shape == (1, cfg.n_q, frames)
        assert OC.decode(sd, codes, cfg).shape == (1, 1, frames * cfg.hop)


def test_rvq_decode_rejects_out_of_range_ids():
    cfg = W.codec_config_tiny()
    sd = W.codec_state_dict(cfg, seed=1)
    bad = torch.full((1, cfg.n_q, 3), cfg.bins, dtype=torch.long)     # e.g. a stray special token
    with pytest.raises(IndexError):
        OC.rvq_decode(sd, bad, cfg)
