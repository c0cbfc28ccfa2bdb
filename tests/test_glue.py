"""CPU: the per-utterance glue's watermark track (`kept_audio_track`) against what the REAL reference
`inference_one_sample` handed to its codec (tests/golden/glue_watermark.npz, made by oracle/make_golden_glue.py from
/root/reference/inference_scale.py:67-78) — SURVEY §8c G9. Bit-exact: it is pure slicing."""
import os

import numpy as np
import pytest
import torch

import ssr_speech_amd  # noqa: F401
from ssr_speech_amd.inference_scale import HOP, kept_audio_track

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "glue_watermark.npz"))
CASES = ["tts", "edit_mid", "edit_start", "edit_two"]


def padded(wav):
    w = torch.from_numpy(wav)
    return torch.nn.functional.pad(w, (0, -w.shape[-1] % HOP))


@pytest.mark.parametrize("name", CASES)
def test_kept_audio_track_equals_reference_new_wav(name):
    want = G[f"{name}_new_wav"]                                   # [1, 1, T'*320]
    got = kept_audio_track(padded(G[f"{name}_wav"]), G[f"{name}_frames"].shape[-1], G[f"{name}_masks"].tolist(), G[f"{name}_ori_masks"].tolist())
    assert got.shape == want.shape[1:]
    np.testing.assert_array_equal(got.numpy(), want[0])


@pytest.mark.parametrize("name", CASES)
def test_track_is_silent_exactly_where_frames_were_generated(name):
    """marks == 1 <=> generated frame (ssr.py:789-803): those hops are zero in the track, every other hop is original audio."""
    marks = G[f"{name}_marks"][0]
    track = G[f"{name}_new_wav"][0, 0].reshape(-1, HOP)
    assert track.shape[0] == marks.shape[0]
    assert not track[marks == 1].any()
    src = padded(G[f"{name}_wav"]).numpy().reshape(-1, HOP)
    kept = track[marks == 0]
    # kept frames keep their order: they are the original frames outside the edited spans
    mi = G[f"{name}_mask_interval"]
    keep_old = np.ones(src.shape[0], bool)
    for a, b in mi:
        keep_old[a:b] = False
    np.testing.assert_array_equal(kept, src[keep_old])


def test_track_rejects_an_interval_whose_length_changed():
    with pytest.raises(ValueError):
        kept_audio_track(torch.zeros(1, 10 * HOP), 12, [(0, 5)], [(0, 4)])
    with pytest.raises(ValueError):
        kept_audio_track(torch.zeros(1, 10 * HOP + 1), 12, [(0, 5)], [(0, 5)])


def test_negative_interval_starts_clamp_to_zero():
    """`max(item[0], 0)` of inference_scale.py:71-72."""
    w = torch.arange(4 * HOP, dtype=torch.float32).view(1, -1)
    got = kept_audio_track(w, 6, [(-1, 0), (4, 6)], [(-1, 0), (2, 4)])
    assert not got[:, : 4 * HOP].any()
    np.testing.assert_array_equal(got[0, 4 * HOP:].numpy(), w[0, 2 * HOP:].numpy())
