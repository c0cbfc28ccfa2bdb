"""ORACLE tooling (build container only) — golden vectors for the per-utterance glue:
run the REAL `inference_one_sample` of /root/reference/inference_scale.py around the REAL tiny reference LM and
record what it hands to the codec's `wmdecode` (the re-assembled waveform `new_wav`, `inference_scale.py:67-78`),
plus the sample offset it cuts for --tts (:85-86).

torchaudio and phonemizer are not installed offline; only their I/O entry points are stood in for
(`torchaudio.load` -> this repo's RIFF reader; the phonemizer symbols are never called because a plain callable
is passed as `text_tokenizer`). The audio tokenizer is a recording double (random codes in, fixed wave out): the
fixture pins the glue's integer/slicing arithmetic, not the codec (which has its own fixtures).

Re-run:  python -m oracle.make_golden_glue      -> tests/golden/glue_watermark.npz
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
import ssr_speech_amd  # noqa: E402,F401
from ssr_speech_amd import weights as W  # noqa: E402
from ssr_speech_amd.data.tokenizer import read_wav, write_wav  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _stub_io_modules():
    if "torchaudio" not in sys.modules:
        ta = types.ModuleType("torchaudio")
        ta.load = lambda fn, frame_offset=0, num_frames=-1: read_wav(fn, frame_offset, num_frames)
        ta.transforms = types.SimpleNamespace(Resample=lambda a, b: (lambda w: w))
        sys.modules["torchaudio"] = ta
    names = ["phonemizer", "phonemizer.backend", "phonemizer.backend.espeak", "phonemizer.backend.espeak.language_switch",
             "phonemizer.backend.espeak.words_mismatch", "phonemizer.punctuation", "phonemizer.separator"]
    for n in names:
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)

    class _Inert:                     # default-argument expressions of the reference's TextTokenizer evaluate these at import time
        def __init__(self, *a, **k):
            pass

        @staticmethod
        def default_marks():
            return ""

    sys.modules["phonemizer.backend"].EspeakBackend = _Inert
    sys.modules["phonemizer.backend.espeak.language_switch"].LanguageSwitch = _Inert
    sys.modules["phonemizer.backend.espeak.words_mismatch"].WordMismatch = _Inert
    sys.modules["phonemizer.punctuation"].Punctuation = _Inert
    sys.modules["phonemizer.separator"].Separator = _Inert


class RecordingTokenizer:
    """Stands in for data.tokenizer.AudioTokenizer: deterministic codes, remembers every wmdecode / decode call."""

    sample_rate, channels = 16000, 1

    def __init__(self, codes):
        self.codes, self.calls = codes, []

    def encode(self, wav):
        assert wav.shape[-1] == self.codes.shape[-1] * 320, (wav.shape, self.codes.shape)
        return self.codes, None, None

    def wmdecode(self, frames, marks, wav, scale):
        self.calls.append(("wmdecode", frames.clone(), marks.clone(), wav.clone()))
        return torch.arange(frames.shape[-1] * 320, dtype=torch.float32).view(1, 1, -1)

    def decode(self, frames, scale):
        self.calls.append(("decode", frames.clone()))
        return torch.arange(frames.shape[-1] * 320, dtype=torch.float32).view(1, 1, -1)


CASES = [
    # name, n_frames, wav tail (samples short of a 320 multiple), mask_interval, tts
    ("tts", 20, 7, [[20, 20]], True),
    ("edit_mid", 30, 0, [[10, 17]], False),
    ("edit_start", 24, 113, [[0, 6]], False),
    ("edit_two", 30, 1, [[5, 9], [18, 22]], False),
]


def main(gold=GOLD):
    _stub_io_modules()
    ssr = ref_import.import_lm()
    import inference_scale as REF   # /root/reference is on sys.path after import_lm()
    args = W.lm_args_tiny()
    out = {}
    tmp = os.path.join("/tmp", "glue_golden")
    os.makedirs(tmp, exist_ok=True)
    for name, n_frames, short, mi, tts in CASES:
        seed = 40 + len(out)
        model = ssr.SSR_Speech(args).eval()
        model.load_state_dict(W.lm_state_dict(args, seed=seed), strict=False)
        g = torch.Generator().manual_seed(seed)
        wav = torch.randn(1, n_frames * 320 - short, generator=g) * 0.2
        fn = os.path.join(tmp, f"{name}.wav")
        write_wav(fn, wav, 16000)
        wav_q, _ = read_wav(fn)                                  # what any reader of the file sees (16-bit PCM)
        codes = torch.randint(0, args.audio_vocab_size, (1, args.n_codebooks, n_frames), generator=g)
        tok = RecordingTokenizer(codes)
        phn2num = {c: i for i, c in enumerate("abcdefghijklmnopqrstuvwxyz")}
        text_tokenizer = lambda texts: [[c for c in t if c != " "] for t in texts]
        decode_config = {"top_k": 1, "top_p": 1.0, "temperature": 1, "stop_repetition": 2, "kvcache": 1, "codec_audio_sr": 16000, "codec_sr": 50}
        seen = {}
        real_inference = model.inference

        def spy(*a, **k):                                        # keep the two interval lists the glue receives from the LM
            r = real_inference(*a, **k)
            seen["masks"], seen["ori_masks"] = r[2], r[3]
            return r

        model.inference = spy
        torch.manual_seed(seed)
        sample = REF.inference_one_sample(model, argparse.Namespace(**vars(args)), phn2num, text_tokenizer, tok, fn, "hello world",
                                          "hello world again and again", torch.LongTensor(mi), 1.5, 2, True, False, True, tts, "cpu", decode_config)
        kind, frames, marks, new_wav = tok.calls[-1]
        assert kind == "wmdecode"
        total = frames.shape[-1] * 320
        out[f"{name}_wav"] = wav_q.numpy()
        out[f"{name}_codes"] = codes.numpy()
        out[f"{name}_mask_interval"] = np.asarray(mi)
        out[f"{name}_tts"] = np.asarray(int(tts))
        out[f"{name}_torch_seed"] = np.asarray(seed)
        out[f"{name}_frames"] = frames.numpy()
        out[f"{name}_marks"] = marks.numpy()
        out[f"{name}_masks"] = np.asarray(seen["masks"]).reshape(-1, 2)
        out[f"{name}_ori_masks"] = np.asarray(seen["ori_masks"]).reshape(-1, 2)
        out[f"{name}_new_wav"] = new_wav.numpy()                 # [1,1,T'*320]
        out[f"{name}_sample_first"] = np.asarray(int(sample[0, 0, 0]))     # = offset cut for --tts (the double returns arange)
        out[f"{name}_sample_len"] = np.asarray(int(sample.shape[-1]))
        print(f"  glue/{name}: frames {tuple(frames.shape)} new_wav nonzero {int((new_wav != 0).sum())}/{total} first {int(sample[0, 0, 0])}")
    out["torch_version"] = np.asarray(torch.__version__)
    np.savez_compressed(os.path.join(gold, "glue_watermark.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
