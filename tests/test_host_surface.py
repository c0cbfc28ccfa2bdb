"""CPU: host-side surfaces that mirror the reference: CLI flags (golden parsed from the reference's inference_v2.py), WAV
reader/writer, watermark wav assembly (inference_scale.py:67-78), checkpoint-config reading."""
import json
import os

import numpy as np
import pytest
import torch

import ssr_speech_amd  # noqa: F401
from ssr_speech_amd import inference_v2 as CLI
from ssr_speech_amd.data import tokenizer as TK
from ssr_speech_amd.inference_scale import kept_audio_track


def test_cli_flag_surface_matches_reference(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "cli_flags.json")))
    mine = {f: kw for f, kw in CLI.REFERENCE_FLAGS}
    assert [r["flag"] for r in ref] == [f for f, _ in CLI.REFERENCE_FLAGS]
    for r in ref:
        kw = mine[r["flag"]]
        assert (kw.get("action") == "store_true") == r["store_true"], r
        if not r["store_true"]:
            assert kw["type"].__name__ == r["type"], r
            if "default" in r:
                assert str(kw.get("default")) == r["default"], (r, kw)
        if r["choices"]:
            assert kw["choices"] == r["choices"]
    a = CLI.parse_args(["--tts", "--top_k", "1", "--language", "en"])
    assert a.tts and a.top_k == 1 and a.top_p == 0.8 and a.temperature == 1 and a.cfg_stride == 1 and a.stop_repetition == 2
    with pytest.raises(SystemExit):
        CLI.parse_args(["--language", "fr"])


def test_wav_roundtrip_and_demo_like_formats(tmp_path):
    x = torch.sin(torch.arange(4000) / 20.0).unsqueeze(0) * 0.5
    p = str(tmp_path / "a.wav")
    TK.write_wav(p, x, 16000)
    y, sr = TK.read_wav(p)
    assert sr == 16000 and y.shape == x.shape and float((y - x).abs().max()) < 1.0 / 32767 + 1e-6
    y2, _ = TK.read_wav(p, frame_offset=100, num_frames=50)
    assert torch.equal(y2, y[:, 100:150])
    # float32 WAVE (format tag 3), as two of the reference's demo prompts are stored
    import struct
    raw = x.numpy().astype("<f4").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 3, 1, 16000, 64000, 4, 32)
    q = str(tmp_path / "f.wav")
    open(q, "wb").write(hdr + b"data" + struct.pack("<I", len(raw)) + raw)
    z, sr = TK.read_wav(q)
    assert sr == 16000 and torch.equal(z, x)
    assert TK.convert_audio(torch.cat([x, -x]), 16000, 16000, 1).abs().max() == 0


def test_watermark_wav_assembly():
    """Edit of frames [3,5) of a 8-frame clip replaced by 4 generated frames: kept audio moves, generated region is zero."""
    hop = 4
    wav = torch.arange(8 * hop, dtype=torch.float32).unsqueeze(0)
    masks = [(0, 3), (7, 10)]          # kept intervals, new coordinates
    ori = [(0, 3), (5, 8)]             # kept intervals, original coordinates
    out = kept_audio_track(wav, 10, masks, ori, hop)
    assert out.shape == (1, 40)
    assert torch.equal(out[0, :12], wav[0, :12]) and torch.equal(out[0, 28:40], wav[0, 20:32]) and out[0, 12:28].abs().sum() == 0


def test_codec_config_from_duck_typed_cfg():
    cfg = {"compression_model": "wmencodec", "sample_rate": 16000, "channels": 1,
           "seanet": {"n_filters": 64, "dimension": 128, "ratios": [8, 5, 4, 2], "lstm": 2, "pad_mode": "reflect"}, "rvq": {"n_q": 4, "bins": 2048}}
    c = TK.codec_config_from_xp_cfg(cfg)
    assert c.ratios == (8, 5, 4, 2) and c.pad_mode == "reflect" and c.hop == 320 and c.frame_rate == 50
    with pytest.raises(KeyError):
        TK.codec_config_from_xp_cfg({"compression_model": "encodec"})


def test_encode_driver_flags_and_txt_format(golden_dir, tmp_path):
    """`data/encode.py` flag surface (reference :5-19) and its on-disk code format (:53-57, :103-108)."""
    from ssr_speech_amd.data import encode as ENC
    ref = json.load(open(os.path.join(golden_dir, "encode_flags.json")))
    a = ENC.parse_args([])
    assert sorted(vars(a)) == sorted(r["flag"][2:] for r in ref)
    for r in ref:
        assert str(getattr(a, r["flag"][2:])) == r["default"], r
        assert type(getattr(ENC.parse_args([r["flag"], "7"]), r["flag"][2:])).__name__ == r["type"], r
    codes = [[1, 22, 333], [4, 5, 6], [7, 8, 9], [2047, 0, 1]]
    fn = str(tmp_path / "seg.txt")
    ENC.write_array_to_txt_file(codes, fn)
    assert open(fn).read() == "1 22 333\n4 5 6\n7 8 9\n2047 0 1"          # K lines, no trailing newline
    assert np.array_equal(ENC.read_codes_txt(fn), np.asarray(codes))
    b = ENC.pad_batch([torch.ones(5), torch.ones(9) * 2, torch.ones(1) * 3])
    assert b.shape == (3, 1, 9) and b[0, 0, 5:].abs().sum() == 0 and b[1].sum() == 18 and b[2, 0, 0] == 3
    TK.write_wav(str(tmp_path / "c.wav"), torch.zeros(1, 1234), 16000)
    clip, dur = ENC.load_clip(str(tmp_path / "c.wav"), 16000)
    assert clip.shape == (1234,) and dur == 1234 / 16000 and round(dur * 50) == 4


def test_split_phonemized_follows_the_reference_symbol_rule():
    """The phonemizer's output line -> LM symbols (reference data/tokenizer.py:59-77 states the rule as the regular expression
    `\\w+|[^\\w\\s]` per word, phone separators dropped, the word separator kept as a symbol). Needs no espeak: the scanner is plain Python."""
    import random
    import re
    from ssr_speech_amd.data.tokenizer import split_phonemized

    def rule(line, w="_", p="|"):
        out = []
        for word in line.split(w):
            out += [tok for tok in re.findall(r"\w+|[^\w\s]", word, re.UNICODE) if tok != p] + [w]
        return out[:-1]

    line = "ɐ m|iː|n? ɹ|ɪ|z|ɜː|v; h|ɪ|z._ð|ə_k|w|ɪ|k,_b|ɹ|aʊ|n"
    assert split_phonemized(line) == ['ɐ', 'm', 'iː', 'n', '?', 'ɹ', 'ɪ', 'z', 'ɜː', 'v', ';', 'h', 'ɪ', 'z', '.', '_', 'ð', 'ə', '_',
                                      'k', 'w', 'ɪ', 'k', ',', '_', 'b', 'ɹ', 'aʊ', 'n']
    assert split_phonemized("") == [] and split_phonemized("_") == ["_"] and split_phonemized("a|b") == ["a", "b"]
    rng = random.Random(4)
    alphabet = "abɐiːɪɜʊŋθðˈˌ|_ ,.;?!-'1"
    for _ in range(3000):
        line = "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 24)))
        assert split_phonemized(line) == rule(line), line
    assert split_phonemized("a-b|c", word_sep="-", phone_sep="|") == ["a", "-", "b", "c"]


def test_edit_spans_follow_the_reference_alignment_to_span_rule(golden_dir):
    """`--mask_spans`: margins, clipping, sort, the 0.2 s merge rule (>=, in the reference's float arithmetic), frame
    rounding and the `maximum 3 editings` error, against cases recorded by executing the reference's own statements
    (oracle/make_golden_spans.py, inference_v2.py:284-317)."""
    import json
    from ssr_speech_amd.inference_v2 import edit_spans, parse_mask_spans
    cases = json.load(open(os.path.join(golden_dir, "edit_spans.json")))["cases"]
    assert len(cases) >= 12
    for c in cases:
        spans = [tuple(s) for s in c["spans"]]
        if "error" in c["expect"]:
            with pytest.raises(RuntimeError, match="maximum 3 editings"):
                edit_spans(spans, c["sub_amount"], c["audio_dur"], c["codec_sr"])
            continue
        morphed, mi = edit_spans(spans, c["sub_amount"], c["audio_dur"], c["codec_sr"])
        assert morphed == c["expect"]["morphed_span"], (c["spans"], morphed)           # exact: the same float operations
        assert mi.dtype == torch.int64 and mi.tolist() == c["expect"]["mask_interval"]
    assert parse_mask_spans("0.8-1.2, 2.5-3.1,") == [(0.8, 1.2), (2.5, 3.1)]
    for bad in ("1.0", "2-1", "-1-2"):
        with pytest.raises(SystemExit):
            parse_mask_spans(bad)
