"""`inference_one_sample` — one utterance through the whole HIP path: phonemes -> ids, prompt wav -> codec codes,
AR decode (`SSR_Speech.inference`), codes -> wav (plain or watermarked decode).

Only the positional signature and the returned tensor are the reference's (`inference_scale.py:18`, `:88`; SURVEY §8
row C1) so that `inference_v2.py`-style drivers can call it unchanged; the body is organised around three stages:

  1. `_phoneme_ids`           text -> LongTensor [1, L]                (ids missing from `phn2num` are dropped, :20-34)
  2. `_prompt_codes`          wav file -> codes [1, T, K], scale       (:36-38)
  3. `_render`                generated codes -> waveform [1, 1, n]    (:63-86), using `kept_audio_track` for the
                              watermark decoder's skip input (:67-78, pinned by tests/golden/glue_watermark.npz)
"""
from __future__ import annotations

import logging
import time
from typing import Sequence, Tuple

import torch
import torch.nn.functional as F

from .data.tokenizer import read_wav, tokenize_audio, tokenize_text

HOP = 320          # samples per codec frame at 16 kHz / 50 Hz; the reference hard-codes it (:66, :69, :76, :86)
log = logging.getLogger(__name__)


def _phoneme_ids(text_tokenizer, phn2num: dict, text: str) -> torch.Tensor:
    ids = [phn2num[p] for p in tokenize_text(text_tokenizer, text=text.strip()) if p in phn2num]
    return torch.tensor(ids, dtype=torch.long).view(1, -1)


def _prompt_codes(audio_tokenizer, audio_fn: str, n_codebooks: int) -> Tuple[torch.Tensor, object]:
    codes, scale, _emb = tokenize_audio(audio_tokenizer, audio_fn)           # [1, K, T]
    frames_first = codes.transpose(2, 1)                                     # [1, T, K] as `inference` wants it
    if frames_first.ndim != 3 or frames_first.shape[0] != 1 or frames_first.shape[2] != n_codebooks:
        raise AssertionError(tuple(frames_first.shape))
    return frames_first[..., :n_codebooks], scale


def kept_audio_track(wav: torch.Tensor, n_frames: int, kept_new: Sequence, kept_old: Sequence, hop: int = HOP) -> torch.Tensor:
    """Waveform the watermark decoder's skip-encoder sees: silence where frames were generated, the ORIGINAL audio where
    frames were kept, moved to where those frames sit in the new utterance (reference `inference_scale.py:67-78`).

    wav       [1, n] original audio, n a multiple of `hop`
    kept_new  intervals (frame units) of kept audio in the NEW frame axis   (`masks` of `SSR_Speech.inference`)
    kept_old  the same intervals in the ORIGINAL frame axis                 (`non_mask_intervals`)
    -> [1, n_frames * hop]

    Done per frame: a gather of whole hop-sized rows, so a kept interval must have the same length on both axes."""
    if wav.ndim != 2 or wav.shape[0] != 1 or wav.shape[1] % hop:
        raise ValueError(f"expected mono audio [1, k*{hop}], got {tuple(wav.shape)}")
    src = wav.reshape(-1, hop)
    track = torch.zeros(n_frames, hop, dtype=wav.dtype)
    for (new_a, new_b), (old_a, old_b) in zip(kept_new, kept_old):
        new_a, old_a = max(int(new_a), 0), max(int(old_a), 0)
        new_b, old_b = int(new_b), int(old_b)
        if new_b - new_a != old_b - old_a:
            raise ValueError(f"kept interval changed length: new [{new_a},{new_b}) vs original [{old_a},{old_b})")
        if new_b > new_a:
            track[new_a:new_b] = src[old_a:old_b]
    return track.reshape(1, n_frames * hop)


def _watermark_inputs(codes, kept_new, kept_old, audio_fn) -> torch.Tensor:
    """The skip-encoder input of `wmdecode` for one utterance (:67-78): [1, 1, frames * HOP] on the codes' device."""
    wav, _sr = read_wav(audio_fn) if isinstance(audio_fn, str) else (audio_fn, None)
    short = -wav.shape[-1] % HOP                                             # zero-extend to whole frames, like the encode side
    if short:
        wav = F.pad(wav, (0, short))
    return kept_audio_track(wav, codes.shape[-1], kept_new, kept_old).unsqueeze(0).to(codes.device)


def _render(audio_tokenizer, codes, marks, kept_new, kept_old, scale, audio_fn, use_watermark: bool) -> torch.Tensor:
    if not use_watermark:
        return audio_tokenizer.decode(codes, scale)
    return audio_tokenizer.wmdecode(codes, marks.to(codes.device), _watermark_inputs(codes, kept_new, kept_old, audio_fn), scale)


def render_many(audio_tokenizer, results: Sequence[tuple], scale, audio_fns, use_watermark: bool, tts: bool):
    """`_render` of several utterances in ONE pass of the codec per length bucket (`AudioTokenizer.decode_batch` /
    `wmdecode_batch`: items of different lengths, each with its own halo, so waveform i equals `_render` of utterance i alone).
    results[i] = the 4-tuple of `SSR_Speech.inference`; audio_fns = one path / [1, n] tensor for all, or a list with one per
    utterance (only read with `use_watermark`). For `tts` the prompt part (the first kept interval) is cut off (:85-86)."""
    if not isinstance(audio_fns, (list, tuple)):
        audio_fns = [audio_fns] * len(results)
    codes = [r[0] for r in results]
    if not use_watermark:
        waves = audio_tokenizer.decode_batch(codes, scale)
    else:
        tracks = [_watermark_inputs(c, kn, ko, fn) for (c, _m, kn, ko), fn in zip(results, audio_fns)]
        waves = audio_tokenizer.wmdecode_batch(codes, [r[1].to(r[0].device) for r in results], tracks, scale)
    if tts:
        waves = [w[..., int(r[2][0][1]) * HOP:] for w, r in zip(waves, results)]
    return waves


@torch.no_grad()
def inference_samples(model, model_args, phn2num, text_tokenizer, audio_tokenizer, audio_fn, prompt_text, target_text, mask_interval,
                      cfg_coef, cfg_stride, aug_text, aug_context, use_watermark, tts, device, decode_config, seeds):
    """`--sample_batch_size N` in one pass: the N samples of ONE utterance (seeds `seeds[i]`) decoded in lock-step through
    `SSR_Speech.inference_batch`, so the weights stream once per step for all of them (the reference loops them one after the
    other, `inference_v2.py:331-358`). Sample i is bit-identical to `inference_one_sample` called after
    `torch.manual_seed(seeds[i])`: the prompt is encoded once (it is the same for every sample) and every sample keeps its
    own RNG stream. `seeds` must be consecutive integers. Returns the list of waveforms [1, 1, n_i]."""
    seeds = [int(s) for s in seeds]
    assert seeds == list(range(seeds[0], seeds[0] + len(seeds))), "seeds must be consecutive (seed + sample index)"
    K = int(model_args.n_codebooks)
    target_ids = _phoneme_ids(text_tokenizer, phn2num, target_text)
    prompt_frames, scale = _prompt_codes(audio_tokenizer, audio_fn, K)
    one = {"x": target_ids, "y": prompt_frames, "mask_interval": mask_interval.unsqueeze(0)}
    t0 = time.perf_counter()
    results = model.inference_batch([one] * len(seeds), top_k=decode_config["top_k"], top_p=decode_config["top_p"],
                                    temperature=decode_config["temperature"], stop_repetition=decode_config["stop_repetition"],
                                    cfg_coef=cfg_coef, cfg_stride=cfg_stride, aug_text=aug_text, seed=seeds[0])
    log.info("AR decode of %d samples in lock-step: %.3f s", len(seeds), time.perf_counter() - t0)
    # all samples through the codec in one ragged pass (they differ in length; each keeps its own halo): same waveforms as one
    # `_render` per sample, which is what the reference's loop does (inference_v2.py:331-358)
    return render_many(audio_tokenizer, results, scale, audio_fn, bool(use_watermark), bool(tts))


@torch.no_grad()
def inference_one_sample(model, model_args, phn2num, text_tokenizer, audio_tokenizer, audio_fn, prompt_text, target_text, mask_interval,
                         cfg_coef, cfg_stride, aug_text, aug_context, use_watermark, tts, device, decode_config):
    """Positional signature of the reference (`inference_scale.py:18`). `aug_context` is accepted and — exactly as in the
    reference, which never forwards it (:42-58) — not handed to `model.inference`. Returns the waveform [1, 1, n]; for
    `tts` the prompt part (the first kept interval) is cut off (:85-86)."""
    K = int(model_args.n_codebooks)
    target_ids = _phoneme_ids(text_tokenizer, phn2num, target_text)
    prompt_ids = _phoneme_ids(text_tokenizer, phn2num, prompt_text)
    prompt_frames, scale = _prompt_codes(audio_tokenizer, audio_fn, K)
    rate = decode_config["codec_sr"]
    log.info("prompt: %d codec frames (%.2f s), %d target phonemes", prompt_frames.shape[1], prompt_frames.shape[1] / rate, target_ids.shape[1])

    t0 = time.perf_counter()
    on_dev = lambda t: t.to(device)
    result = model.inference(
        on_dev(target_ids), on_dev(torch.tensor([target_ids.shape[1]])), on_dev(prompt_ids), on_dev(torch.tensor([prompt_ids.shape[1]])),
        on_dev(prompt_frames), on_dev(prompt_frames), mask_interval=on_dev(mask_interval.unsqueeze(0)),
        top_k=decode_config["top_k"], top_p=decode_config["top_p"], temperature=decode_config["temperature"],
        stop_repetition=decode_config["stop_repetition"], kvcache=decode_config["kvcache"],
        cfg_coef=cfg_coef, cfg_stride=cfg_stride, aug_text=aug_text)
    codes, marks, kept_new, kept_old = result
    if isinstance(codes, tuple):                                             # tolerated by the reference too (:60-61)
        codes = codes[0]
    log.info("AR decode: %.3f s for %d frames (%.2f s of audio)", time.perf_counter() - t0, codes.shape[-1], codes.shape[-1] / rate)

    wave = _render(audio_tokenizer, codes, marks, kept_new, kept_old, scale, audio_fn, bool(use_watermark))
    if tts:
        wave = wave[..., int(kept_new[0][1]) * HOP:]
    return wave
