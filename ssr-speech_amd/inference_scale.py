"""`inference_one_sample` — the reference's per-utterance glue (`inference_scale.py:17-88`) with the same positional
signature: phonemes -> ids, wav -> codes (HIP codec), `model.inference` (HIP decode engine), watermark wav assembly,
codes -> wav (HIP codec)."""
from __future__ import annotations

import logging
import time

import torch
import torch.nn.functional as F

from .data.tokenizer import read_wav, tokenize_audio, tokenize_text


def assemble_watermark_wav(wav: torch.Tensor, n_frames: int, masks, ori_masks, hop: int = 320) -> torch.Tensor:
    """inference_scale.py:67-78: the original audio in the kept regions (moved to their new positions), zeros where
    audio was generated; `masks` are kept intervals in NEW frame coordinates, `ori_masks` in ORIGINAL coordinates."""
    new_wav = torch.zeros(1, n_frames * hop)
    ori = [(max(a, 0), b) for a, b in ori_masks]
    new = [(max(a, 0), b) for a, b in masks]
    for i in range(len(ori)):
        new_wav[:, new[i][0] * hop: new[i][1] * hop] = wav[:, ori[i][0] * hop: ori[i][1] * hop]
    return new_wav


@torch.no_grad()
def inference_one_sample(model, model_args, phn2num, text_tokenizer, audio_tokenizer, audio_fn, prompt_text, target_text, mask_interval,
                         cfg_coef, cfg_stride, aug_text, aug_context, use_watermark, tts, device, decode_config):
    # phonemize (inference_scale.py:20-34): phonemes missing from phn2num are silently dropped
    text_tokens = [phn2num[phn] for phn in tokenize_text(text_tokenizer, text=target_text.strip()) if phn in phn2num]
    text_tokens = torch.LongTensor(text_tokens).unsqueeze(0)
    text_tokens_lens = torch.LongTensor([text_tokens.shape[-1]])
    prompt_text_tokens = [phn2num[phn] for phn in tokenize_text(text_tokenizer, text=prompt_text.strip()) if phn in phn2num]
    prompt_text_tokens = torch.LongTensor(prompt_text_tokens).unsqueeze(0)
    prompt_text_tokens_lens = torch.LongTensor([prompt_text_tokens.shape[-1]])

    encoded_frames, scale, emb = tokenize_audio(audio_tokenizer, audio_fn)
    original_audio = encoded_frames.transpose(2, 1)  # [1,T,K]
    assert original_audio.ndim == 3 and original_audio.shape[0] == 1 and original_audio.shape[2] == model_args.n_codebooks, original_audio.shape
    logging.info(f"with direct encodec encoding before input, original audio length: {original_audio.shape[1]} codec frames, "
                 f"which is {original_audio.shape[1] / decode_config['codec_sr']:.2f} sec.")

    stime = time.time()
    encoded_frames, marks, masks, ori_masks = model.inference(
        text_tokens.to(device), text_tokens_lens.to(device), prompt_text_tokens.to(device), prompt_text_tokens_lens.to(device),
        original_audio[..., :model_args.n_codebooks].to(device), original_audio[..., :model_args.n_codebooks].to(device),
        mask_interval=mask_interval.unsqueeze(0).to(device), top_k=decode_config['top_k'], top_p=decode_config['top_p'],
        temperature=decode_config['temperature'], stop_repetition=decode_config['stop_repetition'], kvcache=decode_config['kvcache'],
        cfg_coef=cfg_coef, cfg_stride=cfg_stride, aug_text=aug_text)
    logging.info(f"inference on one sample take: {time.time() - stime:.4f} sec.")
    if type(encoded_frames) == tuple:
        encoded_frames = encoded_frames[0]
    logging.info(f"generated encoded_frames.shape: {encoded_frames.shape}, which is {encoded_frames.shape[-1] / decode_config['codec_sr']} sec.")

    if use_watermark:
        multiple = 320
        wav, sr = read_wav(audio_fn)
        padding_length = (multiple - (wav.shape[-1] % multiple)) % multiple
        if padding_length > 0:
            wav = F.pad(wav, (0, padding_length), "constant", 0)
        new_wav = assemble_watermark_wav(wav, encoded_frames.shape[-1], masks, ori_masks, 320)
        generated_sample = audio_tokenizer.wmdecode(encoded_frames, marks.to(encoded_frames.device), new_wav.unsqueeze(0).to(encoded_frames.device), scale)
    else:
        generated_sample = audio_tokenizer.decode(encoded_frames, scale)
    if tts:
        generated_sample = generated_sample[:, :, masks[0][1] * 320:]
    return generated_sample
