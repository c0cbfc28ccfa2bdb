"""Batched dataset codec-encode driver — the reference's only batched codec call site.

Mirrors `data/encode.py` of the reference: same flags (:5-19), same manifest (a JSON list of
`{"segment_id": str, "wav": path}`, sliced `[start:end]`, :66-69), same batching (zero `pad_sequence` to the longest
clip of the batch, :99), same on-disk format — one `<save_dir>/<dataset_name>/<save_tag>/<segment_id>.txt` per segment,
K lines of space-separated code ids, no trailing newline, truncated to `round(duration * model_code_sr)` frames, existing
files left alone (:53-57, :103-108).

What differs, deliberately:
* the codec is this package's HIP `WMEncodecModel` (through `AudioTokenizer`), not `WMCompressionSolver.model_from_checkpoint`;
* WAV reading is the in-tree RIFF reader, resampling `data/resample.py` (torchaudio's windowed-sinc `Resample` as a HIP polyphase filter);
* under `torch.distributed.run` (RANK / WORLD_SIZE in the environment) the `[start:end]` slice is split further into
  contiguous per-rank shards — no collective, every rank writes its own files (the reference shards by hand with
  `--start/--end`); `--n_workers` is accepted and ignored (clips are read on the main thread: the encode is ~2000x real time);
* the reference computes `duration = resampled_length / ORIGINAL_sr` (:79-80), which over-counts frames whenever a clip
  is resampled; here `duration = resampled_length / model_sr` (identical for clips already at `model_sr`).
"""
import argparse
import json
import logging
import os
from typing import List, Sequence, Tuple

import numpy as np
import torch

from ..dp import shard_range
from .tokenizer import AudioTokenizer, convert_audio, read_wav


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="encode the dataset using encodec model")
    p.add_argument("--json_path", type=str, default=None)
    p.add_argument("--save_dir", type=str, default=None)
    p.add_argument("--save_tag", type=str, default="wmencodec")
    p.add_argument("--dataset_name", type=str, default=None)
    p.add_argument("--encodec_model_path", type=str, default=None)
    p.add_argument("--n_workers", type=int, default=8, help="accepted for compatibility; unused")
    p.add_argument("--batch_size", type=int, default=64, help="clips per codec batch (per process)")
    p.add_argument("--model_sr", type=int, default=16000, help="encodec input audio sample rate")
    p.add_argument("--downsample_rate", type=int, default=320, help="encodec downsample rate")
    p.add_argument("--model_code_sr", type=int, default=50, help="encodec model code sample rate")
    p.add_argument("--start", type=int, default=0, help="start index for parallel processing")
    p.add_argument("--end", type=int, default=500000, help="end index for parallel processing")
    return p.parse_args(argv)


def write_array_to_txt_file(array: Sequence[Sequence[int]], filename: str) -> None:
    """data/encode.py:53-57 — K lines, space separated, no newline after the last one."""
    with open(filename, "w") as f:
        f.write("\n".join(" ".join(map(str, row)) for row in array))


def read_codes_txt(filename: str) -> np.ndarray:
    """Inverse of `write_array_to_txt_file` (what the reference's dataset reader does, data/dataset.py): [K, T] int64."""
    with open(filename) as f:
        return np.asarray([[int(t) for t in line.split()] for line in f.read().split("\n")], dtype=np.int64)


def pad_batch(clips: List[torch.Tensor]) -> torch.Tensor:
    """`pad_sequence(batch_first=True).unsqueeze(1)` (:99): [n_i] float32 -> [B, 1, max n] zero padded."""
    n = max(int(c.shape[0]) for c in clips)
    out = torch.zeros(len(clips), 1, n, dtype=torch.float32)
    for i, c in enumerate(clips):
        out[i, 0, : c.shape[0]] = c
    return out


def load_clip(path: str, model_sr: int) -> Tuple[torch.Tensor, float]:
    """data/encode.py:74-80: mono clip at `model_sr` and its duration in seconds."""
    audio, sr = read_wav(path)
    if sr != model_sr:
        audio = convert_audio(audio, sr, model_sr, audio.shape[0])
    audio = audio.squeeze()
    if audio.ndim != 1:
        raise ValueError(f"{path}: expected a mono clip, got {tuple(audio.shape)}")     # the reference's pad_sequence would mis-batch it
    return audio.to(torch.float32), audio.shape[0] / float(model_sr)


def encode_manifest(tokenizer: AudioTokenizer, items: List[dict], codes_save_root: str, batch_size: int = 64, model_sr: int = 16000,
                    model_code_sr: int = 50) -> int:
    """Encode `items` (already sliced / sharded) batch by batch; returns the number of files written."""
    os.makedirs(codes_save_root, exist_ok=True)
    written = 0
    for b0 in range(0, len(items), batch_size):
        batch = items[b0 : b0 + batch_size]
        clips, durs = zip(*(load_clip(it["wav"], model_sr) for it in batch))
        padded = pad_batch(list(clips))
        with torch.no_grad():
            codes = tokenizer.encode(padded)[0].cpu()                                   # [B, K, T'] int64
        for i, dur in enumerate(durs):
            save_fn = os.path.join(codes_save_root, batch[i]["segment_id"] + ".txt")
            if not os.path.exists(save_fn):
                actual_len = round(dur * model_code_sr)
                write_array_to_txt_file(codes[i, :, :actual_len].tolist(), save_fn)
                written += 1
    return written


def main(argv=None) -> int:
    logging.basicConfig(format="%(asctime)s [%(levelname)s] %(filename)s:%(lineno)d || %(message)s", level=logging.INFO)
    args = parse_args(argv)
    with open(args.json_path, "r") as f:
        data = json.load(f)[args.start : args.end]
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    lo, hi = shard_range(len(data), world, rank)
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    tokenizer = AudioTokenizer(device=device, signature=args.encodec_model_path)
    assert tokenizer.sample_rate == args.model_sr, (tokenizer.sample_rate, args.model_sr)
    assert tokenizer.sample_rate // tokenizer.codec.frame_rate == args.downsample_rate == args.model_sr // args.model_code_sr, args
    root = os.path.join(args.save_dir, args.dataset_name, args.save_tag)
    logging.info(f"encodec encoding... rank {rank}/{world}: items [{lo}:{hi}) of {len(data)} -> {root}")
    n = encode_manifest(tokenizer, data[lo:hi], root, args.batch_size, args.model_sr, args.model_code_sr)
    logging.info(f"rank {rank}: wrote {n} files")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
