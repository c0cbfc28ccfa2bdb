"""CLI with the reference's `inference_v2.py` argument surface (flags, types, defaults, choices: `inference_v2.py:158-188`)
driving the HIP path. The reference derives the edit span with WhisperX ASR + forced alignment and espeak
phonemisation (`inference_v2.py:216-327`), third-party models that are outside this package's scope; here the span is
given explicitly (extra flags `--mask_start/--mask_end` in seconds, or `--prompt_end` for --tts) and the text goes
through whatever phonemiser `TextTokenizer` finds (or `--phoneme_ids` directly).

Outputs follow the reference: `{output_dir}/{savename}_new_seed{seed+num}.wav` (+ `_orig.wav`), one per `--sample_batch_size`.
"""
from __future__ import annotations

import argparse
import os
import random
import shutil
import time

import numpy as np
import torch

# (flag, argparse kwargs) — the reference's surface, in its order
REFERENCE_FLAGS = [
    ("--sub_amount", dict(type=float, default=0.12)),
    ("--codec_audio_sr", dict(type=int, default=16000)),
    ("--codec_sr", dict(type=int, default=50)),
    ("--top_k", dict(type=int, default=0)),
    ("--top_p", dict(type=float, default=0.8)),
    ("--temperature", dict(type=int, default=1)),
    ("--kvcache", dict(type=int, default=1)),
    ("--seed", dict(type=int, default=1)),
    ("--stop_repetition", dict(type=int, default=2)),
    ("--sample_batch_size", dict(type=int, default=1)),
    ("--cfg_coef", dict(type=float, default=1.5)),
    ("--cfg_stride", dict(type=int, default=1)),
    ("--aug_text", dict(action="store_true")),
    ("--aug_context", dict(action="store_true")),
    ("--use_watermark", dict(action="store_true")),
    ("--tts", dict(action="store_true")),
    ("--prompt_length", dict(type=int, default=3)),
    ("--language", dict(type=str, choices=["en", "zh"])),
    ("--model_path", dict(type=str, default=None)),
    ("--codec_path", dict(type=str, default=None)),
    ("--orig_audio", dict(type=str, default=None)),
    ("--orig_transcript", dict(type=str, default=None)),
    ("--target_transcript", dict(type=str, default=None)),
    ("--temp_folder", dict(type=str, default=None)),
    ("--output_dir", dict(type=str, default=None)),
    ("--savename", dict(type=str, default=None)),
    ("--whisper_model_name", dict(type=str, choices=["base.en", "base"], default="base.en")),
]
# additions of this package (span given explicitly instead of by ASR/alignment)
EXTRA_FLAGS = [
    ("--mask_start", dict(type=float, default=None, help="edit span start in seconds (speech editing)")),
    ("--mask_end", dict(type=float, default=None, help="edit span end in seconds (speech editing)")),
    ("--mask_spans", dict(type=str, default=None, help="speech editing with up to max_n_spans (3) edits: 'a-b,c-d[,e-f]' in seconds — what the "
                                                       "reference derives from the word alignment of the edited transcript (one span per edit)")),
    ("--prompt_end", dict(type=float, default=None, help="--tts: cut the prompt audio at this time in seconds (default --prompt_length)")),
    ("--phoneme_ids", dict(type=str, default=None, help="comma separated phoneme ids of the target transcript (skips espeak)")),
    ("--prompt_phoneme_ids", dict(type=str, default=None, help="comma separated phoneme ids of the prompt transcript")),
    ("--manifest", dict(type=str, default=None, help="--tts only: JSON list of utterances {orig_audio, savename, target_transcript | phoneme_ids, "
                                                     "orig_transcript, prompt_end}; all of them are decoded in lock-step (sharded over the ranks of a "
                                                     "torchrun job) and rendered in one ragged codec pass per rank")),
]


def edit_spans(spans, sub_amount: float, audio_dur: float, codec_sr: int, max_n_spans: int = 3, threshold: float = 0.2):
    """Word-aligned edit intervals [(start, end)] in seconds -> (morphed spans in seconds, mask_interval [M, 2] in codec frames), the way
    the reference turns its alignment into the model's spans (inference_v2.py:284-317): more than `max_n_spans` edits is an error BEFORE
    anything is merged (:284-285); every interval grows by `sub_amount` on both sides, clipped to the file (:307-308); intervals that
    come within `threshold` seconds of each other after sorting by their start become one (:293-305); frames = round(seconds * codec_sr)."""
    if len(spans) > max_n_spans:
        raise RuntimeError(f"Current model only supports maximum {max_n_spans} editings")
    if not spans:
        raise ValueError("no edit span given")
    grown = sorted(([max(float(a) - sub_amount, 0), min(float(b) + sub_amount, audio_dur)] for a, b in spans), key=lambda x: x[0])
    merged = [grown[0]]
    for nxt in grown[1:]:
        if merged[-1][1] >= nxt[0] - threshold:
            merged[-1][1] = max(merged[-1][1], nxt[1])
        else:
            merged.append(nxt)
    return merged, torch.LongTensor([[round(a * codec_sr), round(b * codec_sr)] for a, b in merged])


def parse_mask_spans(text: str):
    """'0.8-1.2,2.5-3.1' -> [(0.8, 1.2), (2.5, 3.1)]"""
    out = []
    for piece in text.split(","):
        piece = piece.strip()
        if not piece:
            continue
        a, sep, b = piece.partition("-")
        if not sep:
            raise SystemExit(f"--mask_spans: '{piece}' is not 'start-end' (seconds)")
        try:
            a, b = float(a), float(b)
        except ValueError:
            raise SystemExit(f"--mask_spans: '{piece}' is not 'start-end' (seconds, both >= 0)")
        if not (0 <= a <= b):
            raise SystemExit(f"--mask_spans: '{piece}' needs 0 <= start <= end")
        out.append((a, b))
    return out


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="inference speech editing")
    for flag, kw in REFERENCE_FLAGS + EXTRA_FLAGS:
        p.add_argument(flag, **kw)
    return p


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def seed_everything(seed: int):
    """inference_v2.py:33-40."""
    os.environ["PYTHONHASHSEED"] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


class _IdTokenizer:
    """Stands in for the phonemiser when ids are given on the command line: yields the ids as 'phonemes'."""

    def __init__(self, table):
        self.table = table

    def __call__(self, texts):
        return [self.table[t.strip()] for t in texts]


def _prompt_16k(path: str, codec_audio_sr: int):
    """--orig_audio as the codec wants it: mono, codec rate (the reference's librosa step, inference_v2.py:216-219)."""
    from .data.tokenizer import read_wav
    wav, sr = read_wav(path)
    if wav.shape[0] > 1:
        wav = wav.mean(0, keepdim=True)
    if sr != codec_audio_sr:
        from .data.resample import resample
        wav, sr = resample(wav, sr, codec_audio_sr).cpu(), codec_audio_sr
    return wav, sr


def run_manifest(args, model, phn2num, audio_tokenizer, device) -> list:
    """`--manifest`: many zero-shot TTS utterances in one job (BASELINE config 4 as a command line). Utterance i is what a single
    `--tts` run with `--seed (seed + i)` produces; the utterances are sharded over the ranks of the process group (if any), decoded in
    lock-step with row refill, their tokens all-gathered, and every rank renders and writes the waveforms of its own shard
    (`dp.synthesize`). Returns the paths this rank wrote."""
    import json
    from . import dp
    from .data.tokenizer import TextTokenizer, tokenize_audio, write_wav
    entries = json.load(open(args.manifest))
    if not args.tts:
        raise SystemExit("--manifest needs --tts (speech editing spans come from one alignment per file)")
    work_dir = args.temp_folder or args.output_dir
    os.makedirs(work_dir, exist_ok=True)
    os.makedirs(args.output_dir, exist_ok=True)
    import torch.distributed as dist
    in_group = dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(), dist.get_rank()) if in_group else (1, 0)
    text_tokenizer = None
    K = int(model.args.n_codebooks)
    # pass 1 (every rank, cheap: no codec call): text ids and prompt length of every entry -> the cost-balanced plan and the bound on
    # every utterance's result length (the all-gather's fixed block layout). Only (path, samples) is kept per entry: a rank holds the
    # waveforms of its OWN shard only (pass 2 reloads them; round 4 kept every prompt of the manifest in memory on every rank).
    ids_all, cut_all, names, costs, caps = [], [], [], [], []
    for i, e in enumerate(entries):
        wav, sr = _prompt_16k(e["orig_audio"], args.codec_audio_sr)
        cut = float(e.get("prompt_end", args.prompt_end if args.prompt_end is not None else args.prompt_length))
        n = int(min(cut, wav.shape[-1] / sr) * sr)
        del wav
        if "phoneme_ids" in e:
            ids = [int(v) for v in e["phoneme_ids"]]
        else:
            if text_tokenizer is None:
                text_tokenizer = TextTokenizer(backend="espeak", language="en-us" if args.language != "zh" else "cmn")
            text = ((e.get("orig_transcript") or "") + " " + e["target_transcript"]).strip()
            ids = [phn2num[p] for p in text_tokenizer([text])[0] if p in phn2num]
        ids_all.append(ids)
        cut_all.append(n)
        names.append(e.get("savename", f"utt{i:05d}"))
        frames = (n + 319) // 320
        costs.append(dp.utterance_cost(len(ids), frames, K))
        caps.append(dp.token_cap(len(ids), frames, 1, K))
    owners = dp.balanced_shards(costs, world)
    # pass 2: only THIS rank's utterances get their prompt loaded, written and encoded (ranks used to write — and race on — the same
    # `{name}_prompt.wav` for every entry and encode all N prompts each). The file keeps the single-run flow's write -> read -> encode
    # round trip (identical prompt codes), under a per-rank name.
    utts = [None] * len(entries)
    for i in owners[rank]:
        wav, sr = _prompt_16k(entries[i]["orig_audio"], args.codec_audio_sr)
        seg = wav[:, :cut_all[i]]
        prompt_fn = os.path.join(work_dir, f"{names[i]}_prompt.wav" if world == 1 else f"{names[i]}_prompt.rank{rank}.wav")
        tmp_fn = prompt_fn + f".tmp{os.getpid()}"
        write_wav(tmp_fn, seg, sr)
        os.replace(tmp_fn, prompt_fn)                                  # atomic: a reader never sees a half-written file
        codes, _scale, _emb = tokenize_audio(audio_tokenizer, prompt_fn)
        frames = codes.shape[-1]
        assert frames <= (cut_all[i] + 319) // 320, (frames, cut_all[i])   # the bound every rank computed in pass 1 holds
        utts[i] = {"x": torch.tensor(ids_all[i], dtype=torch.long).view(1, -1), "y": codes.transpose(2, 1).cpu(),
                   "mask_interval": torch.LongTensor([[[frames, frames]]]), "wav": prompt_fn}
    out_names = [f"{names[i]}_new_seed{args.seed + i}" for i in range(len(entries))]
    waves, mine, _tokens = dp.synthesize(model, audio_tokenizer, utts, seed=args.seed, use_watermark=bool(args.use_watermark), tts=True,
                                         output_dir=args.output_dir, names=out_names, sample_rate=args.codec_audio_sr, costs=costs, caps=caps,
                                         top_k=args.top_k, top_p=args.top_p, temperature=args.temperature, stop_repetition=args.stop_repetition,
                                         cfg_coef=args.cfg_coef, cfg_stride=args.cfg_stride, aug_text=args.aug_text)
    assert list(mine) == list(owners[rank])
    return [os.path.join(args.output_dir, out_names[i] + ".wav") for i in mine]


def main(argv=None):
    args = parse_args(argv)
    seed_everything(args.seed)
    from .data.tokenizer import AudioTokenizer, TextTokenizer, read_wav, write_wav
    from .inference_scale import inference_one_sample, inference_samples
    from .models.ssr import SSR_Speech

    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and torch.cuda.is_available():        # torchrun: this rank's GPU before anything is allocated
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    device = "cuda" if torch.cuda.is_available() else "cpu"
    ckpt = torch.load(args.model_path, map_location="cpu", weights_only=False)      # inference_v2.py:197-204
    model = SSR_Speech(ckpt["config"])
    model.load_state_dict(ckpt["model"])
    config = vars(model.args)
    phn2num = ckpt["phn2num"]
    model.to(device)
    model.eval()
    audio_tokenizer = AudioTokenizer(device=device, signature=args.codec_path)        # :205

    if args.manifest is not None:
        import torch.distributed as dist
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1 and not dist.is_initialized():                                   # launched by torchrun: one rank per GPU over RCCL
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl")
        t0 = time.time()
        written = run_manifest(args, model, phn2num, audio_tokenizer, device)
        print(f"Running time: {time.time() - t0:.4f} s ({len(written)} utterances on this rank)")
        return written

    start_time = time.time()
    os.makedirs(args.output_dir, exist_ok=True)
    wav, sr = read_wav(args.orig_audio)
    if wav.shape[0] > 1:
        wav = wav.mean(0, keepdim=True)                                               # librosa.load(mono=True) of the reference (:218)
    if sr != args.codec_audio_sr:
        # the reference resamples the prompt with librosa (soxr) here (:216-219); this package resamples with the same windowed-sinc
        # filter the tokenizer uses (data/resample.py = torchaudio's Resample): same band limit, not the same samples
        from .data.resample import resample
        wav, sr = resample(wav, sr, args.codec_audio_sr).cpu(), args.codec_audio_sr
    audio_dur = wav.shape[-1] / sr
    # From here on the utterance is the CONVERTED audio (mono, codec rate), in both modes: the reference overwrites `audio_fn` with
    # the 16 kHz mono file it writes to temp_folder (:216-219), and tokenize_audio / the watermark glue / `_orig.wav` all read that
    # file — never the original-rate input (whose 320-sample hops would not be codec frames).
    work_dir = args.temp_folder or args.output_dir
    os.makedirs(work_dir, exist_ok=True)
    if args.tts:
        cut = args.prompt_end if args.prompt_end is not None else float(args.prompt_length)
        cut = min(cut, audio_dur)
        n = int(cut * sr)
        audio_fn = os.path.join(work_dir, f"{args.savename}_prompt.wav")
        write_wav(audio_fn, wav[:, :n], sr)
        frames = round(n / sr * args.codec_sr)
        mask_interval = torch.LongTensor([[frames, frames]])                         # :320-326: empty span at the prompt's end
        prompt_text = args.orig_transcript or ""
        target_text = (prompt_text + " " + args.target_transcript).strip()           # :273
    else:
        if args.mask_spans is not None:
            spans = parse_mask_spans(args.mask_spans)
        elif args.mask_start is not None and args.mask_end is not None:
            spans = [(args.mask_start, args.mask_end)]
        else:
            raise SystemExit("speech editing without WhisperX needs --mask_spans 'a-b[,c-d[,e-f]]' or --mask_start and --mask_end (seconds)")
        audio_fn = os.path.join(work_dir, f"{args.savename}_16k.wav")
        write_wav(audio_fn, wav, sr)
        # :284-317: the alignment's edit intervals -> margins, merge, frames (here the intervals come from the command line)
        morphed_span, mask_interval = edit_spans(spans, args.sub_amount, audio_dur, args.codec_sr, int(config.get("max_n_spans", 3)))
        torch.save(morphed_span, os.path.join(args.output_dir, f"{args.savename}_mask.pt"))
        prompt_text = args.orig_transcript or ""
        target_text = args.target_transcript

    if args.phoneme_ids is not None:
        inv = {v: k for k, v in phn2num.items()}
        table = {target_text.strip(): [inv[int(i)] for i in args.phoneme_ids.split(",") if i.strip()]}
        table[prompt_text.strip()] = [inv[int(i)] for i in (args.prompt_phoneme_ids or "").split(",") if i.strip()]
        text_tokenizer = _IdTokenizer(table)
    else:
        text_tokenizer = TextTokenizer(backend="espeak", language="en-us" if args.language != "zh" else "cmn")
    decode_config = {"top_k": args.top_k, "top_p": args.top_p, "temperature": args.temperature, "stop_repetition": args.stop_repetition,
                     "kvcache": args.kvcache, "codec_audio_sr": args.codec_audio_sr, "codec_sr": args.codec_sr}
    shutil.copyfile(audio_fn, os.path.join(args.output_dir, f"{args.savename}_orig.wav"))   # :357-358 (the prompt actually used)
    common = (model, argparse.Namespace(**config), phn2num, text_tokenizer, audio_tokenizer, audio_fn, prompt_text, target_text, mask_interval,
              args.cfg_coef, args.cfg_stride, args.aug_text, args.aug_context, args.use_watermark, args.tts, device, decode_config)
    seeds = [args.seed + num for num in range(args.sample_batch_size)]               # :331-332: sample `num` runs under seed + num
    if len(seeds) > 1:
        # all samples of the utterance in ONE lock-step decode (same outputs as the reference's sequential loop :331-358)
        seed_everything(seeds[-1])
        waves = inference_samples(*common, seeds=seeds)
    else:
        seed_everything(seeds[0])
        waves = [inference_one_sample(*common)]
    for s_, new_audio in zip(seeds, waves):
        write_wav(os.path.join(args.output_dir, f"{args.savename}_new_seed{s_}.wav"), new_audio[0].cpu(), args.codec_audio_sr)
    print(f"Running time: {time.time() - start_time:.4f} s")                         # :360-363


if __name__ == "__main__":
    main()
