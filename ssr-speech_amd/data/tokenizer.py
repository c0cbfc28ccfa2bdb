"""`AudioTokenizer`, `tokenize_audio`, `TextTokenizer`, `tokenize_text` — the reference's `data/tokenizer.py:31-159`
surface on top of the HIP codec. Audio I/O is a minimal RIFF/WAVE reader (torchaudio is not a dependency);
text phonemisation needs the external `phonemizer`/espeak-ng stack exactly like the reference (out of scope here:
any callable `tokenizer([text]) -> [[phonemes]]` can be passed instead).
"""
from __future__ import annotations

import struct
from typing import Any, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from ..codec.wmencodec import WMEncodecModel
from ..weights import CodecConfig


# ----------------------------------------------------------------------------- checkpoint / config reading
def _cfg_get(cfg, dotted: str, default=None):
    """Duck-typed read of an OmegaConf DictConfig / dict / namespace (omegaconf is not required)."""
    cur = cfg
    for part in dotted.split("."):
        if cur is None:
            return default
        if isinstance(cur, dict):
            cur = cur.get(part, None)
        else:
            try:
                cur = getattr(cur, part)
            except Exception:
                try:
                    cur = cur[part]
                except Exception:
                    return default
    return default if cur is None else cur


def codec_config_from_xp_cfg(cfg) -> CodecConfig:
    """Fields the reference's builder reads (`audiocraft/models/builders.py:100-115`, `solvers/wmcompression.py:281-315`)."""
    cm = _cfg_get(cfg, "compression_model", "wmencodec")
    if cm != "wmencodec":
        raise KeyError(f"Unexpected compression model {cm}")          # builders.py:114
    # Architecture switches this implementation does not have: refuse them instead of silently producing wrong audio / codes
    # (the reference ships e.g. config/model/encodec/encodec_base_causal.yaml). Absent keys mean the reference's defaults.
    fixed = [("encodec.causal", False), ("seanet.causal", False), ("encodec.renormalize", False), ("seanet.true_skip", True),
             ("seanet.norm", "weight_norm"), ("seanet.activation", "ELU"), ("seanet.dilation_base", 2), ("seanet.decoder.final_activation", None),
             ("seanet.decoder.trim_right_ratio", 1.0), ("seanet.n_residual_layers", 1), ("seanet.disable_norm_outer_blocks", 0)]
    for key, want in fixed:
        got = _cfg_get(cfg, key, want)
        if got != want and not (want is None and got in ("", "null", "None")):
            raise NotImplementedError(f"codec checkpoint has {key}={got!r}; ssr_speech_amd implements {key}={want!r} only")
    ratios = _cfg_get(cfg, "seanet.ratios", [8, 5, 4, 2])
    return CodecConfig(
        channels=int(_cfg_get(cfg, "channels", 1)), dimension=int(_cfg_get(cfg, "seanet.dimension", 128)),
        n_filters=int(_cfg_get(cfg, "seanet.n_filters", 64)), n_residual_layers=int(_cfg_get(cfg, "seanet.n_residual_layers", 1)),
        ratios=tuple(int(r) for r in ratios), kernel_size=int(_cfg_get(cfg, "seanet.kernel_size", 7)),
        residual_kernel_size=int(_cfg_get(cfg, "seanet.residual_kernel_size", 3)), last_kernel_size=int(_cfg_get(cfg, "seanet.last_kernel_size", 7)),
        compress=int(_cfg_get(cfg, "seanet.compress", 2)), lstm=int(_cfg_get(cfg, "seanet.lstm", 2)),
        pad_mode=str(_cfg_get(cfg, "seanet.pad_mode", "constant")), n_q=int(_cfg_get(cfg, "rvq.n_q", 4)), bins=int(_cfg_get(cfg, "rvq.bins", 2048)),
        sample_rate=int(_cfg_get(cfg, "sample_rate", 16000)))


def load_codec_checkpoint(path: str) -> Tuple[CodecConfig, dict]:
    """Accepts the reference's layout {'xp.cfg': DictConfig, 'best_state': {'model': state_dict}} (unpickling it needs
    `omegaconf` installed, as in the reference) and this repo's plain layout {'codec_config': dict, 'model': state_dict}."""
    state = torch.load(path, "cpu", weights_only=False)
    assert state is not None and "exported" not in state, "When loading an exported checkpoint, use the //pretrained/ prefix."  # wmcompression.py:303-304
    if "xp.cfg" in state:
        return codec_config_from_xp_cfg(state["xp.cfg"]), state["best_state"]["model"]
    return CodecConfig(**state["codec_config"]), state["model"]


# ----------------------------------------------------------------------------- audio I/O
def read_wav(path: str, frame_offset: int = 0, num_frames: int = -1) -> Tuple[torch.Tensor, int]:
    """Minimal RIFF/WAVE reader -> (float32 [channels, n] in [-1,1], sample_rate), like torchaudio.load(normalize=True).
    PCM 8/16/32-bit and IEEE float32."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, raw = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            raw = body
        pos += 8 + size + (size & 1)
    if fmt is None or raw is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, ch, sr, bits = fmt
    if tag == 3 and bits == 32:
        x = np.frombuffer(raw, dtype="<f4").astype(np.float32)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif tag == 1 and bits == 8:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag={tag} bits={bits}")
    x = x[: (len(x) // ch) * ch].reshape(-1, ch).T
    if frame_offset > 0:
        x = x[:, frame_offset:]
    if num_frames is not None and num_frames >= 0:
        x = x[:, :num_frames]
    return torch.from_numpy(np.ascontiguousarray(x)), sr


def write_wav(path: str, wav: torch.Tensor, sample_rate: int) -> None:
    """float32 [channels, n] -> 16-bit PCM WAVE (what torchaudio.save writes by default for float input is float32;
    16-bit keeps files small and portable)."""
    x = wav.detach().cpu().to(torch.float32).clamp(-1, 1).numpy()
    pcm = (x.T * 32767.0).round().astype("<i2").tobytes()
    ch = x.shape[0]
    hdr = b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, ch, sample_rate, sample_rate * ch * 2, ch * 2, 16)
    with open(path, "wb") as f:
        f.write(hdr + b"data" + struct.pack("<I", len(pcm)) + pcm)


def convert_audio(wav: torch.Tensor, sr: int, target_sr: int, target_channels: int) -> torch.Tensor:
    """Channel layout, then sample rate (the contract of data/tokenizer.py:82-97): stereo -> mono is the mean of the two channels, mono ->
    n channels repeats the samples (as views), anything else keeps its channels; the rate conversion is this package's GPU resampler
    (`torchaudio.transforms.Resample(sr, target_sr)` in the reference, :96), skipped when there is nothing to convert."""
    n_in = int(wav.shape[0])
    if n_in not in (1, 2):
        raise AssertionError("Audio must be mono or stereo.")
    if n_in == 2 and target_channels == 1:
        mixed = wav.mean(dim=0, keepdim=True)
    elif n_in == 1 and target_channels > 1:
        mixed = wav.expand(target_channels, wav.shape[-1])
    else:
        mixed = wav
    if sr == target_sr:
        return mixed
    from .resample import resample
    return resample(mixed, sr, target_sr)


# ----------------------------------------------------------------------------- tokenizers
class AudioTokenizer:
    """EnCodec audio (data/tokenizer.py:99-138)."""

    def __init__(self, device: Any = None, signature=None, *, config: Optional[CodecConfig] = None, state_dict: Optional[dict] = None) -> None:
        if signature is not None:
            config, state_dict = load_codec_checkpoint(signature)
        if config is None or state_dict is None:
            raise ValueError("AudioTokenizer needs `signature=<checkpoint path>` or (`config`, `state_dict`)")
        if not device:
            device = torch.device("cpu")
            if torch.cuda.is_available():
                device = torch.device("cuda:0")
        self._device = torch.device(device)
        self.codec = WMEncodecModel(config, state_dict, self._device)
        self.sample_rate = self.codec.sample_rate
        self.channels = self.codec.channels

    @property
    def device(self):
        return self._device

    def encode(self, wav: torch.Tensor):
        return self.codec.encode(wav.to(self.device))

    def decode(self, frames: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
        return self.codec.decode(frames, scale)

    def wmdecode(self, frames: torch.Tensor, marks: torch.Tensor, wav: torch.Tensor, scale: torch.Tensor):
        # the reference discards the detector output here (tokenizer.py:133): skip computing it
        out, _ = self.codec.wmdecode(frames.to(self.device), marks.to(self.device), wav.to(self.device), scale, with_mark=False)
        return out

    def detect_watermark(self, wav: torch.Tensor):
        return self.codec.detect_watermark(wav.to(self.device))

    # -- several utterances of different lengths in one pass (not in the reference, which decodes one at a time: inference_v2.py:331-358)
    def decode_batch(self, frames: Sequence[torch.Tensor], scale=None) -> List[torch.Tensor]:
        """frames[i]: [1, K, T_i] -> list of [1, 1, T_i * 320]; item i equals `decode(frames[i], scale)`."""
        return self.codec.decode_ragged(list(frames), scale)

    def wmdecode_batch(self, frames: Sequence[torch.Tensor], marks: Sequence[torch.Tensor], wavs: Sequence[torch.Tensor], scale=None) -> List[torch.Tensor]:
        """item i equals `wmdecode(frames[i], marks[i], wavs[i], scale)`."""
        out, _ = self.codec.wmdecode_ragged(list(frames), list(marks), list(wavs), scale, with_mark=False)
        return out


def tokenize_audio(tokenizer: AudioTokenizer, audio_path: str, offset=-1, num_frames=-1, multiple=320):
    """File (or the window [offset, offset + num_frames) of it) -> (codes [1, K, T'], scale, emb) (data/tokenizer.py:141-159). The samples
    are placed at the front of a zero buffer whose length is the next multiple of `multiple` (whole codec hops, before any channel /
    rate conversion, as the reference pads), converted to the codec's layout and encoded as a batch of one."""
    window = {"frame_offset": offset, "num_frames": num_frames} if (offset != -1 and num_frames != -1) else {}
    samples, sr = read_wav(audio_path, **window)
    n = int(samples.shape[-1])
    whole = -(-n // multiple) * multiple
    if whole != n:
        buf = samples.new_zeros(samples.shape[:-1] + (whole,))
        buf[..., :n] = samples
        samples = buf
    batch = convert_audio(samples, sr, tokenizer.sample_rate, tokenizer.channels)[None]
    with torch.no_grad():
        return tokenizer.encode(batch)


def split_phonemized(phonemized: str, word_sep: str = "_", phone_sep: str = "|") -> List[str]:
    """One phonemized utterance -> the symbol list the LM's phoneme table is indexed by (what the reference's tokenizer returns,
    data/tokenizer.py:59-77): inside every word a maximal run of letters / digits / modifier letters is ONE symbol (a phone such as
    `iː`), every other non-blank character a symbol of its own (punctuation), phone separators vanish, and the word separator itself
    stands between two words as a symbol. A character scanner instead of a regular expression: no dependency on `re`'s Unicode tables
    beyond `str.isalnum`, and usable (and tested, tests/test_host_surface.py) without espeak installed."""
    symbols: List[str] = []
    words = phonemized.split(word_sep)
    for wi, word in enumerate(words):
        run = ""
        for ch in word:
            if ch.isalnum() or ch == "_":                 # part of a phone
                run += ch
                continue
            if run:
                symbols.append(run)
                run = ""
            if not ch.isspace() and ch != phone_sep:      # a punctuation mark (the phone separator only delimits)
                symbols.append(ch)
        if run:
            symbols.append(run)
        if wi + 1 < len(words):
            symbols.append(word_sep)
    kept = sum(len(sym) for sym in symbols)
    if kept != len(phonemized) - phonemized.count(phone_sep) - sum(ch.isspace() for ch in phonemized):
        raise ValueError(f"phonemizer output does not round-trip: {phonemized!r}")
    return symbols


class TextTokenizer:
    """Text -> phoneme symbols through the external `phonemizer` / espeak-ng stack (reference data/tokenizer.py:31-80). Only the
    backend call needs that stack; the symbol splitting is `split_phonemized` above. The CLI's `--phoneme_ids` bypasses this class."""

    WORD_SEP, SYLLABLE_SEP, PHONE_SEP = "_", "-", "|"

    def __init__(self, language="en-us", backend="espeak", separator=None, preserve_punctuation=True, punctuation_marks=None,
                 with_stress=False, tie=False, language_switch="keep-flags", words_mismatch="ignore") -> None:
        try:
            from phonemizer.backend import EspeakBackend
            from phonemizer.punctuation import Punctuation
            from phonemizer.separator import Separator
        except ImportError as e:
            raise RuntimeError("TextTokenizer needs the `phonemizer` package and espeak-ng (reference data/tokenizer.py:8-10); "
                               "pass phoneme ids directly (--phoneme_ids) on a box without them") from e
        if backend != "espeak":
            raise ValueError("only the espeak backend is wired up (as in the reference)")
        self.separator = separator if separator is not None else Separator(word=self.WORD_SEP, syllable=self.SYLLABLE_SEP, phone=self.PHONE_SEP)
        marks = Punctuation.default_marks() if punctuation_marks is None else punctuation_marks
        self.backend = EspeakBackend(language, punctuation_marks=marks, preserve_punctuation=preserve_punctuation, with_stress=with_stress,
                                     tie=tie, language_switch=language_switch, words_mismatch=words_mismatch)

    def __call__(self, text, strip=True) -> List[List[str]]:
        batch = [text] if isinstance(text, str) else list(text)
        lines = self.backend.phonemize(batch, separator=self.separator, strip=strip, njobs=1)
        return [split_phonemized(line, self.separator.word, self.separator.phone) for line in lines]


def tokenize_text(tokenizer, text: str) -> List[str]:
    """data/tokenizer.py:93-96."""
    phonemes = tokenizer([text.strip()])
    return phonemes[0]
