"""Fixture for BASELINE config 3 ("English 830M speech editing, demo/84_121550_000074_000000.wav single-span edit"): the reference's
demo prompt as a DATA file under tests/golden/ (16-bit PCM, what the reference's own pipeline writes after its 16 kHz conversion,
inference_v2.py:216-219) plus the facts the survey recorded about it (126,880 samples -> 397 codec frames of 320).
Round 6 adds the prompt BASELINE configs 1-2 name, demo/5895_34622_000026_000002.wav: its first 160 frames (51,200 samples = 3.2 s, cut on
a 320 multiple as SURVEY 8(d) config 1 prescribes: "no aligner available => fixed 160 frames") as 16-bit PCM.
Run in the build container only (needs /root/reference):  python oracle/make_golden_demo.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssr_speech_amd  # noqa: E402,F401
from ssr_speech_amd.data.tokenizer import read_wav, write_wav  # noqa: E402

SRC = "/root/reference/demo/84_121550_000074_000000.wav"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

if __name__ == "__main__":
    wav, sr = read_wav(SRC)
    assert sr == 16000 and wav.shape == (1, 126880), (sr, wav.shape)
    write_wav(os.path.join(OUT, "demo_84_121550_000074_000000.wav"), wav, sr)
    back, _ = read_wav(os.path.join(OUT, "demo_84_121550_000074_000000.wav"))
    json.dump({"source": "demo/84_121550_000074_000000.wav of the reference repository (audio data, 16 kHz mono)", "samples": int(wav.shape[1]),
               "frames_320": (int(wav.shape[1]) + 319) // 320, "sample_rate": sr, "peak": round(float(wav.abs().max()), 4),
               "max_abs_quantisation_error": float((back - wav).abs().max())},
              open(os.path.join(OUT, "demo_84_121550_000074_000000.json"), "w"), indent=1)
    # configs 1-2: the TTS demo prompt, first 160 frames
    src2 = "/root/reference/demo/5895_34622_000026_000002.wav"
    wav2, sr2 = read_wav(src2)
    assert sr2 == 16000 and wav2.shape == (1, 125920), (sr2, wav2.shape)
    cut = wav2[:, : 160 * 320]
    name2 = "demo_5895_34622_000026_000002_160f"
    write_wav(os.path.join(OUT, name2 + ".wav"), cut, sr2)
    back2, _ = read_wav(os.path.join(OUT, name2 + ".wav"))
    json.dump({"source": "demo/5895_34622_000026_000002.wav of the reference repository (audio data, 16 kHz mono), samples [0, 51200)",
               "source_samples": int(wav2.shape[1]), "samples": int(cut.shape[1]), "frames_320": 160, "sample_rate": sr2,
               "peak": round(float(cut.abs().max()), 4), "max_abs_quantisation_error": float((back2 - cut).abs().max())},
              open(os.path.join(OUT, name2 + ".json"), "w"), indent=1)
    print("wrote", OUT)
