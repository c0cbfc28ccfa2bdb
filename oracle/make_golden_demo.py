"""Fixture for BASELINE config 3 ("English 830M speech editing, demo/84_121550_000074_000000.wav single-span edit"): the reference's
demo prompt as a DATA file under tests/golden/ (16-bit PCM, what the reference's own pipeline writes after its 16 kHz conversion,
inference_v2.py:216-219) plus the facts the survey recorded about it (126,880 samples -> 397 codec frames of 320).
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
    print("wrote", OUT)
