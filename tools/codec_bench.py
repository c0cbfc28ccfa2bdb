"""wmencodec throughput on one MI355X (BASELINE config 5 shape: 16 kHz x 30 s clips, full SEANet config, synthetic weights):
encode (SEANet encoder + LSTM + RVQ search) and decode (dequant + LSTM + SEANet decoder), per chunk of B clips.
Usage: python tools/codec_bench.py [B] [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ssr_speech_amd  # noqa: E402,F401
from ssr_speech_amd import weights as W  # noqa: E402
from ssr_speech_amd.codec.wmencodec import WMEncodecModel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
cfg = W.codec_config_full()
m = WMEncodecModel(cfg, W.codec_state_dict(cfg, seed=0), "cuda")
g = torch.Generator().manual_seed(0)
wav = (torch.randn(B, 1, int(secs * 16000), generator=g) * 0.1).cuda()
for _ in range(1):
    codes, _, emb = m.encode(wav)
    out = m.decode(codes)
torch.cuda.synchronize()
t0 = time.perf_counter(); codes, _, emb = m.encode(wav); torch.cuda.synchronize(); t1 = time.perf_counter()
out = m.decode(codes); torch.cuda.synchronize(); t2 = time.perf_counter()
audio_s = B * secs
GF = 6.97      # GFLOP per audio-second, encode and decode each
print(f"B={B} x {secs:.0f}s: encode {1000*(t1-t0):.1f} ms ({audio_s/(t1-t0):.0f} audio-s/s, {GF*audio_s/(t1-t0)/1e3:.1f} TFLOP/s) | "
      f"decode {1000*(t2-t1):.1f} ms ({audio_s/(t2-t1):.0f} audio-s/s, {GF*audio_s/(t2-t1)/1e3:.1f} TFLOP/s) | peak mem {torch.cuda.max_memory_allocated()/1e9:.1f} GB")
