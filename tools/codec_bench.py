"""wmencodec throughput on one MI355X (BASELINE config 5 shape: 16 kHz x 30 s clips, full SEANet config, synthetic weights):
encode (SEANet encoder + LSTM + RVQ search), decode (dequant + LSTM + SEANet decoder) and — with `wm` — wmdecode (skip encoder +
label-conditioned decoder, with and without the detector pass; marks = second half ones), per chunk of B clips.
Usage: python tools/codec_bench.py [B] [seconds] [wm] [lanes]"""
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
if len(sys.argv) > 3 and sys.argv[3] == "wm":
    if len(sys.argv) > 4:
        m.lanes = int(sys.argv[4])
    Tf = codes.shape[-1]
    marks = torch.zeros(B, Tf, dtype=torch.long, device="cuda")
    marks[:, Tf // 2:] = 1
    del out, emb
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    for with_mark, gf in ((False, 14.5), (True, 21.69)):            # GFLOP per audio-second
        m.wmdecode(codes, marks, wav, with_mark=with_mark)           # untimed pass: allocator holds the blocks
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w, mk = m.wmdecode(codes, marks, wav, with_mark=with_mark)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print(f"B={B} x {secs:.0f}s: wmdecode(with_mark={with_mark}, lanes={m.lanes}) {1000*(t1-t0):.1f} ms ({audio_s/(t1-t0):.0f} audio-s/s, "
              f"{gf*audio_s/(t1-t0)/1e3:.1f} TFLOP/s of {gf} GFLOP/audio-s) | peak mem {torch.cuda.max_memory_allocated()/1e9:.1f} GB")
        del w, mk
