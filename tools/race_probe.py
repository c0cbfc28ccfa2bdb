"""Round 5: what makes the multi-stream codec stress test fail (once in ~6 processes, always in its FIRST concurrent round, always the
detector output `mark` of the middle caller)? One process, many "first rounds": per mode, N trials of one concurrent round of
encode -> decode -> wmdecode on three caller streams against the same calls made alone.
  python tools/race_probe.py [trials]
modes: base        = fresh caller streams + fresh side streams + emptied allocator cache before every trial (a first round's conditions)
       nopipe      = base with the two-stream LSTM pipeline off (LSTM_CHUNK above T)
       keepcache   = fresh streams, allocator cache kept
       keepstreams = streams kept, allocator cache emptied
       warm        = streams and cache kept (a later round's conditions)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ssr_speech_amd  # noqa: E402,F401
from ssr_speech_amd import weights as W  # noqa: E402
from ssr_speech_amd.codec.wmencodec import WMEncodecModel  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = W.codec_config_full()
m = WMEncodecModel(cfg, W.codec_state_dict(cfg, seed=21), "cuda")
g = torch.Generator().manual_seed(19)
n = cfg.hop * 70 + 11
Bs = (9, 7, 9)
wavs = [(torch.randn(b, 1, n, generator=g) * 0.2).cuda() for b in Bs]
labels = [torch.randint(0, 2, (b, 71), generator=g).cuda() for b in Bs]
tracks = [torch.nn.functional.pad(w, (0, 71 * cfg.hop - n)) for w in wavs]
names = ("codes", "emb", "dec", "wm", "mark")


def call(i):
    codes, _, emb = m.encode(wavs[i])
    dec = m.decode(codes)
    wm, mark = m.wmdecode(codes, labels[i], tracks[i])
    return codes, emb, dec, wm, mark


alone = [call(i) for i in range(3)]
torch.cuda.synchronize()
chunk0 = m.LSTM_CHUNK
for mode in ("base", "nopipe", "keepcache", "keepstreams", "warm", "base"):
    m.LSTM_CHUNK = 10 ** 9 if mode == "nopipe" else chunk0
    streams = [torch.cuda.Stream() for _ in range(3)]
    fails = []
    for trial in range(trials):
        if mode in ("base", "nopipe", "keepcache"):
            streams = [torch.cuda.Stream() for _ in range(3)]
            m._side_streams.clear()
            m._keep.clear()
        if mode in ("base", "nopipe", "keepstreams"):
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        got = [None] * 3
        for i in (0, 1, 2):
            streams[i].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(streams[i]):
                got[i] = call(i)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        torch.cuda.synchronize()
        for i in range(3):
            for k, (a, b) in enumerate(zip(alone[i], got[i])):
                if not torch.equal(a, b):
                    idx = (a != b).nonzero()
                    fails.append(f"trial {trial} caller {i} {names[k]}: {idx.shape[0]} of {a.numel()} differ, max {float((a.float() - b.float()).abs().max()):.3g}, first {idx[0].tolist()}")
        del got
    print(f"mode {mode:12s}: {len(fails)} failing outputs in {trials} trials")
    for f in fails[:6]:
        print("    " + f)
