"""Does any codec kernel READ workspace memory that no producer wrote? (round 5: the multi-stream stress test failed once inside a full
suite run and ten times not in a fresh process — results that depend on what the allocator's blocks held before are the usual cause.)
Every call is made twice: with plain torch.empty workspaces and with SSRHIP_POISON_ALLOC=1 (activation / state buffers pre-filled with
NaN; `wmencodec._empty`). Outputs must be finite and identical.
  python tools/poison_check.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ssr_speech_amd  # noqa: E402,F401
from ssr_speech_amd import weights as W  # noqa: E402
from ssr_speech_amd.codec.wmencodec import WMEncodecModel  # noqa: E402

cfg = W.codec_config_full()
m = WMEncodecModel(cfg, W.codec_state_dict(cfg, seed=21), "cuda")
g = torch.Generator().manual_seed(19)
bad = 0
for B, n in ((7, cfg.hop * 70 + 11), (9, cfg.hop * 70 + 11), (3, cfg.hop * 100), (1, cfg.hop * 33 + 5), (17, cfg.hop * 64)):
    T = (n + cfg.hop - 1) // cfg.hop
    wav = (torch.randn(B, 1, n, generator=g) * 0.2).cuda()
    label = torch.randint(0, 2, (B, T), generator=g).cuda()
    track = torch.nn.functional.pad(wav, (0, T * cfg.hop - n))

    def call():
        codes, _, emb = m.encode(wav)
        dec = m.decode(codes)
        wm, mark = m.wmdecode(codes, label, track)
        torch.cuda.synchronize()
        return codes, emb, dec, wm, mark

    os.environ["SSRHIP_POISON_ALLOC"] = "0"
    plain = call()
    os.environ["SSRHIP_POISON_ALLOC"] = "1"
    pois = call()
    os.environ["SSRHIP_POISON_ALLOC"] = "0"
    for name, a, b in zip(("codes", "emb", "dec", "wm", "mark"), plain, pois):
        nan = int((~torch.isfinite(b.float())).sum())
        diff = int((a != b).sum())
        flag = "" if (nan == 0 and diff == 0) else "   <-- uninitialised read"
        bad += bool(flag)
        where = ""
        if diff:
            idx = (a != b).nonzero()
            where = f" first {idx[0].tolist()} last {idx[-1].tolist()}"
        print(f"B={B:3d} n={n:6d} T={T:4d} {name:6s} shape {tuple(a.shape)}: non-finite with poison {nan}, elements differing {diff}{where}{flag}")
print("RESULT:", "clean" if bad == 0 else f"{bad} outputs depend on unwritten memory")
