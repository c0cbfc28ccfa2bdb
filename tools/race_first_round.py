"""Round 5: one process = the calls alone, then ONE concurrent round (the only round the stress test ever failed in), with a map of where
`mark` differs if it does. Run many processes: `for i in $(seq 12); do python tools/race_first_round.py; done`."""
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
n = cfg.hop * 70 + 11
Bs = (9, 7, 9)
wavs = [(torch.randn(b, 1, n, generator=g) * 0.2).cuda() for b in Bs]
labels = [torch.randint(0, 2, (b, 71), generator=g).cuda() for b in Bs]
tracks = [torch.nn.functional.pad(w, (0, 71 * cfg.hop - n)) for w in wavs]
names = ("codes", "emb", "dec", "wm", "mark")
if len(sys.argv) > 1 and sys.argv[1] == "nopipe":
    m.LSTM_CHUNK = 10 ** 9


# every layer output of the DETECTOR pass (wm_encoder) is kept, for the calls alone and for the concurrent round: if `mark` differs, the
# first layer whose output differs names the kernel (the buffers are only referenced, not copied: no extra launches)
orig_run = m._run
rec = None


def run_rec(nodes, x, after=None):
    if nodes is not m.wm_encoder.nodes or rec is None:
        return orig_run(nodes, x, after)
    for idx in range(len(nodes)):
        x = orig_run(nodes[idx: idx + 1], x, after=(nodes[idx + 1] if idx + 1 < len(nodes) else after))
        rec.append((nodes[idx][1], x))
    return x


m._run = run_rec


def call(i):
    codes, _, emb = m.encode(wavs[i])
    dec = m.decode(codes)
    wm, mark = m.wmdecode(codes, labels[i], tracks[i])
    return codes, emb, dec, wm, mark


alone, layers_alone = [], []
for i in range(3):
    rec = []
    alone.append(call(i))
    layers_alone.append(rec)
rec = None
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(3)]
got = [None] * 3
layers_got = []
for i in (0, 1, 2):
    streams[i].wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(streams[i]):
        rec = []
        got[i] = call(i)
        layers_got.append(rec)
rec = None
for st in streams:
    torch.cuda.current_stream().wait_stream(st)
torch.cuda.synchronize()
ok = True
for i in range(3):
    for k, (a, b) in enumerate(zip(alone[i], got[i])):
        if not torch.equal(a, b):
            ok = False
            d = (a.float() - b.float()).abs()
            print(f"FAIL caller {i} {names[k]} shape {tuple(a.shape)}: {int((d > 0).sum())} differ, max {float(d.max()):.3g}")
            if d.dim() == 3 and names[k] == "mark":
                for bi in range(d.shape[0]):
                    per_t = d[bi].amax(dim=-1)
                    nz = (per_t > 0).nonzero().flatten().tolist()
                    print(f"    item {bi}: frames differing {len(nz)} of {per_t.numel()}, first {nz[0] if nz else None}, last {nz[-1] if nz else None}, "
                          f"max at t={int(per_t.argmax())} ({float(per_t.max()):.3g}); per-frame max (x1e4): {' '.join(f'{v * 1e4:.0f}' for v in per_t.tolist())}")
if not ok:
    for i in range(3):
        for li, ((kind, a), (_, b)) in enumerate(zip(layers_alone[i], layers_got[i])):
            da, db = a.data, b.data
            if da.shape == db.shape and not torch.equal(da, db):
                w = (da != db).nonzero()
                rows = w[:, 1] - a.padL
                print(f"    caller {i}: FIRST differing detector layer = #{li} ({kind}), output [{a.B}][{a.T}][{a.C}]: {w.shape[0]} values differ, max "
                      f"{float((da - db).abs().max()):.3g}; items {sorted(set(w[:, 0].tolist()))}, rows {int(rows.min())}..{int(rows.max())}, channels {int(w[:, 2].min())}..{int(w[:, 2].max())}; "
                      f"layers before it: {[k for k, _ in layers_alone[i][:li]]}")
                seen = {}
                for it, r, ch in w.tolist():
                    seen.setdefault((it, r), []).append(ch)
                for (it, r), chs in list(seen.items())[:5]:
                    print(f"        item {it} row {r - a.padL}: channels {chs}; alone {[round(float(da[it, r, c_]), 4) for c_ in chs[:6]]} concurrent {[round(float(db[it, r, c_]), 4) for c_ in chs[:6]]}")
                break
print("first concurrent round:", "identical" if ok else "DIFFERENT")
