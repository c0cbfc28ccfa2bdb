"""Round 5: the multi-stream failure was localised to ONE 64-row tile of a split-GEMM launch in the detector's encoder. This
lab replays exactly those launches under load: a solo detector pass records the inputs of every `_conv` call at the 2840 -> 568 and 568-frame
levels (the convolutions that run as gemm_split_dma_kernel<64, ..>); each recorded call is then repeated R times on stream A while
streams B and C run whole encode / decode / wmdecode passes of other batches, and every output is compared bit for bit with the solo
result. Mismatches are reported with the tile coordinates (item, 64-row tile, 128-column tile).
  python tools/gemm_race_lab.py [rounds] [repeats per round]
"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ssr_speech_amd  # noqa: E402,F401
from ssr_speech_amd import weights as W  # noqa: E402
from ssr_speech_amd.codec.wmencodec import WMEncodecModel  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg = W.codec_config_full()
m = WMEncodecModel(cfg, W.codec_state_dict(cfg, seed=21), "cuda")
g = torch.Generator().manual_seed(19)
n = cfg.hop * 70 + 11
Bs = (9, 7, 9)
wavs = [(torch.randn(b, 1, n, generator=g) * 0.2).cuda() for b in Bs]
labels = [torch.randint(0, 2, (b, 71), generator=g).cuda() for b in Bs]
tracks = [torch.nn.functional.pad(w, (0, 71 * cfg.hop - n)) for w in wavs]

# ---- record the detector's GEMM-shaped convolutions on caller 1's watermarked audio
codes1 = m.encode(wavs[1])[0]
wm1, _ = m.wmdecode(codes1, labels[1], tracks[1])
torch.cuda.synchronize()
recorded = []
orig_conv = m._conv


def spy(c, x, nxt, R=None, post_elu=False):
    out = orig_conv(c, x, nxt, R=R, post_elu=post_elu)
    if x.T in (2840, 568) or out.T == 568:
        xc = copy.copy(x)
        xc.data = x.data.clone()
        Rc = None
        if R is not None:
            Rc = copy.copy(R)
            Rc.data = R.data.clone()
        recorded.append((c, xc, nxt, Rc, post_elu, out.data.clone()))
    return out


m._conv = spy
mk = m._run(m.wm_encoder.nodes, m._input_tm(wm1, m.wm_encoder.nodes[0]), after=None)
m._conv = orig_conv
torch.cuda.synchronize()
print(f"recorded {len(recorded)} convolution calls: " + ", ".join(f"[{x.T}x{c.Cin}*{c.k}/s{c.s} -> {c.Cout}{' +R' if R is not None else ''}]" for c, x, _, R, _, _ in recorded))

sA, sB, sC = (torch.cuda.Stream() for _ in range(3))
total = bad = 0
for rnd in range(rounds):
    outs = []
    for st in (sA, sB, sC):
        st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(sB):
        cb = m.encode(wavs[2])[0]
        db = m.decode(cb)
        wb = m.wmdecode(cb, labels[2], tracks[2])
    with torch.cuda.stream(sC):
        cc = m.encode(wavs[0])[0]
        wc = m.wmdecode(cc, labels[0], tracks[0])
    with torch.cuda.stream(sA):
        for r in range(reps):
            for idx, (c, x, nxt, R, post, ref) in enumerate(recorded):
                outs.append((idx, orig_conv(c, x, nxt, R=R, post_elu=post)))
    for st in (sA, sB, sC):
        torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    for idx, o in outs:
        total += 1
        ref = recorded[idx][5]
        if not torch.equal(o.data, ref):
            bad += 1
            d = (o.data != ref)
            where = d.nonzero()
            items = sorted(set(where[:, 0].tolist()))
            rows = where[:, 1] - o.padL
            cols = where[:, 2]
            print(f"round {rnd}: call {idx} ({recorded[idx][1].T} rows in): {int(d.sum())} values differ, max {float((o.data - ref).abs().max()):.3g}; items {items}, "
                  f"rows {int(rows.min())}..{int(rows.max())} (64-row tiles {int(rows.min()) // 64}..{int(rows.max()) // 64}), columns {int(cols.min())}..{int(cols.max())}")
    del outs
print(f"RESULT: {bad} of {total} replayed launches differ from their solo result")
