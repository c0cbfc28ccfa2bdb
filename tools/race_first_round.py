"""Round 5: one process = the calls alone, then ONE concurrent round (the only round the stress test ever failed in), with a map of where
`mark` differs if it does. Run many processes: `for i in $(seq 12); do python tools/race_first_round.py; done`, or, round 6, through
`tools/race_trials.py` (arms x trials, failures / trials with a Wilson interval). Round-6 switches (each names ONE suspected trigger):
  prewarm-queues=N   create N streams, run a kernel on each and synchronize BEFORE any codec work: every hardware queue HIP will ever
                     give this process exists before the first codec kernel runs (no run-list rebuild under load)
  early-streams      create the three caller streams and the model's side streams (and touch them) before the calls made alone
  warm-alloc         before the concurrent round, allocate and free on every caller stream what a call allocates (no kernels):
                     the round runs without a single hipMalloc
  bg-malloc          (with warm-alloc) a host thread maps fresh device memory (torch.empty of ever new sizes on a private stream, never
                     touched by a kernel) for as long as the round is being enqueued and run: is hipMalloc BESIDE running kernels enough?
  rounds=N           N - 1 more concurrent rounds behind the first one (outputs compared, no layer map): with the codec's sizing passes on
                     (SSRHIP_CODEC_PRESIZE, the default since round 6) the first use of a stream synchronises the device, so the
                     real overlap of the three callers happens from the second round on
  nopipe             the LSTM layers on ONE stream (no side stream)
  nolayers           do not keep the detector's layer outputs (the plain stress test's memory picture)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import ssr_speech_amd  # noqa: E402,F401
from ssr_speech_amd import weights as W  # noqa: E402
from ssr_speech_amd.codec.wmencodec import WMEncodecModel  # noqa: E402

opts = {a.split("=")[0]: (a.split("=") + ["1"])[1] for a in sys.argv[1:]}
_held = []
if "prewarm-queues" in opts:
    for _ in range(int(opts["prewarm-queues"])):
        s_ = torch.cuda.Stream()
        with torch.cuda.stream(s_):
            _held.append((s_, torch.zeros(64, device="cuda") + 1))
    torch.cuda.synchronize()
cfg = W.codec_config_full()
_sd_path = "/dev/shm/ssr_race_codec_sd21.pt"               # the GPU boxes generate 144 M parameters in ~5 s: once per box, not once per trial
if os.path.exists(_sd_path):
    _sd = torch.load(_sd_path, mmap=True)
else:
    _sd = W.codec_state_dict(cfg, seed=21)
    torch.save(_sd, _sd_path + f".{os.getpid()}")
    os.replace(_sd_path + f".{os.getpid()}", _sd_path)
m = WMEncodecModel(cfg, _sd, "cuda")
g = torch.Generator().manual_seed(19)
n = cfg.hop * 70 + 11
Bs = (9, 7, 9)
wavs = [(torch.randn(b, 1, n, generator=g) * 0.2).cuda() for b in Bs]
labels = [torch.randint(0, 2, (b, 71), generator=g).cuda() for b in Bs]
tracks = [torch.nn.functional.pad(w, (0, 71 * cfg.hop - n)) for w in wavs]
names = ("codes", "emb", "dec", "wm", "mark")
m.lstm_pipe_min_b = int(opts.get("pipe-min-b", "1"))       # the two-stream LSTM pipeline for every caller, as in all trials of round 6 (product: 8)
if "nopipe" in opts:
    m.LSTM_CHUNK = 10 ** 9
streams = [torch.cuda.Stream() for _ in range(3)]
if "early-streams" in opts:
    for st in streams:
        with torch.cuda.stream(st):
            _held.append(torch.zeros(64, device="cuda") + 1)
            sd_ = m._side_stream()
            with torch.cuda.stream(sd_):
                _held.append(torch.zeros(64, device="cuda") + 1)
    torch.cuda.synchronize()


# every layer output of the DETECTOR pass (wm_encoder) is kept, for the calls alone and for the concurrent round: if `mark` differs, the
# first layer whose output differs names the kernel (the buffers are only referenced, not copied: no extra launches)
orig_run = m._run
rec = None


def run_rec(nodes, x, after=None):
    if nodes is not m.wm_encoder.nodes or rec is None or "nolayers" in opts:
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
if "warm-alloc" in opts:
    # what one call allocates, on every caller stream (and its side stream's pool is never allocated from: wmencodec._lstm allocates on
    # the caller's stream only): a generous superset, then freed into that stream's pool
    for st in streams:
        with torch.cuda.stream(st):
            big = [torch.empty(64 << 20, dtype=torch.uint8, device="cuda") for _ in range(40)]
            mid = [torch.empty(8 << 20, dtype=torch.uint8, device="cuda") for _ in range(40)]
            small = [torch.empty(256 << 10, dtype=torch.uint8, device="cuda") for _ in range(64)]
            del big, mid, small
    torch.cuda.synchronize()
n_malloc0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
bg_stop, bg_thread, bg_count = False, None, [0]
if "bg-malloc" in opts:
    import threading
    bg_stream = torch.cuda.Stream()

    def bg_loop():
        held = []
        i = 0
        while not bg_stop:
            with torch.cuda.stream(bg_stream):
                held.append(torch.empty((40 << 20) + i * (2 << 20), dtype=torch.uint8, device="cuda"))    # a size nobody freed: a fresh hipMalloc
            i += 1
            bg_count[0] += 1
            if len(held) >= 48:
                held.clear()                       # back to the bg stream's pool (no hipFree): the next sizes are larger, so still fresh
        held.clear()

    bg_thread = threading.Thread(target=bg_loop, daemon=True)
    bg_thread.start()
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
if bg_thread is not None:
    bg_stop = True
    bg_thread.join()
    print(f"background thread made {bg_count[0]} fresh allocations during the round")
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
                # where do the wrong values come from? (a) the same lanes' result of ANOTHER row of the same buffer (a stale / replayed store-data
                # beat: rows r +- 16 k are the same lanes' other iterations), (b) one value for the whole group (an input sample that
                # overwrote the register), (c) neither
                for (it, r), chs in list(seen.items())[:8]:
                    bad = db[it, r, chs]
                    src = None
                    for d in range(-256, 257):
                        if d != 0 and 0 <= r + d < da.shape[1] and torch.equal(da[it, r + d, chs], bad):
                            src = d
                            break
                    same = bool((bad == bad[0]).all())
                    tl = (r - a.padL) % 16
                    print(f"        item {it} row {r - a.padL} (row % 16 = {tl}: lanes {16 * (tl % 4)}..{16 * (tl % 4) + 15} of wave {tl // 4}): wrong values == the alone pass's row {('%+d' % src) if src is not None else 'none within +-256'}; "
                          f"all {len(chs)} equal each other: {same}; max |diff| {float((bad - da[it, r, chs]).abs().max()):.3g}")
                for (it, r), chs in list(seen.items())[:5]:
                    print(f"        item {it} row {r - a.padL}: channels {chs}; alone {[round(float(da[it, r, c_]), 4) for c_ in chs[:6]]} concurrent {[round(float(db[it, r, c_]), 4) for c_ in chs[:6]]}")
                break
for rnd in range(1, int(opts.get("rounds", "1"))):
    more = [None] * 3
    for i in ((0, 1, 2) if rnd % 2 == 0 else (2, 0, 1)):
        streams[i].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(streams[i]):
            more[i] = call(i)
    for st in streams:
        torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    for i in range(3):
        for k, (a, b) in enumerate(zip(alone[i], more[i])):
            if not torch.equal(a, b):
                ok = False
                print(f"FAIL round {rnd} caller {i} {names[k]}: {int((a != b).sum())} differ, max {float((a.float() - b.float()).abs().max()):.3g}")
print(f"codec: {m.sizing_passes} sizing passes, {m.mallocs_in_flight} driver allocations during sized calls")
print(f"device mallocs during the round: {torch.cuda.memory_stats().get('num_device_alloc', 0) - n_malloc0}")
print("first concurrent round:", "identical" if ok else "DIFFERENT")
sys.exit(0 if ok else 3)
