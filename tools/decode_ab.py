"""A/B of decode-step variants in ONE process on ONE box (round 5): the 830M weights are generated once, then for every variant — a set
of environment knobs the C side reads when a launch is ENQUEUED, i.e. at graph capture — a fresh DecodeEngine is built, started on
bench.py's config-2 input and timed exactly like bench.py's headline (W untimed steps, K timed, wall clock around `eng.decode`), plus
the per-category graph-chained launch times. Variants alternate over `--reps` rounds; per variant the list of ms/step is printed with
its minimum and median (box-to-box spread is 3-5 %, so only same-box alternating runs can resolve a 1 % change).

  python tools/decode_ab.py [--utts U] [--steps K] [--warmup W] [--reps R] name:KNOB=v,KNOB=v ...
  e.g. python tools/decode_ab.py base:SSRHIP_GEMV_SEGU=0,SSRHIP_ATTN_PIN=0 segu4: segu2:SSRHIP_GEMV_SEGU=2
"""
import argparse
import dataclasses
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ssr_speech_amd  # noqa: E402,F401
from ssr_speech_amd import layout as LY  # noqa: E402
from ssr_speech_amd import weights as W  # noqa: E402
from ssr_speech_amd.engine import DecodeEngine, DecodeKnobs, LMWeightsArena  # noqa: E402
from bench import synth_inputs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--utts", type=int, default=1)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--greedy", action="store_true", help="top_k=1 instead of bench.py's sampling knobs (tokens comparable across variants)")
ap.add_argument("variants", nargs="+")
a = ap.parse_args()

variants = []
for v in a.variants:
    name, _, kn = v.partition(":")
    variants.append((name, dict(kv.split("=", 1) for kv in kn.split(",") if kv)))
all_knobs = sorted({k for _, d in variants for k in d})

dev = torch.device("cuda", 0)
args_lm = W.lm_args_830m()
arena = LMWeightsArena(args_lm, W.lm_state_dict(args_lm, seed=0, device=dev), dev)
x, y, unc = synth_inputs(args_lm, 0)
L, N = x.shape[1], y.shape[1]
total = a.warmup + a.steps
cated, _, num_task, _ = LY.build_layout(y[0].T.numpy(), np.asarray([[N, N]]), args_lm)
T0 = cated.shape[1]
U = a.utts
text_rows = []
for u in range(U):
    xu, _, uu = (x, y, unc) if u == 0 else synth_inputs(args_lm, u)
    text_rows += [xu[0].numpy(), uu[0].numpy()]
kn = DecodeKnobs(top_k=1 if a.greedy else 40, top_p=1.0 if a.greedy else 0.8, temperature=1.0, stop_repetition=2, cfg_coef=1.5, cfg_stride=5,
                 use_cfg=True, text_len=L, n_spans=num_task, seed=2024)

res = {name: {"ms": [], "gemv": [], "attn": [], "sample": [], "tok": None} for name, _ in variants}
for rep in range(a.reps):
    for name, knobs in variants:
        for k in all_knobs:
            os.environ.pop(k, None)
        os.environ.update(knobs)
        eng = DecodeEngine(arena, U, True, ((L + T0 + total + 8 + 1023) // 1024) * 1024, ((total + 255) // 256) * 256)
        eng.start(text_rows, [cated] * U, [dataclasses.replace(kn, seed=2024 + u) for u in range(U)], noise=None)
        torch.cuda.synchronize()
        eng.decode(a.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.decode(a.steps)
        torch.cuda.synchronize()
        ms = 1000 * (time.perf_counter() - t0) / a.steps
        n_done = int(eng.states()[0].n_steps)
        tok = eng.generated[0, :n_done].cpu().numpy().copy()
        r = res[name]
        r["ms"].append(ms)
        r["gemv"].append(eng.time_category("gemv", 50)[0])
        r["attn"].append(eng.time_category("attn", 50)[0])
        if r["tok"] is None:
            r["tok"] = tok
        del eng
        torch.cuda.empty_cache()

base = variants[0][0]
print(f"# {U} utterance(s) x CFG = {2 * U} rows, {a.steps} timed steps after {a.warmup}, {a.reps} alternating rounds; ms per step (wall), us per launch (graph-chained)")
for name, knobs in variants:
    r = res[name]
    same = "" if r["tok"] is None or res[base]["tok"] is None else f"  tokens == {base}: {bool(np.array_equal(r['tok'], res[base]['tok']))}"
    print(f"{name:14s} ms/step min {min(r['ms']):.4f} med {statistics.median(r['ms']):.4f}  all {' '.join(f'{v:.4f}' for v in r['ms'])} | "
          f"gemv {min(r['gemv']):.3f} attn {min(r['attn']):.3f} us{same}   [{' '.join(f'{k}={v}' for k, v in knobs.items()) or 'defaults'}]")
