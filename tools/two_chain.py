"""Experiment: N independent decode chains (each 1 utterance x CFG = 2 rows, its own engine and hipGraph) on N HIP streams at
once, vs the same utterances batched into one engine. Does a second chain fill the HBM idle time of the first (ramp / tail of
every kernel)?   Usage: python tools/two_chain.py [n_chains] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ssr_speech_amd  # noqa: E402,F401
from ssr_speech_amd import layout as LY, weights as W  # noqa: E402
from ssr_speech_amd.engine import DecodeEngine, DecodeKnobs, LMWeightsArena  # noqa: E402

n_chains = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda")
args = W.lm_args_830m()
arena = LMWeightsArena(args, W.lm_state_dict(args, seed=0, device=dev), dev)
L, N = 130, 160


def inputs(i):
    g = torch.Generator().manual_seed(2024 + i)
    x = torch.randint(0, 100, (1, L), generator=g)
    y = torch.randint(0, 2048, (1, N, 4), generator=g)
    unc = torch.randint(0, 101, (1, L), generator=g)
    cated, _, num_task, _ = LY.build_layout(y[0].T.numpy(), np.asarray([[N, N]]), args)
    kn = DecodeKnobs(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=2, cfg_coef=1.5, cfg_stride=5, use_cfg=True, text_len=L, n_spans=num_task, seed=i)
    return [x[0].numpy(), unc[0].numpy()], cated, kn


engs, streams = [], []
for i in range(n_chains):
    rows, cated, kn = inputs(i)
    e = DecodeEngine(arena, 1, True, 1024, 512)
    e.start(rows, [cated], [kn])
    e.decode(20)
    engs.append(e)
    streams.append(torch.cuda.Stream(dev))
torch.cuda.synchronize()
# one chain alone
t0 = time.perf_counter(); engs[0].decode(steps); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"1 chain : {1000 * (t1 - t0) / steps:.4f} ms/step -> {4 * steps / (t1 - t0):.0f} codec-tokens/s")
# n chains at once (graph launches enqueued round-robin in slices so that neither stream runs dry)
t0 = time.perf_counter()
for s0 in range(0, steps, 25):
    for e, st in zip(engs, streams):
        with torch.cuda.stream(st):
            e.decode(min(25, steps - s0))
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"{n_chains} chains: {1000 * (t1 - t0) / steps:.4f} ms per step of every chain -> {4 * n_chains * steps / (t1 - t0):.0f} codec-tokens/s")
# the same utterances in one engine
rows_all, cols, kns = [], [], []
for i in range(n_chains):
    rows, cated, kn = inputs(i)
    rows_all += rows; cols.append(cated); kns.append(kn)
eb = DecodeEngine(arena, n_chains, True, 1024, 512)
eb.start(rows_all, cols, kns)
eb.decode(20); torch.cuda.synchronize()
t0 = time.perf_counter(); eb.decode(steps); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"1 engine, {2 * n_chains} rows: {1000 * (t1 - t0) / steps:.4f} ms/step -> {4 * n_chains * steps / (t1 - t0):.0f} codec-tokens/s")
