"""Where do the ~18 ms of DecodeEngine.start() (prefill of the 598 prompt rows of the bench configuration) go?
Usage: python tools/prefill_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ssr_speech_amd  # noqa: E402,F401
from ssr_speech_amd import layout as LY, weights as W  # noqa: E402
from ssr_speech_amd.engine import DecodeEngine, DecodeKnobs, LMWeightsArena  # noqa: E402

dev = torch.device("cuda")
args = W.lm_args_830m()
sd = W.lm_state_dict(args, seed=0, device=dev)
arena = LMWeightsArena(args, sd, dev)
g = torch.Generator().manual_seed(2024)
L, N = 130, 160
x = torch.randint(0, 100, (1, L), generator=g)
y = torch.randint(0, 2048, (1, N, 4), generator=g)
unc = torch.randint(0, 101, (1, L), generator=g)
cated, _, num_task, _ = LY.build_layout(y[0].T.numpy(), np.asarray([[N, N]]), args)
eng = DecodeEngine(arena, 1, True, 1024, 512)
kn = DecodeKnobs(top_k=40, top_p=0.8, temperature=1.0, stop_repetition=2, cfg_coef=1.5, cfg_stride=5, use_cfg=True, text_len=L, n_spans=num_task, seed=1)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.start([x[0].numpy(), unc[0].numpy()], [cated], [kn], noise=None)
    t1 = time.perf_counter()            # host side returned (launches enqueued)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"start(): host {1000 * (t1 - t0):.2f} ms, + wait for the GPU {1000 * (t2 - t1):.2f} ms, total {1000 * (t2 - t0):.2f} ms")
