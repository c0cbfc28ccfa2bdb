"""Where the time of ONE segment-GEMV launch goes: per-workgroup wall_clock64 stamps (100 MHz) of the last launch of a dependent chain
of LN+QKV (6144 x 2048), LN+FFN1 (8192 x 2048) and FFN2 (2048 x 8192) launches at B = 2. usage: python tools/gemv_prof.py"""
import ctypes as C, math, sys
import numpy as np, torch
sys.path.insert(0, ".")
import ssr_speech_amd  # noqa
from ssr_speech_amd import _lib

L = _lib.lib()
L.ssrhip_debug_gemv_prof.argtypes = [C.c_void_p]
L.ssrhip_debug_gemv_prof.restype = None
B, D, F = 2, 2048, 8192
g = torch.Generator().manual_seed(0)
mk = lambda n, k: [(torch.randn(n, k, generator=g) / math.sqrt(k)).cuda() for _ in range(4)]
Wq, W1, W2 = mk(3 * D, D), mk(F, D), mk(D, F)
bq, b1, b2 = torch.randn(3 * D).cuda(), torch.randn(F).cuda(), torch.randn(D).cuda()
x = torch.randn(B, D, generator=g).cuda()
q = torch.zeros(B, 3 * D).cuda()
h = torch.zeros(B, F).cuda()
prof = torch.zeros(512 * 8, dtype=torch.int64).cuda()
st = _lib.stream_ptr()


def call(W, bias, xin, yout, N, K, pro, act, epi):
    a = _lib.GemvArgs()
    a.W, a.bias, a.x, a.y = W.data_ptr(), bias.data_ptr(), xin.data_ptr(), yout.data_ptr()
    a.B, a.N, a.K, a.groups, a.x_stride, a.y_stride = B, N, K, 1, K, N
    a.pro, a.act, a.epi, a.ln_eps = pro, act, epi, 1e-5
    _lib.check(L.ssrhip_gemv(C.byref(a), st))


def layer(i):
    call(Wq[i % 4], bq, x, q, 3 * D, D, _lib.PRO_LAYERNORM, 0, 0)
    call(W1[i % 4], b1, x, h, F, D, _lib.PRO_LAYERNORM, 1, 0)
    call(W2[i % 4], b2, h, x, D, F, 0, 0, 1)
    x.mul_(0.5)


names = ["start", "first loads issued", "prologue done", "stream done (own wave)", "barrier passed", "end"]
for which, label in ((0, "LN+QKV 6144x2048 (50.4 MB)"), (1, "LN+FFN1 8192x2048 (67.1 MB)"), (2, "FFN2 2048x8192 (67.1 MB)")):
    for i in range(6):
        layer(i)
    torch.cuda.synchronize()
    # the profiled launch runs right behind its usual predecessor
    if which == 0:
        call(W2[0], b2, h, x, D, F, 0, 0, 1)
    elif which == 1:
        call(Wq[0], bq, x, q, 3 * D, D, _lib.PRO_LAYERNORM, 0, 0)
    else:
        call(W1[0], b1, x, h, F, D, _lib.PRO_LAYERNORM, 1, 0)
    L.ssrhip_debug_gemv_prof(prof.data_ptr())
    if which == 0:
        call(Wq[1], bq, x, q, 3 * D, D, _lib.PRO_LAYERNORM, 0, 0)
    elif which == 1:
        call(W1[1], b1, x, h, F, D, _lib.PRO_LAYERNORM, 1, 0)
    else:
        call(W2[1], b2, h, x, D, F, 0, 0, 1)
    L.ssrhip_debug_gemv_prof(None)
    torch.cuda.synchronize()
    P = prof.cpu().numpy().reshape(512, 8).astype(np.float64) / 100.0
    t0 = P[:, 0].min()
    print(label)
    for i, nm in enumerate(names):
        col = P[:, i] - t0
        print(f"  {nm:24s} min {col.min():6.2f}  p10 {np.percentile(col, 10):6.2f}  med {np.median(col):6.2f}  p90 {np.percentile(col, 90):6.2f}  max {col.max():6.2f}")
    # who finishes late? by XCD guess (b % 8), by dispatch half (b // 256: the second workgroup of a CU), and the raw series
    sd = P[:, 3] - t0
    print("  stream-done by b % 8       :", " ".join(f"{sd[i::8].mean():5.2f}" for i in range(8)))
    print("  stream-done by b // 64     :", " ".join(f"{sd[i * 64:(i + 1) * 64].mean():5.2f}" for i in range(8)))
    fi = P[:, 1] - t0
    print("  first-loads-issued by b//64:", " ".join(f"{fi[i * 64:(i + 1) * 64].mean():5.2f}" for i in range(8)))
    print("  start by b // 64           :", " ".join(f"{(P[i * 64:(i + 1) * 64, 0] - t0).mean():5.2f}" for i in range(8)))
    print("  stream-done, b = 0..31     :", " ".join(f"{v:4.1f}" for v in sd[:32]))
    print("  stream-done, b = 256..287  :", " ".join(f"{v:4.1f}" for v in sd[256:288]))
