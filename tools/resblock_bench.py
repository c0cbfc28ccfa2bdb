"""The fused SEANetResnetBlock kernels alone, at the shapes they have in config 5 (C = 128: T = 240,000 per clip; C = 64: T = 480,000),
B clips, random data: ms per call and TFLOP/s / GB/s. usage: python tools/resblock_bench.py [B] [planes 0|1]"""
import ctypes as C, math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssr_speech_amd  # noqa
from ssr_speech_amd import _lib

L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
use_planes = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
g = torch.Generator().manual_seed(0)
for Cc, T in ((128, 240000), (64, 480000)):
    Hh = Cc // 2
    x = (torch.randn(B, T + 2, Cc, generator=g) * 0.5).cuda()
    y = torch.empty(B, T, Cc, device="cuda")
    w3 = (torch.randn(Hh, 3 * Cc, generator=g) / math.sqrt(3 * Cc)).cuda()
    w1 = (torch.randn(Cc, Hh, generator=g) / math.sqrt(Hh)).cuda()
    b3, b1 = torch.zeros(Hh).cuda(), torch.zeros(Cc).cuda()
    kp = torch.arange(Hh)
    perm = 16 * (kp // 16) + (kp % 8) % 4 + 8 * ((kp % 8) // 4) + 4 * ((kp // 8) % 2)
    w1p = w1[:, perm.cuda()].contiguous()
    p3 = torch.empty(3, Hh, 3 * Cc, dtype=torch.int16, device="cuda")
    p1 = torch.empty(3, Cc, Hh, dtype=torch.int16, device="cuda")
    _lib.check(L.ssrhip_split_weights(w3.data_ptr(), p3.data_ptr(), w3.numel(), _lib.stream_ptr()))
    _lib.check(L.ssrhip_split_weights(w1p.data_ptr(), p1.data_ptr(), w1p.numel(), _lib.stream_ptr()))
    a = _lib.ResblockArgs()
    a.x, a.y, a.w3, a.b3, a.w1, a.b1 = x.data_ptr(), y.data_ptr(), w3.data_ptr(), b3.data_ptr(), w1.data_ptr(), b1.data_ptr()
    a.B, a.T, a.C, a.x_bstride, a.y_bstride, a.out_act = B, T, Cc, (T + 2) * Cc, T * Cc, _lib.ACT_ELU
    if use_planes:
        a.w3_split, a.w1_split = p3.data_ptr(), p1.data_ptr()
    for _ in range(2):
        _lib.check(L.ssrhip_resblock(C.byref(a), _lib.stream_ptr()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        _lib.check(L.ssrhip_resblock(C.byref(a), _lib.stream_ptr()))
    torch.cuda.synchronize()
    ms = 1000 * (time.perf_counter() - t0) / n
    flop = 2.0 * B * T * (3 * Cc * Hh + Hh * Cc)
    gb = 2.0 * B * T * Cc * 4 / 1e9
    print(f"C={Cc} B={B} T={T}: {ms:.2f} ms per call = {flop / ms / 1e9:.1f} TFLOP/s fp32-equivalent, {gb / ms * 1e3:.0f} GB/s of activation traffic (x256/B: {ms * 256 / B:.1f} ms at 256 clips)")
    del x, y
