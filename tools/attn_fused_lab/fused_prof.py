"""Where the time of ONE fused attention + out-projection launch (csrc/attn_fused.hip) goes: per-workgroup wall_clock64 stamps
(100 MHz) of the last launch of a chain [QKV-sized GEMV -> fused] x n at the 830M layer shape, context ~ctx.
usage: python tools/fused_prof.py [ctx]"""
import ctypes as C, math, sys
import numpy as np, torch
sys.path.insert(0, ".")
import ssr_speech_amd  # noqa
from ssr_speech_amd import _lib

L = _lib.lib()
L.ssrhip_debug_fused_prof.argtypes = [C.c_void_p]
L.ssrhip_debug_fused_prof.restype = None
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 314
B, D, hd, n_layer = 2, 2048, 128, 2
H = D // hd
max_pages = 8
g = torch.Generator().manual_seed(0)
n_pages = B * max_pages
pool = torch.randn(n_pages, n_layer, 2, H, 128, hd, generator=g).cuda()
table = torch.arange(n_pages - 1, -1, -1, dtype=torch.int32).view(B, max_pages).cuda()
lens = torch.tensor([ctx, ctx - 3], dtype=torch.int32).cuda()
q = torch.randn(B, D, generator=g).cuda()
Ws = [(torch.randn(D, D, generator=g) / math.sqrt(D)).cuda() for _ in range(8)]
Wq = [(torch.randn(3 * D, D, generator=g) / math.sqrt(D)).cuda() for _ in range(4)]
bo = torch.randn(D, generator=g).cuda()
y = torch.zeros(B, D).cuda()
x = torch.randn(B, D, generator=g).cuda()
qkv = torch.zeros(B, 3 * D).cuda()
part_o = torch.zeros(B * H * max_pages * hd).cuda()
part_ml = torch.zeros(B * H * max_pages * 2).cuda()
sync = torch.zeros(L.ssrhip_attn_outproj_sync_words(), dtype=torch.int32).cuda()
prof = torch.zeros(256 * 8, dtype=torch.int64).cuda()
at = _lib.AttnArgs()
at.q, at.q_stride = q.data_ptr(), 0
at.kv = _lib.KV(pool.data_ptr(), table.data_ptr(), max_pages, n_layer, H, hd)
at.layer, at.row_seq, at.row_len, at.R, at.max_splits = 1, 0, lens.data_ptr(), B, max_pages
at.scale, at.part_o, at.part_ml = 1.0 / math.sqrt(hd), part_o.data_ptr(), part_ml.data_ptr()
st = _lib.stream_ptr()


def gemv_qkv(i):
    a = _lib.GemvArgs()
    a.W, a.x, a.y = Wq[i % 4].data_ptr(), x.data_ptr(), qkv.data_ptr()
    a.B, a.N, a.K, a.groups, a.x_stride, a.y_stride = B, 3 * D, D, 1, D, 3 * D
    a.pro, a.ln_eps = _lib.PRO_LAYERNORM, 1e-5
    _lib.check(L.ssrhip_gemv(C.byref(a), st))


def fused(i):
    ga = _lib.GemvArgs()
    ga.W, ga.bias, ga.y, ga.B, ga.N, ga.K, ga.groups, ga.x_stride, ga.y_stride = Ws[i % 8].data_ptr(), bo.data_ptr(), y.data_ptr(), B, D, D, 1, D, D
    ga.pro, ga.act, ga.epi = _lib.PRO_ATTN_COMBINE, 0, _lib.EPI_RESIDUAL
    ga.part_o, ga.part_ml, ga.max_splits, ga.row_len, ga.kv = part_o.data_ptr(), part_ml.data_ptr(), max_pages, lens.data_ptr(), at.kv
    _lib.check(L.ssrhip_attn_outproj(C.byref(at), C.byref(ga), sync.data_ptr(), st))


for i in range(20):
    gemv_qkv(i); fused(i)
torch.cuda.synchronize()
L.ssrhip_debug_fused_prof(prof.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
gemv_qkv(0)
e0.record(); fused(1); e1.record()
torch.cuda.synchronize()
L.ssrhip_debug_fused_prof(None)
P = prof.cpu().numpy().reshape(256, 8).astype(np.float64) / 100.0     # us
t0 = P[:, 0].min()
n_items = int(sum((int(v) + 127) // 128 for v in lens.cpu()) * H)
names = ["start", "loads issued", "items done (publish)", "wait passed", "merge done (x in LDS)", "end"]
print(f"ctx {ctx}: {n_items} items; event-timed launch {1000 * e0.elapsed_time(e1):.1f} us; stamps relative to the first workgroup's start")
for i, nm in enumerate(names):
    col = P[:, i] - t0
    it, no = col[:n_items], col[n_items:]
    print(f"  {nm:24s} item WGs: min {it.min():6.2f} med {np.median(it):6.2f} max {it.max():6.2f} | others: min {no.min():6.2f} med {np.median(no):6.2f} max {no.max():6.2f}")
print("  nonzero sync words after:", sync.cpu().nonzero().flatten().tolist())
