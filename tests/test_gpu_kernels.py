"""GPU: each HIP kernel called through the C-ABI against a plain PyTorch fp32 CPU reference of the
same op (and, for the sampler, against the oracle's state machine). Tolerances are written per test:
fp32 everywhere, differences come only from summation order."""
import ctypes as C
import os
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ssr_speech_amd  # noqa: F401
from ssr_speech_amd import _lib
from ssr_speech_amd import weights as W
from oracle import lm as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return _lib.lib()


def dev(t):
    return t.to("cuda").contiguous()


def sync():
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------ GEMV
@pytest.mark.parametrize("B,N,K", [(2, 512, 2048), (2, 96, 8192), (1, 100, 1024), (4, 77, 128), (2, 64, 512), (2, 130, 4096), (1, 2056, 1024)])
@pytest.mark.parametrize("pro,act,epi", [(0, 0, 0), (1, 1, 0), (1, 2, 0), (0, 0, 1)])
def test_gemv_matches_torch(L, B, N, K, pro, act, epi):
    if pro == 1 and K > 4096:
        pytest.skip("LayerNorm prologue is only used with K = d_model")
    g = torch.Generator().manual_seed(B * 1000 + N + K + pro + act + epi)
    Wt = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    x = torch.randn(B, K, generator=g) * 1.5 + 0.3
    lw, lb = 1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    y0 = torch.randn(B, N, generator=g)
    xin = F.layer_norm(x, (K,), lw, lb, 1e-5) if pro == 1 else x
    ref = F.linear(xin, Wt, bias)
    ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    ref = y0 + ref if epi == 1 else ref
    dW, db, dx, dlw, dlb, dy = dev(Wt), dev(bias), dev(x), dev(lw), dev(lb), dev(y0.clone())
    a = _lib.GemvArgs()
    a.W, a.bias, a.x, a.y = dW.data_ptr(), db.data_ptr(), dx.data_ptr(), dy.data_ptr()
    a.B, a.N, a.K, a.groups, a.x_stride, a.y_stride = B, N, K, 1, K, N
    a.pro, a.act, a.epi = pro, act, epi
    a.ln_w, a.ln_b, a.ln_eps = dlw.data_ptr(), dlb.data_ptr(), 1e-5
    _lib.check(L.ssrhip_gemv(C.byref(a), _lib.stream_ptr()))
    sync()
    # fp32, values O(1): summation-order noise only
    torch.testing.assert_close(dy.cpu(), ref, rtol=2e-5, atol=2e-5)


def test_gemv_grouped_heads(L):
    """K groups with their own weights/inputs (second Linear of the prediction heads, ssr.py:177)."""
    g = torch.Generator().manual_seed(3)
    G, B, N, K = 4, 2, 72, 1024
    Wt = torch.randn(G, N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(G, N, generator=g)
    x = torch.randn(B, G, K, generator=g)
    ref = torch.stack([F.linear(x[:, k], Wt[k], bias[k]) for k in range(G)], 1)      # [B,G,N]
    dW, db, dx = dev(Wt), dev(bias), dev(x)
    dy = torch.zeros(B, G, N, device="cuda")
    a = _lib.GemvArgs()
    a.W, a.bias, a.x, a.y = dW.data_ptr(), db.data_ptr(), dx.data_ptr(), dy.data_ptr()
    a.B, a.N, a.K, a.groups, a.x_stride, a.y_stride = B, N, K, G, G * K, G * N
    _lib.check(L.ssrhip_gemv(C.byref(a), _lib.stream_ptr()))
    sync()
    torch.testing.assert_close(dy.cpu(), ref, rtol=2e-5, atol=2e-5)


# ------------------------------------------------------------------------------------------ GEMV, 5..16 rows (MFMA path)
@pytest.mark.parametrize("B", [5, 8, 16])
@pytest.mark.parametrize("N,K", [(512, 2048), (96, 8192), (100, 1024), (77, 128), (2056, 1024), (130, 4096), (48, 16)])
@pytest.mark.parametrize("pro,act,epi", [(0, 0, 0), (1, 1, 0), (1, 2, 0), (0, 0, 1)])
@pytest.mark.parametrize("wt", [0, 1])
def test_gemv_mfma_rows_matches_torch(L, B, N, K, pro, act, epi, wt):
    """B > 4 rows go to gemv_mfma.hip (v_mfma_f32_16x16x4_f32); LayerNorm gamma/beta must be folded by the caller.
    wt=1: W handed over in the streaming order (SSRHIP_WTILED_INDEX, `engine.to_streaming_order`) the decode engine uses."""
    from ssr_speech_amd.engine import to_streaming_order
    if pro == 1 and K > 4096:
        pytest.skip("LayerNorm prologue is only used with K = d_model")
    g = torch.Generator().manual_seed(B * 1000 + N + K + pro + act + epi)
    Wt = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    x = torch.randn(B, K, generator=g) * 1.5 + 0.3
    lw, lb = 1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    y0 = torch.randn(B, N, generator=g)
    xin = F.layer_norm(x, (K,), lw, lb, 1e-5) if pro == 1 else x
    ref = F.linear(xin, Wt, bias)
    ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    ref = y0 + ref if epi == 1 else ref
    if pro == 1:      # what LMWeightsArena does at load time
        Wf = (Wt.double() * lw.double()[None, :]).float()
        bf = (bias.double() + Wt.double() @ lb.double()).float()
    else:
        Wf, bf = Wt, bias
    dW, db, dx, dy = dev(to_streaming_order(Wf) if wt else Wf), dev(bf), dev(x), dev(y0.clone())
    a = _lib.GemvArgs()
    a.W, a.bias, a.x, a.y = dW.data_ptr(), db.data_ptr(), dx.data_ptr(), dy.data_ptr()
    a.B, a.N, a.K, a.groups, a.x_stride, a.y_stride = B, N, K, 1, K, N
    a.pro, a.act, a.epi = pro, act, epi
    a.ln_eps = 1e-5
    a.w_tiled = wt
    _lib.check(L.ssrhip_gemv(C.byref(a), _lib.stream_ptr()))
    sync()
    torch.testing.assert_close(dy.cpu(), ref, rtol=3e-5, atol=3e-5)


def _to_tiled(t):
    """[B<=16, K] -> the 16-column tiled layout of include/ssrhip.h (SSRHIP_TILED): [K/4][16][4]."""
    B, K = t.shape
    out = torch.zeros(K // 4, 16, 4)
    out[:, :B, :] = t.view(B, K // 4, 4).permute(1, 0, 2)
    return out.contiguous()


def _from_tiled(t, B, K):
    return t.view(K // 4, 16, 4)[:, :B, :].permute(1, 0, 2).reshape(B, K)


@pytest.mark.parametrize("B", [6, 16])
@pytest.mark.parametrize("G,N,K,pro,act,epi", [(1, 512, 2048, 1, 1, 0), (1, 2048, 8192, 0, 0, 1), (4, 72, 1024, 0, 0, 0), (1, 4096, 2048, 1, 2, 0)])
@pytest.mark.parametrize("wt", [0, 1])
def test_gemv_mfma_tiled_activations(L, B, G, N, K, pro, act, epi, wt):
    """x and/or y in the tiled layout the 5..16-row decode step keeps its activations in; wt=1: grouped W in streaming order."""
    from ssr_speech_amd.engine import to_streaming_order
    g = torch.Generator().manual_seed(B + N + K)
    Wt = torch.randn(G, N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(G, N, generator=g)
    x = torch.randn(B, G, K, generator=g) * 1.2 + 0.2
    y0 = torch.randn(B, G, N, generator=g)
    xin = F.layer_norm(x, (K,), None, None, 1e-5) if pro == 1 else x
    ref = torch.stack([F.linear(xin[:, k], Wt[k], bias[k]) for k in range(G)], 1)
    ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    ref = y0 + ref if epi == 1 else ref
    dW, db = dev(to_streaming_order(Wt) if wt else Wt), dev(bias)
    dx = dev(_to_tiled(x.reshape(B, G * K)))
    dy = dev(_to_tiled(y0.reshape(B, G * N)))
    a = _lib.GemvArgs()
    a.W, a.bias, a.x, a.y = dW.data_ptr(), db.data_ptr(), dx.data_ptr(), dy.data_ptr()
    a.B, a.N, a.K, a.groups, a.x_stride, a.y_stride = B, N, K, G, 0, 0
    a.pro, a.act, a.epi, a.ln_eps = pro, act, epi, 1e-5
    a.x_tiled, a.y_tiled, a.w_tiled = 1, 1, wt
    _lib.check(L.ssrhip_gemv(C.byref(a), _lib.stream_ptr()))
    sync()
    got = _from_tiled(dy.cpu(), B, G * N).reshape(B, G, N)
    torch.testing.assert_close(got, ref, rtol=3e-5, atol=3e-5)
    a.B = 4
    assert L.ssrhip_gemv(C.byref(a), _lib.stream_ptr()) != 0       # tiled operands are a 5..16-row feature


@pytest.mark.parametrize("B", [5, 16])
@pytest.mark.parametrize("N,act,epi", [(6144, 0, 2), (8192, 1, 0), (4096, 2, 0), (2048, 0, 0)])
def test_gemv_rows_edge_kernel_is_bit_identical_to_the_eight_wave_kernel(L, monkeypatch, B, N, act, epi):
    """Round 6 (`gemv_rows_edge_kernel`, csrc/gemv_mfma.hip): the 5..16-row LayerNorm launches at K = 2048 with the LayerNorm on four extra
    waves (x' handed over through LDS) and every weight request posted at entry. It computes the statistics in the same slices, order
    and expressions as `gemv_rows_xreg_kernel`, so outputs — q, the appended K / V rows, the FFN hidden, the head hidden — must be
    BIT-identical with the knob on and off (SSRHIP_GEMVM_EDGE=1, read at every launch; the default is the 8-wave kernel: the edge form
    measured slower, profiles/r06_microbench/gemvm_bench_16_edge.log), and within the GEMV tolerance of torch.
    Shapes: QKV with the cache append (3 units per workgroup: a 16-row and an 8-row tile), FFN1 (two 16-row tiles), the head MLP (one
    tile), and N = 2048 (one 8-row unit per workgroup)."""
    from ssr_speech_amd.engine import to_streaming_order
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the row split assumes 256 workgroups")
    K = 2048
    g = torch.Generator().manual_seed(N + B)
    Wt = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    x = torch.randn(B, K, generator=g) * 1.3 + 0.4
    ref = F.linear(F.layer_norm(x, (K,), None, None, 1e-5), Wt, bias)
    ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    dW, db, dx = dev(to_streaming_order(Wt)), dev(bias), dev(_to_tiled(x))
    H, hd, n_layer, max_pages, layer = 16, 128, 2, 3, 1
    pool, table = _make_cache(B, max_pages, n_layer, H, hd, g)
    pos = torch.randint(0, max_pages * _lib.PAGE, (B,), generator=g).to(torch.int32)
    dtable, dpos = dev(table), dev(pos)

    def run():
        dpool = dev(pool.clone())
        if epi == 2:
            dy = torch.zeros(B, K, device="cuda")
        else:
            dy = dev(_to_tiled(torch.zeros(B, N)))
        a = _lib.GemvArgs()
        a.W, a.bias, a.x, a.y = dW.data_ptr(), db.data_ptr(), dx.data_ptr(), dy.data_ptr()
        a.B, a.N, a.K, a.groups, a.x_stride, a.y_stride = B, N, K, 1, 0, (K if epi == 2 else 0)
        a.pro, a.act, a.epi, a.ln_eps = _lib.PRO_LAYERNORM, act, epi, 1e-5
        a.x_tiled, a.y_tiled, a.w_tiled = 1, (0 if epi == 2 else 1), 1
        if epi == 2:
            a.kv = _lib.KV(dpool.data_ptr(), dtable.data_ptr(), max_pages, n_layer, H, hd)
            a.layer, a.kv_pos = layer, dpos.data_ptr()
        _lib.check(L.ssrhip_gemv(C.byref(a), _lib.stream_ptr()))
        sync()
        return dy.cpu(), dpool.cpu()

    monkeypatch.setenv("SSRHIP_GEMVM_EDGE", "1")
    y_edge, pool_edge = run()
    monkeypatch.delenv("SSRHIP_GEMVM_EDGE")
    y_old, pool_old = run()
    assert torch.equal(y_edge, y_old) and torch.equal(pool_edge, pool_old)
    # the other request-order experiment of the round (first weight requests in front of the x requests): same arithmetic, same bits
    monkeypatch.setenv("SSRHIP_GEMVM_WFIRST", "1")
    y_wf, pool_wf = run()
    monkeypatch.delenv("SSRHIP_GEMVM_WFIRST")
    assert torch.equal(y_wf, y_old) and torch.equal(pool_wf, pool_old)
    if epi == 2:
        torch.testing.assert_close(y_edge, ref[:, :K], rtol=3e-5, atol=3e-5)
        for b in range(B):
            p_ = int(pos[b])
            page = int(table[b, p_ // _lib.PAGE])
            for which in (0, 1):
                got = pool_edge[page, layer, which, :, p_ % _lib.PAGE, :].reshape(-1)
                torch.testing.assert_close(got, ref[b, (1 + which) * K:(2 + which) * K], rtol=3e-5, atol=3e-5)
    else:
        torch.testing.assert_close(_from_tiled(y_edge, B, N), ref, rtol=3e-5, atol=3e-5)


def test_gemv_mfma_grouped_heads_and_qkv_append(L):
    g = torch.Generator().manual_seed(4)
    G, B, N, K = 4, 11, 72, 1024
    Wt = torch.randn(G, N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(G, N, generator=g)
    x = torch.randn(B, G, K, generator=g)
    ref = torch.stack([F.linear(x[:, k], Wt[k], bias[k]) for k in range(G)], 1)
    dW, db, dx = dev(Wt), dev(bias), dev(x)
    dy = torch.zeros(B, G, N, device="cuda")
    a = _lib.GemvArgs()
    a.W, a.bias, a.x, a.y = dW.data_ptr(), db.data_ptr(), dx.data_ptr(), dy.data_ptr()
    a.B, a.N, a.K, a.groups, a.x_stride, a.y_stride = B, N, K, G, G * K, G * N
    _lib.check(L.ssrhip_gemv(C.byref(a), _lib.stream_ptr()))
    sync()
    torch.testing.assert_close(dy.cpu(), ref, rtol=3e-5, atol=3e-5)
    # QKV append, 16 rows at different cache positions
    B, D, H, hd, n_layer, max_pages, layer = 16, 256, 4, 64, 2, 3, 1
    pool, table = _make_cache(B, max_pages, n_layer, H, hd, g)
    Wt = torch.randn(3 * D, D, generator=g) / math.sqrt(D)
    bias = torch.randn(3 * D, generator=g)
    x = torch.randn(B, D, generator=g)
    pos = torch.randint(0, max_pages * _lib.PAGE, (B,), generator=g).to(torch.int32)
    ref = F.linear(F.layer_norm(x, (D,), None, None, 1e-5), Wt, bias)
    dpool, dtable, dW, db, dx, dpos = dev(pool), dev(table), dev(Wt), dev(bias), dev(x), dev(pos)
    dq = torch.zeros(B, D, device="cuda")
    a = _lib.GemvArgs()
    a.W, a.bias, a.x, a.y = dW.data_ptr(), db.data_ptr(), dx.data_ptr(), dq.data_ptr()
    a.B, a.N, a.K, a.groups, a.x_stride, a.y_stride = B, 3 * D, D, 1, D, D
    a.pro, a.act, a.epi, a.ln_eps = _lib.PRO_LAYERNORM, 0, _lib.EPI_QKV_APPEND, 1e-5
    a.kv = _lib.KV(dpool.data_ptr(), dtable.data_ptr(), max_pages, n_layer, H, hd)
    a.layer, a.kv_pos = layer, dpos.data_ptr()
    _lib.check(L.ssrhip_gemv(C.byref(a), _lib.stream_ptr()))
    sync()
    torch.testing.assert_close(dq.cpu(), ref[:, :D], rtol=3e-5, atol=3e-5)
    newpool = dpool.cpu()
    mask = torch.ones_like(pool, dtype=torch.bool)
    for b in range(B):
        p = int(pos[b])
        page = int(table[b, p // _lib.PAGE])
        for which in (0, 1):
            got = newpool[page, layer, which, :, p % _lib.PAGE, :].reshape(-1)
            torch.testing.assert_close(got, ref[b, (1 + which) * D:(2 + which) * D], rtol=3e-5, atol=3e-5)
        mask[page, layer, :, :, p % _lib.PAGE, :] = False
    assert torch.equal(newpool[mask], pool[mask])



# ------------------------------------------------------------------------------------------ 2-row segment kernel at the step's shapes
@pytest.mark.parametrize("seg", ["1", "0"])
@pytest.mark.parametrize("B", [1, 2, 4])
@pytest.mark.parametrize("N,K,groups,pro,act,epi", [(6144, 2048, 1, 1, 0, 0), (8192, 2048, 1, 1, 1, 0), (2048, 8192, 1, 0, 0, 1), (4096, 2048, 1, 1, 2, 0),
                                                     (2056, 1024, 4, 0, 0, 0), (2048, 2048, 1, 0, 0, 1), (1000, 4096, 1, 1, 0, 1), (515, 2048, 2, 0, 1, 0)])
def test_gemv_step_shapes_both_kernels(L, monkeypatch, seg, B, N, K, groups, pro, act, epi):
    """The six GEMV shapes of the 830M decode step (plus a ragged N and a K = 4096 one) through `ssrhip_gemv` with the segment kernel
    (SSRHIP_GEMV_SEG=1, default) and with the row-per-wave kernels (=0), LayerNorm folded as the engine does (ln_w = NULL)."""
    import subprocess, sys, os
    if seg == "0" and pro == 1 and K > 2048:
        pytest.skip("the row-per-wave kernels take a folded LayerNorm only up to K = 2048 (they refuse it; the segment kernel covers it)")
    # the switch is read once per process: run the comparison in a child process
    code = f"""
import ctypes as C, math, sys, torch, torch.nn.functional as F
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
import ssr_speech_amd
from ssr_speech_amd import _lib
L = _lib.lib()
B, N, K, G, pro, act, epi = {B}, {N}, {K}, {groups}, {pro}, {act}, {epi}
g = torch.Generator().manual_seed(B * 7 + N + K)
Wt = torch.randn(G, N, K, generator=g) / math.sqrt(K)
bias = torch.randn(G, N, generator=g)
x = torch.randn(B, G, K, generator=g) * 1.5 + 0.3
y0 = torch.randn(B, G, N, generator=g)
xin = F.layer_norm(x, (K,), None, None, 1e-5) if pro == 1 else x
ref = torch.stack([F.linear(xin[:, k].double(), Wt[k].double(), bias[k].double()) for k in range(G)], 1)
ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
ref = (y0.double() + ref) if epi == 1 else ref
dW, db, dx, dy = Wt.cuda(), bias.cuda(), x.cuda().contiguous(), y0.clone().cuda().contiguous()
a = _lib.GemvArgs()
a.W, a.bias, a.x, a.y = dW.data_ptr(), db.data_ptr(), dx.data_ptr(), dy.data_ptr()
a.B, a.N, a.K, a.groups, a.x_stride, a.y_stride = B, N, K, G, G * K, G * N
a.pro, a.act, a.epi, a.ln_eps = pro, act, epi, 1e-5
_lib.check(L.ssrhip_gemv(C.byref(a), _lib.stream_ptr()))
torch.cuda.synchronize()
err = float((dy.cpu().double() - ref).abs().max())
print("ERR", err)
"""
    env = dict(os.environ, SSRHIP_GEMV_SEG=seg)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    err = float(out.stdout.strip().split("ERR")[-1])
    assert err < 3e-5, err


def test_gemv_segu_kernel_is_bit_identical_to_the_segment_kernel(L, tmp_path):
    """Round 5's `gemv_segu_kernel` (one 8-wave workgroup per CU, NUW units per wave as straight-line code, 2 or 4 in flight) performs per
    unit, per segment and per output the operations of `gemv_seg_kernel` in the same order: the outputs of the step's four shapes it takes
    (LN+QKV with the K/V append, LN+FFN1+ReLU, FFN2+residual, LN+head-MLP1+GELU; 2 rows) must be BIT-identical with the knob at 0 / 2 / 4.
    The knob is read once per process: three child processes write their outputs, the parent compares."""
    import subprocess, sys
    code = f"""
import ctypes as C, math, sys, numpy as np, torch
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
import ssr_speech_amd
from ssr_speech_amd import _lib
L = _lib.lib()
out = {{}}
B = 2
for name, N, K, pro, act, epi in [("qkv", 6144, 2048, 1, 0, 2), ("ffn1", 8192, 2048, 1, 1, 0), ("ffn2", 2048, 8192, 0, 0, 1), ("head1", 4096, 2048, 1, 2, 0)]:
    g = torch.Generator().manual_seed(N + K)
    Wt = (torch.randn(N, K, generator=g) / math.sqrt(K)).cuda()
    bias = torch.randn(N, generator=g).cuda()
    x = (torch.randn(B, K, generator=g) * 1.5 + 0.3).cuda()
    y = torch.randn(B, K if epi == 2 else N, generator=g).cuda()
    a = _lib.GemvArgs()
    a.W, a.bias, a.x, a.y = Wt.data_ptr(), bias.data_ptr(), x.data_ptr(), y.data_ptr()
    a.B, a.N, a.K, a.groups, a.x_stride, a.y_stride = B, N, K, 1, K, (K if epi == 2 else N)
    a.pro, a.act, a.epi, a.ln_eps = pro, act, epi, 1e-5
    if epi == 2:
        H, hd, n_layer, max_pages = 16, 128, 2, 4
        pool = torch.zeros(2 * max_pages + 1, n_layer, 2, H, _lib.PAGE, hd, device="cuda")
        table = torch.tensor([[5, 2, 7, 1], [0, 6, 3, 4]], dtype=torch.int32, device="cuda")
        pos = torch.tensor([130, 300], dtype=torch.int32, device="cuda")
        a.kv = _lib.KV(pool.data_ptr(), table.data_ptr(), max_pages, n_layer, H, hd)
        a.layer, a.kv_pos = 1, pos.data_ptr()
    _lib.check(L.ssrhip_gemv(C.byref(a), _lib.stream_ptr()))
    torch.cuda.synchronize()
    out[name] = y.cpu().numpy()
    if epi == 2:
        out["pool"] = pool.cpu().numpy()
np.savez(sys.argv[1], **out)
"""
    res = {}
    for knob in ("0", "2", "4"):
        f = str(tmp_path / f"segu{knob}.npz")
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, SSRHIP_GEMV_SEGU=knob), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[knob] = dict(np.load(f))
    assert np.abs(res["0"]["pool"]).sum() > 0                        # the append landed somewhere
    for knob in ("2", "4"):
        for key, want in res["0"].items():
            assert np.array_equal(res[knob][key], want), (knob, key, float(np.abs(res[knob][key] - want).max()))


def test_gemv_pair_launch_is_bit_identical_to_the_two_launches(L):
    """Round 5's `gemv_pair_kernel` (csrc/gemv.hip): FFN2 + residual and the LayerNorm + Linear that consumes it as ONE launch, the
    all-to-all edge between them inside the kernel (tagged granules, write-through stores, gather). For each of the three consumers of
    the 830M step's shape family — LN + QKV with the K/V append (N = 6144), LN + head-MLP1 + GELU (4096), LN + FFN1 + ReLU (8192) — a
    chain of 7 pairs (the three granule buffers cycled the way the engine cycles them, closed over the chain and replayed twice: stale
    tags would show) must leave the residual stream, the consumer's output and the cache BIT-identical to 14 separate ssrhip_gemv calls."""
    if torch.cuda.get_device_properties(0).multi_processor_count < 256:
        pytest.skip("the pair launch needs 256 CUs")
    B, D, F = 2, 2048, 8192
    ws = torch.zeros(_lib_pair_ws_bytes() // 4, dtype=torch.int32, device="cuda")
    for name, N, act, epi in [("qkv", 6144, 0, 2), ("head1", 4096, 2, 0), ("ffn1", 8192, 1, 0)]:
        g = torch.Generator().manual_seed(N)
        W2 = [(torch.randn(D, F, generator=g) / math.sqrt(F)).cuda() for _ in range(2)]
        b2 = torch.randn(D, generator=g).cuda()
        Wn = [(torch.randn(N, D, generator=g) / math.sqrt(D)).cuda() for _ in range(2)]
        bn = torch.randn(N, generator=g).cuda()
        h = (torch.randn(B, F, generator=g) * 0.7).cuda()
        x0 = (torch.randn(B, D, generator=g) * 1.5 + 0.3)
        H, hd, n_layer, max_pages = 16, 128, 2, 4
        table = torch.tensor([[5, 2, 7, 1], [0, 6, 3, 4]], dtype=torch.int32, device="cuda")
        pos = torch.tensor([130, 300], dtype=torch.int32, device="cuda")
        res = {}
        for form in ("two", "pair"):
            x = x0.clone().cuda()
            y = torch.zeros(B, D if epi == 2 else N, device="cuda")
            pool = torch.zeros(2 * max_pages + 1, n_layer, 2, H, _lib.PAGE, hd, device="cuda")
            n_pairs, i_pair = 7, 0
            buf = lambda i: 1 if (i == n_pairs - 1 and n_pairs % 3 == 1) else i % 3
            for rep in range(2):
                for i in range(n_pairs):
                    a = _lib.GemvArgs()
                    a.W, a.bias, a.x, a.y = W2[i % 2].data_ptr(), b2.data_ptr(), h.data_ptr(), x.data_ptr()
                    a.B, a.N, a.K, a.groups, a.x_stride, a.y_stride = B, D, F, 1, F, D
                    a.pro, a.act, a.epi = 0, 0, 1
                    b = _lib.GemvArgs()
                    b.W, b.bias, b.x, b.y = Wn[i % 2].data_ptr(), bn.data_ptr(), x.data_ptr(), y.data_ptr()
                    b.B, b.N, b.K, b.groups, b.x_stride, b.y_stride = B, N, D, 1, D, (D if epi == 2 else N)
                    b.pro, b.act, b.epi, b.ln_eps = 1, act, epi, 1e-5
                    if epi == 2:
                        b.kv = _lib.KV(pool.data_ptr(), table.data_ptr(), max_pages, n_layer, H, hd)
                        b.layer, b.kv_pos = i % 2, pos.data_ptr()
                    assert L.ssrhip_gemv_pair_applicable(C.byref(a), C.byref(b)) == 1
                    if form == "two":
                        _lib.check(L.ssrhip_gemv(C.byref(a), _lib.stream_ptr()))
                        _lib.check(L.ssrhip_gemv(C.byref(b), _lib.stream_ptr()))
                    else:
                        rc = L.ssrhip_gemv_pair(C.byref(a), C.byref(b), ws.data_ptr(), buf(i), buf((i + 1) % n_pairs), _lib.stream_ptr())
                        assert rc == 0, rc
            torch.cuda.synchronize()
            res[form] = (x.cpu().numpy(), y.cpu().numpy(), pool.cpu().numpy())
        assert L.ssrhip_gemv_pair_status(ws.data_ptr(), _lib.stream_ptr()) == 0
        for k, (got, want) in enumerate(zip(res["pair"], res["two"])):
            assert np.isfinite(want).all()
            assert np.array_equal(got, want), (name, k, float(np.abs(got - want).max()))
        if epi == 2:
            assert np.abs(res["two"][2]).sum() > 0
    # ---- the other pair of the step: split-KV merge + out-projection + residual, then LN + FFN1 + ReLU (gemv_pair_merge_kernel); contexts of
    # 8 / 2 pages (beyond the 6 prefetched ones / short) and 3 / 5 pages, odd max_splits too
    H, hd = 16, 128
    for MS, lens in ((8, [1000, 129]), (8, [300, 640]), (7, [7 * _lib.PAGE, 1])):
        g = torch.Generator().manual_seed(MS + lens[0])
        Wo = [(torch.randn(D, D, generator=g) / math.sqrt(D)).cuda() for _ in range(2)]
        bo = torch.randn(D, generator=g).cuda()
        W1 = [(torch.randn(F, D, generator=g) / math.sqrt(D)).cuda() for _ in range(2)]
        b1 = torch.randn(F, generator=g).cuda()
        part_o = torch.randn(B, H, MS, hd, generator=g).cuda()
        part_ml = torch.stack([torch.randn(B, H, MS, generator=g) * 2, torch.rand(B, H, MS, generator=g) + 0.5], dim=-1).contiguous().cuda()
        for b_ in range(B):                                          # beyond the row's pages the buffers hold garbage the kernel must not use
            n = (lens[b_] + _lib.PAGE - 1) // _lib.PAGE
            part_o[b_, :, n:] = float("nan")
            part_ml[b_, :, n:] = float("nan")
        dlen = torch.tensor(lens, dtype=torch.int32, device="cuda")
        x0 = (torch.randn(B, D, generator=g) * 1.5 + 0.3)
        res = {}
        for form in ("two", "pair"):
            x = x0.clone().cuda()
            y = torch.zeros(B, F, device="cuda")
            n_pairs = 5
            buf = lambda i: 1 if (i == n_pairs - 1 and n_pairs % 3 == 1) else i % 3
            for rep in range(2):
                for i in range(n_pairs):
                    a = _lib.GemvArgs()
                    a.W, a.bias, a.y = Wo[i % 2].data_ptr(), bo.data_ptr(), x.data_ptr()
                    a.B, a.N, a.K, a.groups, a.x_stride, a.y_stride = B, D, D, 1, D, D
                    a.pro, a.act, a.epi = _lib.PRO_ATTN_COMBINE, 0, _lib.EPI_RESIDUAL
                    a.part_o, a.part_ml, a.max_splits, a.row_len = part_o.data_ptr(), part_ml.data_ptr(), MS, dlen.data_ptr()
                    a.kv = _lib.KV(0, 0, MS, 1, H, hd)
                    b = _lib.GemvArgs()
                    b.W, b.bias, b.x, b.y = W1[i % 2].data_ptr(), b1.data_ptr(), x.data_ptr(), y.data_ptr()
                    b.B, b.N, b.K, b.groups, b.x_stride, b.y_stride = B, F, D, 1, D, F
                    b.pro, b.act, b.epi, b.ln_eps = 1, 1, 0, 1e-5
                    assert L.ssrhip_gemv_pair_applicable(C.byref(a), C.byref(b)) == 1
                    if form == "two":
                        _lib.check(L.ssrhip_gemv(C.byref(a), _lib.stream_ptr()))
                        _lib.check(L.ssrhip_gemv(C.byref(b), _lib.stream_ptr()))
                    else:
                        rc = L.ssrhip_gemv_pair(C.byref(a), C.byref(b), ws.data_ptr(), buf(i), buf((i + 1) % n_pairs), _lib.stream_ptr())
                        assert rc == 0, rc
            torch.cuda.synchronize()
            res[form] = (x.cpu().numpy(), y.cpu().numpy())
        assert L.ssrhip_gemv_pair_status(ws.data_ptr(), _lib.stream_ptr()) == 0
        for k, (got, want) in enumerate(zip(res["pair"], res["two"])):
            assert np.isfinite(want).all()
            assert np.array_equal(got, want), ("merge", MS, lens, k, float(np.abs(got - want).max()))
    # shapes that do not qualify are refused, nothing is launched
    a = _lib.GemvArgs(); b = _lib.GemvArgs()
    assert L.ssrhip_gemv_pair_applicable(C.byref(a), C.byref(b)) == 0


def _lib_pair_ws_bytes():
    return 3 * 4096 * 8 + 64                                        # include/ssrhip.h SSRHIP_PAIR_WS_BYTES


@pytest.mark.parametrize("max_pages", [8, 7, 5, 1])
@pytest.mark.parametrize("B", [1, 2, 4])
def test_gemv_seg_combine_and_qkv_append_at_2048(L, B, max_pages):
    """Segment kernel, the two launches with special plumbing at d_model = 2048 / 16 heads: (a) LayerNorm (folded) + QKV with the K/V
    rows appended into the paged cache, (b) split-KV merge prologue (contexts of 1..8 pages: beyond the 6 prefetched ones) + residual."""
    g = torch.Generator().manual_seed(40 + B)
    D, H, hd, n_layer, layer = 2048, 16, 128, 2, 1                 # max_pages: even, odd (the (m, l) pairs are read one by one), a single page
    cap = max_pages * _lib.PAGE
    pool, table = _make_cache(B, max_pages, n_layer, H, hd, g)
    # ---- (a)
    Wt = torch.randn(3 * D, D, generator=g) / math.sqrt(D)
    bias = torch.randn(3 * D, generator=g)
    x = torch.randn(B, D, generator=g) * 1.3 + 0.2
    pos = torch.tensor([min(p, cap - 1) for p in [130, 7, 1023, 512][:B]], dtype=torch.int32)
    ref = F.linear(F.layer_norm(x, (D,), None, None, 1e-5).double(), Wt.double(), bias.double()).float()
    dpool, dtable, dW, db, dx, dpos = dev(pool), dev(table), dev(Wt), dev(bias), dev(x), dev(pos)
    dq = torch.zeros(B, D, device="cuda")
    a = _lib.GemvArgs()
    a.W, a.bias, a.x, a.y = dW.data_ptr(), db.data_ptr(), dx.data_ptr(), dq.data_ptr()
    a.B, a.N, a.K, a.groups, a.x_stride, a.y_stride = B, 3 * D, D, 1, D, D
    a.pro, a.act, a.epi, a.ln_eps = _lib.PRO_LAYERNORM, 0, _lib.EPI_QKV_APPEND, 1e-5
    a.kv = _lib.KV(dpool.data_ptr(), dtable.data_ptr(), max_pages, n_layer, H, hd)
    a.layer, a.kv_pos = layer, dpos.data_ptr()
    _lib.check(L.ssrhip_gemv(C.byref(a), _lib.stream_ptr()))
    sync()
    torch.testing.assert_close(dq.cpu(), ref[:, :D], rtol=3e-5, atol=3e-5)
    newpool = dpool.cpu()
    mask = torch.ones_like(pool, dtype=torch.bool)
    for b in range(B):
        p = int(pos[b])
        page = int(table[b, p // _lib.PAGE])
        for which in (0, 1):
            got = newpool[page, layer, which, :, p % _lib.PAGE, :].reshape(-1)
            torch.testing.assert_close(got, ref[b, (1 + which) * D:(2 + which) * D], rtol=3e-5, atol=3e-5)
        mask[page, layer, :, :, p % _lib.PAGE, :] = False
    assert torch.equal(newpool[mask], pool[mask])
    # ---- (b)
    for lens in ([1000, 129, 1, 640][:B], [5, 1024, 900, 257][:B], [cap, cap - 1, cap, 1][:B]):
        lens = [min(v, cap) for v in lens]
        q = torch.randn(B, D, generator=g)
        att = torch.zeros(B, D)
        for r, ln in enumerate(lens):
            for h in range(H):
                k = _gather(pool, table, r, layer, 0, h, ln)
                v = _gather(pool, table, r, layer, 1, h, ln)
                att[r, h * hd:(h + 1) * hd] = F.scaled_dot_product_attention(q[r, h * hd:(h + 1) * hd].view(1, 1, 1, hd), k.view(1, 1, ln, hd), v.view(1, 1, ln, hd)).view(-1)
        dlen = dev(torch.tensor(lens, dtype=torch.int32))
        part_o = torch.zeros(B * H * max_pages * hd, device="cuda")
        part_ml = torch.zeros(B * H * max_pages * 2, device="cuda")
        at = _lib.AttnArgs()
        dq2 = dev(q)
        at.q, at.q_stride = dq2.data_ptr(), 0
        keep = dev(pool)
        at.kv = _lib.KV(keep.data_ptr(), dtable.data_ptr(), max_pages, n_layer, H, hd)
        at.layer, at.row_seq, at.row_len, at.R, at.max_splits = layer, 0, dlen.data_ptr(), B, max_pages
        at.scale, at.part_o, at.part_ml = 1.0 / math.sqrt(hd), part_o.data_ptr(), part_ml.data_ptr()
        _lib.check(L.ssrhip_attn_decode(C.byref(at), _lib.stream_ptr()))
        Wo = torch.randn(D, D, generator=g) / math.sqrt(D)
        bo = torch.randn(D, generator=g)
        y0 = torch.randn(B, D, generator=g)
        dWo, dbo, dy = dev(Wo), dev(bo), dev(y0.clone())
        ga = _lib.GemvArgs()
        ga.W, ga.bias, ga.y, ga.B, ga.N, ga.K, ga.groups, ga.x_stride, ga.y_stride = dWo.data_ptr(), dbo.data_ptr(), dy.data_ptr(), B, D, D, 1, D, D
        ga.pro, ga.act, ga.epi = _lib.PRO_ATTN_COMBINE, 0, _lib.EPI_RESIDUAL
        ga.part_o, ga.part_ml, ga.max_splits, ga.row_len, ga.kv = part_o.data_ptr(), part_ml.data_ptr(), max_pages, dlen.data_ptr(), at.kv
        _lib.check(L.ssrhip_gemv(C.byref(ga), _lib.stream_ptr()))
        sync()
        want = y0 + F.linear(att.double(), Wo.double(), bo.double()).float()
        torch.testing.assert_close(dy.cpu(), want, rtol=3e-5, atol=3e-5)

# ------------------------------------------------------------------------------------------ attention
def _make_cache(n_seq, max_pages, n_layer, H, hd, g):
    n_pages = n_seq * max_pages
    pool = torch.randn(n_pages, n_layer, 2, H, _lib.PAGE, hd, generator=g)
    # a non-trivial page table: reversed physical order
    table = torch.arange(n_pages - 1, -1, -1, dtype=torch.int32).view(n_seq, max_pages)
    return pool, table


def _gather(pool, table, seq, layer, which, h, length):
    rows = []
    for p in range((length + _lib.PAGE - 1) // _lib.PAGE):
        rows.append(pool[table[seq, p], layer, which, h])
    return torch.cat(rows, 0)[:length]


@pytest.mark.parametrize("hd", [64, 128])
def test_attention_decode_and_combine(L, hd):
    g = torch.Generator().manual_seed(hd)
    H, n_layer, max_pages, layer = 4, 2, 4, 1
    lens = [1, 5, 128, 129, 300, 512, 16, 17, 255, 256, 257, 385]
    R = len(lens)
    pool, table = _make_cache(R, max_pages, n_layer, H, hd, g)
    q = torch.randn(R, H * hd, generator=g)
    ref = torch.zeros(R, H * hd)
    for r, ln in enumerate(lens):
        for h in range(H):
            k = _gather(pool, table, r, layer, 0, h, ln)
            v = _gather(pool, table, r, layer, 1, h, ln)
            o = F.scaled_dot_product_attention(q[r, h * hd:(h + 1) * hd].view(1, 1, 1, hd), k.view(1, 1, ln, hd), v.view(1, 1, ln, hd))
            ref[r, h * hd:(h + 1) * hd] = o.view(-1)
    dpool, dtable, dq = dev(pool), dev(table), dev(q)
    dlen = dev(torch.tensor(lens, dtype=torch.int32))
    part_o = torch.zeros(R * H * max_pages * hd, device="cuda")
    part_ml = torch.zeros(R * H * max_pages * 2, device="cuda")
    out = torch.zeros(R, H * hd, device="cuda")
    a = _lib.AttnArgs()
    a.q, a.q_stride = dq.data_ptr(), 0
    a.kv = _lib.KV(dpool.data_ptr(), dtable.data_ptr(), max_pages, n_layer, H, hd)
    a.layer, a.row_seq, a.row_len, a.R, a.max_splits = layer, 0, dlen.data_ptr(), R, max_pages
    a.scale, a.part_o, a.part_ml = 1.0 / math.sqrt(hd), part_o.data_ptr(), part_ml.data_ptr()
    _lib.check(L.ssrhip_attn_decode(C.byref(a), _lib.stream_ptr()))
    _lib.check(L.ssrhip_attn_combine(C.byref(a), out.data_ptr(), _lib.stream_ptr()))
    sync()
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-5, atol=2e-5)
    # the fused walk over the pages (one workgroup per (row, head), online softmax, no partials): same result, row-major and tiled
    for tiled in (0, 1):
        out2 = torch.full((16 * H * hd,), float("nan"), device="cuda")
        a.out_tiled, a.part_o, a.part_ml = tiled, 0, 0
        _lib.check(L.ssrhip_attn_rows(C.byref(a), out2.data_ptr(), _lib.stream_ptr()))
        sync()
        got2 = _from_tiled(out2.cpu(), R, H * hd) if tiled else out2.cpu()[: R * H * hd].view(R, H * hd)
        torch.testing.assert_close(got2, ref, rtol=2e-5, atol=2e-5)
    a.out_tiled, a.part_o, a.part_ml = 0, part_o.data_ptr(), part_ml.data_ptr()
    assert L.ssrhip_attn_rows(C.byref(a), dq.data_ptr(), _lib.stream_ptr()) != 0          # out must not alias q
    # the same partials merged inside the out-projection GEMV prologue
    Wt = torch.randn(48, H * hd, generator=g) / math.sqrt(H * hd)
    for r0 in (0, 2, 4):
        dW = dev(Wt)
        dy = torch.zeros(2, 48, device="cuda")
        ga = _lib.GemvArgs()
        ga.W, ga.y, ga.B, ga.N, ga.K, ga.groups, ga.x_stride, ga.y_stride = dW.data_ptr(), dy.data_ptr(), 2, 48, H * hd, 1, H * hd, 48
        ga.pro, ga.act, ga.epi = _lib.PRO_ATTN_COMBINE, 0, 0
        ga.part_o = part_o.data_ptr() + 4 * r0 * H * max_pages * hd
        ga.part_ml = part_ml.data_ptr() + 4 * r0 * H * max_pages * 2
        ga.max_splits, ga.row_len, ga.kv = max_pages, dlen.data_ptr() + 4 * r0, a.kv
        _lib.check(L.ssrhip_gemv(C.byref(ga), _lib.stream_ptr()))
        sync()
        torch.testing.assert_close(dy.cpu(), F.linear(ref[r0:r0 + 2], Wt), rtol=2e-5, atol=2e-5)


def test_attention_rows_beyond_64_pages(L):
    """ssrhip_attn_rows keeps a row's page ids in 4 VGPRs (64 pages each, picked with v_readlane): contexts that cross the 64-,
    128- and 192-page register boundaries (8,192 / 16,384 / 24,576 positions), a shuffled table, both register-pair parities of
    the two-page pipeline (odd and even page counts)."""
    g = torch.Generator().manual_seed(77)
    H, hd, n_layer, layer = 2, 64, 1, 0
    lens = [8191, 8193, 16385 + 128, 24577, 300, 1]
    R = len(lens)
    max_pages = (max(lens) + _lib.PAGE - 1) // _lib.PAGE
    n_pages = R * max_pages
    pool = torch.randn(n_pages, n_layer, 2, H, _lib.PAGE, hd, generator=g)
    table = torch.randperm(n_pages, generator=g).to(torch.int32).view(R, max_pages)
    q = torch.randn(R, H * hd, generator=g)
    ref = torch.zeros(R, H * hd)
    for r, ln in enumerate(lens):
        for h in range(H):
            k = _gather(pool, table, r, layer, 0, h, ln)
            v = _gather(pool, table, r, layer, 1, h, ln)
            ref[r, h * hd:(h + 1) * hd] = F.scaled_dot_product_attention(q[r, h * hd:(h + 1) * hd].view(1, 1, 1, hd), k.view(1, 1, ln, hd), v.view(1, 1, ln, hd)).view(-1)
    dpool, dtable, dq = dev(pool), dev(table), dev(q)
    dlen = dev(torch.tensor(lens, dtype=torch.int32))
    out = torch.full((R, H * hd), float("nan"), device="cuda")
    a = _lib.AttnArgs()
    a.q, a.q_stride = dq.data_ptr(), 0
    a.kv = _lib.KV(dpool.data_ptr(), dtable.data_ptr(), max_pages, n_layer, H, hd)
    a.layer, a.row_seq, a.row_len, a.R, a.max_splits = layer, 0, dlen.data_ptr(), R, max_pages
    a.scale = 1.0 / math.sqrt(hd)
    _lib.check(L.ssrhip_attn_rows(C.byref(a), out.data_ptr(), _lib.stream_ptr()))
    sync()
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-5, atol=2e-5)
    a.kv = _lib.KV(dpool.data_ptr(), dtable.data_ptr(), 257, n_layer, H, hd)
    assert L.ssrhip_attn_rows(C.byref(a), out.data_ptr(), _lib.stream_ptr()) != 0          # more than 256 pages per row: refused, not mis-read


@pytest.mark.parametrize("hd", [64, 128])
def test_attention_prefill_tiled_matches_torch(L, hd):
    """ssrhip_attn_prefill: causal attention of whole prompts from the paged cache (K/V tiles in LDS, both products on the matrix
    core) vs F.scaled_dot_product_attention(is_causal=True) per (sequence, head): ragged lengths around the 32-key tile, the
    128-query block and the 128-position page boundaries, a shuffled page table, q taken from a packed qkv buffer."""
    g = torch.Generator().manual_seed(100 + hd)
    H, n_layer, max_pages, layer = 3, 2, 4, 1
    lens = [1, 33, 130, 300, 128, 129, 257, 64]
    n_seq = len(lens)
    D = H * hd
    pool, table = _make_cache(n_seq, max_pages, n_layer, H, hd, g)
    perm = torch.randperm(n_seq * max_pages, generator=g).to(torch.int32).view(n_seq, max_pages)
    table = perm
    R = sum(lens)
    qkv = torch.randn(R, 3 * D, generator=g)
    starts = [0]
    for ln in lens:
        starts.append(starts[-1] + ln)
    ref = torch.zeros(R, D)
    for s_, ln in enumerate(lens):
        for h in range(H):
            k = _gather(pool, table, s_, layer, 0, h, ln)
            v = _gather(pool, table, s_, layer, 1, h, ln)
            q = qkv[starts[s_]:starts[s_ + 1], h * hd:(h + 1) * hd]
            o = F.scaled_dot_product_attention(q.view(1, 1, ln, hd), k.view(1, 1, ln, hd), v.view(1, 1, ln, hd), is_causal=True)
            ref[starts[s_]:starts[s_ + 1], h * hd:(h + 1) * hd] = o.view(ln, hd)
    dpool, dtable, dq = dev(pool), dev(table), dev(qkv)
    dstart = dev(torch.tensor(starts, dtype=torch.int32))
    out = torch.full((R, D), float("nan"), device="cuda")
    a = _lib.AttnArgs()
    a.q, a.q_stride = dq.data_ptr(), 3 * D
    a.kv = _lib.KV(dpool.data_ptr(), dtable.data_ptr(), max_pages, n_layer, H, hd)
    a.layer, a.scale = layer, 1.0 / math.sqrt(hd)
    _lib.check(L.ssrhip_attn_prefill(C.byref(a), dstart.data_ptr(), n_seq, max(lens), out.data_ptr(), _lib.stream_ptr()))
    sync()
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-5, atol=2e-5)


def test_qkv_append_writes_cache(L):
    g = torch.Generator().manual_seed(9)
    B, D, H, hd, n_layer, max_pages, layer = 2, 256, 2, 128, 2, 3, 1
    pool, table = _make_cache(B, max_pages, n_layer, H, hd, g)
    Wt = torch.randn(3 * D, D, generator=g) / math.sqrt(D)
    bias = torch.randn(3 * D, generator=g)
    x = torch.randn(B, D, generator=g)
    lw, lb = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    pos = torch.tensor([130, 7], dtype=torch.int32)
    ref = F.linear(F.layer_norm(x, (D,), lw, lb, 1e-5), Wt, bias)
    dpool, dtable, dW, db, dx, dlw, dlb, dpos = dev(pool), dev(table), dev(Wt), dev(bias), dev(x), dev(lw), dev(lb), dev(pos)
    dq = torch.zeros(B, D, device="cuda")
    a = _lib.GemvArgs()
    a.W, a.bias, a.x, a.y = dW.data_ptr(), db.data_ptr(), dx.data_ptr(), dq.data_ptr()
    a.B, a.N, a.K, a.groups, a.x_stride, a.y_stride = B, 3 * D, D, 1, D, D
    a.pro, a.act, a.epi = _lib.PRO_LAYERNORM, 0, _lib.EPI_QKV_APPEND
    a.ln_w, a.ln_b, a.ln_eps = dlw.data_ptr(), dlb.data_ptr(), 1e-5
    a.kv = _lib.KV(dpool.data_ptr(), dtable.data_ptr(), max_pages, n_layer, H, hd)
    a.layer, a.kv_pos = layer, dpos.data_ptr()
    _lib.check(L.ssrhip_gemv(C.byref(a), _lib.stream_ptr()))
    sync()
    torch.testing.assert_close(dq.cpu(), ref[:, :D], rtol=2e-5, atol=2e-5)
    newpool = dpool.cpu()
    for b in range(B):
        p = int(pos[b])
        page = int(table[b, p // _lib.PAGE])
        for which in (0, 1):
            got = newpool[page, layer, which, :, p % _lib.PAGE, :].reshape(-1)
            torch.testing.assert_close(got, ref[b, (1 + which) * D:(2 + which) * D], rtol=2e-5, atol=2e-5)
    # nothing else in the pool was touched
    mask = torch.ones_like(pool, dtype=torch.bool)
    for b in range(B):
        p = int(pos[b])
        mask[int(table[b, p // _lib.PAGE]), layer, :, :, p % _lib.PAGE, :] = False
    assert torch.equal(newpool[mask], pool[mask])


# ------------------------------------------------------------------------------------------ embed
def test_embed_matches_oracle(L):
    args = W.lm_args_tiny()
    sd = W.lm_state_dict(args, seed=4)
    sd["audio_positional_embedding.alpha"] = torch.tensor([0.7])
    sd["text_positional_embedding.alpha"] = torch.tensor([1.3])
    K, D = args.n_codebooks, args.d_model
    card = args.audio_vocab_size + args.n_special + args.max_n_spans
    g = torch.Generator().manual_seed(1)
    R = 9
    tok = torch.randint(0, card, (R, 4), generator=g, dtype=torch.int32)
    kind = torch.tensor([0, 1, 1, 0, 1, 1, 1, 0, 1], dtype=torch.int32)
    tok[kind == 0, 0] = torch.randint(0, args.text_vocab_size + 1, (int((kind == 0).sum()),), generator=g, dtype=torch.int32)
    pos = torch.randint(0, 500, (R,), generator=g, dtype=torch.int32)
    pe = O.sine_pe(512, D)
    ref = torch.zeros(R, D)
    for r in range(R):
        if kind[r] == 0:
            e = F.embedding(tok[r, 0].long(), sd["text_embedding.word_embeddings.weight"])
            ref[r] = e * 1.0 + sd["text_positional_embedding.alpha"] * pe[pos[r]]
        else:
            e = torch.stack([F.embedding(tok[r, k].long(), sd[f"audio_embedding.{k}.word_embeddings.weight"]) for k in range(K)]).sum(0)
            ref[r] = e * 1.0 + sd["audio_positional_embedding.alpha"] * pe[pos[r]]
    a = _lib.EmbedArgs()
    te = dev(sd["text_embedding.word_embeddings.weight"])
    ae = dev(torch.stack([sd[f"audio_embedding.{k}.word_embeddings.weight"] for k in range(K)]))
    dpe, dtok, dpos, dkind = dev(pe), dev(tok), dev(pos), dev(kind)
    out = torch.zeros(R, D, device="cuda")
    a.text_emb, a.audio_emb, a.pe, a.alpha_text, a.alpha_audio = te.data_ptr(), ae.data_ptr(), dpe.data_ptr(), 1.3, 0.7
    a.tok, a.pos, a.kind, a.R, a.D, a.K, a.card, a.out = dtok.data_ptr(), dpos.data_ptr(), dkind.data_ptr(), R, D, K, card, out.data_ptr()
    _lib.check(L.ssrhip_embed(C.byref(a), _lib.stream_ptr()))
    sync()
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------------------------------ GEMM / LN / scatter
@pytest.mark.parametrize("M,N,K,act,res", [(600, 384, 2048, 0, 0), (70, 130, 64, 1, 0), (64, 128, 512, 0, 1), (333, 72, 128, 2, 1), (5, 2056, 1024, 0, 0),
                                           (5000, 32, 192, 0, 1), (1000, 64, 448, 2, 0), (777, 1, 448, 0, 0), (300, 33, 64, 1, 1),   # N <= 64: the tall 256-row tiles
                                           (2100, 2048, 256, 1, 1)])   # >= 256 wide tiles: 64x128 (the smaller shapes above use 64x64)
def test_gemm_matches_torch(L, M, N, K, act, res):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    Wt = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    c0 = torch.randn(M, N, generator=g)
    ref = F.linear(A, Wt, bias)
    ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    ref = c0 + ref if res else ref
    dA, dW, db, dC = dev(A), dev(Wt), dev(bias), dev(c0.clone())
    a = _lib.GemmArgs(dA.data_ptr(), dW.data_ptr(), db.data_ptr(), dC.data_ptr(), M, N, K, K, N, act, res)
    _lib.check(L.ssrhip_gemm(C.byref(a), _lib.stream_ptr()))
    sync()
    torch.testing.assert_close(dC.cpu(), ref, rtol=3e-5, atol=3e-5)


@pytest.mark.parametrize("M,N,K,act,res", [(3000, 2056, 1000, 2, 1), (2300, 1280, 36, 0, 0), (50000, 128, 260, 1, 0)])
def test_gemm_pipelined_tile_matches_torch(L, M, N, K, act, res):
    """Shapes whose 128 x 128 tiling fills the GPU (>= 384 workgroups) take `gemm_pipe_kernel` (128 x 128 x 32, two LDS buffers,
    XCD-aware tile order): ragged M / N tiles, a K that is no multiple of the 32-wide k-tile (incl. K < 64), every epilogue."""
    test_gemm_matches_torch(L, M, N, K, act, res)


def test_gemm_batched_strided_view_with_every_codec_extension(L):
    """The codec's use of ssrhip_gemm in one call (include/ssrhip.h ssrhip_gemm_args): batch in grid.z with per-item strides, A a
    STRIDED VIEW (lda < K: the im2col rows of a stride-2 convolution overlap), ELU on load, skip tensor R, the transposed
    convolution's time mask, and the per-row class bias — large enough for the pipelined tile, against plain torch ops."""
    g = torch.Generator().manual_seed(77)
    B, Cin, k, s, Cout, T = 3, 32, 4, 2, 160, 20000
    rows = T * s + k                                            # time rows of the (already padded) input
    x = torch.randn(B, rows, Cin, generator=g)
    Wt = torch.randn(Cout, k * Cin, generator=g) / math.sqrt(k * Cin)
    bias = torch.randn(Cout, generator=g)
    R = torch.randn(B, T, Cout, generator=g)
    cls_bias = torch.randn(2, Cout, generator=g)
    rep = 50
    cls = torch.randint(0, 2, (B, T // rep), generator=g, dtype=torch.int32)
    view = torch.stack([F.elu(x)[:, t * s: t * s + k].reshape(B, k * Cin) for t in range(0, T, 997)], 1)     # spot rows of the im2col matrix
    ref_rows = list(range(0, T, 997))
    ref = view @ Wt.t() + bias + R[:, ref_rows] + cls_bias[cls.long()[:, [t // rep for t in ref_rows]]]
    dx, dW, db, dR, dcb, dcl = dev(x), dev(Wt), dev(bias), dev(R), dev(cls_bias), dev(cls)
    out = torch.full((B, T, Cout), -7.0, device="cuda")
    a = _lib.GemmArgs()
    a.A, a.W, a.bias, a.C = dx.data_ptr(), dW.data_ptr(), db.data_ptr(), out.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldc = T, Cout, k * Cin, s * Cin, Cout
    a.act_in, a.R, a.ldr, a.batch = _lib.ACT_ELU, dR.data_ptr(), Cout, B
    a.strideA, a.strideC, a.strideR = rows * Cin, T * Cout, T * Cout
    a.rbias, a.rclass, a.rrep, a.rclass_stride = dcb.data_ptr(), dcl.data_ptr(), rep, T // rep
    lo, hi = 3, T - 5                                           # time mask in units of tm_c = Cout floats: rows outside [lo, hi) stay untouched
    a.tm_c, a.tm_lo, a.tm_hi = Cout, lo, hi
    _lib.check(L.ssrhip_gemm(C.byref(a), _lib.stream_ptr()))
    sync()
    got = out.cpu()
    keep = [i for i, t in enumerate(ref_rows) if lo <= t < hi]
    torch.testing.assert_close(got[:, [ref_rows[i] for i in keep]], ref[:, keep], rtol=3e-5, atol=3e-5)
    assert (got[:, :lo] == -7.0).all() and (got[:, hi:] == -7.0).all()


def _split_planes(L, dW):
    planes = torch.empty(3, dW.shape[0], dW.shape[1], dtype=torch.int16, device="cuda")
    _lib.check(L.ssrhip_split_weights(dW.data_ptr(), planes.data_ptr(), dW.numel(), _lib.stream_ptr()))
    return planes


def test_split_weights_is_an_exact_three_way_split(L):
    """w == w0 + w1 + w2 EXACTLY (bf16 pieces, fp32 sums in any order of the two small ones first), for random full-mantissa values,
    tiny and huge magnitudes, zeros and negative values (magnitudes whose 2^-16 residual would be an fp32 subnormal, |w| < ~1e-33, lose the third piece: subnormals are flushed)."""
    g = torch.Generator().manual_seed(5)
    w = torch.randn(64, 96, generator=g) * torch.logspace(-20, 20, 64 * 96).view(64, 96)[torch.randperm(64, generator=g)]
    w[0, :8] = 0.0
    w[1, :8] = torch.tensor([1.0, -1.0, 3.0e-29, 65504.0, -1.0000001, 0.33333334, 1e-30, -7.5])
    planes = _split_planes(L, dev(w)).cpu()
    pieces = (planes.to(torch.int32) << 16).view(torch.float32)           # bf16 bits -> fp32
    recon = (pieces[2].double() + pieces[1].double() + pieces[0].double())
    assert torch.equal(recon.float(), w) and torch.equal(recon, w.double())      # exact, not merely close


@pytest.mark.parametrize("M,N,K,act,res", [(3000, 2056, 1000, 2, 1), (4096, 512, 2048, 0, 0), (50000, 128, 264, 1, 0), (300, 200, 64, 0, 1), (77, 1030, 8, 2, 0)])
def test_gemm_split_bf16x3_is_as_accurate_as_the_fp32_chain(L, M, N, K, act, res):
    """csrc/gemm_split.hip (taken when the caller supplies the bf16 weight planes and the grid is large): fp32 operands split exactly
    into three bf16 pieces, six cross products on the bf16 matrix cores. Against an fp64 reference its error must be no worse than
    1.5x the exact fp32 MFMA chain's (it is usually smaller: its products are exact), and within the GEMM's usual 3e-5 of torch."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    Wt = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    c0 = torch.randn(M, N, generator=g)
    ref64 = A.double() @ Wt.double().t() + bias.double()
    ref64 = F.relu(ref64) if act == 1 else (F.gelu(ref64) if act == 2 else ref64)
    ref64 = c0.double() + ref64 if res else ref64
    dA, dW, db = dev(A), dev(Wt), dev(bias)
    planes = _split_planes(L, dW)
    outs = []
    for use_split in (False, True):
        dC = dev(c0.clone())
        a = _lib.GemmArgs(dA.data_ptr(), dW.data_ptr(), db.data_ptr(), dC.data_ptr(), M, N, K, K, N, act, res)
        a.W_split = planes.data_ptr() if use_split else 0
        _lib.check(L.ssrhip_gemm(C.byref(a), _lib.stream_ptr()))
        sync()
        outs.append(dC.cpu())
    exact, split = outs
    assert not torch.equal(exact, split)                                   # really another kernel
    e_exact = (exact.double() - ref64).abs().max().item()
    e_split = (split.double() - ref64).abs().max().item()
    assert e_split <= 1.5 * e_exact + 1e-7, (e_split, e_exact)
    torch.testing.assert_close(split, ref64.float(), rtol=3e-5, atol=3e-5)


def test_gemm_split_takes_every_codec_extension(L, monkeypatch):
    """The split kernel behind the same call as `test_gemm_batched_strided_view_with_every_codec_extension` (strided view, ELU on load,
    skip tensor, time mask, per-row class bias, batch): equal to the exact kernel within the GEMM tolerance."""
    g = torch.Generator().manual_seed(78)
    B, Cin, k, s, Cout, T = 3, 32, 4, 2, 160, 20000
    rows = T * s + k
    x = torch.randn(B, rows, Cin, generator=g)
    Wt = torch.randn(Cout, k * Cin, generator=g) / math.sqrt(k * Cin)
    bias = torch.randn(Cout, generator=g)
    R = torch.randn(B, T, Cout, generator=g)
    cls_bias = torch.randn(2, Cout, generator=g)
    rep = 50
    cls = torch.randint(0, 2, (B, T // rep), generator=g, dtype=torch.int32)
    dx, dW, db, dR, dcb, dcl = dev(x), dev(Wt), dev(bias), dev(R), dev(cls_bias), dev(cls)
    planes = _split_planes(L, dW)
    outs = []
    for use_split in (False, True):
        out = torch.full((B, T, Cout), -7.0, device="cuda")
        a = _lib.GemmArgs()
        a.A, a.W, a.bias, a.C = dx.data_ptr(), dW.data_ptr(), db.data_ptr(), out.data_ptr()
        a.M, a.N, a.K, a.lda, a.ldc = T, Cout, k * Cin, s * Cin, Cout
        a.act_in, a.R, a.ldr, a.batch = _lib.ACT_ELU, dR.data_ptr(), Cout, B
        a.strideA, a.strideC, a.strideR = rows * Cin, T * Cout, T * Cout
        a.rbias, a.rclass, a.rrep, a.rclass_stride = dcb.data_ptr(), dcl.data_ptr(), rep, T // rep
        a.tm_c, a.tm_lo, a.tm_hi = Cout, 3, T - 5
        a.W_split = planes.data_ptr() if use_split else 0
        _lib.check(L.ssrhip_gemm(C.byref(a), _lib.stream_ptr()))
        sync()
        outs.append(out.cpu())
    assert not torch.equal(outs[0], outs[1])
    torch.testing.assert_close(outs[1], outs[0], rtol=3e-5, atol=3e-5)
    assert (outs[1][:, :3] == -7.0).all() and (outs[1][:, T - 5:] == -7.0).all()


def test_gemm_is_transpose_safe(L):
    """A = I with an ASYMMETRIC W: catches a swapped C/D lane map (cdna guide §3)."""
    n = 96
    A = torch.eye(n)
    Wt = torch.arange(n * n, dtype=torch.float32).view(n, n) / 100.0      # W[i][j] != W[j][i]
    dA, dW = dev(A), dev(Wt)
    dC = torch.zeros(n, n, device="cuda")
    a = _lib.GemmArgs(dA.data_ptr(), dW.data_ptr(), 0, dC.data_ptr(), n, n, n, n, n, 0, 0)
    _lib.check(L.ssrhip_gemm(C.byref(a), _lib.stream_ptr()))
    sync()
    assert torch.equal(dC.cpu(), Wt.t().contiguous())


def test_layernorm_and_kv_scatter(L):
    g = torch.Generator().manual_seed(2)
    R, D, H, hd = 37, 256, 2, 128
    x = torch.randn(R, D, generator=g) * 2 + 0.5
    w, b = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    dx, dw, db = dev(x), dev(w), dev(b)
    y = torch.zeros(R, D, device="cuda")
    _lib.check(L.ssrhip_layernorm(dx.data_ptr(), dw.data_ptr(), db.data_ptr(), 1e-5, y.data_ptr(), R, D, _lib.stream_ptr()))
    sync()
    torch.testing.assert_close(y.cpu(), F.layer_norm(x, (D,), w, b, 1e-5), rtol=1e-5, atol=1e-5)
    n_layer, max_pages, layer = 2, 2, 0
    pool, table = _make_cache(2, max_pages, n_layer, H, hd, g)
    qkv = torch.randn(R, 3 * D, generator=g)
    seq = torch.tensor([r % 2 for r in range(R)], dtype=torch.int32)
    pos = torch.tensor([100 + r for r in range(R)], dtype=torch.int32)
    dpool, dtable, dqkv, dseq, dpos = dev(pool), dev(table), dev(qkv), dev(seq), dev(pos)
    kv = _lib.KV(dpool.data_ptr(), dtable.data_ptr(), max_pages, n_layer, H, hd)
    _lib.check(L.ssrhip_kv_scatter(dqkv.data_ptr(), C.byref(kv), layer, dseq.data_ptr(), dpos.data_ptr(), R, _lib.stream_ptr()))
    sync()
    newpool = dpool.cpu()
    for r in range(R):
        p = int(pos[r])
        page = int(table[int(seq[r]), p // _lib.PAGE])
        for which in (0, 1):
            assert torch.equal(newpool[page, layer, which, :, p % _lib.PAGE, :].reshape(-1), qkv[r, (1 + which) * D:(2 + which) * D])


# ------------------------------------------------------------------------------------------ sampler
def _run_sampler_script(L, args, logits_seq, knobs, noise_seq, text_len, audio_pos0, n_spans=1):
    """Drive ssrhip_sample step by step on scripted logits; return (samples [S,K], dbg logits [S,K,card], states)."""
    K = args.n_codebooks
    card = args.audio_vocab_size + args.n_special + args.max_n_spans
    use_cfg = knobs["aug_text"]
    Brows = 2 if use_cfg else 1
    S = len(logits_seq)
    cfg = _lib.SamplerCfg()
    cfg.top_k, cfg.top_p, cfg.temperature, cfg.stop_repetition = knobs["top_k"], knobs["top_p"], knobs["temperature"], knobs["stop_repetition"]
    cfg.cfg_coef, cfg.cfg_one_minus, cfg.cfg_stride, cfg.use_cfg = knobs["cfg_coef"], 1 - knobs["cfg_coef"], knobs["cfg_stride"], int(use_cfg)
    sil = knobs["silence_tokens"]
    cfg.n_silence = len(sil)
    for i, s in enumerate(sil):
        cfg.silence[i] = s
    cfg.text_len, cfg.n_spans = text_len, n_spans
    cfg.empty_token, cfg.eog, cfg.eos, cfg.sos, cfg.mts, cfg.max_n_spans = args.empty_token, args.eog, args.eos, args.sos, args.mts, args.max_n_spans
    cfg.max_steps = S
    cfg.use_noise = int(noise_seq is not None)
    st = _lib.SamplerState()
    st.num_cfg_tag, st.prev_token, st.audio_pos = 1, -1, audio_pos0
    dcfg = torch.frombuffer(bytearray(bytes(cfg)), dtype=torch.uint8).cuda()
    dst = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).cuda()
    dnoise = dev(torch.stack(noise_seq).unsqueeze(0)) if noise_seq is not None else None
    gen = torch.zeros(1, S, K, dtype=torch.int32, device="cuda")
    next_tok = torch.zeros(Brows, 4, dtype=torch.int32, device="cuda")
    next_pos = torch.zeros(Brows, dtype=torch.int32, device="cuda")
    kv_pos = torch.zeros(Brows, dtype=torch.int32, device="cuda")
    row_len = torch.zeros(Brows, dtype=torch.int32, device="cuda")
    dbg = torch.zeros(1, K, card, device="cuda")
    outs, dbgs = [], []
    for s in range(S):
        dl = dev(logits_seq[s])
        a = _lib.SampleArgs()
        a.logits, a.n_utt, a.K, a.card = dl.data_ptr(), 1, K, card
        a.cfg, a.state = dcfg.data_ptr(), dst.data_ptr()
        a.noise = dnoise.data_ptr() if dnoise is not None else 0
        a.generated, a.next_tok, a.next_pos, a.kv_pos, a.row_len, a.dbg_logits = gen.data_ptr(), next_tok.data_ptr(), next_pos.data_ptr(), kv_pos.data_ptr(), row_len.data_ptr(), dbg.data_ptr()
        _lib.check(L.ssrhip_sample(C.byref(a), _lib.stream_ptr()))
        sync()
        state = _lib.SamplerState.from_buffer_copy(bytes(dst.cpu().numpy().tobytes()))
        dbgs.append(dbg[0].cpu().clone())
        if state.done:
            break
    return gen[0, : state.n_steps].cpu().long(), dbgs, state


@pytest.mark.parametrize("case", ["greedy_cfg", "topk_topp", "topp_temp", "silence", "nocfg_topk"])
def test_sampler_state_machine_matches_oracle(L, case):
    args = W.lm_args_tiny()
    K = args.n_codebooks
    card = args.audio_vocab_size + args.n_special + args.max_n_spans
    knobs = dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=2, silence_tokens=[3, 7, 11], cfg_coef=1.5, cfg_stride=3, aug_text=True)
    if case == "topk_topp":
        knobs.update(top_k=12, top_p=0.8)
    elif case == "topp_temp":
        knobs.update(top_k=0, top_p=0.7, temperature=2.0, cfg_stride=1, cfg_coef=1.3)
    elif case == "silence":
        knobs.update(top_k=1, stop_repetition=1, cfg_stride=1)
    elif case == "nocfg_topk":
        knobs.update(top_k=5, top_p=0.95, aug_text=False)
    g = torch.Generator().manual_seed({'greedy_cfg': 1, 'topk_topp': 2, 'topp_temp': 3, 'silence': 4, 'nocfg_topk': 5}[case])
    S, text_len, audio_pos0 = 40, 3, 8      # cap triggers when audio_pos+1 > 30
    Brows = 2 if knobs["aug_text"] else 1
    logits_seq, noise_seq = [], []
    for s in range(S):
        lg = torch.randn(Brows, K, 1, card, generator=g) * 2.0
        if case == "silence" and 2 <= s < 12:
            lg[:, 0, 0, 7] = 9.0           # keep emitting silence token 7 until the penalty bites
        if case == "greedy_cfg" and s == 15:
            lg[:, 0, 0, args.eog] = 50.0   # argmax == eog stop rule
        logits_seq.append(lg)
        noise_seq.append(torch.empty(K, card).exponential_(1, generator=g))
    # oracle
    st = O.SpanState()
    ref_samples, ref_logits = [], []
    rec = {}
    for s in range(S):
        smp = O.step_logits_to_samples(logits_seq[s].clone(), st, args, audio_pos0 + s + 1, text_len,
                                       top_k=knobs["top_k"], top_p=knobs["top_p"], temperature=knobs["temperature"],
                                       stop_repetition=knobs["stop_repetition"], silence_tokens=knobs["silence_tokens"],
                                       cfg_coef=knobs["cfg_coef"], cfg_stride=knobs["cfg_stride"], aug_text=knobs["aug_text"],
                                       noise=noise_seq[s], rec=rec)
        ref_samples.append(smp.squeeze(-1).clone())
        if st.num_eog == K:
            break
    ref = torch.stack(ref_samples)
    got, dbgs, state = _run_sampler_script(L, args, [l.squeeze(2) for l in logits_seq], knobs, noise_seq, text_len, audio_pos0)
    assert state.done == 1
    assert torch.equal(got, ref), (got, ref)
    for s in range(len(ref)):
        torch.testing.assert_close(dbgs[s], rec["edited_logits"][s], rtol=0, atol=0)   # edits + CFG combine are exact


@pytest.mark.parametrize("case", ["greedy_cfg", "topk_topp", "topp_temp", "silence", "nocfg_topk"])
def test_sampler_state_machine_matches_reference_script(L, golden_dir, case):
    """the same five scripts, but the expected tokens / edited logits come from the REFERENCE's own decode loop
    (tests/golden/sampler_script.npz, oracle/make_golden.py::make_state_machine), not from the oracle."""
    from tests_script_knobs import SCRIPT_KNOBS
    g = np.load(os.path.join(golden_dir, "sampler_script.npz"))
    args = W.lm_args_tiny()
    knobs = dict(SCRIPT_KNOBS[case], silence_tokens=[3, 7, 11])
    logits, noise = torch.from_numpy(g[f"{case}_logits"]), torch.from_numpy(g[f"{case}_noise"])
    got, dbgs, state = _run_sampler_script(L, args, [l for l in logits], knobs, [n for n in noise], int(g[f"{case}_text_len"]), int(g[f"{case}_audio_pos0"]))
    assert state.done == 1 and state.n_steps == logits.shape[0]
    assert np.array_equal(got.numpy(), g[f"{case}_samples"]), (got, g[f"{case}_samples"])
    for s in range(logits.shape[0]):
        np.testing.assert_array_equal(dbgs[s].numpy(), g[f"{case}_edited_logits"][s])     # CFG combine + edits are exact


def _full_card_script(case, top_k, S=8):
    args = W.lm_args_830m()
    K = args.n_codebooks
    card = args.audio_vocab_size + args.n_special + args.max_n_spans
    g = torch.Generator().manual_seed(hash((case, top_k)) % 1000)
    logits_seq, noise_seq = [], []
    for s in range(S):
        lg = torch.randn(2, K, 1, card, generator=g) * (0.05 if case == "narrow" else 2.0)
        if case == "ties":
            lg = torch.round(lg * 2) / 2
        if case == "flat":
            lg = torch.zeros_like(lg)
        lg[..., args.eog] -= 30.0                          # keep the scripted run going
        logits_seq.append(lg)
        noise_seq.append(torch.empty(K, card).exponential_(1, generator=g))
    return args, logits_seq, noise_seq


# top-p on tied logits is not compared with the reference: its sorted-prefix cut falls INSIDE a group of equal logits at a
# position decided by torch.sort's tie order, which no order-free selection can (or should) reproduce. top-k keeps ties whole.
@pytest.mark.parametrize("case,top_k,top_p", [("normal", 40, 0.8), ("normal", 0, 0.8), ("normal", 300, 1.0), ("normal", 40, 0.999),
                                              ("narrow", 40, 0.8), ("narrow", 0, 0.5), ("ties", 40, 1.0), ("ties", 300, 1.0), ("flat", 40, 1.0)])
def test_sampler_full_card_selection_paths(L, case, top_k, top_p):
    """card = 2056 (the 830M shape: 9 logits per lane): the value-bin selection on ordinary logits ("normal", "narrow"), and the
    radix fallback when one bin holds more than 64 elements ("ties": logits rounded to halves, "flat": all equal). Tokens
    must equal the oracle's for the same Exp(1) noise."""
    args, logits_seq, noise_seq = _full_card_script(case, top_k)
    K = args.n_codebooks
    sil = [1388, 1898, 131]
    knobs = dict(top_k=top_k, top_p=top_p, temperature=1.0, stop_repetition=2, silence_tokens=sil, cfg_coef=1.5, cfg_stride=2, aug_text=True)
    text_len, audio_pos0 = 100, 8
    st = O.SpanState()
    ref = []
    for s in range(len(logits_seq)):
        smp = O.step_logits_to_samples(logits_seq[s].clone(), st, args, audio_pos0 + s + 1, text_len, top_k=top_k, top_p=top_p, temperature=1.0,
                                       stop_repetition=2, silence_tokens=sil, cfg_coef=1.5, cfg_stride=2, aug_text=True, noise=noise_seq[s])
        ref.append(smp.squeeze(-1).clone())
    got, _, state = _run_sampler_script(L, args, [l.squeeze(2) for l in logits_seq], knobs, noise_seq, text_len, audio_pos0)
    assert torch.equal(got, torch.stack(ref)), (case, got, torch.stack(ref))


@pytest.mark.parametrize("case,top_k,top_p", [("normal", 40, 0.8), ("narrow", 0, 0.6), ("ties", 40, 0.8), ("flat", 7, 0.3)])
def test_sampler_bin_path_equals_radix_path(L, monkeypatch, case, top_k, top_p):
    """The value-bin selection and the radix selection define the same thresholds with the same integer arithmetic: the
    sampled tokens are identical, ties or not (SSRHIP_SAMPLE_RADIX=1 forces the radix path)."""
    args, logits_seq, noise_seq = _full_card_script(case, top_k)
    knobs = dict(top_k=top_k, top_p=top_p, temperature=1.0, stop_repetition=2, silence_tokens=[1388, 1898, 131], cfg_coef=1.5, cfg_stride=2, aug_text=True)
    seq = [l.squeeze(2) for l in logits_seq]
    got, _, _ = _run_sampler_script(L, args, seq, knobs, noise_seq, 100, 8)
    monkeypatch.setenv("SSRHIP_SAMPLE_RADIX", "1")
    ref, _, _ = _run_sampler_script(L, args, seq, knobs, noise_seq, 100, 8)
    assert torch.equal(got, ref)


def test_sampler_filter_golden(L, golden_dir):
    """top-k / top-p keep-sets against the reference's top_k_top_p_filtering (golden), via the sampled
    token: with noise == 1 the draw is the argmax of the filtered distribution; with huge noise on the
    kept set's complement nothing outside the keep-set can ever be drawn."""
    import os
    gd = np.load(os.path.join(golden_dir, "sampler.npz"))
    base = torch.from_numpy(gd["logits"])                         # [4,72]
    args = W.lm_args_tiny()
    K, card = 4, 72
    for k in (0, 1, 3, 10):
        for p in (1.0, 0.9, 0.5, 0.05):
            # rows 1 and 2 contain exact ties (kept/removed by sort order in the reference): skip them here
            for trial in range(6):
                g = torch.Generator().manual_seed(trial)
                noise = torch.empty(K, card).exponential_(1, generator=g)
                knobs = dict(top_k=k, top_p=p, temperature=1.0, stop_repetition=-1, silence_tokens=[], cfg_coef=1.0, cfg_stride=1, aug_text=False)
                # neutralise the state machine: start past the "first K-1 steps" rule by faking num_gen via 4 warm-up steps
                seq = [torch.zeros(1, K, card) for _ in range(3)] + [base.clone().unsqueeze(0)]
                nz = [torch.ones(K, card)] * 3 + [noise]
                got, dbgs, state = _run_sampler_script(L, args, seq, knobs, nz, text_len=100, audio_pos0=0)
                edited = dbgs[3]
                ref_f = O.top_k_top_p_filtering(edited.clone(), top_k=k, top_p=p)
                probs = torch.softmax(ref_f, -1)
                want = torch.argmax(probs / noise, -1)
                for row in (0, 3):
                    assert got[3, row] == want[row], (k, p, trial, row)


def test_gemm_split_result_of_an_item_does_not_depend_on_the_batch(L):
    """The split GEMM picks 64- or 128-row tiles by the size of the grid (i.e. by the batch); per output element both do the same
    arithmetic, so item 0 of a batch of 12 (128-row tiles) must be BIT-identical to the same item computed alone (64-row tiles)."""
    g = torch.Generator().manual_seed(91)
    M, N, K, B = 1500, 512, 1024, 12
    A = torch.randn(B, M, K, generator=g)
    Wt = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    dA, dW, db = dev(A), dev(Wt), dev(bias)
    planes = _split_planes(L, dW)

    def run(batch):
        out = torch.zeros(batch, M, N, device="cuda")
        a = _lib.GemmArgs()
        a.A, a.W, a.bias, a.C = dA.data_ptr(), dW.data_ptr(), db.data_ptr(), out.data_ptr()
        a.M, a.N, a.K, a.lda, a.ldc = M, N, K, K, N
        a.act_in, a.batch, a.strideA, a.strideC = _lib.ACT_ELU, batch, M * K, M * N
        a.W_split = planes.data_ptr()
        _lib.check(L.ssrhip_gemm(C.byref(a), _lib.stream_ptr()))
        sync()
        return out.cpu()

    alone, many = run(1), run(B)
    assert torch.equal(alone[0], many[0])


@pytest.mark.parametrize("M,N,K,B,tm", [(1500, 512, 1024, 12, False), (777, 1280, 256, 5, False), (130, 640, 64, 3, False), (901, 512, 128, 7, True)])
def test_gemm_split_xcd_tile_order_changes_nothing(L, monkeypatch, M, N, K, B, tm):
    """Round 6: `gemm_split_dma_kernel` takes its tile from an XCD-aware remap of the launch index (the N-tiles of a row-tile on one XCD:
    one fabric read of A instead of one per XCD). The remap only decides WHICH workgroup computes a tile: with it and without it
    (SSRHIP_GEMM_XCD=0, read at every launch) the outputs must be BIT-identical — grids whose tile count is not a multiple of 8, ragged
    last tiles in M and N, batches, the residual add, and the transposed convolutions' masked epilogue (`tm`)."""
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(B, M, K, generator=g)
    Wt = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    R = torch.randn(B, M, N, generator=g)
    dA, dW, db, dR = dev(A), dev(Wt), dev(bias), dev(R)
    planes = _split_planes(L, dW)

    def run():
        out = torch.full((B, M, N), -3.0, device="cuda")
        a = _lib.GemmArgs()
        a.A, a.W, a.bias, a.C = dA.data_ptr(), dW.data_ptr(), db.data_ptr(), out.data_ptr()
        a.M, a.N, a.K, a.lda, a.ldc = M, N, K, K, N
        a.act_in, a.batch, a.strideA, a.strideC = _lib.ACT_ELU, B, M * K, M * N
        if tm:
            a.tm_c, a.tm_lo, a.tm_hi = N // 2, 3, 2 * M - 5
        else:
            a.R, a.ldr, a.strideR = dR.data_ptr(), N, M * N
        a.W_split = planes.data_ptr()
        _lib.check(L.ssrhip_gemm(C.byref(a), _lib.stream_ptr()))
        sync()
        return out.cpu()

    with_remap = run()
    monkeypatch.setenv("SSRHIP_GEMM_XCD", "0")
    plain = run()
    monkeypatch.delenv("SSRHIP_GEMM_XCD")
    assert torch.equal(with_remap, plain)
    if tm:
        assert (with_remap.view(B, 2 * M, N // 2)[:, :3] == -3.0).all() and (with_remap.view(B, 2 * M, N // 2)[:, 2 * M - 5:] == -3.0).all()
    ref = torch.nn.functional.elu(A[1].double()) @ Wt.double().t() + bias.double() + (0 if tm else R[1].double())
    got = with_remap[1].double()
    if tm:
        keep = torch.zeros(2 * M, dtype=torch.bool)
        keep[3:2 * M - 5] = True
        torch.testing.assert_close(got.view(2 * M, N // 2)[keep], ref.view(2 * M, N // 2)[keep], rtol=3e-5, atol=3e-5)
    else:
        torch.testing.assert_close(got, ref, rtol=3e-5, atol=3e-5)


@pytest.mark.parametrize("Cc", [64, 128])
@pytest.mark.parametrize("T", [1, 127, 128, 129, 1000])
@pytest.mark.parametrize("out_act", [0, _lib.ACT_ELU])
def test_resblock_split_dma_kernel_matches_fp64(L, Cc, T, out_act):
    """csrc/resblock_split.hip (round 4: SEANetResnetBlock at C = 64 / 128 on the bf16 matrix cores, fp32 operands split exactly, weights by
    DMA, the intermediate handed from stage 1 to stage 2 in registers through a column permutation of W1) against an fp64 evaluation of
    y = x + b1 + W1 . ELU(b3 + W3 . ELU(x[t-1:t+2])) (modules/seanet.py:16-60): tiles that end inside a 128-step block, a single time step,
    three items with their own halos, with and without the ELU-on-store epilogue. Error no larger than the fp32 paths' (2e-5 on O(1) data);
    and the plane-less call (round-3 kernels) agrees to the same tolerance."""
    g = torch.Generator().manual_seed(1000 * Cc + T)
    Hh, B = Cc // 2, 3
    x = torch.randn(B, T + 2, Cc, generator=g)                         # rows 0 and T + 1: the halo (any values: a reflect / zero pad)
    w3 = torch.randn(Hh, 3, Cc, generator=g) / math.sqrt(3 * Cc)
    w1 = torch.randn(Cc, Hh, generator=g) / math.sqrt(Hh)
    b3, b1 = torch.randn(Hh, generator=g) * 0.1, torch.randn(Cc, generator=g) * 0.1
    xd = x.double()
    win = torch.cat([xd[:, 0:T], xd[:, 1:T + 1], xd[:, 2:T + 2]], dim=2)          # [B][T][3C]: taps t-1, t, t+1
    hmid = F.elu(F.elu(win) @ w3.reshape(Hh, 3 * Cc).double().t() + b3.double())
    want = xd[:, 1:T + 1] + hmid @ w1.double().t() + b1.double()
    if out_act:
        want = F.elu(want)
    dx, dw3, dw1, db3, db1 = dev(x), dev(w3.reshape(Hh, 3 * Cc).contiguous()), dev(w1), dev(b3), dev(b1)
    kp = torch.arange(Hh)
    perm = 16 * (kp // 16) + (kp % 8) % 4 + 8 * ((kp % 8) // 4) + 4 * ((kp // 8) % 2)
    assert sorted(perm.tolist()) == list(range(Hh))
    dw1p = dev(w1[:, perm].contiguous())
    p3 = torch.empty(3, Hh, 3 * Cc, dtype=torch.int16, device="cuda")
    p1 = torch.empty(3, Cc, Hh, dtype=torch.int16, device="cuda")
    _lib.check(L.ssrhip_split_weights(dw3.data_ptr(), p3.data_ptr(), dw3.numel(), _lib.stream_ptr()))
    _lib.check(L.ssrhip_split_weights(dw1p.data_ptr(), p1.data_ptr(), dw1p.numel(), _lib.stream_ptr()))
    outs = []
    for planes in (True, False):
        y = torch.full((B, T, Cc), float("nan"), device="cuda")
        a = _lib.ResblockArgs()
        a.x, a.y, a.w3, a.b3, a.w1, a.b1 = dx.data_ptr(), y.data_ptr(), dw3.data_ptr(), db3.data_ptr(), dw1.data_ptr(), db1.data_ptr()
        a.B, a.T, a.C, a.x_bstride, a.y_bstride, a.out_act = B, T, Cc, (T + 2) * Cc, T * Cc, out_act
        if planes:
            a.w3_split, a.w1_split = p3.data_ptr(), p1.data_ptr()
        _lib.check(L.ssrhip_resblock(C.byref(a), _lib.stream_ptr()))
        sync()
        outs.append(y.cpu())
        torch.testing.assert_close(y.cpu().double(), want, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(outs[0], outs[1], rtol=2e-5, atol=2e-5)
    a.w3_split, a.w1_split = p3.data_ptr(), 0
    assert L.ssrhip_resblock(C.byref(a), _lib.stream_ptr()) != 0                           # the two plane pointers come together


@pytest.mark.parametrize("B,Cc,T", [(70, 128, 6), (64, 256, 4), (33, 1024, 3)])
@pytest.mark.parametrize("skip,out_act", [(False, 0), (True, _lib.ACT_ELU)])
def test_lstm_split_step_matches_fp64(L, B, Cc, T, skip, out_act):
    """csrc/lstm_split.hip (LSTM recurrence on the bf16 matrix cores, both operands exactly split, streamed in MFMA fragment order; the codec's
    default from 128 items up) through `ssrhip_lstm_layer` with `w_split` / `hsplit`, in two time windows (the chunked two-stream pipeline's
    calling pattern), against an fp64 evaluation of torch.nn.LSTM's cell (gates i f g o; modules/lstm.py:10-25) — and against the fp32
    matrix-pipe kernel the same call takes without the planes. A batch that does not fill its last 64-row group, three widths
    (C / 64 = 2, 4, 16 k-steps per wave: both prefetch depths), skip + ELU-on-store epilogue. Error no larger than the fp32 path's."""
    from ssr_speech_amd.codec.wmencodec import pack_lstm_whh_planes
    g = torch.Generator().manual_seed(B + Cc + T + int(skip))
    whh = torch.randn(4 * Cc, Cc, generator=g) / math.sqrt(Cc)
    gin = torch.randn(B, T, 4 * Cc, generator=g)
    xs = torch.randn(B, T, Cc, generator=g)
    h, c = torch.zeros(B, Cc, dtype=torch.float64), torch.zeros(B, Cc, dtype=torch.float64)
    want = []
    for t in range(T):
        gt = gin[:, t].double() + h @ whh.double().t()
        i, f, gg, o = gt[:, :Cc], gt[:, Cc:2 * Cc], gt[:, 2 * Cc:3 * Cc], gt[:, 3 * Cc:]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        y = h + xs[:, t].double() if skip else h
        want.append(F.elu(y) if out_act else y)
    want = torch.stack(want, 1)
    dw, dgin, dxs = dev(whh), dev(gin), dev(xs)
    packed_fp32 = dev(whh.view(4, Cc // 4, 4, Cc // 16, 4, 4).permute(1, 3, 4, 2, 0, 5).contiguous())        # ssrhip_lstm_args.w_packed
    planes = torch.empty(3, 4 * Cc, Cc, dtype=torch.int16, device="cuda")
    _lib.check(L.ssrhip_split_weights(dw.data_ptr(), planes.data_ptr(), dw.numel(), _lib.stream_ptr()))
    wsplit = pack_lstm_whh_planes(planes)
    rows = (B + 15) // 16 * 16
    outs = []
    for split in (True, False):
        out = torch.full((B, T, Cc), float("nan"), device="cuda")
        hbuf, cbuf = torch.zeros(2, rows, Cc, device="cuda"), torch.zeros(B, Cc, device="cuda")
        hs = torch.full((2 * ((B + 63) // 64) * 64 * Cc * 3,), 0x7FC0, dtype=torch.int16, device="cuda")     # NaNs: the library has to zero it at t = 0
        for t0, t1 in ((0, T // 2), (T // 2, T)):
            a = _lib.LstmArgs()
            a.gin, a.w_hh, a.out = dgin.data_ptr(), packed_fp32.data_ptr(), out.data_ptr()
            a.w_packed = 1
            a.skip = dxs.data_ptr() if skip else 0
            a.hbuf, a.cbuf, a.gates = hbuf.data_ptr(), cbuf.data_ptr(), 0
            a.B, a.T, a.C = B, T, Cc
            a.gin_bstride, a.out_bstride, a.skip_bstride = T * 4 * Cc, T * Cc, T * Cc
            a.t_begin, a.t_end, a.out_act = t0, t1, out_act
            if split:
                a.w_split, a.hsplit = wsplit.data_ptr(), hs.data_ptr()
            _lib.check(L.ssrhip_lstm_layer(C.byref(a), _lib.stream_ptr()))
        sync()
        outs.append(out.cpu())
        torch.testing.assert_close(out.cpu().double(), want, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(outs[0], outs[1], rtol=2e-5, atol=2e-5)
    a.w_split, a.hsplit = wsplit.data_ptr(), 0
    assert L.ssrhip_lstm_layer(C.byref(a), _lib.stream_ptr()) != 0                                          # the two pointers come together


@pytest.mark.parametrize("wide", ["0", "1"])
def test_both_epilogue_forms_of_the_codec_kernels_pass_their_parity_tests(wide):
    """The split DMA GEMM and the residual-block kernel each have a dword and a 16-byte epilogue (same arithmetic; `SSRHIP_EPILOGUE_WIDE`,
    read once per process) and each DEFAULTS to a different one — so the suite above exercises one form per kernel. This runs the two
    kernels' parity tests again in a child process with the knob forced either way."""
    import subprocess, sys
    env = dict(os.environ, SSRHIP_EPILOGUE_WIDE=wide)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_kernels.py"), "-x", "-q", "-k",
                          "resblock_split_dma or gemm_split or gemm_batched_strided"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "failed" not in out.stdout, out.stdout[-2000:]
