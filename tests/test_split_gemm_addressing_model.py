"""Host-side replay of `gemm_split_dma_kernel`'s addressing (csrc/gemm_split.hip): the kernel reads A and the three W planes through
buffer descriptors whose extent ends with the tile's last valid row, and sends a lane out of range (offset 0x80000000) when its k is
past K. No GPU: the same integer formulas in NumPy, checked for the properties the kernel relies on —

  * a lane that should read element (m, k) reads exactly A[m * lda + k .. + 4) (or W[n * K + k .. + 8) of its plane), and that access lies
    inside the descriptor;
  * a lane whose k chunk starts at or past K is out of range; with lda >= K every row past M (past N for W) is out of range as well, i.e.
    reads as zero (for overlapping views, lda < K, such rows may read a neighbour's data — their outputs are never stored);
  * together the lanes of a workgroup cover the BM x 32 (128 x 32) tile exactly once;
  * every in-range offset stays below the out-of-range marker when the launcher's 2 GiB guard holds.
The GPU side of the same contract is `tests/test_gpu_kernels.py` (tails in M, N and K against fp64)."""
import numpy as np
import pytest

OOB = 0x80000000


def _a_lanes(BM, M, K, lda, m0, k0):
    """(offset or OOB, row, k) of every A load of one k-tile: 512 threads, 8 per row (4 floats each), 64 rows per pass, BM / 64 passes."""
    t = np.arange(512)
    lr, lc = t >> 3, (t & 7) * 4
    rows_a = min(BM, M - m0)
    extent = ((rows_a - 1) * lda + K) * 4                                # num_records of the descriptor based at A + m0 * lda
    out = []
    for i in range(BM // 64):
        kin = (k0 + lc) < K
        off = np.where(kin, ((lr + 64 * i) * lda + lc) * 4 + k0 * 4, OOB)
        out.append((off, lr + 64 * i, k0 + lc))
    return out, extent, rows_a


def _w_lanes(N, K, n0, k0):
    """W by DMA: wave w, lane l -> row 16 w + l / 4, chunk (l & 3) ^ ((l >> 4) & 3) (8 bf16 = 16 bytes)."""
    t = np.arange(512)
    wave, lane = t >> 6, t & 63
    wrow = 16 * wave + (lane >> 2)
    wchunk = (lane & 3) ^ ((lane >> 4) & 3)
    rows_w = min(128, N - n0)
    extent = rows_w * K * 2
    kin = (k0 + wchunk * 8) < K
    off = np.where(kin, (wrow * K + wchunk * 8) * 2 + k0 * 2, OOB)
    return off, wrow, k0 + wchunk * 8, extent, rows_w


@pytest.mark.parametrize("BM", [128, 64])
@pytest.mark.parametrize("M,N,K,lda", [(3000, 2056, 1000, 1000), (50000, 128, 264, 264), (300, 200, 64, 96), (1500, 512, 384, 128), (77, 1030, 8, 8)])
def test_a_operand_addressing(BM, M, N, K, lda):
    for m0 in {0, (M - 1) // BM * BM}:                                   # first and last row block
        for k0 in {0, (K - 1) // 32 * 32}:                               # first and last k-tile
            lanes, extent, rows_a = _a_lanes(BM, M, K, lda, m0, k0)
            seen = set()
            for off, row, k in lanes:
                for o, r, kk in zip(off.tolist(), row.tolist(), k.tolist()):
                    if kk >= K:
                        assert o == OOB
                        continue
                    assert o < OOB
                    if r < rows_a:                                       # a row of the problem: the right address, inside the extent
                        assert o == (r * lda + kk) * 4 and o + 16 <= extent
                        assert (r, kk) not in seen
                        seen.add((r, kk))
                    elif lda >= K:                                       # a row past M: reads as zero
                        assert o >= extent
            want = {(r, kk) for r in range(rows_a) for kk in range(k0, min(k0 + 32, K), 4)}
            assert seen == want


@pytest.mark.parametrize("N,K", [(2056, 1000), (128, 264), (200, 64), (1030, 8), (4096, 1024)])
def test_w_planes_addressing(N, K):
    for n0 in {0, (N - 1) // 128 * 128}:
        for k0 in {0, (K - 1) // 32 * 32}:
            off, row, k, extent, rows_w = _w_lanes(N, K, n0, k0)
            seen = set()
            for o, r, kk in zip(off.tolist(), row.tolist(), k.tolist()):
                if kk >= K:
                    assert o == OOB
                elif r < rows_w:
                    assert o == (r * K + kk) * 2 and o + 16 <= extent and (r, kk) not in seen
                    seen.add((r, kk))
                else:
                    assert extent <= o < OOB                             # a row past N: the DMA writes zeros
            assert seen == {(r, kk) for r in range(rows_w) for kk in range(k0, min(k0 + 32, K), 8)}


def test_the_launchers_guard_keeps_every_offset_below_the_marker():
    # ssrhip_gemm_split_launch takes the DMA kernels only if (127 * lda + K) * 4 and 128 * K * 2 stay below 0x7FFFFFF0
    for lda, K in [(4_000_000, 4_000_000), (1_000_000, 3_000_000), (4096, 4096)]:
        ok = (127 * lda + K) * 4 < 0x7FFFFFF0 and 128 * K * 2 < 0x7FFFFFF0
        biggest_a = (127 * lda + (K - 4)) * 4 + 16                       # last row, last chunk, last byte + 1
        biggest_w = (127 * K + (K - 8)) * 2 + 16
        if ok:
            assert biggest_a <= 0x7FFFFFF0 + 16 and biggest_a < OOB and biggest_w < OOB
    assert not ((127 * 5_000_000 + 4096) * 4 < 0x7FFFFFF0)              # a row pitch that does not fit is refused (4-wave kernel instead)


def _xcd_tile(nx, ny, nz, gn=0):
    """csrc/gemm_split.hip xcd_tile in NumPy: launch index w (x fastest) -> (bx, by, bz) of the logical tile it takes; gn = N-tile group width."""
    total = nx * ny * nz
    w = np.arange(total, dtype=np.int64)
    c, j, q, r = w & 7, w >> 3, total >> 3, total & 7
    L = c * q + np.minimum(c, r) + j
    if gn == 0 or gn > nx:
        gn = nx
    per_group = gn * ny * nz
    g = L // per_group
    idx = L - g * per_group
    width = np.where(g * gn + gn <= nx, gn, nx - g * gn)
    bx = g * gn + idx % width
    t2 = idx // width
    return bx, t2 % ny, t2 // ny, c, L


@pytest.mark.parametrize("nx,ny,nz,gn", [(4, 469, 256, 0), (10, 94, 256, 3), (10, 94, 3, 4), (1, 7, 1, 0), (3, 1, 1, 2), (2, 5, 3, 1), (17, 13, 11, 5), (8, 8, 8, 8),
                                         (5, 1, 9, 2), (64, 47, 2, 13)])
def test_xcd_tile_order_is_a_bijection_that_keeps_a_row_tile_on_one_xcd(nx, ny, nz, gn):
    """Round 6: workgroup w runs on XCD w % 8; the remap must (a) hit every tile of the grid exactly once for any grid
    and any group width, including totals that are not multiples of 8 and a narrower last group, (b) give every XCD a contiguous range of
    the logical order, in which (c) the N-tiles of one (row-tile, item) that belong to the same group are neighbours, and (d) a group's
    N-tiles are swept over all row-tiles and items before the next group starts (the W tiles of a group stay in that XCD's L2)."""
    bx, by, bz, xcd, L = _xcd_tile(nx, ny, nz, gn)
    total = nx * ny * nz
    assert bx.min() >= 0 and bx.max() < nx and by.min() >= 0 and by.max() < ny and bz.min() >= 0 and bz.max() < nz
    lin = bx + nx * (by + ny * bz)
    assert np.array_equal(np.sort(lin), np.arange(total))                 # a bijection
    for c in range(8):
        Lc = L[xcd == c]
        if len(Lc):
            assert np.array_equal(Lc, np.arange(Lc[0], Lc[0] + len(Lc)))  # contiguous, in launch order
    g_eff = nx if (gn == 0 or gn > nx) else gn
    order = np.argsort(L)
    bxo, byo, bzo = bx[order], by[order], bz[order]
    grp = bxo // g_eff
    assert np.all(np.diff(grp) >= 0)                                      # (d) groups in order, each finished before the next
    same_row = (byo[1:] == byo[:-1]) & (bzo[1:] == bzo[:-1]) & (grp[1:] == grp[:-1])
    assert np.all(bxo[1:][same_row] == bxo[:-1][same_row] + 1)            # (c) x fastest inside a group
    n_runs = len(set(zip(grp.tolist(), byo.tolist(), bzo.tolist())))            # one run of neighbours per (group, row-tile, item)
    assert same_row.sum() == total - n_runs
