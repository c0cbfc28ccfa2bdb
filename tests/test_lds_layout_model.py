"""The LDS layouts of the split GEMM kernels (csrc/gemm_split.hip) checked against the bank model of
/opt/skills/guides/MI355X_MICROARCH.md §LDS (64 banks of 4 B; `ds_read_b128` is serviced in four fixed 16-lane groups with bank =
(addr / 4) mod 64; `ds_write_b64` in four groups of 16 contiguous lanes with bank = (addr / 4) mod 32; a second distinct address on a
busy bank inside a group costs one more LDS cycle). Host-side arithmetic only: the same index formulas as the kernels, no GPU.

What it pins:
  * `gemm_split_dma_kernel`: XOR-swizzled 64-byte rows (chunk c of row r in slot c ^ ((r >> 2) & 3)) are conflict-free for the fragment
    reads and for the A stores, and the DMA's lane -> (row, chunk) choice fills exactly the slots the readers look in;
  * `gemm_split_kernel` (first generation, padded 80-byte rows): reads conflict-free, every store 2-way — 48 extra LDS cycles per wave
    and k-tile of the 8-wave arrangement, which is what rocprofv3 counted (profiles/r03_split_gemm_pmc.md: SQ_LDS_BANK_CONFLICT
    50.7 M = 48 x 1,048,576 wave-k-tiles on 4096^3)."""
import itertools

READ_B128_GROUPS = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
    [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63],
]


def _extra_cycles(addrs_by_lane, groups, width, nbanks):
    """Extra LDS cycles of one wave instruction: per lane group, (max number of DISTINCT addresses that need one bank) - 1."""
    extra = 0
    for g in groups:
        need = {}
        for lane in g:
            a = addrs_by_lane[lane]
            for d in range(width // 4):
                need.setdefault(((a // 4) + d) % nbanks, set()).add(a)
        extra += max(len(v) for v in need.values()) - 1
    return extra


def _read_extra(addr_of_lane):
    return _extra_cycles([addr_of_lane(l) for l in range(64)], READ_B128_GROUPS, 16, 64)


def _write_extra(addrs, width):
    lanes = 16 if width == 8 else 8                       # ds_write_b64: 4 x 16 contiguous lanes; ds_write_b128: 8 x 8
    groups = [list(range(i, i + lanes)) for i in range(0, 64, lanes)]
    return _extra_cycles(addrs, groups, width, 32)


def test_dma_kernel_swizzled_rows_are_conflict_free_for_reads_and_stores():
    # fragment reads (mma_tile): lane (li, lh) of a wave reads row base + li, chunk (kk / 8 + lh), 16 bytes, of a [rows][64 B] plane
    for base_row, kk in itertools.product(range(0, 128, 32), (0, 16)):
        def addr(l, base_row=base_row, kk=kk):
            li, lh = l & 31, l >> 5
            row = base_row + li
            return row * 64 + ((((kk >> 3) + lh) ^ ((li >> 2) & 3)) << 4)
        assert _read_extra(addr) == 0, (base_row, kk)
    # A stores (store_a): thread t -> row lr (+ 64 i), k-offset lc = (t & 7) * 4: 8 bytes at slot (lc >> 3) ^ ((lr >> 2) & 3), half (lc >> 2) & 1
    for wave, i in itertools.product(range(8), range(2)):
        addrs = []
        for lane in range(64):
            t = wave * 64 + lane
            lr, lc = t >> 3, (t & 7) * 4
            addrs.append((lr + 64 * i) * 64 + ((((lc >> 3) ^ ((lr >> 2) & 3)) << 4) + ((lc >> 2) & 1) * 8))
        assert _write_extra(addrs, 8) == 0, (wave, i)
        assert len(set(addrs)) == 64


def test_dma_lane_placement_matches_what_the_fragment_reads_expect():
    # dma_w: wave w, lane l lands in slot l of the wave's KiB (lane-linear), i.e. row 16 w + l / 4, slot l % 4, and fetches chunk
    # (l & 3) ^ ((l >> 4) & 3) of that row; the readers look for chunk c of row r in slot c ^ ((r >> 2) & 3)
    holds = {}
    for wave, lane in itertools.product(range(8), range(64)):
        row, slot = 16 * wave + (lane >> 2), lane & 3
        chunk = (lane & 3) ^ ((lane >> 4) & 3)
        assert (wave * 1024 + lane * 16) == row * 64 + slot * 16          # lane-linear KiB == [row][64 B] image
        holds[(row, slot)] = chunk
    assert len(holds) == 128 * 4
    for row, chunk in itertools.product(range(128), range(4)):
        assert holds[(row, chunk ^ ((row >> 2) & 3))] == chunk


def test_first_generation_padded_rows_read_clean_but_store_two_way():
    PITCH = 80                                                             # bytes: 32 bf16 + 16 B of padding
    for base_row, kk in itertools.product(range(0, 128, 32), (0, 16)):
        assert _read_extra(lambda l, b=base_row, k=kk: (b + (l & 31)) * PITCH + k * 2 + (l >> 5) * 16) == 0
    # the 8-wave arrangement measured in profiles/r03_split_gemm_pmc.md ("db" kernel): per wave and k-tile 6 ds_write_b64 (A: rows lr, lr + 64,
    # three planes) and 3 ds_write_b128 (W: row t >> 2, chunk t & 3, three planes)
    extra = 0
    for lane0 in (0,):
        a_addrs = [((lane0 * 64 + lane) >> 3) * PITCH + ((lane0 * 64 + lane) & 7) * 8 for lane in range(64)]
        w_addrs = [((lane0 * 64 + lane) >> 2) * PITCH + ((lane0 * 64 + lane) & 3) * 16 for lane in range(64)]
        extra = 6 * _write_extra(a_addrs, 8) + 3 * _write_extra(w_addrs, 16)
    assert extra == 48
    wave_k_tiles = (4096 // 128) ** 2 * (4096 // 32) * 8                   # 4096^3: workgroups x k-tiles x waves
    assert abs(extra * wave_k_tiles - 50_724_864) / 50_724_864 < 0.01     # SQ_LDS_BANK_CONFLICT of that run


def test_epilogue_transposition_through_lds_is_conflict_free():
    """The 16-byte epilogues (gemm_split_dma_kernel, and resblock_split_dma_kernel behind SSRHIP_EPILOGUE_WIDE=1) turn a 32 x 32
    accumulator block through a wave-private [32 rows][32 dwords] tile: lane (li, lh) stores register r at row (r & 3) + 8 (r >> 2) + 4 lh,
    column li (`ds_write_b32`: two groups of 32 lanes, bank = dword address mod 32), then lane l reads the float4 at row l / 8 + 8 p,
    columns 4 (l % 8) .. + 3 (`ds_read_b128` lane groups). Unpadded 128-byte rows are conflict-free for both — and every element written
    is read exactly once."""
    written = set()
    for r in range(16):
        addrs = [(((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)) * 4 for l in range(64)]
        groups = [list(range(0, 32)), list(range(32, 64))]
        assert _extra_cycles(addrs, groups, 4, 32) == 0, r
        written |= set(addrs)
    assert len(written) == 32 * 32
    read = set()
    for p in range(4):
        def addr(l, p=p):
            return (((l >> 3) + 8 * p) * 32 + (l & 7) * 4) * 4
        assert _read_extra(addr) == 0, p
        for l in range(64):
            read |= {addr(l) + 4 * d for d in range(4)}
    assert read == written


def test_split_lstm_partial_tiles_in_lds():
    """lstm_step_split_kernel (csrc/lstm_split.hip): the four waves' partial 64 x 64 tiles meet in LDS as part[wave][n][65 floats]. The
    accumulator stores (lane li = column n, register r = row m: `ds_write_b32`) are conflict-free with the 65-float row stride (64 would
    put all 32 lanes of a group on one bank); the finishing reads (lane = unit u of batch row bl: 16 lanes share a row) are 2-way."""
    PS = 65
    for mb, nb, r in itertools.product(range(2), range(2), range(16)):
        addrs = [((nb * 32 + (l & 31)) * PS + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 4 for l in range(64)]
        assert _extra_cycles(addrs, [list(range(0, 32)), list(range(32, 64))], 4, 32) == 0, (mb, nb, r)
        bad = [((nb * 32 + (l & 31)) * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 4 for l in range(64)]
        assert _extra_cycles(bad, [list(range(0, 32)), list(range(32, 64))], 4, 32) == 62        # the unpadded form: 32-way twice
    for i, g in itertools.product(range(4), range(4)):
        addrs = []
        for l in range(64):
            p = l + 256 * i                                        # wave 0's lanes; the other waves differ by a multiple of 4 batch rows
            u, bl = p & 15, p >> 4
            addrs.append((bl * PS + g * 16 + u) * 4)
        assert _extra_cycles(addrs, [list(range(0, 32)), list(range(32, 64))], 4, 32) <= 2, (i, g)
