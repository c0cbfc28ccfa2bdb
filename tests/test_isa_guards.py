"""CPU (hipcc cross-compiles gfx950 without a GPU): code-generation properties the decode GEMV's speed depends on, read off the ISA.
They were found by reading `hipcc -S` output while tuning and are easy to lose with an innocent edit:

  * every `gemv_seg_kernel` variant fits 128 VGPRs (16 waves per CU) with NO scratch (a spill puts memory traffic in the prologue);
  * its streaming loop re-requests each 16-byte piece in place: four non-temporal `global_load_dwordx4` per iteration, each behind an
    `s_waitcnt vmcnt(3)` — never a `vmcnt(0)` drain inside the loop (what a conditional re-request produced);
  * the row-per-wave kernels and the matrix-core kernels of the 830M shapes stay spill-free too.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ssr-speech_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def _asm(tmp_path_factory, name):
    out = tmp_path_factory.mktemp("isa") / (name + ".s")
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", f"-I{ROOT}/include", f"-I{CSRC}", "-ffp-contract=off", "-S", "--cuda-device-only",
           os.path.join(CSRC, name + ".hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    return open(out).read()


def _kernel_meta(asm):
    """symbol -> (vgpr_count, private_segment_fixed_size) from the .amdhsa metadata block."""
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", asm, re.S):
        body = m.group(2)
        v = re.search(r"\.vgpr_count:\s+(\d+)", body)
        p = re.search(r"\.private_segment_fixed_size:\s+(\d+)", asm[m.start() - 400:m.end()])
        if v and p:
            meta[m.group(1)] = (int(v.group(1)), int(p.group(1)))
    return meta


def _body(asm, symbol):
    start = asm.index("\n" + symbol + ":")
    return asm[start:asm.index("s_endpgm", start)]


@pytest.fixture(scope="module")
def gemv_asm(tmp_path_factory):
    return _asm(tmp_path_factory, "gemv")


def test_segment_kernel_fits_128_vgprs_without_scratch(gemv_asm):
    meta = {k: v for k, v in _kernel_meta(gemv_asm).items() if "gemv_seg_kernel" in k}
    assert len(meta) == 15, sorted(meta)                             # B in {1,2,4} x prologue in {none, LayerNorm, split-KV merge}, + two units at entry for B <= 2
    for sym, (vgpr, scratch) in meta.items():
        one_per_cu = "Li2ELb1E" in sym                                # split-KV merge + two units at entry: launched at one workgroup per CU
        assert vgpr <= (256 if one_per_cu else 128), (sym, vgpr)
        assert scratch == 0, (sym, scratch)


def test_segment_kernel_loop_rerequests_in_place(gemv_asm):
    for sym in (s for s in _kernel_meta(gemv_asm) if "gemv_seg_kernel" in s and "ILi2E" in s and "Lb0E" in s):   # the in-place variants (TWO = false)
        body = _body(gemv_asm, sym)
        loops = [m.start() for m in re.finditer(r"=>This Inner Loop Header", body)]
        assert loops, sym
        found = False
        for lo in loops:
            # the loop's text runs to its backward branch: the first s_cbranch after the header that targets a label defined before it
            seg = body[lo:lo + 12000]
            nt = [m.start() for m in re.finditer(r"global_load_dwordx4 [^\n]* nt", seg)]
            if len(nt) >= 4:
                inner = seg[:nt[3]]
                if inner.count("s_waitcnt vmcnt(3)") >= 4 and "s_waitcnt vmcnt(0)" not in inner:
                    found = True
        assert found, f"{sym}: no loop with four in-place non-temporal re-requests behind vmcnt(3) waits"


def test_round5_decode_kernels_keep_their_request_order_in_the_isa(gemv_asm, tmp_path_factory):
    """Round 5 read four inefficiencies of the 2-row decode step off the ISA; each is easy to bring back with an innocent edit:
      * a runtime profiling hook at kernel entry (exec-mask branch + global store) split the kernel-argument fetch into two dependent scalar
        round trips and pushed the kv_pos -> page-table chain of the QKV launch off the scalar path (vector loads behind `vmcnt(0)`): the
        straight-line `gemv_segu_kernel` variants must reach their first load behind ONE `s_waitcnt lgkmcnt`, never use v_readfirstlane,
        fit 128 VGPRs without scratch, and drain (`vmcnt(0)`) only once — behind their last unit;
      * the split-KV merge prologue of the out-projection drained its whole weight slice before the merge arithmetic (a loop pre-header's
        `vmcnt(0)` and a false dependency through a v_pk_fma register pair): between its first two barriers no `vmcnt(0)` outside the
        cold path's asm statements;
      * the decode attention requested its V rows behind the LAST K row: with VAT = 8 all 16 K requests precede the first wait, the 16 V
        requests follow in one run."""
    meta = _kernel_meta(gemv_asm)
    segu = [k for k in meta if "gemv_segu_kernel" in k]
    assert len(segu) == 12, sorted(segu)                              # 2 rows x {no prologue, LayerNorm} x {4, 6, 8} units per wave x depth {2, 4}
    for sym in segu:
        vgpr, scratch = meta[sym]
        assert vgpr <= 128 and scratch == 0, (sym, vgpr, scratch)
        body = _body(gemv_asm, sym)
        head = body[:body.index("global_load")]
        assert head.count("s_waitcnt lgkmcnt") == 1, (sym, head.count("s_waitcnt lgkmcnt"))
        assert "v_readfirstlane" not in body, sym
        assert body.count("vmcnt(0)") == 1, (sym, body.count("vmcnt(0)"))
    merge = [k for k in meta if "gemv_seg_kernelILi2ELi2ELb1E" in k]
    assert len(merge) == 1, merge
    body = _body(gemv_asm, merge[0])
    bars = [m.start() for m in re.finditer(r"s_barrier", body)]
    assert len(bars) >= 2
    hot = re.sub(r";;#ASMSTART.*?;;#ASMEND", "", body[bars[0]:bars[1]], flags=re.S)
    assert "vmcnt(0)" not in hot, "the out-projection's merge prologue drains its weight loads again"
    attn = _asm(tmp_path_factory, "attn")
    sym = next(k for k in _kernel_meta(attn) if "attn_decode_kernelILi128ELb0ELi8E" in k)
    seq = []
    for ln in _body(attn, sym).split("\n"):
        t = ln.strip()
        if t.startswith("global_load_dwordx4") and t.endswith("nt"):
            seq.append("L")
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            seq.append("w")
    joined = "".join(seq)
    assert joined.startswith("L" * 16 + "w"), joined[:40]            # all K rows in flight before the first wait
    assert "L" * 16 in joined[17:], joined                            # the V rows in one run
    # 16-row LayerNorm launches: the LayerNorm waits for the x slice ONLY — never for one of the 16 weight requests posted behind it (a
    # runtime switch around the weight requests made hipcc lose count: `vmcnt(7) .. vmcnt(0)` in front of the LayerNorm, +1 us per launch)
    mf = _asm(tmp_path_factory, "gemv_mfma")
    sym = next(k for k in _kernel_meta(mf) if "gemv_rows_xreg_kernelILi1ELi16ELi16ELb0ELb0E" in k)      # (the shipped order: x requests first; ..ELb1E is the opt-in experiment)
    body = _whole_body(mf, sym)
    head = body[:body.index("s_barrier")]
    waits = [int(m.group(1)) for m in re.finditer(r"s_waitcnt[^\n]*vmcnt\((\d+)\)", head)]
    assert waits and min(waits) >= 16, waits


def test_pair_kernels_keep_exact_wait_counts_and_the_scalar_path(gemv_asm):
    """The paired launches of the 2-row step (gemv_pair_kernel<4|6|8>, gemv_pair_merge_kernel<8>) run 12 waves per workgroup — three per
    SIMD, 168 VGPRs at most — as two branch arms, one per role. Read off the ISA while building them, each easy to lose again:
      * written as a series of `if (wave < 8)` blocks the streaming role's units were guarded by vmcnt(4 PF - 1) instead of vmcnt(15) (the
        wait-count pass merges the "block skipped" path into every join): with one arm per role every re-request sits behind
        `s_waitcnt vmcnt(15)` — (NUW - 4) x 4 of them per phase plus the first of the tail — and the kernel drains exactly twice;
      * with the edge role's arm laid out FIRST its stores made the streaming arm's uniform loads (row_len of the merge; the kv_pos chain)
        "possibly clobbered": vector loads behind vmcnt(0) in front of the first weight request. The streaming arm comes first (its first
        branch is taken by the edge waves), row_len is an s_load, the KV-append chain lives in the edge arm;
      * inside `if (t < B * H)` the merge's (m, l) loads were sunk behind the weight requests and their first use drained the queue: the six
        global_load_dwordx2 precede the first non-temporal load."""
    meta = _kernel_meta(gemv_asm)
    pairs = sorted(k for k in meta if "gemv_pair" in k)
    assert len(pairs) == 5, pairs            # gemv_pair_kernel<4|6|8>, gemv_pair_merge_kernel<8, 0|2> (round 6: B's first two units at entry)
    for sym in pairs:
        vgpr, scratch = meta[sym]
        assert vgpr <= 168 and scratch == 0, (sym, vgpr, scratch)
        body = _whole_body(gemv_asm, sym)
        merge = "gemv_pair_merge" in sym
        nuwb = int(re.search(r"kernelILi(\d)E", sym).group(1))
        # the streaming arm: from the first non-temporal load to the end of the function or the edge arm's sweep, whichever the layout puts last
        first_nt = body.index(" nt\n")
        stream = body[:body.index("sc1")] if body.index("sc1") > first_nt else body[first_nt:]
        # per phase: (units - 4) x 4 re-requests + the first wait of the tail; the merge walks its prefetched partials 19, 19, 18, 18 .. down
        early = merge and "ILi8ELi2E" in sym     # the merge form with B's units 0 and 1 requested at entry: 8 more loads in flight under the
        want15 = ((1 if early else 2) if merge else (8 - 4) * 4 + 1) + (nuwb - 4) * 4 + 1      # merge, whose waits walk 33, 32, .. instead of 25, 24, ..
        assert stream.count("s_waitcnt vmcnt(15)") == want15, (sym, stream.count("s_waitcnt vmcnt(15)"), want15)
        if merge:
            assert body.index("sc1") > first_nt, "the edge role's arm precedes the streaming arm again"
            head = body[:first_nt]
            assert len(re.findall(r"global_load_dwordx2 ", head)) >= 6, "the (m, l) loads are behind the weight requests again"
            assert not re.search(r"s_waitcnt[^\n]*vmcnt", head), "a vector-memory wait in front of the first weight request (row_len off the scalar path?)"
        else:
            hot = re.sub(r";;#ASMSTART.*?;;#ASMEND", "", stream, flags=re.S)
            assert hot.count("vmcnt(0)") <= 2 + (1 if nuwb == 4 else 0), (sym, hot.count("vmcnt(0)"))


def test_residual_block_dma_wait_counts_the_loads_the_compiler_really_issued(tmp_path_factory):
    """`resblock_split_dma_kernel` completes its weight-tile DMA (inline asm, invisible to hipcc's wait-count pass) through a
    hand-counted `s_waitcnt vmcnt(CNT * tiles_after(u) + NEL)`; at tap 1 the count assumes that exactly NEL = 5 compiler-issued loads of
    the next ELU(x) tile are YOUNGER than the tile being waited for. If hipcc merged, sank or dropped one of them the wait would be one
    too loose and a stale weight tile would be consumed silently. The shipped ring depth is 2: the tap-1 wait is `vmcnt(5)`; between the
    workgroup barrier in front of the DMA requests and that wait there must be the DMA instructions followed by exactly five
    `global_load_dwordx4` and nothing else that counts."""
    asm = _asm(tmp_path_factory, "resblock_split")
    syms = [k for k in _kernel_meta(asm) if "resblock_split_dma_kernel" in k and "ELi2ELi0E" in k]
    assert len(syms) == 2, syms                                      # 64 and 128 channels, ring of 2
    for sym in syms:
        body = _whole_body(asm, sym)
        waits = [m.start() for m in re.finditer(r"s_waitcnt vmcnt\(5\)", body)]
        assert waits, sym
        seen = 0
        for w in waits:
            seg = body[body.rindex("s_barrier", 0, w):w]
            ops = re.findall(r"(buffer_load_dwordx4[^\n]*lds|global_load_dword\S*|global_store\S*|buffer_store\S*)", seg)
            dma = [o for o in ops if o.startswith("buffer_load")]
            if not dma:
                continue                                              # a vmcnt(5) of an epilogue's descending sequence
            seen += 1
            rest = [o for o in ops if not o.startswith("buffer_load")]
            assert ops[:len(dma)] == dma, (sym, ops)                  # the DMA requests first ...
            assert rest == ["global_load_dwordx4"] * 5, (sym, rest)    # ... then exactly NEL loads of the next ELU(x) tile
        assert seen >= 1, sym


def test_generic_and_matrix_core_gemv_kernels_do_not_spill(gemv_asm, tmp_path_factory):
    assert not [k for k in _kernel_meta(gemv_asm) if "gemv_fast_kernel" in k]          # the intermediate generation is gone (round 3)
    generic = {k: v for k, v in _kernel_meta(gemv_asm).items() if "gemv_kernel" in k and "ILi2E" in k}
    assert generic
    for sym, (vgpr, scratch) in generic.items():          # the fallback for odd shapes: a few bytes of scratch in one variant are tolerated, real spills are not
        assert scratch <= 16 and vgpr <= 256, (sym, vgpr, scratch)
    mfma = _kernel_meta(_asm(tmp_path_factory, "gemv_mfma"))
    used = [k for k in mfma if ("gemv_rows_xreg_kernel" in k and "Li16ELi16E" in k) or "gemv_rows_stream_kernel" in k]
    assert len(used) >= 4, sorted(mfma)
    for sym in used:
        assert mfma[sym][1] == 0 and mfma[sym][0] <= 256, (sym, mfma[sym])


def test_split_gemm_kernels_use_the_bf16_matrix_core_and_do_not_spill(tmp_path_factory):
    """Round 3's split kernels (csrc/gemm_split.hip, resblock_chain_split_kernel): two workgroups per CU need <= 256 VGPRs and no
    scratch; the k-loop must really issue v_mfma_f32_32x32x16_bf16 (six per 16 k-values and accumulator) and v_cvt_pk_bf16_f32 for the
    on-the-fly split — and no fp32 MFMA at all."""
    asm = _asm(tmp_path_factory, "gemm_split")
    meta = {k: v for k, v in _kernel_meta(asm).items() if "gemm_split_kernel" in k}
    assert len(meta) == 4, sorted(meta)                              # 128- / 64-row tiles x ELU on load or not
    for sym, (vgpr, scratch) in meta.items():
        # (the 4-row variants keep one 32-byte stack object — the per-row length arrays — and spill nothing)
        assert vgpr <= 256 and scratch <= (32 if "ILi4E" in sym else 0), (sym, vgpr, scratch)
        body = _body(asm, sym)
        n_mfma = body.count("v_mfma_f32_32x32x16_bf16")
        assert n_mfma >= 24 and n_mfma % 6 == 0, (sym, n_mfma)
        assert "v_cvt_pk_bf16_f32" in body and "v_mfma_f32_32x32x2_f32" not in body, sym
    # the 8-wave DMA kernel: four waves per SIMD need <= 128 VGPRs; W reaches LDS through buffer_load ... lds (three planes per tile, in the
    # prologue and in the loop), A through branch-free buffer loads (no exec-mask branch around a load anywhere in the k-loop)
    dma = {k: v for k, v in _kernel_meta(asm).items() if "gemm_split_dma_kernel" in k}
    assert len(dma) == 6, sorted(dma)                                # 128- / 64-row tiles x ELU on load or not, + the two opt-in time-mask instantiations
    for sym, (vgpr, scratch) in dma.items():
        assert vgpr <= 128 and scratch == 0, (sym, vgpr, scratch)
        body = _body(asm, sym)
        n_mfma = body.count("v_mfma_f32_32x32x16_bf16")
        assert n_mfma >= 12 and n_mfma % 6 == 0, (sym, n_mfma)           # 2 k-steps x 6 products x 1 or 2 accumulators
        n_dma = len([ln for ln in body.splitlines() if "buffer_load_dwordx4" in ln and ln.rstrip().endswith("lds")])
        assert n_dma >= 6, (sym, n_dma)
        assert "v_cvt_pk_bf16_f32" in body and "v_mfma_f32_32x32x2_f32" not in body and "s_setprio 1" in body, sym
    rb = _asm(tmp_path_factory, "resblock")
    chain = {k: v for k, v in _kernel_meta(rb).items() if "resblock_chain_split_kernel" in k}
    assert len(chain) == 1
    for sym, (vgpr, scratch) in chain.items():
        assert vgpr <= 256 and scratch == 0, (sym, vgpr, scratch)
        body = _body(rb, sym)
        assert body.count("v_mfma_f32_32x32x16_bf16") >= 36 and "v_mfma_f32_32x32x2_f32" not in body, sym


def test_segment_kernel_prologue_has_no_integer_division(gemv_asm):
    """Round 4: the row partition floor(N * blockIdx / G) was a 64-bit division — ~330 emulation instructions at the head of every wave,
    in front of the first weight request. The host passes N / G and N % G now; the first non-temporal load must come early."""
    for sym in (s for s in _kernel_meta(gemv_asm) if "gemv_seg_kernel" in s and "ILi2E" in s and "Li2ELb" not in s):   # no-prologue and LayerNorm variants
        lines = [l for l in _body(gemv_asm, sym).split("\n") if l.strip() and not l.strip().startswith(";") and not l.strip().endswith(":")]
        first = next(i for i, l in enumerate(lines) if "global_load_dwordx4" in l and " nt" in l)
        assert first < 120, (sym, first)
        assert not any("v_rcp_iflag_f32" in l for l in lines[:first]), sym          # the signature of an emulated integer division


def test_split_gemm_drains_its_dma_before_the_tile_barrier(tmp_path_factory):
    """gemm_split_dma_kernel: W tiles arrive in LDS by DMA issued by OTHER waves; the k-loop must wait for its own DMA (vmcnt(0)) before the
    barrier that publishes the tile (until round 4 only the compiler's conservative wait placement guaranteed it)."""
    asm = _asm(tmp_path_factory, "gemm_split")
    syms = [k for k in _kernel_meta(asm) if "gemm_split_dma_kernel" in k]
    assert len(syms) == 6, syms
    for sym in syms:
        body = _body(asm, sym)
        lo = body.index("=>This Inner Loop Header")
        loop = body[lo:]
        first_mfma = loop.index("v_mfma_f32_32x32x16")
        head = loop[:first_mfma]                                   # loop head .. first MFMA: store_a, the drain, the second barrier, the next loads
        bars = [m.start() for m in re.finditer(r"s_barrier", head)]
        assert len(bars) >= 2, sym
        assert "s_waitcnt vmcnt(0)" in head[bars[0]:bars[1]], sym


def _serialized_loads(body):
    """loads that are waited for (`vmcnt(0)`) within four instructions of their issue = dependent memory round trips (tools/isa_scan.py)"""
    n, last = 0, -99
    lines = [ln.strip() for ln in body.split("\n")]
    lines = [t for t in lines if t and not t.startswith(";") and not t.startswith(".")]
    for i, t in enumerate(lines):
        if (t.startswith("global_load") or t.startswith("buffer_load")) and " lds" not in t:
            last = i
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t and i - last <= 4:
            n, last = n + 1, -99
    return n


def _whole_body(asm, symbol):
    """the kernel's whole text (a kernel with several exits has several s_endpgm: `_body` stops at the first)"""
    start = asm.index("\n" + symbol + ":")
    return asm[start:asm.index(".Lfunc_end", start)]


def _longest_run(body, prefix):
    """most `prefix` instructions issued with no `s_waitcnt vmcnt` and no branch between them"""
    best = run = 0
    for ln in body.split("\n"):
        t = ln.strip()
        if t.startswith(prefix + " "):
            run += 1
            best = max(best, run)
        elif t.startswith("s_waitcnt") and "vmcnt" in t or t.startswith("s_cbranch"):
            run = 0
    return best


def test_codec_epilogues_request_the_added_operand_a_block_at_a_time(tmp_path_factory):
    """The residual-block kernel and the split DMA GEMM add an operand behind the accumulators (x, or R of a residual block's second
    convolution). Written as `load -> add -> store` per output under a row predicate this compiles to ONE dependent HBM round trip per
    output — 64 per lane, 30 of a workgroup's 49 us in the residual block, found late in round 4 with tools/resblock_lab.hip. The whole-tile
    paths must issue a block's loads together (16 dword loads, or 4 dwordx4 in the 16-byte form); the only serialized loads left are
    those of the general / ragged-tile loops (one per output there: the count is exact, so a regression of the fast paths shows)."""
    asm = _asm(tmp_path_factory, "resblock_split")
    syms = [k for k in _kernel_meta(asm) if "resblock_split_dma_kernel" in k]
    assert len(syms) == 4, syms
    for sym in syms:
        body = _whole_body(asm, sym)
        per_lane = 64 if "ILi128E" in sym else 32
        assert _longest_run(body, "global_load_dword") >= 16, sym                 # dword form: a block's 16 residual values at once
        assert _longest_run(body, "global_load_dwordx4") >= 4, sym                # 16-byte form (and the x tile loads)
        assert _serialized_loads(body) == per_lane, (sym, _serialized_loads(body))   # the ragged last tile's plain loop, nothing else
    asm = _asm(tmp_path_factory, "gemm_split")
    syms = [k for k in _kernel_meta(asm) if "gemm_split_dma_kernel" in k]
    assert len(syms) == 6, syms
    for sym in syms:
        body = _whole_body(asm, sym)
        mt = 2 if "ILi128E" in sym else 1
        assert _longest_run(body, "global_load_dword") >= 16, sym
        assert body.count("global_store_dwordx4 ") >= 4 * mt, sym                  # the 16-byte form's stores
        # general loop: up to 4 loads per output (C, R, class id -> class bias), 16 outputs per block; + a handful in the prologue
        assert _serialized_loads(body) <= 4 * 16 * mt + 4, (sym, _serialized_loads(body))


def test_split_lstm_step_keeps_its_prefetch_distance(tmp_path_factory):
    """lstm_step_split_kernel (csrc/lstm_split.hip, opt-in): one wave per SIMD streams both MFMA operands from L2 with nobody to hide a
    round trip behind, so the refills must stay DEPTH - 1 MFMA blocks ahead: no scratch, the bf16 matrix instruction, and no `vmcnt(0)`
    inside the k-loop (what a branch around the refill produced in the first version: every iteration drained the whole ring)."""
    asm = _asm(tmp_path_factory, "lstm_split")
    meta = {k: v for k, v in _kernel_meta(asm).items() if "lstm_step_split_kernel" in k}
    assert len(meta) == 2, sorted(meta)
    for sym, (vgpr, scratch) in meta.items():
        assert scratch == 0 and vgpr <= 512, (sym, vgpr, scratch)
        body = _whole_body(asm, sym)
        lo = body.index("=>This Inner Loop Header")
        loop = body[lo:body.index("s_barrier", lo)]
        last_back = loop.rindex("s_cbranch")                        # the loop's back edge
        loop = loop[:last_back]
        assert loop.count("v_mfma_f32_32x32x16_bf16") >= 48, sym
        assert "vmcnt(0)" not in loop, sym
