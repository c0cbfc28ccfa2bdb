// gemv_mfma.hip — the fused weight-streaming GEMV for 5..16 rows (several utterances x CFG rows decoded in lock-step on one
// GPU: SURVEY §8d config 4 / §8e "B_local utterances x 2 CFG rows are batched into one GEMV pass so weights are read once").
//
//   y[b][n] = epi( sum_k pro(x)[b][k] * W[n][k] + bias[n] ),   b < B <= 16
//
// Still HBM-bound (2*B FLOP per 4 weight bytes = 8 FLOP/B at B=16), but 16 FMAs per weight element no longer fit the VALU
// budget of a streaming wave, nor does x fit its registers — so the tile goes to the matrix core:
//   * v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain): A = 16 weight rows x 4 k, B = 4 k x 16 batch columns, D = 16x16 fp32.
//     Lane l supplies A[row l%16][k-slot l/16] and B[k-slot l/16][column l%16]; one global_load_dwordx4 of W per lane feeds
//     4 MFMAs (the float4's elements are 4 consecutive k; A and B use the same k permutation, so no shuffles are needed);
//   * a workgroup owns 16 weight rows; its NW <= 8 waves split K in 512-float slices (32 MFMA k-steps each), so a wave keeps
//     its whole x slice — 16 columns x 512 floats = 32 float4 per lane — in VGPRs (loaded once from L2, reused for nothing
//     else: there is no LDS on the operand path), and streams W with 16 x 1 KiB loads in flight (rolling: a k-step's registers are refilled right after its 4 MFMAs);
//   * the first 16 W loads are in flight DURING the LayerNorm prologue, which is the reference's two-pass LayerNorm computed on
//     the register-resident x (column sums across the 4 k-slot lanes by permlane swaps, across waves through 128 B of LDS);
//   * the NW partial 16x16 tiles are added in wave order through LDS (deterministic) and wave 0 runs the epilogue: each lane
//     holds 4 consecutive output rows of one batch column => float4 bias / residual / store / KV-append.
// K > NW*512 (FFN2, K = 8192) loops over chunks. Rows beyond N and columns beyond B are handled by clamping the load
// addresses (never by predication: see gemv.hip) and masking the stores.
#include <stdlib.h>
#include "common.h"

namespace {

typedef float f4v __attribute__((ext_vector_type(4)));

// Phase time stamps of the rows-per-workgroup kernels exist only in a -DSSR_GEMVM_PROFILE build (tools/gemvm_lab.hip includes this file
// with it): wave 0 of every workgroup leaves wall_clock64 (100 MHz) at: 0 entry, 6 x requests issued, 7 first weight requests issued, 1 x
// has arrived (and the epilogue operands are requested), 2 LayerNorm done, 3 last MFMA issued, 4 behind the partial-tile barrier, 5
// epilogue stores issued. Nothing of this is in the library build.
#ifdef SSR_GEMVM_PROFILE
__device__ long long* g_gemvm_prof = nullptr;
#define MSTAMP(i) do { if (g_gemvm_prof && threadIdx.x == 0) g_gemvm_prof[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define MSTAMP(i) do { } while (0)
#endif

struct GemvM {
  ssrhip_gemv_args a;
  int nw;       // waves per workgroup
  int steps;    // K / 16 MFMA k-steps in total
  int nchunk;   // ceil(steps / (nw * 32))
  int hd;
  int rows;     // weight rows per workgroup: 16, or 8 (tile rows 8..15 duplicate 0..7) when N/16 workgroups would leave CUs idle
};

constexpr int SPW = 32;   // k-steps per wave per chunk (512 floats of K)

constexpr int DEPTH = 16;  // weight loads in flight per wave (16 KiB)

// sum over the 4 lanes that share lane%16 (the 4 k-slots of one batch column)
__device__ __forceinline__ float kslot_sum(float v) {
  v += xor16_f(v);
  v += xor32_f(v);
  return v;
}

// Epilogue of one 16x16 output tile: this lane holds rows r0 = row0 + 4*(lane/16) .. r0+3 of batch column c = lane%16
// (only the first `tile_rows` rows of the tile are real: 16, or 8 when the tile's rows 8..15 duplicate 0..7).
// Split in two so that everything the epilogue has to FETCH — bias, the residual, and for the QKV launch the cache address
// (kv_pos -> page table -> pool: two dependent loads) — is requested before the weight loop and has long arrived when the
// last MFMA retires; otherwise that latency chain (1-2 us) sits in the tail of every launch with the HBM idle.
struct TileEpi {
  float* dst;
  float bias[4], res[4];
  int nvalid;        // 0: this lane stores nothing
  int kv_which, kv_cc, kv_pos, kv_page;   // QKV launch, K / V rows (kv_which = 1 | 2): dst is resolved in tile_epilogue_finish
};

// kv_pos of this lane's batch column for the QKV launch (0 otherwise): request it BEFORE the x / W loads (see tile_epilogue_fetch)
__device__ __forceinline__ int tile_kvpos(const ssrhip_gemv_args& a, int lane) {
  return (a.epi == SSRHIP_EPI_QKV_APPEND) ? a.kv_pos[min(lane & 15, a.B - 1)] : 0;
}

__device__ __forceinline__ TileEpi tile_epilogue_fetch(const ssrhip_gemv_args& a, int hd, int grp, int row0, int tile_rows, int lane, int kvpos) {
  TileEpi e;
  const int c = lane & 15, ks = lane >> 4;
  const int N = a.N, K = a.K, B = a.B;
  const int r0 = row0 + ks * 4;
  e.dst = nullptr;
  e.kv_which = 0; e.kv_cc = 0; e.kv_pos = kvpos; e.kv_page = 0;
  // QKV launch: the address of a K / V row needs kv_pos[c] -> page table -> pool, two DEPENDENT loads. `kvpos` was requested by the caller
  // as the wave's OLDEST load (in front of x and W); the table entry is requested here by EVERY lane (branch-free: q rows and idle lanes
  // read a valid entry they never use) and first used in tile_epilogue_finish — so neither wait drains anything. Rounds 2-4 requested
  // both here, back to back, under the lane's row predicate: each was followed by `s_waitcnt vmcnt(0)`, i.e. every wave of the LN + QKV
  // launch drained its x slice and its first 16 weight loads — twice — before the LayerNorm could start; and a load under a divergent
  // branch makes hipcc wait for it (`vmcnt(0)`) in the OTHER branch before it may reuse the destination register (read off the ISA,
  // round 5; the 2-row kernel had the same disease, csrc/gemv.hip).
  if (a.epi == SSRHIP_EPI_QKV_APPEND) e.kv_page = a.kv.table[(size_t)min(c, B - 1) * a.kv.max_pages + (kvpos / SSRHIP_PAGE)];
  const bool live = c < B && r0 < N && ks * 4 < tile_rows;
  e.nvalid = live ? min(4, N - r0) : 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { e.bias[j] = 0.f; e.res[j] = 0.f; }
  if (!live) return e;
  if (a.epi == SSRHIP_EPI_QKV_APPEND) {
    const int D = K, which = r0 / D, cc = r0 % D;
    e.kv_which = which;                                                // 0: a q row (plain store below); 1 | 2: resolved in tile_epilogue_finish
    e.kv_cc = cc;
    e.dst = a.y + (size_t)c * a.y_stride + cc;
  } else if (a.y_tiled) {
    e.dst = a.y + (size_t)grp * N * 16 + SSRHIP_TILED(c, r0);
  } else {
    e.dst = a.y + (size_t)c * a.y_stride + (size_t)grp * N + r0;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j < e.nvalid) {
      if (a.bias) e.bias[j] = a.bias[(size_t)grp * N + r0 + j];
      if (a.epi == SSRHIP_EPI_RESIDUAL) e.res[j] = e.dst[j];
    }
  }
  return e;
}

__device__ __forceinline__ void tile_epilogue_finish(const ssrhip_gemv_args& a, const TileEpi& e0, f4v acc, int hd) {
  if (e0.nvalid == 0) return;
  TileEpi e = e0;
  if (e.kv_which) {
    const size_t off = ((((size_t)e.kv_page * a.kv.n_layer + a.layer) * 2 + (e.kv_which - 1)) * a.kv.n_head + e.kv_cc / hd) * SSRHIP_PAGE + (e.kv_pos % SSRHIP_PAGE);
    e.dst = a.kv.pool + off * a.kv.head_dim + (e.kv_cc % hd);
  }
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] += e.bias[j];
    if (a.act == SSRHIP_ACT_RELU) v[j] = fmaxf(v[j], 0.f);
    else if (a.act == SSRHIP_ACT_GELU_ERF) v[j] = 0.5f * v[j] * (1.0f + erff(v[j] * 0.70710678118654752440f));
    v[j] = e.res[j] + v[j];                         // res == 0 unless EPI_RESIDUAL (same operand order as the fused add: y + v)
  }
  if (e.nvalid == 4 && ((reinterpret_cast<size_t>(e.dst) & 15) == 0)) {
    *reinterpret_cast<float4*>(e.dst) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < e.nvalid) e.dst[j] = v[j];
  }
}

__device__ __forceinline__ void tile_epilogue(const ssrhip_gemv_args& a, int hd, int grp, int row0, int tile_rows, int lane, f4v acc) {
  const TileEpi e = tile_epilogue_fetch(a, hd, grp, row0, tile_rows, lane, tile_kvpos(a, lane));
  tile_epilogue_finish(a, e, acc, hd);
}

template <int PRO>
__global__ __launch_bounds__(512) void gemv_mfma_kernel(const GemvM p) {
  __shared__ float red[2][8][16];
  __shared__ f4v tile[8][64];
  const ssrhip_gemv_args& a = p.a;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, ks = lane >> 4;
  const int grp = blockIdx.y, row0 = blockIdx.x * p.rows;
  const int N = a.N, K = a.K, B = a.B;
  const int last = p.steps - 1;
  const float* wbase = a.W + (size_t)grp * N * K;                      // uniform
  const unsigned wvoff = (unsigned)min(row0 + (c & (p.rows - 1)), N - 1) * (unsigned)K + ks * 4;
  // row-major x: 16 rows x 64 B per wave instruction; tiled x (SSRHIP_TILED): one contiguous KiB per wave instruction
  const float* xbase = a.x_tiled ? a.x + (size_t)grp * K * 16 : a.x + (size_t)grp * K;
  const unsigned xvoff = a.x_tiled ? (unsigned)(ks * 16 + c) * 4 : (unsigned)min(c, B - 1) * (unsigned)a.x_stride + ks * 4;
  const int xstep = a.x_tiled ? 256 : 16;      // floats per k-step

  f4v acc = {0.f, 0.f, 0.f, 0.f};
  for (int chunk = 0; chunk < p.nchunk; ++chunk) {
    const int tbase = (chunk * p.nw + wave) * SPW;
    // x slice first (L2 hits), then the first 16 weight loads (HBM): loads return in order, so waiting for x leaves the
    // weights in flight. The scheduling barriers / register pins keep hipcc from sinking the x loads to their first use
    // (which would serialise one L2 round trip per MFMA group).
    float4 w[DEPTH];
    float4 xr[SPW];
#pragma unroll
    for (int t = 0; t < SPW; ++t) xr[t] = ld4(xbase + min(tbase + t, last) * xstep + xvoff);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) w[i] = ld_nt(wbase + min(tbase + i, last) * 16 + wvoff);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < SPW; ++t) asm volatile("" : "+v"(xr[t].x), "+v"(xr[t].y), "+v"(xr[t].z), "+v"(xr[t].w));
#pragma unroll
    for (int t = 0; t < SPW; ++t)
      if (tbase + t > last) xr[t] = make_float4(0.f, 0.f, 0.f, 0.f);

    if (PRO == SSRHIP_PRO_LAYERNORM) {
      // two-pass LayerNorm over the whole row (one chunk: K <= nw*512), gamma/beta folded into W/bias by the caller
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < SPW; ++t) s += (xr[t].x + xr[t].y) + (xr[t].z + xr[t].w);
      s = kslot_sum(s);
      if (ks == 0) red[0][wave][c] = s;
      __syncthreads();
      float mean = 0.f;
      for (int w = 0; w < p.nw; ++w) mean += red[0][w][c];
      mean /= (float)K;
      float q = 0.f;
#pragma unroll
      for (int t = 0; t < SPW; ++t) {
        if (tbase + t <= last) {
          const float dx = xr[t].x - mean, dy = xr[t].y - mean, dz = xr[t].z - mean, dw = xr[t].w - mean;
          q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
      }
      q = kslot_sum(q);
      if (ks == 0) red[1][wave][c] = q;
      __syncthreads();
      float var = 0.f;
      for (int w = 0; w < p.nw; ++w) var += red[1][w][c];
      var /= (float)K;
      const float rstd = 1.0f / sqrtf(var + a.ln_eps);
#pragma unroll
      for (int t = 0; t < SPW; ++t) {
        if (tbase + t <= last) {
          xr[t].x = (xr[t].x - mean) * rstd;
          xr[t].y = (xr[t].y - mean) * rstd;
          xr[t].z = (xr[t].z - mean) * rstd;
          xr[t].w = (xr[t].w - mean) * rstd;
        }
      }
    }

    // rolling pipeline: consume k-step t (4 MFMAs), immediately refill its registers with k-step t+DEPTH
#pragma unroll
    for (int t = 0; t < SPW; ++t) {
      const float4 wv = w[t % DEPTH], xv = xr[t];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv.w, acc, 0, 0, 0);
      if (t + DEPTH < SPW) w[t % DEPTH] = ld_nt(wbase + min(tbase + t + DEPTH, last) * 16 + wvoff);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // K-slices of the tile: added in wave order by wave 0
  if (p.nw > 1) {
    tile[wave][lane] = acc;
    __syncthreads();
    if (wave != 0) return;
    acc = tile[0][lane];
    for (int w = 1; w < p.nw; ++w) acc += tile[w][lane];
  }

  tile_epilogue(a, p.hd, grp, row0, p.rows, lane, acc);
}


// =====================================================================================================================
// Version 2 (default): every workgroup owns a CONTIGUOUS BLOCK OF ROWS sized so that all CUs stream the same number of
// bytes, and keeps its x operand across those rows.
//
// Why: in the per-tile kernel above a workgroup re-reads its 16 x K slice of x for every 16 (or 8) weight rows, so the L2->CU
// x traffic equals (or doubles) the weight traffic, and N/16 workgroups rarely divide evenly over 256 CUs (QKV: 384 -> half
// of the CUs stream twice as much as the others). rocprofv3 at 16 rows: QKV 20.3 us, FFN1 20.5 us, FFN2 26 us per launch for
// 50 / 67 / 67 MB (2.5..3.3 TB/s).
//
//   * rows are dealt in 8-row units: workgroup b of `wgs` gets units [b*U/wgs, (b+1)*U/wgs) (counts differ by at most one);
//     two units form a full 16-row MFMA tile, an odd last unit runs as an 8-row tile whose rows 8..15 duplicate 0..7
//     (costs matrix-core cycles, no HBM bytes);
//   * the NW <= 8 waves of the workgroup split K; `gemv_rows_xreg_kernel` (K <= 2048, and every LayerNorm launch up to K = 4096) keeps the wave's
//     x slice in VGPRs for ALL of the workgroup's tiles (loaded once per launch: 128 KB per CU at K = 2048 instead of one
//     slice per 16 rows); `gemv_rows_stream_kernel` (larger K: FFN2) streams x beside W with its own rolling loads, one
//     uninterrupted pipeline over the whole K range (the per-tile kernel drained and refilled it per 4096-float chunk);
//   * the weight pipeline never drains between tiles: the refill of k-step t+DEPTH crosses into the next tile; the last tile is
//     a separate code path without the cross-tile refills, so no load is predicated and none is wasted;
//   * per-tile partial sums of the waves go to LDS once (no barrier inside the row loop); after ONE barrier the waves run the
//     epilogues of different tiles in parallel, adding the K-slices in wave order (deterministic).
// =====================================================================================================================
struct GemvR {
  ssrhip_gemv_args a;
  int nw;       // waves per workgroup (K split)
  int steps;    // K / 16 MFMA k-steps in total
  int spw;      // k-steps per wave (stream kernel: multiple of 16)
  int units;    // ceil(N / 8) 8-row units per group
  int wgs;      // workgroups per group (gridDim.x)
  int hd;
};

constexpr int MAXT = 4;     // 16-row tiles per workgroup (LDS: MAXT x 8 waves x 1 KiB of partial sums)

// Where a 16-row launch spends its time (round 5, VERDICT r4 item 3: "attribute, do not guess") — the stamps of tools/gemvm_lab.hip on an
// LN + QKV launch (us after entry, median over 256 workgroups, wave 0), and an experiment built on them that did NOT pay:
//     as shipped:       x requested 1.1 | first 16 weight requests POSTED 5.1 | x seen 6.6 | LayerNorm done 8.4 | last MFMA 12.1 | end 15.2
//     LayerNorm first:  x requested 1.1 | x seen 1.7 | LayerNorm done 3.7 | weight requests posted 7.2 | last MFMA 12.6 | end 15.5
// (1) the broadcast of x is NOT slow — 128 KB are in the registers of all eight waves 1.7 us after entry when nothing is queued in front of
// them (tools/xbcast_lab.hip: 1.5 us for every CU reading the same 128 KB behind a kernel boundary); (2) POSTING 16 KB of weight requests
// per wave takes 3.5-4 us in either order: a CU accepts HBM misses only at the rate earlier ones return (~36 GB/s), the wave is in-order,
// so as shipped it sits in the request phase with its x long arrived, and the LayerNorm barrier waits for the slowest wave to get through;
// (3) a bare s_barrier between the two request phases (all x requests in front of all weight requests) changed nothing, waiting for x
// without moving the LayerNorm did not either; (4) with the LayerNorm in FRONT of the weight requests the matrix work starts 1.1 us earlier
// on a stream that started 2.6 us later — a wash: tools/decode_ab.py 1.3857 -> 1.3748 and 1.3769 -> 1.3609 ms/step for it, gemvm_bench
// 966.6 -> 980.0 us per step's GEMVs against it. Not kept — and its runtime switch alone cost 1 us per LayerNorm launch while it was in
// the source: with the weight requests under a (uniform) branch hipcc can no longer count them and waits for ALL of them before the
// LayerNorm (`vmcnt(7) .. vmcnt(0)` where the straight-line code has `vmcnt(23) .. vmcnt(16)`; kernel trace 16.09 -> 17.13 us; logs under
// profiles/r05_microbench/). What a 16-row launch pays over the chain floor (9.5 us for 50 MB) is: ~1 us to request x, ~4 us of a stream
// that runs at 4.6-5.6 TB/s instead of 7.3 while the waves alternate between blocked request phases and matrix work, and ~3 us of tail (the
// slowest of the eight K-slice waves reaches the partial-tile barrier 2 us after wave 0; merge + epilogue 1.1 us).

__device__ __forceinline__ f4v mfma4(float a, float b, f4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// per-lane pointer to (row of this lane in tile `tile`, k-slot of this lane) of the weight matrix
__device__ __forceinline__ const float* tile_wptr(const float* wbase, int row_lo, int nun, int tile, int c, int ks, int N, int K, int w_tiled) {
  const int rows = (2 * tile + 1 < nun) ? 16 : 8;
  const int rr = row_lo + tile * 16 + (c & (rows - 1));
  if (w_tiled) return wbase + (size_t)(rr >> 3) * 8 * K + (ks * 8 + (rr & 7)) * 4;   // streaming order: see SSRHIP_WTILED_INDEX (units are zero-padded)
  return wbase + (size_t)min(rr, N - 1) * K + ks * 4;
}

// PAIR (host-selected when every workgroup owns exactly ONE 8-row unit: out-proj, FFN2 — and the weights are in streaming order):
// an 8-row tile in the 16-row MFMA duplicates its rows, so a wave-level load would carry only 512 useful bytes. Instead lanes
// c >= 8 load the NEXT k-step's block of the same 8 rows (adjacent in memory: one contiguous KiB per load, half as many loads), the
// k-steps are consumed in pairs — MFMAs against x[2i] are right in tile rows 0..7, MFMAs against x[2i+1] in rows 8..15 — and the
// two half-results are added across lanes l <-> l+32 at the end. Same matrix-core work, half the load instructions.
// WFIRST (round 6 experiment, SSRHIP_GEMVM_WFIRST=1, LayerNorm launches): the first DEP weight requests in FRONT of the x requests — the HBM
// stream starts ~1 us earlier; x (L2 hits) then returns behind the first 16 KiB of weights, which the LayerNorm waited for anyway.
template <int PRO, int SPWX, int DEP = 16, bool PAIR = false, bool WFIRST = false>
__global__ __launch_bounds__(512) void gemv_rows_xreg_kernel(const GemvR p) {
  static_assert(SPWX >= DEP && SPWX % DEP == 0, "SPWX must be a multiple of the pipeline depth");
  __shared__ float red[2][8][16];
  __shared__ f4v part[MAXT][8][64];
  const ssrhip_gemv_args& a = p.a;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, ks = lane >> 4;
  const int grp = blockIdx.y;
  const int N = a.N, K = a.K, B = a.B;
  const int u_lo = (int)((long long)blockIdx.x * p.units / p.wgs), u_hi = (int)((long long)(blockIdx.x + 1) * p.units / p.wgs);
  const int nun = u_hi - u_lo;
  if (nun <= 0) return;                                                 // uniform; only when wgs > units
  const int ntile = (nun + 1) >> 1;
  const int row_lo = u_lo * 8;
  const int last = p.steps - 1;
  const int tbase = wave * SPWX;
  const float* wbase = a.W + (size_t)grp * (a.w_tiled ? (size_t)p.units * 8 : (size_t)N) * K;
  const int wstep = a.w_tiled ? 128 : 16;        // floats between consecutive k-steps of one lane
  const float* xbase = a.x_tiled ? a.x + (size_t)grp * K * 16 : a.x + (size_t)grp * K;
  const unsigned xvoff = a.x_tiled ? (unsigned)(ks * 16 + c) * 4 : (unsigned)min(c, B - 1) * (unsigned)a.x_stride + ks * 4;
  const int xstep = a.x_tiled ? 256 : 16;

  MSTAMP(0);
  float4 w[DEP];
  float4 xr[SPWX];
  const int kvpos = tile_kvpos(a, lane);                                // the wave's oldest load (QKV launch only)
  __builtin_amdgcn_sched_barrier(0);
  const float* wp = tile_wptr(wbase, row_lo, nun, 0, c, ks, N, K, a.w_tiled) + (PAIR ? (c >> 3) * 128 : 0);
  if (WFIRST && !PAIR) {
#pragma unroll
    for (int i = 0; i < DEP; ++i) w[i] = ld_nt(wp + min(tbase + i, last) * wstep);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int t = 0; t < SPWX; ++t) xr[t] = ld4(xbase + min(tbase + t, last) * xstep + xvoff);
  __builtin_amdgcn_sched_barrier(0);
  MSTAMP(6);
  if (PAIR) {
#pragma unroll
    for (int i = 0; i < SPWX / 2; ++i) w[i] = ld_nt(wp + min(tbase + 2 * i, last - 1) * wstep);   // host: steps even, SPWX / 2 <= DEP
  } else if (!WFIRST) {
#pragma unroll
    for (int i = 0; i < DEP; ++i) w[i] = ld_nt(wp + min(tbase + i, last) * wstep);
  }
  __builtin_amdgcn_sched_barrier(0);
  MSTAMP(7);
  // what this wave's epilogue (tile `wave`) will need, requested now (behind the first weight loads, used after the last MFMA)
  const bool epi_mine = wave < ntile;
  const TileEpi epi0 = tile_epilogue_fetch(a, p.hd, grp, row_lo + wave * 16, epi_mine ? ((2 * wave + 1 < nun) ? 16 : 8) : 0, lane, kvpos);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < SPWX; ++t) asm volatile("" : "+v"(xr[t].x), "+v"(xr[t].y), "+v"(xr[t].z), "+v"(xr[t].w));
#pragma unroll
  for (int t = 0; t < SPWX; ++t)
    if (tbase + t > last) xr[t] = make_float4(0.f, 0.f, 0.f, 0.f);

  MSTAMP(1);
  if (PRO == SSRHIP_PRO_LAYERNORM) {
    // LayerNorm on the register-resident x (gamma / beta folded into W / bias by the caller) with ONE workgroup barrier: every
    // wave computes the two-pass mean / sum of squared deviations of ITS K-slice, the slices are merged with the exact
    // pairwise-update identity  M2 = sum_w M2_w + sum_w n_w (mean_w - mean)^2  (Chan et al.) — as accurate as the reference's
    // two-pass over the whole row. The first DEP weight loads are in flight meanwhile.
    const int nval = max(0, min(SPWX, last + 1 - tbase)) * 16;       // floats of K in this wave's slice (uniform)
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < SPWX; ++t) s += (xr[t].x + xr[t].y) + (xr[t].z + xr[t].w);
    s = kslot_sum(s);
    const float mw = nval > 0 ? s / (float)nval : 0.f;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < SPWX; ++t) {
      if (tbase + t <= last) {
        const float dx = xr[t].x - mw, dy = xr[t].y - mw, dz = xr[t].z - mw, dw = xr[t].w - mw;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
    }
    q = kslot_sum(q);
    if (ks == 0) { red[0][wave][c] = mw; red[1][wave][c] = q; }
    __syncthreads();
    float mean = 0.f;
    for (int v = 0; v < p.nw; ++v) mean += red[0][v][c] * (float)(max(0, min(SPWX, last + 1 - v * SPWX)) * 16);
    mean /= (float)K;
    float var = 0.f;
    for (int v = 0; v < p.nw; ++v) {
      const float d = red[0][v][c] - mean;
      var += red[1][v][c] + (float)(max(0, min(SPWX, last + 1 - v * SPWX)) * 16) * d * d;
    }
    var /= (float)K;
    const float rstd = 1.0f / sqrtf(var + a.ln_eps);
#pragma unroll
    for (int t = 0; t < SPWX; ++t) {
      if (tbase + t <= last) {
        xr[t].x = (xr[t].x - mean) * rstd;
        xr[t].y = (xr[t].y - mean) * rstd;
        xr[t].z = (xr[t].z - mean) * rstd;
        xr[t].w = (xr[t].w - mean) * rstd;
      }
    }
  }

  MSTAMP(2);
  if (PAIR) {
    f4v aA = {0.f, 0.f, 0.f, 0.f}, aB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < SPWX / 2; ++i) {
      const float4 wv = w[i], xa = xr[2 * i], xb = xr[2 * i + 1];
      aA = mfma4(wv.x, xa.x, aA);
      aB = mfma4(wv.x, xb.x, aB);
      aA = mfma4(wv.y, xa.y, aA);
      aB = mfma4(wv.y, xb.y, aB);
      aA = mfma4(wv.z, xa.z, aA);
      aB = mfma4(wv.z, xb.z, aB);
      aA = mfma4(wv.w, xa.w, aA);
      aB = mfma4(wv.w, xb.w, aB);
    }
    f4v acc;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = aA[e] + xor32_f(aB[e]);          // rows 0..7 (lanes < 32) = own rows + rows 8..15 of lane + 32
    MSTAMP(3);
    part[0][wave][lane] = acc;
    __syncthreads();
    MSTAMP(4);
    if (wave == 0) {
      f4v sum = part[0][0][lane];
      for (int v = 1; v < p.nw; ++v) sum += part[0][v][lane];
      tile_epilogue_finish(a, epi0, sum, p.hd);
    }
    MSTAMP(5);
    return;
  }
  // all tiles but the last: the refills past this tile's k-range fetch the head of the next tile
  for (int tile = 0; tile < ntile - 1; ++tile) {
    const float* wn = tile_wptr(wbase, row_lo, nun, tile + 1, c, ks, N, K, a.w_tiled);
    f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < SPWX; ++t) {
      const float4 wv = w[t % DEP], xv = xr[t];
      a0 = mfma4(wv.x, xv.x, a0);
      a1 = mfma4(wv.y, xv.y, a1);
      a0 = mfma4(wv.z, xv.z, a0);
      a1 = mfma4(wv.w, xv.w, a1);
      if (t + DEP < SPWX) w[t % DEP] = ld_nt(wp + min(tbase + t + DEP, last) * wstep);
      else w[t % DEP] = ld_nt(wn + min(tbase + t + DEP - SPWX, last) * wstep);
      __builtin_amdgcn_sched_barrier(0);
    }
    part[tile][wave][lane] = a0 + a1;
    wp = wn;
  }
  {
    f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < SPWX; ++t) {
      const float4 wv = w[t % DEP], xv = xr[t];
      a0 = mfma4(wv.x, xv.x, a0);
      a1 = mfma4(wv.y, xv.y, a1);
      a0 = mfma4(wv.z, xv.z, a0);
      a1 = mfma4(wv.w, xv.w, a1);
      if (t + DEP < SPWX) w[t % DEP] = ld_nt(wp + min(tbase + t + DEP, last) * wstep);
      __builtin_amdgcn_sched_barrier(0);
    }
    part[ntile - 1][wave][lane] = a0 + a1;
  }
  MSTAMP(3);
  __syncthreads();
  MSTAMP(4);
  for (int tile = wave; tile < ntile; tile += p.nw) {
    f4v acc = part[tile][0][lane];
    for (int v = 1; v < p.nw; ++v) acc += part[tile][v][lane];
    if (tile == wave) tile_epilogue_finish(a, epi0, acc, p.hd);
    else tile_epilogue(a, p.hd, grp, row_lo + tile * 16, (2 * tile + 1 < nun) ? 16 : 8, lane, acc);
  }
  MSTAMP(5);
}

// Round 6: the LayerNorm launches at K = 2048 with the LayerNorm OFF the streaming waves. Round 5's stamps
// (above) say where such a launch loses against the chain floor: a wave is in-order, posting its weight requests blocks it for ~4 us (the CU
// accepts misses at the rate earlier ones return), and everything else a streaming wave has to do — fetch its x slice, the two LayerNorm
// passes, the statistics barrier, the normalisation — happens either in front of the requests (then HBM starts late) or behind the first 16
// of them (then HBM runs dry while the LayerNorm computes: as shipped, ~2 us with nothing requested). Three re-orderings inside the
// 8-wave kernel did not pay. Here the workgroup has 12 waves:
//   * waves 0-7 (streaming) post ALL weight requests of the workgroup's NT tiles at entry — 2 x 16 KiB per wave, nothing in front of them,
//     nothing between them — and then wait;
//   * waves 8-11 (edge) fetch x (32 k-steps = 32 KiB each, L2 hits, their own request queues), compute the LayerNorm statistics in
//     EXACTLY the eight 256-column slices, the order and the expressions of gemv_rows_xreg_kernel (so the normalised x, and with it every
//     output, is bit-identical to that kernel), normalise, and leave x' in LDS in k-step order (128 KiB of the CU's 160);
//   * two workgroup barriers (statistics complete, x' complete) — the streaming waves reach them when their requests are posted, the
//     edge waves long before — then the streaming waves read their 16 KiB slice of x' from LDS k-step by k-step and run the MFMA chains of
//     both tiles on weights that are resident or in flight. Partial tiles, merge order and epilogue as in gemv_rows_xreg_kernel.
// 12 waves are 3 per SIMD: 168 VGPRs per wave, of which the resident weights take 128 (NT = 2) — x' therefore stays in LDS and is read
// one k-step ahead. Host-selected when every workgroup owns exactly NT tiles (QKV: 3 units, FFN1: 4 units -> NT = 2; head MLP: 2 units ->
// NT = 1), x is tiled and K = 2048.
// MEASURED (profiles/r06_microbench/gemvm_bench_16_edge.log, bench_n1_8utts_edge*.json; bit-identical outputs): it LOSES — LN + QKV 17.5
// against 15.3-16.0 us, LN + FFN1 20.3 against 15.8-16.0, head MLP 12.9 against 12.2; the 16-row step 1.521 against 1.410 ms. With every
// request posted up front the posting phase simply lasts as long as the stream (FFN1: 262 KB per CU in ~12 us = 5.6 TB/s chip-wide —
// that, not 7.3 TB/s, is what this access pattern gets), and the matrix work that the 8-wave kernel overlaps with its refills now waits
// behind all of it. The premise "the LayerNorm is what delays the stream" was wrong: the stream is the critical path, the 8-wave kernel
// already hides the LayerNorm under the first 128 KB per CU. Kept as an OPT-IN (SSRHIP_GEMVM_EDGE=1, read at every launch) with its
// bit-identity test, like the other recorded negatives.
constexpr int EDGE_XS_BYTES = 128 * 1024;                 // x': 128 k-steps x 1 KiB
constexpr int edge_lds(int NT) { return EDGE_XS_BYTES + NT * 8 * 1024 + 2 * 8 * 16 * 4; }

template <int NT>
__global__ __launch_bounds__(768) void gemv_rows_edge_kernel(const GemvR p) {
  constexpr int SW = 8, SPWX = 16;                        // streaming waves, k-steps per streaming wave (K = 2048: 128 k-steps)
  extern __shared__ __attribute__((aligned(1024))) char esm[];
  float4* const xs = reinterpret_cast<float4*>(esm);                                  // [128][64]
  f4v* const part = reinterpret_cast<f4v*>(esm + EDGE_XS_BYTES);                       // [NT][8][64]
  float* const red = reinterpret_cast<float*>(esm + EDGE_XS_BYTES + NT * 8 * 1024);    // [2][8][16]
  const ssrhip_gemv_args& a = p.a;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, ks = lane >> 4;
  const int grp = blockIdx.y;
  const int N = a.N, K = a.K;
  const int u_lo = (int)((long long)blockIdx.x * p.units / p.wgs), u_hi = (int)((long long)(blockIdx.x + 1) * p.units / p.wgs);
  const int nun = u_hi - u_lo;                                          // host: 2 NT - 1 or 2 NT for every workgroup
  const int row_lo = u_lo * 8;
  if (wave >= SW) {
    // ---------------- edge: x -> LayerNorm -> x' in LDS
    const int e = wave - SW;
    const float* xp = a.x + (size_t)grp * K * 16 + (size_t)(32 * e) * 256 + (unsigned)(ks * 16 + c) * 4;   // tiled x: one KiB per k-step
    float4 xr[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) xr[t] = ld4(xp + t * 256);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {                                       // slice v = 2 e + h: what streaming wave v computes in the 8-wave kernel
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < SPWX; ++t) s += (xr[16 * h + t].x + xr[16 * h + t].y) + (xr[16 * h + t].z + xr[16 * h + t].w);
      s = kslot_sum(s);
      const float mw = s / (float)(SPWX * 16);
      float q = 0.f;
#pragma unroll
      for (int t = 0; t < SPWX; ++t) {
        const float dx = xr[16 * h + t].x - mw, dy = xr[16 * h + t].y - mw, dz = xr[16 * h + t].z - mw, dw = xr[16 * h + t].w - mw;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
      q = kslot_sum(q);
      if (ks == 0) { red[(2 * e + h) * 16 + c] = mw; red[128 + (2 * e + h) * 16 + c] = q; }
    }
    __syncthreads();                                                    // (A) the eight slices' statistics are in LDS
    float mean = 0.f;
    for (int v = 0; v < SW; ++v) mean += red[v * 16 + c] * (float)(SPWX * 16);
    mean /= (float)K;
    float var = 0.f;
    for (int v = 0; v < SW; ++v) {
      const float d = red[v * 16 + c] - mean;
      var += red[128 + v * 16 + c] + (float)(SPWX * 16) * d * d;
    }
    var /= (float)K;
    const float rstd = 1.0f / sqrtf(var + a.ln_eps);
#pragma unroll
    for (int t = 0; t < 32; ++t)
      xs[(32 * e + t) * 64 + lane] = make_float4((xr[t].x - mean) * rstd, (xr[t].y - mean) * rstd, (xr[t].z - mean) * rstd, (xr[t].w - mean) * rstd);
    __syncthreads();                                                    // (B) x' complete
    return;
  }
  // ---------------- streaming: every weight request of this wave, now
  const int tbase = wave * SPWX;
  const float* wbase = a.W + (size_t)grp * (a.w_tiled ? (size_t)p.units * 8 : (size_t)N) * K;
  const int wstep = a.w_tiled ? 128 : 16;
  const int kvpos = tile_kvpos(a, lane);                                // the wave's oldest load (QKV launch only)
  __builtin_amdgcn_sched_barrier(0);
  float4 w[NT][SPWX];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const float* wp = tile_wptr(wbase, row_lo, nun, i, c, ks, N, K, a.w_tiled);
#pragma unroll
    for (int t = 0; t < SPWX; ++t) w[i][t] = ld_nt(wp + (tbase + t) * wstep);
  }
  __builtin_amdgcn_sched_barrier(0);
  const int ntile = (nun + 1) >> 1;
  const TileEpi epi0 = tile_epilogue_fetch(a, p.hd, grp, row_lo + wave * 16, wave < ntile ? ((2 * wave + 1 < nun) ? 16 : 8) : 0, lane, kvpos);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();                                                      // (A)
  __syncthreads();                                                      // (B)
  const float4* xw = xs + (size_t)tbase * 64 + lane;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    float4 xv = xw[0];
#pragma unroll
    for (int t = 0; t < SPWX; ++t) {
      const float4 wv = w[i][t], xc = xv;
      if (t + 1 < SPWX) xv = xw[(t + 1) * 64];                          // one k-step ahead (LDS)
      a0 = mfma4(wv.x, xc.x, a0);
      a1 = mfma4(wv.y, xc.y, a1);
      a0 = mfma4(wv.z, xc.z, a0);
      a1 = mfma4(wv.w, xc.w, a1);
    }
    part[(i * 8 + wave) * 64 + lane] = a0 + a1;
  }
  __syncthreads();                                                      // (C) partial tiles complete (the edge waves have left)
  if (wave < ntile) {
    f4v acc = part[(wave * 8) * 64 + lane];
    for (int v = 1; v < SW; ++v) acc += part[(wave * 8 + v) * 64 + lane];
    tile_epilogue_finish(a, epi0, acc, p.hd);
  }
}

// K > 2048 without a LayerNorm prologue (FFN2, K = 8192): x no longer fits the registers of 8 waves, so it is streamed like W: per k-step one KiB of W (HBM) and one
// KiB of x (L2; the tiled layout makes it one contiguous KiB per wave instruction), 16 of each in flight per wave.
template <bool PAIR>
__global__ __launch_bounds__(512) void gemv_rows_stream_kernel(const GemvR p) {
  constexpr int DEP = 16;
  __shared__ f4v part[MAXT][8][64];
  const ssrhip_gemv_args& a = p.a;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, ks = lane >> 4;
  const int grp = blockIdx.y;
  const int N = a.N, K = a.K, B = a.B;
  const int u_lo = (int)((long long)blockIdx.x * p.units / p.wgs), u_hi = (int)((long long)(blockIdx.x + 1) * p.units / p.wgs);
  const int nun = u_hi - u_lo;
  if (nun <= 0) return;
  const int ntile = (nun + 1) >> 1;
  const int row_lo = u_lo * 8;
  const int last = p.steps - 1;
  const int tbase = wave * p.spw;
  const int ngrp = p.spw / DEP;                  // groups of DEP k-steps per tile for this wave
  const float* wbase = a.W + (size_t)grp * (a.w_tiled ? (size_t)p.units * 8 : (size_t)N) * K;
  const int wstep = a.w_tiled ? 128 : 16;
  const float* xp = (a.x_tiled ? a.x + (size_t)grp * K * 16 : a.x + (size_t)grp * K) +
                    (a.x_tiled ? (unsigned)(ks * 16 + c) * 4 : (unsigned)min(c, B - 1) * (unsigned)a.x_stride + ks * 4);
  const int xstep = a.x_tiled ? 256 : 16;

  MSTAMP(0);
  float4 w[DEP], xr[DEP];
  const int kvpos = tile_kvpos(a, lane);                                // the wave's oldest load (QKV launch only)
  __builtin_amdgcn_sched_barrier(0);
  const float* wp = tile_wptr(wbase, row_lo, nun, 0, c, ks, N, K, a.w_tiled) + (PAIR ? (c >> 3) * 128 : 0);
  if (PAIR) {
    // one 8-row unit per workgroup (see gemv_rows_xreg_kernel): per group of 16 k-steps 8 weight loads (k-step pairs) + 16 x loads
    float4 wq[DEP / 2];
#pragma unroll
    for (int i = 0; i < DEP; ++i) xr[i] = ld4(xp + min(tbase + i, last) * xstep);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < DEP / 2; ++i) wq[i] = ld_nt(wp + min(tbase + 2 * i, last - 1) * wstep);
    __builtin_amdgcn_sched_barrier(0);
    const TileEpi epi0 = tile_epilogue_fetch(a, p.hd, grp, row_lo, wave == 0 ? 8 : 0, lane, kvpos);
    __builtin_amdgcn_sched_barrier(0);
    MSTAMP(1);
    f4v aA = {0.f, 0.f, 0.f, 0.f}, aB = {0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < ngrp - 1; ++g) {                                   // all groups but the last: refill for group g + 1
      const int kb = tbase + g * DEP, kbn = kb + DEP;
#pragma unroll
      for (int i = 0; i < DEP / 2; ++i) {
        const float4 wv = wq[i];
        float4 xa = xr[2 * i], xb = xr[2 * i + 1];
        if (kb + 2 * i > last) { xa = make_float4(0.f, 0.f, 0.f, 0.f); xb = xa; }   // steps is even: a pair is in or out as a whole
        aA = mfma4(wv.x, xa.x, aA);
        aB = mfma4(wv.x, xb.x, aB);
        aA = mfma4(wv.y, xa.y, aA);
        aB = mfma4(wv.y, xb.y, aB);
        aA = mfma4(wv.z, xa.z, aA);
        aB = mfma4(wv.z, xb.z, aB);
        aA = mfma4(wv.w, xa.w, aA);
        aB = mfma4(wv.w, xb.w, aB);
        xr[2 * i] = ld4(xp + min(kbn + 2 * i, last) * xstep);
        xr[2 * i + 1] = ld4(xp + min(kbn + 2 * i + 1, last) * xstep);
        wq[i] = ld_nt(wp + min(kbn + 2 * i, last - 1) * wstep);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {
      const int kb = tbase + (ngrp - 1) * DEP;
#pragma unroll
      for (int i = 0; i < DEP / 2; ++i) {
        const float4 wv = wq[i];
        float4 xa = xr[2 * i], xb = xr[2 * i + 1];
        if (kb + 2 * i > last) { xa = make_float4(0.f, 0.f, 0.f, 0.f); xb = xa; }
        aA = mfma4(wv.x, xa.x, aA);
        aB = mfma4(wv.x, xb.x, aB);
        aA = mfma4(wv.y, xa.y, aA);
        aB = mfma4(wv.y, xb.y, aB);
        aA = mfma4(wv.z, xa.z, aA);
        aB = mfma4(wv.z, xb.z, aB);
        aA = mfma4(wv.w, xa.w, aA);
        aB = mfma4(wv.w, xb.w, aB);
      }
    }
    f4v acc;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = aA[e] + xor32_f(aB[e]);
    MSTAMP(3);
    part[0][wave][lane] = acc;
    __syncthreads();
    MSTAMP(4);
    if (wave == 0) {
      f4v sum = part[0][0][lane];
      for (int v = 1; v < p.nw; ++v) sum += part[0][v][lane];
      tile_epilogue_finish(a, epi0, sum, p.hd);
    }
    MSTAMP(5);
    return;
  }
#pragma unroll
  for (int i = 0; i < DEP; ++i) xr[i] = ld4(xp + min(tbase + i, last) * xstep);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < DEP; ++i) w[i] = ld_nt(wp + min(tbase + i, last) * wstep);
  __builtin_amdgcn_sched_barrier(0);
  const bool epi_mine = wave < ntile;
  const TileEpi epi0 = tile_epilogue_fetch(a, p.hd, grp, row_lo + wave * 16, epi_mine ? ((2 * wave + 1 < nun) ? 16 : 8) : 0, lane, kvpos);
  __builtin_amdgcn_sched_barrier(0);
  MSTAMP(1);
  const int total = ntile * ngrp;
  f4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  int tile = 0, kg = 0;
  for (int g = 0; g < total - 1; ++g) {
    // group g = (tile, kg); the refills fetch group g + 1
    int tile_n = tile, kg_n = kg + 1;
    if (kg_n == ngrp) { kg_n = 0; tile_n = tile + 1; }
    const float* wn = (tile_n == tile) ? wp : tile_wptr(wbase, row_lo, nun, tile_n, c, ks, N, K, a.w_tiled);
    const int kb = tbase + kg * DEP, kbn = tbase + kg_n * DEP;
#pragma unroll
    for (int t = 0; t < DEP; ++t) {
      const float4 wv = w[t];
      float4 xv = xr[t];
      if (kb + t > last) xv = make_float4(0.f, 0.f, 0.f, 0.f);           // uniform: k-steps past the end of K contribute nothing
      a0 = mfma4(wv.x, xv.x, a0);
      a1 = mfma4(wv.y, xv.y, a1);
      a0 = mfma4(wv.z, xv.z, a0);
      a1 = mfma4(wv.w, xv.w, a1);
      const int kk = min(kbn + t, last);
      xr[t] = ld4(xp + kk * xstep);
      w[t] = ld_nt(wn + kk * wstep);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kg_n == 0) {                                                      // uniform: tile finished
      part[tile][wave][lane] = a0 + a1;
      a0 = (f4v){0.f, 0.f, 0.f, 0.f};
      a1 = (f4v){0.f, 0.f, 0.f, 0.f};
    }
    tile = tile_n; kg = kg_n; wp = wn;
  }
  {
    const int kb = tbase + kg * DEP;
#pragma unroll
    for (int t = 0; t < DEP; ++t) {
      const float4 wv = w[t];
      float4 xv = xr[t];
      if (kb + t > last) xv = make_float4(0.f, 0.f, 0.f, 0.f);
      a0 = mfma4(wv.x, xv.x, a0);
      a1 = mfma4(wv.y, xv.y, a1);
      a0 = mfma4(wv.z, xv.z, a0);
      a1 = mfma4(wv.w, xv.w, a1);
    }
    part[ntile - 1][wave][lane] = a0 + a1;
  }
  MSTAMP(3);
  __syncthreads();
  MSTAMP(4);
  for (int t2 = wave; t2 < ntile; t2 += p.nw) {
    f4v acc = part[t2][0][lane];
    for (int v = 1; v < p.nw; ++v) acc += part[t2][v][lane];
    if (t2 == wave) tile_epilogue_finish(a, epi0, acc, p.hd);
    else tile_epilogue(a, p.hd, grp, row_lo + t2 * 16, (2 * t2 + 1 < nun) ? 16 : 8, lane, acc);
  }
  MSTAMP(5);
}

}  // namespace

// called by ssrhip_gemv for 4 < B <= 16 (arguments already validated there)
int ssrhip_gemv_mfma_launch(const ssrhip_gemv_args* a, hipStream_t s) {
  SSR_REQUIRE(a->B > 0 && a->B <= 16, "ssrhip_gemv: B=%d rows > 16", a->B);
  SSR_REQUIRE(a->K % 16 == 0, "ssrhip_gemv (B>4): K=%d must be a multiple of 16", a->K);
  SSR_REQUIRE(a->pro == SSRHIP_PRO_NONE || a->pro == SSRHIP_PRO_LAYERNORM,
              "ssrhip_gemv (B>4): the split-KV combine prologue is not fused; run ssrhip_attn_combine first");
  SSR_REQUIRE(a->x, "ssrhip_gemv: x is null");
  SSR_REQUIRE(!a->y_tiled || (a->N % 4 == 0 && a->epi != SSRHIP_EPI_QKV_APPEND), "ssrhip_gemv: tiled y needs N %% 4 == 0 and is not available for the q output");
  static int g_ver = -1, g_cus = 0, g_wpc = 0, g_dep8 = 0, g_nopair = 0;
  if (g_ver < 0) {
    const char* e = getenv("SSRHIP_GEMVM_V");          // 1: the per-tile kernel (round 1), 2: rows-per-workgroup kernels (default)
    g_ver = (e && atoi(e) == 1) ? 1 : 2;
    int dev = 0, cu = 0;
    g_cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cu > 0) ? cu : 256;
    const char* w = getenv("SSRHIP_GEMVM_WPC");        // tuning knob: 512-thread workgroups per CU (default 1)
    g_wpc = (w && atoi(w) >= 1 && atoi(w) <= 4) ? atoi(w) : 1;
    g_nopair = getenv("SSRHIP_GEMVM_NOPAIR") != nullptr;   // A/B knob: 8-row tiles with duplicated rows instead of k-step pairs
    g_dep8 = getenv("SSRHIP_GEMVM_DEP8") != nullptr;   // experiment: 8 instead of 16 weight loads in flight per wave (LayerNorm launches)
  }
  const int hd = a->kv.head_dim > 0 ? a->kv.head_dim : 1;
  if (a->pro == SSRHIP_PRO_LAYERNORM) {
    SSR_REQUIRE(a->K <= 4096, "ssrhip_gemv (B>4): LayerNorm prologue needs K <= 4096");
    SSR_REQUIRE(!a->ln_w && !a->ln_b, "ssrhip_gemv (B>4): LayerNorm gamma/beta must be folded into W/bias (ln_w == ln_b == NULL)");
  }
  if (a->epi == SSRHIP_EPI_QKV_APPEND) {
    SSR_REQUIRE(a->N == 3 * a->K && a->groups == 1 && a->kv.pool && a->kv.table && a->kv_pos && a->kv.head_dim > 0 && a->kv.head_dim % 4 == 0,
                "ssrhip_gemv: QKV epilogue needs N==3K and a kv cache");
  }
  SSR_REQUIRE(!a->w_tiled || g_ver == 2, "ssrhip_gemv (B>4): the streaming-order weight layout needs the rows-per-workgroup kernels (SSRHIP_GEMVM_V=2)");
  if (g_ver == 2) {
    GemvR r;
    r.a = *a;
    r.steps = a->K / 16;
    r.hd = hd;
    r.units = (a->N + 7) / 8;
    const bool xreg = a->K <= 2048 || a->pro == SSRHIP_PRO_LAYERNORM;   // x slice of every wave in registers; else streamed beside W
    const int spwx = a->K <= 2048 ? 16 : 32;             // k-steps of x a wave keeps in registers
    if (xreg) {
      r.nw = (r.steps + spwx - 1) / spwx;
      r.spw = spwx;
    } else {
      r.nw = 8;
      r.spw = ((r.steps + 7) / 8 + 15) / 16 * 16;
    }
    // one 512-thread workgroup per CU (two when the K split leaves it <= 4 waves), all groups together
    int target = g_cus * g_wpc * (r.nw <= 4 ? 2 : 1) / a->groups;
    if (target < 1) target = 1;
    r.wgs = r.units < target ? r.units : target;
    const int need = (r.units + 2 * MAXT - 1) / (2 * MAXT);   // LDS holds MAXT tiles of partials per workgroup
    if (r.wgs < need) r.wgs = need;
    SSR_REQUIRE(r.wgs <= 65535 * 32, "ssrhip_gemv (B>4): N too large");
    dim3 grid(r.wgs, a->groups), block(r.nw * 64);
    // every workgroup owns exactly one 8-row unit (out-proj, FFN2) and the weights are in streaming order: k-step pairs per load
    const bool pair = a->w_tiled && r.units <= r.wgs && r.steps % 2 == 0 && !g_nopair;
    // round 6, opt-in (measured slower): LayerNorm launches with the LayerNorm on four extra waves and every weight request posted at entry
    const char* ee = getenv("SSRHIP_GEMVM_EDGE");
    const int per = r.wgs > 0 ? r.units / r.wgs : 0;
    if (a->pro == SSRHIP_PRO_LAYERNORM && a->K == 2048 && a->x_tiled && ee && ee[0] == '1' && g_wpc == 1 && a->groups == 1 &&
        r.units % r.wgs == 0 && per >= 1 && per <= 4) {
      static ssr_once_per_device once;
      if (once.need()) {
        SSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_rows_edge_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, edge_lds(1)));
        SSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_rows_edge_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, edge_lds(2)));
      }
      if (per <= 2) hipLaunchKernelGGL(gemv_rows_edge_kernel<1>, grid, dim3(768), edge_lds(1), s, r);
      else hipLaunchKernelGGL(gemv_rows_edge_kernel<2>, grid, dim3(768), edge_lds(2), s, r);
      SSR_LAUNCH_CHECK();
      return 0;
    }
    if (!xreg && pair) hipLaunchKernelGGL(gemv_rows_stream_kernel<true>, grid, block, 0, s, r);
    else if (!xreg) hipLaunchKernelGGL(gemv_rows_stream_kernel<false>, grid, block, 0, s, r);
    else if (pair && a->pro == SSRHIP_PRO_NONE && spwx == 16) hipLaunchKernelGGL((gemv_rows_xreg_kernel<SSRHIP_PRO_NONE, 16, 16, true>), grid, block, 0, s, r);
    else if (g_dep8 && a->pro == SSRHIP_PRO_LAYERNORM && spwx == 16) hipLaunchKernelGGL((gemv_rows_xreg_kernel<SSRHIP_PRO_LAYERNORM, 16, 8>), grid, block, 0, s, r);
    else if (a->pro == SSRHIP_PRO_LAYERNORM && spwx == 16 && getenv("SSRHIP_GEMVM_WFIRST") && getenv("SSRHIP_GEMVM_WFIRST")[0] == '1')
      hipLaunchKernelGGL((gemv_rows_xreg_kernel<SSRHIP_PRO_LAYERNORM, 16, 16, false, true>), grid, block, 0, s, r);
    else if (a->pro == SSRHIP_PRO_LAYERNORM && spwx == 16) hipLaunchKernelGGL((gemv_rows_xreg_kernel<SSRHIP_PRO_LAYERNORM, 16>), grid, block, 0, s, r);
    else if (a->pro == SSRHIP_PRO_LAYERNORM) hipLaunchKernelGGL((gemv_rows_xreg_kernel<SSRHIP_PRO_LAYERNORM, 32>), grid, block, 0, s, r);
    else if (spwx == 16) hipLaunchKernelGGL((gemv_rows_xreg_kernel<SSRHIP_PRO_NONE, 16>), grid, block, 0, s, r);
    else hipLaunchKernelGGL((gemv_rows_xreg_kernel<SSRHIP_PRO_NONE, 32>), grid, block, 0, s, r);
    SSR_LAUNCH_CHECK();
    return 0;
  }
  GemvM p;
  p.a = *a;
  p.steps = a->K / 16;
  p.nw = (p.steps + SPW - 1) / SPW;
  if (p.nw > 8) p.nw = 8;
  p.nchunk = (p.steps + p.nw * SPW - 1) / (p.nw * SPW);
  p.hd = hd;
  // 16-row tiles unless that gives fewer workgroups than CUs (out-proj, FFN2: N/16 = 128) — then 8-row tiles
  p.rows = (a->N / 16) * a->groups >= 256 ? 16 : 8;
  if (const char* e = getenv("SSRHIP_GEMVM_ROWS")) { const int v = atoi(e); if (v == 8 || v == 16) p.rows = v; }   // tuning knob
  dim3 grid((a->N + p.rows - 1) / p.rows, a->groups);
  if (a->pro == SSRHIP_PRO_LAYERNORM) hipLaunchKernelGGL((gemv_mfma_kernel<SSRHIP_PRO_LAYERNORM>), grid, dim3(p.nw * 64), 0, s, p);
  else hipLaunchKernelGGL((gemv_mfma_kernel<SSRHIP_PRO_NONE>), grid, dim3(p.nw * 64), 0, s, p);
  SSR_LAUNCH_CHECK();
  return 0;
}
