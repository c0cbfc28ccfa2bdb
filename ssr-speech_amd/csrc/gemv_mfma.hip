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

  // epilogue: this lane holds rows r0..r0+3 of batch column c
  const int r0 = row0 + ks * 4;
  if (c >= B || r0 >= N || ks * 4 >= p.rows) return;
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  const int nvalid = min(4, N - r0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j < nvalid) {
      if (a.bias) v[j] += a.bias[(size_t)grp * N + r0 + j];
      if (a.act == SSRHIP_ACT_RELU) v[j] = fmaxf(v[j], 0.f);
      else if (a.act == SSRHIP_ACT_GELU_ERF) v[j] = 0.5f * v[j] * (1.0f + erff(v[j] * 0.70710678118654752440f));
    }
  }
  float* dst;
  if (a.epi == SSRHIP_EPI_QKV_APPEND) {
    const int D = K, which = r0 / D, cc = r0 % D;
    if (which == 0) dst = a.y + (size_t)c * a.y_stride + cc;
    else dst = kv_addr(a.kv, c, a.layer, which - 1, cc / p.hd, a.kv_pos[c]) + (cc % p.hd);
  } else if (a.y_tiled) {
    dst = a.y + (size_t)grp * N * 16 + SSRHIP_TILED(c, r0);
  } else {
    dst = a.y + (size_t)c * a.y_stride + (size_t)grp * N + r0;
  }
  if (a.epi == SSRHIP_EPI_RESIDUAL) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nvalid) v[j] = dst[j] + v[j];
  }
  if (nvalid == 4 && ((reinterpret_cast<size_t>(dst) & 15) == 0)) {
    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nvalid) dst[j] = v[j];
  }
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
  GemvM p;
  p.a = *a;
  p.steps = a->K / 16;
  p.nw = (p.steps + SPW - 1) / SPW;
  if (p.nw > 8) p.nw = 8;
  p.nchunk = (p.steps + p.nw * SPW - 1) / (p.nw * SPW);
  p.hd = a->kv.head_dim > 0 ? a->kv.head_dim : 1;
  if (a->pro == SSRHIP_PRO_LAYERNORM) {
    SSR_REQUIRE(p.nchunk == 1, "ssrhip_gemv (B>4): LayerNorm prologue needs K <= 4096");
    SSR_REQUIRE(!a->ln_w && !a->ln_b, "ssrhip_gemv (B>4): LayerNorm gamma/beta must be folded into W/bias (ln_w == ln_b == NULL)");
  }
  if (a->epi == SSRHIP_EPI_QKV_APPEND) {
    SSR_REQUIRE(a->N == 3 * a->K && a->groups == 1 && a->kv.pool && a->kv.table && a->kv_pos && a->kv.head_dim > 0 && a->kv.head_dim % 4 == 0,
                "ssrhip_gemv: QKV epilogue needs N==3K and a kv cache");
  }
  // 16-row tiles unless that gives fewer workgroups than CUs (out-proj, FFN2: N/16 = 128) — then 8-row tiles: the duplicate
  // tile rows cost MFMA cycles (not the bound) but no HBM bytes. Measured (tools/gemvm_bench): out-proj 8.7 -> 7.6 us,
  // FFN2 31 -> 26 us; for N/16 >= 256 the 16-row tile is faster (QKV 17 vs 21 us).
  p.rows = (a->N / 16) * a->groups >= 256 ? 16 : 8;
  if (const char* e = getenv("SSRHIP_GEMVM_ROWS")) { const int v = atoi(e); if (v == 8 || v == 16) p.rows = v; }   // tuning knob
  dim3 grid((a->N + p.rows - 1) / p.rows, a->groups);
  if (a->pro == SSRHIP_PRO_LAYERNORM) hipLaunchKernelGGL((gemv_mfma_kernel<SSRHIP_PRO_LAYERNORM>), grid, dim3(p.nw * 64), 0, s, p);
  else hipLaunchKernelGGL((gemv_mfma_kernel<SSRHIP_PRO_NONE>), grid, dim3(p.nw * 64), 0, s, p);
  SSR_LAUNCH_CHECK();
  return 0;
}
