// gemm.hip — dense fp32 GEMM on the CDNA4 matrix cores for the prefill rows (and the codec),
// plus row LayerNorm and the prefill KV scatter.
//
//   C[M][N] = epi( A[M][K] . W[N][K]^T + bias[N] )      (both operands K-contiguous)
//
// v_mfma_f32_32x32x2_f32: exact fp32 (bit-identical to a k-ordered fmaf chain), 157 TF peak = the fp32
// vector rate, but one VGPR per operand per lane and the VALU stays free for the epilogue.
// Block tile 64(M) x 128(N) x 16(K), 4 waves side by side along N (each 64x32 = two 32x32 accumulators).
// LDS rows are padded to 20 floats: a lane's ds_read_b128 (4 consecutive k of one row) then hits 16
// distinct 16-byte slots per 16-lane group -> conflict-free. The k-pairs fed to one MFMA are (t, t+4):
// any permutation of k is legal as long as A and W use the same one.
#include <stdio.h>
#include <stdlib.h>
#include "common.h"

bool ssrhip_gemm_split_eligible(const ssrhip_gemm_args* a);          // gemm_split.hip
int ssrhip_gemm_split_launch(const ssrhip_gemm_args* a, hipStream_t s);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef SSR_GEMM_SINGLE_BUF
#define SSR_GEMM_SINGLE_BUF 1      // 0: double-buffered LDS tiles, one barrier per k-tile — measured equal (4096^3: 100.5 vs 99.0 TFLOP/s;
#endif                             // codec 32 x 30 s: 86.6 vs 86.7 ms), so the smaller LDS footprint stays the default
#ifndef SSR_GEMM_BK
#define SSR_GEMM_BK 16
#endif
constexpr int BK = SSR_GEMM_BK, LDSW = BK + 4;
constexpr int TPR = BK / 4, RPP = 256 / TPR;      // loader: threads per tile row, rows per pass of the 256 threads

__device__ __forceinline__ float act_fn(float v, int act) {
  if (act == SSRHIP_ACT_RELU) return fmaxf(v, 0.f);
  if (act == SSRHIP_ACT_GELU_ERF) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  return v;
}

// 4 waves arranged MW x NW; each wave owns MT x NT accumulator blocks of 32x32. Block tile = (MW*MT*32) x (NW*NT*32) x 16.
//   <1,4,2,1>  64 x 128 : the general shape (prefill, wide convolutions, LSTM input GEMM)
//   <2,2,1,1>  64 x  64 : when 64 x 128 tiles would not even give one workgroup per CU (prefill of a single prompt: M ~ 600,
//                         N = 2048 -> 160 workgroups; with 64 x 64 tiles 320)
//   <4,1,2,2> 256 x  64 : N <= 64  (SEANet layers with 64 output channels at 16 kHz / 8 kHz: rows are plentiful, columns are not)
//   <4,1,2,1> 256 x  32 : N <= 32  (ResBlock bottlenecks 64 -> 32, the final 64 -> 1 convolution)
// With the wide tile a 32-column layer wastes 3 of 4 waves on zero columns (measured 23 TFLOP/s useful on those layers).
template <int MW, int NW, int MT, int NT>
__global__ __launch_bounds__(256) void gemm_kernel(const ssrhip_gemm_args a0) {
  constexpr int BM = MW * MT * 32, BN = NW * NT * 32;
  constexpr int LA = BM / RPP, LW = (BN + RPP - 1) / RPP;       // float4 loads per thread per k-tile
  constexpr int NBUF = SSR_GEMM_SINGLE_BUF ? 1 : 2;
  const int li_ = threadIdx.x & 31, lh_ = (threadIdx.x & 63) >> 5;
  // NBUF == 2: double-buffered LDS tiles, ONE barrier per k-tile (the next tile is written into the other buffer after this tile's
  // MFMAs); NBUF == 1 (default, see SSR_GEMM_SINGLE_BUF): one buffer, two barriers per k-tile. Several workgroups share a CU, so the
  // barrier stalls are already hidden: both forms keep the matrix core busy ~65 % of the time (profiles/r02_mfma_util_*).
  __shared__ __attribute__((aligned(16))) float As[NBUF][BM * LDSW];
  __shared__ __attribute__((aligned(16))) float Ws[NBUF][BN * LDSW];
  ssrhip_gemm_args a = a0;
  {   // batched problems: grid.z
    const size_t z = blockIdx.z;
    a.A += z * (size_t)a.strideA;
    a.C += z * (size_t)a.strideC;
    if (a.R) a.R += z * (size_t)a.strideR;
    if (a.rclass) a.rclass += z * (size_t)a.rclass_stride;
  }
  const int t = threadIdx.x, wave = t >> 6;
  const int wm = wave / NW, wn = wave % NW;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int lr = t / TPR, lc = (t % TPR) * 4;            // loader: row, first k column
  const int M = a.M, N = a.N, K = a.K;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[LA], rw[LW];
  auto gload = [&](int k0) {
    const bool kin = (k0 + lc) < K;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int m = m0 + lr + RPP * i;
      ra[i] = (kin && m < M) ? ld4(a.A + (size_t)m * a.lda + k0 + lc) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.act_in == SSRHIP_ACT_ELU) { ra[i].x = elu1(ra[i].x); ra[i].y = elu1(ra[i].y); ra[i].z = elu1(ra[i].z); ra[i].w = elu1(ra[i].w); }
    }
#pragma unroll
    for (int i = 0; i < LW; ++i) {
      const int n = n0 + lr + RPP * i;
      rw[i] = (kin && n < N && (lr + RPP * i) < BN) ? ld4(a.W + (size_t)n * K + k0 + lc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto lds_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LA; ++i) *reinterpret_cast<float4*>(&As[buf][(lr + RPP * i) * LDSW + lc]) = ra[i];
#pragma unroll
    for (int i = 0; i < LW; ++i)
      if (lr + RPP * i < BN) *reinterpret_cast<float4*>(&Ws[buf][(lr + RPP * i) * LDSW + lc]) = rw[i];
  };
  auto mma_tile = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < BK; kk += 8) {
      float4 b4[NT], a4[MT];
#pragma unroll
      for (int j = 0; j < NT; ++j) b4[j] = *reinterpret_cast<const float4*>(&Ws[buf][((wn * NT + j) * 32 + li_) * LDSW + kk + lh_ * 4]);
#pragma unroll
      for (int i = 0; i < MT; ++i) a4[i] = *reinterpret_cast<const float4*>(&As[buf][((wm * MT + i) * 32 + li_) * LDSW + kk + lh_ * 4]);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, b4[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, b4[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, b4[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, b4[j].w, acc[i][j], 0, 0, 0);
    }
  };
  gload(0);
  if (NBUF == 2) {
    lds_store(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += BK, buf ^= 1) {
      const bool more = k0 + BK < K;
      if (more) gload(k0 + BK);                        // in flight during this tile's MFMAs
      mma_tile(buf);
      if (more) lds_store(buf ^ 1);                    // the other buffer: its last readers passed the previous barrier
      __syncthreads();
    }
  } else {
    for (int k0 = 0; k0 < K; k0 += BK) {
      __syncthreads();                                 // previous tile fully consumed
      lds_store(0);
      __syncthreads();
      if (k0 + BK < K) gload(k0 + BK);                 // prefetch next tile under the MFMAs
      mma_tile(0);
    }
  }
  // epilogue: C/D layout of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int li = li_, lh = lh_;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + (wn * NT + j) * 32 + li;
    if (n >= N) continue;
    const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M) {
          if (a.tm_c > 0) {   // transposed-conv trimming: rows of the full output outside [tm_lo, tm_hi) are not stored
            const long u = ((long)m * N + n) / a.tm_c;
            if (u < a.tm_lo || u >= a.tm_hi) continue;
          }
          float v = act_fn(acc[mt][j][r] + bias, a.act);
          float* c = a.C + (size_t)m * a.ldc + n;
          if (a.residual) v += *c;
          if (a.R) v += a.R[(size_t)m * a.ldr + n];
          if (a.rbias) v += a.rbias[(size_t)a.rclass[m / a.rrep] * N + n];
          if (a.act_out == SSRHIP_ACT_ELU) v = elu1(v);          // the consumer's ELU-on-load, done once here
          *c = v;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const float* w, const float* b, float eps, float* y, int D) {
  __shared__ float red[8];
  const int r = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* xr = x + (size_t)r * D;
  float s = 0.f;
  for (int k = t * 4; k < D; k += 1024) { const float4 v = ld4(xr + k); s += (v.x + v.y) + (v.z + v.w); }
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)D;
  float q = 0.f;
  for (int k = t * 4; k < D; k += 1024) {
    const float4 v = ld4(xr + k);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
  q = wave_sum(q);
  if (lane == 0) red[4 + wave] = q;
  __syncthreads();
  const float var = ((red[4] + red[5]) + (red[6] + red[7])) / (float)D;
  const float rstd = 1.0f / sqrtf(var + eps);
  for (int k = t * 4; k < D; k += 1024) {
    float4 v = ld4(xr + k);
    const float4 ww = ld4(w + k), bb = ld4(b + k);
    v.x = (v.x - mean) * rstd * ww.x + bb.x;
    v.y = (v.y - mean) * rstd * ww.y + bb.y;
    v.z = (v.z - mean) * rstd * ww.z + bb.z;
    v.w = (v.w - mean) * rstd * ww.w + bb.w;
    *reinterpret_cast<float4*>(y + (size_t)r * D + k) = v;
  }
}

__global__ __launch_bounds__(256) void kv_scatter_kernel(const float* qkv, const ssrhip_kv kv, int layer, const int* row_seq,
                                                         const int* row_pos, int D) {
  const int r = blockIdx.x;
  const int seq = row_seq ? row_seq[r] : r;
  const int pos = row_pos[r];
  const int hd = kv.head_dim;
  for (int c = threadIdx.x * 4; c < 2 * D; c += 1024) {
    const int which = c / D, cc = c % D;
    const float4 v = ld4(qkv + (size_t)r * 3 * D + D + c);
    float* dst = kv_addr(kv, seq, layer, which, cc / hd, pos) + (cc % hd);
    *reinterpret_cast<float4*>(dst) = v;
  }
}

}  // namespace

extern "C" int ssrhip_gemm(const ssrhip_gemm_args* a, ssrhip_stream_t stream) {
  SSR_REQUIRE(a && a->A && a->W && a->C, "ssrhip_gemm: null argument");
  SSR_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0 && a->K % 4 == 0 && a->lda % 4 == 0, "ssrhip_gemm: K and lda must be multiples of 4");
  SSR_REQUIRE(!a->rbias || (a->rclass && a->rrep > 0), "ssrhip_gemm: rbias needs rclass and rrep > 0");
  if (ssrhip_gemm_split_eligible(a)) return ssrhip_gemm_split_launch(a, (hipStream_t)stream);    // caller supplied bf16 weight planes
  int bm = a->N <= 64 ? 256 : 64, bn = a->N <= 32 ? 32 : (a->N <= 64 ? 64 : 128);
  const long wide_wgs = (long)((a->N + 127) / 128) * ((a->M + 63) / 64) * (a->batch > 1 ? a->batch : 1);
  const bool small_grid = a->N > 64 && wide_wgs < 256;
  if (small_grid) bn = 64;
  SSR_REQUIRE(a->batch <= 65535 && (a->M + bm - 1) / bm <= 65535, "ssrhip_gemm: grid too large (M=%d batch=%d)", a->M, a->batch);
  dim3 grid((a->N + bn - 1) / bn, (a->M + bm - 1) / bm, a->batch > 1 ? a->batch : 1);
  ssr_gemm_log(a, grid.x, grid.y, grid.z, 0);
  // Experiment knob (off): 128 x 128 tiles (each wave 64 x 64: twice the FLOPs per operand byte staged through LDS). Measured
  // SLOWER on the codec (32 clips x 30 s: encode 86.6 -> 96.7 ms, decode 88.5 -> 98.2 ms): one workgroup per CU no longer
  // hides the two barriers per k-tile.
  static const int big_tiles = getenv("SSRHIP_GEMM_BIG") ? atoi(getenv("SSRHIP_GEMM_BIG")) : 0;
  const long big_wgs = (long)((a->N + 127) / 128) * ((a->M + 127) / 128) * (a->batch > 1 ? a->batch : 1);
  if (big_tiles && a->N >= 128 && big_wgs >= 512) {
    dim3 g2((a->N + 127) / 128, (a->M + 127) / 128, a->batch > 1 ? a->batch : 1);
    SSR_REQUIRE(g2.y <= 65535, "ssrhip_gemm: grid too large (M=%d)", a->M);
    hipLaunchKernelGGL((gemm_kernel<2, 2, 2, 2>), g2, dim3(256), 0, (hipStream_t)stream, *a);
    SSR_LAUNCH_CHECK();
    return 0;
  }
  if (small_grid) hipLaunchKernelGGL((gemm_kernel<2, 2, 1, 1>), grid, dim3(256), 0, (hipStream_t)stream, *a);
  else if (bn == 32) hipLaunchKernelGGL((gemm_kernel<4, 1, 2, 1>), grid, dim3(256), 0, (hipStream_t)stream, *a);
  else if (bn == 64) hipLaunchKernelGGL((gemm_kernel<4, 1, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream, *a);
  else hipLaunchKernelGGL((gemm_kernel<1, 4, 2, 1>), grid, dim3(256), 0, (hipStream_t)stream, *a);
  SSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int ssrhip_layernorm(const float* x, const float* w, const float* b, float eps, float* y, int32_t R, int32_t D,
                                ssrhip_stream_t stream) {
  SSR_REQUIRE(x && w && b && y && R > 0 && D > 0 && D % 4 == 0, "ssrhip_layernorm: bad argument");
  hipLaunchKernelGGL(layernorm_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, x, w, b, eps, y, D);
  SSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int ssrhip_kv_scatter(const float* qkv, const ssrhip_kv* kv, int32_t layer, const int32_t* row_seq,
                                 const int32_t* row_pos, int32_t R, ssrhip_stream_t stream) {
  SSR_REQUIRE(qkv && kv && kv->pool && kv->table && row_pos && R > 0, "ssrhip_kv_scatter: bad argument");
  SSR_REQUIRE(kv->head_dim % 4 == 0, "ssrhip_kv_scatter: head_dim must be a multiple of 4");
  hipLaunchKernelGGL(kv_scatter_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, qkv, *kv, layer, row_seq, row_pos,
                     kv->n_head * kv->head_dim);
  SSR_LAUNCH_CHECK();
  return 0;
}
