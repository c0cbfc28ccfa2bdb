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
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 128, BK = 16, LDSW = BK + 4;

__device__ __forceinline__ float act_fn(float v, int act) {
  if (act == SSRHIP_ACT_RELU) return fmaxf(v, 0.f);
  if (act == SSRHIP_ACT_GELU_ERF) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  return v;
}

__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : expm1f(v); }   // nn.ELU(alpha=1)

__global__ __launch_bounds__(256) void gemm_kernel(const ssrhip_gemm_args a0) {
  __shared__ __attribute__((aligned(16))) float As[BM * LDSW];
  __shared__ __attribute__((aligned(16))) float Ws[BN * LDSW];
  ssrhip_gemm_args a = a0;
  {   // batched problems: grid.z
    const size_t z = blockIdx.z;
    a.A += z * (size_t)a.strideA;
    a.C += z * (size_t)a.strideC;
    if (a.R) a.R += z * (size_t)a.strideR;
  }
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int lr = t >> 2, lc = (t & 3) * 4;            // loader: row, first k column
  const int M = a.M, N = a.N, K = a.K;

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }

  auto gload = [&](int k0, float4& ra, float4& rw0, float4& rw1) {
    const bool kin = (k0 + lc) < K;
    ra = (kin && (m0 + lr) < M) ? ld4(a.A + (size_t)(m0 + lr) * a.lda + k0 + lc) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.act_in == SSRHIP_ACT_ELU) { ra.x = elu1(ra.x); ra.y = elu1(ra.y); ra.z = elu1(ra.z); ra.w = elu1(ra.w); }
    rw0 = (kin && (n0 + lr) < N) ? ld4(a.W + (size_t)(n0 + lr) * K + k0 + lc) : make_float4(0.f, 0.f, 0.f, 0.f);
    rw1 = (kin && (n0 + lr + 64) < N) ? ld4(a.W + (size_t)(n0 + lr + 64) * K + k0 + lc) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  float4 ra, rw0, rw1;
  gload(0, ra, rw0, rw1);
  const int li = lane & 31, lh = lane >> 5;
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();                                   // previous tile fully consumed
    *reinterpret_cast<float4*>(&As[lr * LDSW + lc]) = ra;
    *reinterpret_cast<float4*>(&Ws[lr * LDSW + lc]) = rw0;
    *reinterpret_cast<float4*>(&Ws[(lr + 64) * LDSW + lc]) = rw1;
    __syncthreads();
    if (k0 + BK < K) gload(k0 + BK, ra, rw0, rw1);     // prefetch next tile under the MFMAs
#pragma unroll
    for (int kk = 0; kk < BK; kk += 8) {
      const float4 b4 = *reinterpret_cast<const float4*>(&Ws[(wave * 32 + li) * LDSW + kk + lh * 4]);
      const float4 a0 = *reinterpret_cast<const float4*>(&As[(li)*LDSW + kk + lh * 4]);
      const float4 a1 = *reinterpret_cast<const float4*>(&As[(32 + li) * LDSW + kk + lh * 4]);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b4.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b4.x, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b4.y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b4.y, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b4.z, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b4.z, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b4.w, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b4.w, acc[1], 0, 0, 0);
    }
  }
  // epilogue: C/D layout of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int n = n0 + wave * 32 + li;
  if (n < N) {
    const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M) {
          if (a.tm_c > 0) {   // transposed-conv trimming: rows of the full output outside [tm_lo, tm_hi) are not stored
            const long u = ((long)m * N + n) / a.tm_c;
            if (u < a.tm_lo || u >= a.tm_hi) continue;
          }
          float v = act_fn(acc[mt][r] + bias, a.act);
          float* c = a.C + (size_t)m * a.ldc + n;
          if (a.residual) v += *c;
          if (a.R) v += a.R[(size_t)m * a.ldr + n];
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
  SSR_REQUIRE(a->batch <= 65535 && (a->M + BM - 1) / BM <= 65535, "ssrhip_gemm: grid too large (M=%d batch=%d)", a->M, a->batch);
  dim3 grid((a->N + BN - 1) / BN, (a->M + BM - 1) / BM, a->batch > 1 ? a->batch : 1);
  hipLaunchKernelGGL(gemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, *a);
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
