// gemv.hip — weight-streaming fused GEMV for the AR decode step (B <= 4 rows), gfx950.
//
//   y[b][n] = epi( sum_k pro(x)[b][k] * W[n][k] + bias[n] )
//
// Roofline: HBM-bound. Every weight byte is read exactly once per step (2*B FLOP per 4 bytes),
// so the kernel is organised around keeping 16-byte non-temporal loads in flight:
//   * a wave owns whole weight rows (or one <=2048-float slice of them when K > 2048); its slice of
//     x (<= 8 float4 per lane per batch row) lives in VGPRs for the whole kernel;
//   * a row is 8 x global_load_dwordx4 per lane (1 KiB per wave-instruction, fully coalesced), two
//     rows are issued back to back before the FMAs so 16 KiB per wave are in flight;
//   * no LDS on the weight path (each weight element is used once: staging would be pure overhead);
//     LDS is used only to share the prologue (LayerNorm / split-KV combine) between the 4 waves and
//     to add the K-slices of one row.
// Replaces F.linear (+LayerNorm / ReLU / GELU / residual) of the reference: see include/ssrhip.h.
#include "common.h"

namespace {

struct GemvK {
  ssrhip_gemv_args a;
  int nslice;     // waves cooperating on one row (K split), 1|2|4
  int slice_len;  // floats per slice (multiple of 4)
  int nch;        // float4 chunks per lane per slice (<= 8)
  int rpw;        // rows per wave-group
  int hd;         // head_dim (QKV epilogue / combine prologue)
};

constexpr int MAXCH = 8;
constexpr int MAX_RPW = 8;

template <int B>
__device__ __forceinline__ void stage_layernorm(const GemvK& p, int g, float* xs, float* red) {
  // LayerNorm of B rows of length K into xs[B][K]; biased variance, eps inside the sqrt
  // (F.layer_norm, models/modules/transformer.py:58-75). Two-pass (mean, then centered squares).
  const int K = p.a.K;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  float s[B];
#pragma unroll
  for (int b = 0; b < B; ++b) {
    s[b] = 0.f;
    const float* xb = p.a.x + (size_t)b * p.a.x_stride + (size_t)g * K;
    for (int k = t * 4; k < K; k += 1024) {
      float4 v = ld4(xb + k);
      *reinterpret_cast<float4*>(xs + b * K + k) = v;
      s[b] += (v.x + v.y) + (v.z + v.w);
    }
    s[b] = wave_sum(s[b]);
    if (lane == 0) red[b * 4 + wave] = s[b];
  }
  __syncthreads();
  float mean[B];
#pragma unroll
  for (int b = 0; b < B; ++b) mean[b] = ((red[b * 4 + 0] + red[b * 4 + 1]) + (red[b * 4 + 2] + red[b * 4 + 3])) / (float)K;
  __syncthreads();
#pragma unroll
  for (int b = 0; b < B; ++b) {
    float q = 0.f;
    for (int k = t * 4; k < K; k += 1024) {
      float4 v = *reinterpret_cast<float4*>(xs + b * K + k);
      float dx = v.x - mean[b], dy = v.y - mean[b], dz = v.z - mean[b], dw = v.w - mean[b];
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    q = wave_sum(q);
    if (lane == 0) red[b * 4 + wave] = q;
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < B; ++b) {
    const float var = ((red[b * 4 + 0] + red[b * 4 + 1]) + (red[b * 4 + 2] + red[b * 4 + 3])) / (float)K;
    const float rstd = 1.0f / sqrtf(var + p.a.ln_eps);
    for (int k = t * 4; k < K; k += 1024) {
      float4 v = *reinterpret_cast<float4*>(xs + b * K + k);
      const float4 w = ld4(p.a.ln_w + k), bb = ld4(p.a.ln_b + k);
      v.x = (v.x - mean[b]) * rstd * w.x + bb.x;
      v.y = (v.y - mean[b]) * rstd * w.y + bb.y;
      v.z = (v.z - mean[b]) * rstd * w.z + bb.z;
      v.w = (v.w - mean[b]) * rstd * w.w + bb.w;
      *reinterpret_cast<float4*>(xs + b * K + k) = v;
    }
  }
  __syncthreads();
}

template <int B>
__device__ __forceinline__ void stage_attn_combine(const GemvK& p, float* xs) {
  // Merge the per-page partials of ssrhip_attn_decode: out = sum_s e^{m_s-M} o_s / sum_s e^{m_s-M} l_s
  const int K = p.a.K, hd = p.hd, H = K / hd, MS = p.a.max_splits;
  for (int e = threadIdx.x * 4; e < K; e += 1024) {
    const int h = e / hd, d = e % hd;
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const int ns = (p.a.row_len[b] + SSRHIP_PAGE - 1) / SSRHIP_PAGE;
      const float* ml = p.a.part_ml + ((size_t)b * H + h) * MS * 2;
      const float* po = p.a.part_o + (((size_t)b * H + h) * MS) * hd + d;
      float M = -INFINITY;
      for (int s = 0; s < ns; ++s) M = fmaxf(M, ml[2 * s]);
      float den = 0.f;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s = 0; s < ns; ++s) {
        const float w = expf(ml[2 * s] - M);
        den = fmaf(w, ml[2 * s + 1], den);
        const float4 o = ld4(po + (size_t)s * hd);
        acc.x = fmaf(w, o.x, acc.x);
        acc.y = fmaf(w, o.y, acc.y);
        acc.z = fmaf(w, o.z, acc.z);
        acc.w = fmaf(w, o.w, acc.w);
      }
      const float inv = 1.0f / den;
      *reinterpret_cast<float4*>(xs + b * K + e) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
  }
  __syncthreads();
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == SSRHIP_ACT_RELU) return fmaxf(v, 0.f);
  if (act == SSRHIP_ACT_GELU_ERF) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  return v;
}

__device__ __forceinline__ void finalize(const GemvK& p, int g, int n, int b, float v) {
  const ssrhip_gemv_args& a = p.a;
  if (a.bias) v += a.bias[(size_t)g * a.N + n];
  v = apply_act(v, a.act);
  if (a.epi == SSRHIP_EPI_STORE) {
    a.y[(size_t)b * a.y_stride + (size_t)g * a.N + n] = v;
  } else if (a.epi == SSRHIP_EPI_RESIDUAL) {
    float* y = a.y + (size_t)b * a.y_stride + (size_t)g * a.N + n;
    *y = *y + v;
  } else {  // QKV append: rows [0,D) -> q, [D,2D) -> k cache, [2D,3D) -> v cache
    const int D = a.K;
    const int which = n / D, c = n % D;
    if (which == 0) {
      a.y[(size_t)b * a.y_stride + c] = v;
    } else {
      float* dst = kv_addr(a.kv, b, a.layer, which - 1, c / p.hd, a.kv_pos[b]);
      dst[c % p.hd] = v;
    }
  }
}

template <int B, bool FULL>
__device__ __forceinline__ void load_row(float4 (&w)[MAXCH], const float* wrow, int lane, int nch, int len) {
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    if (i < nch) {
      const int k = (i * 64 + lane) * 4;
      if (FULL || k < len) w[i] = ld_nt(wrow + k);
      else w[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

template <int B>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvK p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const ssrhip_gemv_args& a = p.a;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.y;
  const int K = a.K, N = a.N;
  const int slice = wave % p.nslice, rg = wave / p.nslice, n_rg = 4 / p.nslice;
  const int k0 = slice * p.slice_len;
  const int len = min(p.slice_len, K - k0);   // floats in this wave's slice (may be <= 0 for tiny K)
  const bool full = (len == p.nch * 256);

  // ---- prologue: this wave's slice of x into registers
  float4 xr[B][MAXCH];
  float* part = smem;                          // [4 waves][MAX_RPW][B] cross-slice partials
  float* xs = smem + 4 * MAX_RPW * B + 16;     // staged x (LayerNorm / combine prologues)
  if (a.pro == SSRHIP_PRO_LAYERNORM) {
    stage_layernorm<B>(p, g, xs, smem);        // `red` aliases `part`: not live yet
  } else if (a.pro == SSRHIP_PRO_ATTN_COMBINE) {
    stage_attn_combine<B>(p, xs);
  }
#pragma unroll
  for (int b = 0; b < B; ++b) {
    const float* xb = (a.pro == SSRHIP_PRO_NONE) ? (a.x + (size_t)b * a.x_stride + (size_t)g * K) : (xs + b * K);
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int k = (i * 64 + lane) * 4;
      xr[b][i] = (i < p.nch && k < len) ? ld4(xb + k0 + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }

  // ---- main loop: rows [n0, n1) of this wave-group, two rows in flight
  const int G = blockIdx.x * n_rg + rg;
  const int n0 = G * p.rpw, n1 = min(N, n0 + p.rpw);
  const float* Wg = a.W + (size_t)g * N * K + k0;
  for (int n = n0; n < n1; n += 2) {
    const bool two = (n + 1 < n1);
    float4 w0[MAXCH], w1[MAXCH];
    if (full) {
      load_row<B, true>(w0, Wg + (size_t)n * K, lane, p.nch, len);
      if (two) load_row<B, true>(w1, Wg + (size_t)(n + 1) * K, lane, p.nch, len);
    } else {
      load_row<B, false>(w0, Wg + (size_t)n * K, lane, p.nch, len);
      if (two) load_row<B, false>(w1, Wg + (size_t)(n + 1) * K, lane, p.nch, len);
    }
    float acc[2][B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int i = 0; i < MAXCH; ++i) {
        if (i < p.nch) {
          s0 = dot4(w0[i], xr[b][i], s0);
          if (two) s1 = dot4(w1[i], xr[b][i], s1);
        }
      }
      acc[0][b] = wave_sum(s0);
      acc[1][b] = two ? wave_sum(s1) : 0.f;
    }
    // lanes 0..2B-1 each own one (row, b) result
    float mine = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int b = 0; b < B; ++b)
        if (lane == r * B + b) mine = acc[r][b];
    if (lane < 2 * B) {
      const int r = lane / B, b = lane % B;
      if (n + r < n1) {
        if (p.nslice == 1) finalize(p, g, n + r, b, mine);
        else part[(wave * MAX_RPW + (n - n0) + r) * B + b] = mine;
      }
    }
  }
  if (p.nslice > 1) {
    __syncthreads();
    // thread t -> (rg, r, b); sum the slices in a fixed order
    const int t = threadIdx.x;
    if (t < n_rg * p.rpw * B) {
      const int b = t % B, r = (t / B) % p.rpw, rg2 = t / (B * p.rpw);
      const int n = (blockIdx.x * n_rg + rg2) * p.rpw + r;
      if (n < N) {
        float v = 0.f;
        for (int s = 0; s < p.nslice; ++s) v += part[((rg2 * p.nslice + s) * MAX_RPW + r) * B + b];
        finalize(p, g, n, b, v);
      }
    }
  }
}

}  // namespace

extern "C" int ssrhip_gemv(const ssrhip_gemv_args* a, ssrhip_stream_t stream) {
  SSR_REQUIRE(a && a->W && a->y, "ssrhip_gemv: null argument");
  SSR_REQUIRE(a->B == 1 || a->B == 2 || a->B == 4, "ssrhip_gemv: B=%d not in {1,2,4}", a->B);
  SSR_REQUIRE(a->K > 0 && a->K % 4 == 0 && a->K <= 8192, "ssrhip_gemv: K=%d must be a multiple of 4, <= 8192", a->K);
  SSR_REQUIRE(a->N > 0 && a->groups >= 1, "ssrhip_gemv: bad N/groups");
  GemvK p;
  p.a = *a;
  p.nslice = a->K <= 2048 ? 1 : (a->K <= 4096 ? 2 : 4);
  p.slice_len = ((a->K + p.nslice - 1) / p.nslice + 3) / 4 * 4;
  p.nch = (p.slice_len + 255) / 256;
  SSR_REQUIRE(p.nch <= MAXCH, "ssrhip_gemv: slice too long");
  const int n_rg = 4 / p.nslice;
  int rpw = (a->N + 1024 * n_rg - 1) / (1024 * n_rg);
  if (rpw > 1 && (rpw & 1)) ++rpw;
  if (rpw > MAX_RPW) rpw = MAX_RPW;
  p.rpw = rpw;
  p.hd = (a->kv.head_dim > 0) ? a->kv.head_dim : 1;
  size_t smem = (4 * MAX_RPW * a->B + 16) * sizeof(float);
  if (a->pro != SSRHIP_PRO_NONE) {
    SSR_REQUIRE(a->groups == 1 || a->pro == SSRHIP_PRO_LAYERNORM, "ssrhip_gemv: combine prologue needs groups==1");
    SSR_REQUIRE((size_t)a->B * a->K * 4 <= 96 * 1024, "ssrhip_gemv: staged prologue needs B*K*4 <= 96 KiB");
    smem += (size_t)a->B * a->K * sizeof(float);
    if (a->pro == SSRHIP_PRO_LAYERNORM) SSR_REQUIRE(a->ln_w && a->ln_b && a->x, "ssrhip_gemv: LayerNorm prologue needs x, ln_w, ln_b");
    if (a->pro == SSRHIP_PRO_ATTN_COMBINE) {
      SSR_REQUIRE(a->part_o && a->part_ml && a->row_len && a->kv.head_dim > 0 && a->K % a->kv.head_dim == 0 && a->kv.head_dim % 4 == 0,
                  "ssrhip_gemv: combine prologue needs part_o, part_ml, row_len, kv.head_dim");
    }
  } else {
    SSR_REQUIRE(a->x, "ssrhip_gemv: x is null");
  }
  if (a->epi == SSRHIP_EPI_QKV_APPEND) {
    SSR_REQUIRE(a->N == 3 * a->K && a->groups == 1 && a->kv.pool && a->kv.table && a->kv_pos && a->kv.head_dim > 0,
                "ssrhip_gemv: QKV epilogue needs N==3K and a kv cache");
  }
  dim3 grid((a->N + rpw * n_rg - 1) / (rpw * n_rg), a->groups);
  hipStream_t s = (hipStream_t)stream;
  switch (a->B) {
    case 1: hipLaunchKernelGGL(gemv_kernel<1>, grid, dim3(256), smem, s, p); break;
    case 2: hipLaunchKernelGGL(gemv_kernel<2>, grid, dim3(256), smem, s, p); break;
    default: hipLaunchKernelGGL(gemv_kernel<4>, grid, dim3(256), smem, s, p); break;
  }
  SSR_LAUNCH_CHECK();
  return 0;
}
