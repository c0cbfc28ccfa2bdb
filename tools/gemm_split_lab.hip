// tools/gemm_split_lab.hip — EXPERIMENT (round 3): fp32-accurate GEMM on the bf16 matrix cores ("bf16x3": every fp32 operand is split
// exactly into three bf16 pieces a = a0 + a1 + a2, and C accumulates the six largest of the nine cross products in fp32:
// a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0; the three dropped ones are <= 2^-24 relative, the size of one fp32 rounding).
// v_mfma_f32_32x32x16_bf16 is 16x the rate of v_mfma_f32_32x32x2_f32, so six of them per 16 k-values cost 192 matrix-pipe cycles
// against 512 for the exact fp32 chain. Measured against the library's ssrhip_gemm (exact fp32 FMA chain) for speed and against an
// fp64 reference for the error of both.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -Issr-speech_amd/csrc tools/gemm_split_lab.hip \
//          -o tools/bin/gemm_split_lab -Lssr-speech_amd/csrc -lssrhip -Wl,-rpath,'$ORIGIN/../../ssr-speech_amd/csrc'
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "common.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
void ssrhip_set_error(const char*, ...) {}

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_fn(float v, int act) {
  if (act == SSRHIP_ACT_RELU) return fmaxf(v, 0.f);
  if (act == SSRHIP_ACT_GELU_ERF) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  return v;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bfx2 __attribute__((ext_vector_type(2)));
// exact three-way split of 4 consecutive k-values (v_cvt_pk_bf16_f32 = round to nearest even, two elements per instruction;
// bf16 -> fp32 is a 16-bit shift; the residuals a - hi and (a - hi) - mid are exact in fp32): piece p of element e in out[p][e]
__device__ __forceinline__ void split4(const float4 v, bf16x4 (&out)[3]) {
  f32x2 r[2] = {{v.x, v.y}, {v.z, v.w}};
#pragma unroll
  for (int p = 0; p < 3; ++p) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bfx2 b = __builtin_convertvector(r[h], bfx2);
      const unsigned bits = __builtin_bit_cast(unsigned, b);
      out[p][2 * h] = (short)(bits & 0xFFFFu);
      out[p][2 * h + 1] = (short)(bits >> 16);
      if (p < 2) {
        const f32x2 back = {__builtin_bit_cast(float, bits << 16), __builtin_bit_cast(float, bits & 0xFFFF0000u)};
        r[h] = r[h] - back;
      }
    }
  }
}

// one-time weight preparation: W fp32 [N][K] -> three bf16 planes [3][N][K]
__global__ void split_weights_kernel(const float* __restrict__ W, short* __restrict__ out, size_t n_elems) {
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n_elems; i += (size_t)gridDim.x * blockDim.x * 4) {
    bf16x4 p[3];
    split4(ld4(W + i), p);
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(out + (size_t)q * n_elems + i) = p[q];
  }
}

// Block tile 128 x 128 x 32, 4 waves (2 x 2), each 64 x 64 = 2 x 2 accumulators of 32 x 32. LDS per piece: rows of 32 bf16 (64 B)
// padded to 80 B: a lane's ds_read_b128 (8 consecutive k of one row) then hits 16 distinct 16-byte slots per 16-lane group.
constexpr int BM = 128, BN = 128, BK = 32, PITCH = 40;              // pitch in bf16 elements (80 B)
template <bool ELU>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const ssrhip_gemm_args a0, const short* __restrict__ Wp, const size_t plane) {
  __shared__ __attribute__((aligned(16))) short As[3][BM * PITCH];
  __shared__ __attribute__((aligned(16))) short Ws[3][BN * PITCH];
  ssrhip_gemm_args a = a0;
  {
    const size_t z = blockIdx.z;
    a.A += z * (size_t)a.strideA;
    a.C += z * (size_t)a.strideC;
    if (a.R) a.R += z * (size_t)a.strideR;
  }
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int lr = t >> 3, lc = (t & 7) * 4;                          // loader: 8 threads per row (32 k), 32 rows per pass
  const int M = a.M, N = a.N, K = a.K;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 ra[4];
  bf16x8 rwp[3][2];                                                 // W: 2 chunks of 8 k per piece (row lw + 64 * i, k-chunk cw)
  const int lw = t >> 2, cw = (t & 3) * 8;
  auto gload = [&](int k0) {
    const bool kin = (k0 + lc) < K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + lr + 32 * i;
      ra[i] = (kin && m < M) ? ld4(a.A + (size_t)m * a.lda + k0 + lc) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (ELU) { ra[i].x = elu1(ra[i].x); ra[i].y = elu1(ra[i].y); ra[i].z = elu1(ra[i].z); ra[i].w = elu1(ra[i].w); }
    }
    const bool kinw = (k0 + cw) < K;                                 // K % 8 == 0 required for the pre-split planes
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n = n0 + lw + 64 * i;
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        rwp[q][i] = (kinw && n < N) ? *reinterpret_cast<const bf16x8*>(Wp + (size_t)q * plane + (size_t)n * K + k0 + cw) : z;
      }
  };
  auto lds_store = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16x4 p[3];
      split4(ra[i], p);
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(&As[q][(lr + 32 * i) * PITCH + lc]) = p[q];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<bf16x8*>(&Ws[q][(lw + 64 * i) * PITCH + cw]) = rwp[q][i];
  };
  auto mma_tile = [&]() {
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      bf16x8 fa[3][2], fb[3][2];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[q][i] = *reinterpret_cast<const bf16x8*>(&As[q][((wm * 2 + i) * 32 + li) * PITCH + kk + lh * 8]);
          fb[q][i] = *reinterpret_cast<const bf16x8*>(&Ws[q][((wn * 2 + i) * 32 + li) * PITCH + kk + lh * 8]);
        }
      // smallest terms first
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int pq = 0; pq < 6; ++pq)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pq]][i], fb[PB[pq]][j], acc[i][j], 0, 0, 0);
    }
  };
  gload(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
    lds_store();
    __syncthreads();
    if (k0 + BK < K) gload(k0 + BK);
    mma_tile();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + (wn * 2 + j) * 32 + li;
    if (n >= N) continue;
    const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * 2 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M) {
          float v = act_fn(acc[mt][j][r] + bias, a.act);
          float* c = a.C + (size_t)m * a.ldc + n;
          if (a.residual) v += *c;
          if (a.R) v += a.R[(size_t)m * a.ldr + n];
          *c = v;
        }
      }
    }
  }
}

short* g_wp = nullptr;
int launch_split(const ssrhip_gemm_args* a, hipStream_t s) {
  dim3 grid((a->N + BN - 1) / BN, (a->M + BM - 1) / BM, a->batch > 1 ? a->batch : 1);
  const size_t plane = (size_t)a->N * a->K;
  if (a->act_in == SSRHIP_ACT_ELU) hipLaunchKernelGGL(gemm_split_kernel<true>, grid, dim3(256), 0, s, *a, g_wp, plane);
  else hipLaunchKernelGGL(gemm_split_kernel<false>, grid, dim3(256), 0, s, *a, g_wp, plane);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) * scale;      // 24 random mantissa bits: every split piece is exercised
  }
}
}  // namespace

int main() {
  struct Shape { const char* name; int M, N, K, batch, act_in; } shapes[] = {
      {"square 4096^3", 4096, 4096, 4096, 1, 0},       {"prefill qkv 598x6144x2048", 598, 6144, 2048, 1, 0},
      {"lstm-in 1500x4096x1024 x32", 1500, 4096, 1024, 32, 0}, {"down2 60000x256x1024 x32", 60000, 256, 1024, 32, 1},
      {"down1 240000x128x256 x32", 240000, 128, 256, 32, 1},   {"down4 1500x1024x8192 x32", 1500, 1024, 8192, 32, 1},
      {"convtr 60000x512x512 x32", 60000, 512, 512, 32, 1},
  };
  const size_t cap = (size_t)32 * 240000 * 256;
  float *A, *W, *C0, *C1, *bias;
  CK(hipMalloc(&A, cap * 4)); CK(hipMalloc(&W, (size_t)8192 * 8192 * 4)); CK(hipMalloc(&C0, cap * 4)); CK(hipMalloc(&C1, cap * 4));
  CK(hipMalloc(&bias, 8192 * 4));
  CK(hipMalloc(&g_wp, (size_t)3 * 8192 * 8192 * 2));
  hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, A, cap, 1u, 0.5f);
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, W, (size_t)8192 * 8192, 2u, 0.02f);
  hipLaunchKernelGGL(fill_kernel, dim3(8), dim3(256), 0, 0, bias, (size_t)8192, 3u, 0.1f);
  CK(hipDeviceSynchronize());
  hipStream_t s; CK(hipStreamCreate(&s));
  {   // error of both kernels against fp64 on a 256 x 256 x 4096 problem (no ELU)
    const int M = 256, N = 256, K = 4096;
    ssrhip_gemm_args a; memset(&a, 0, sizeof(a));
    a.A = A; a.W = W; a.bias = bias; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = N;
    a.C = C0; ssrhip_gemm(&a, s);
    hipLaunchKernelGGL(split_weights_kernel, dim3(2048), dim3(256), 0, s, W, g_wp, (size_t)N * K);
    a.C = C1; launch_split(&a, s);
    CK(hipStreamSynchronize(s));
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N), h0((size_t)M * N), h1((size_t)M * N);
    CK(hipMemcpy(hA.data(), A, hA.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hW.data(), W, hW.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), bias, N * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
    double e0 = 0, e1 = 0, s0 = 0, s1 = 0, scale = 0;
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) {
        double ref = hb[n], mag = 0;
        for (int k = 0; k < K; ++k) { const double p = (double)hA[(size_t)m * K + k] * hW[(size_t)n * K + k]; ref += p; mag += fabs(p); }
        const double d0 = fabs(h0[(size_t)m * N + n] - ref), d1 = fabs(h1[(size_t)m * N + n] - ref);
        e0 = fmax(e0, d0 / mag); e1 = fmax(e1, d1 / mag); s0 += d0 / mag; s1 += d1 / mag; scale = fmax(scale, mag);
      }
    printf("error vs fp64 (256x256x4096, relative to sum|a.b|): exact fp32 MFMA chain max %.3e mean %.3e | bf16x3 six products max %.3e mean %.3e\n",
           e0, s0 / (M * N), e1, s1 / (M * N));
  }
  for (auto& sh : shapes) {
    ssrhip_gemm_args a; memset(&a, 0, sizeof(a));
    a.A = A; a.W = W; a.bias = bias; a.M = sh.M; a.N = sh.N; a.K = sh.K; a.lda = sh.K; a.ldc = sh.N; a.act_in = sh.act_in ? SSRHIP_ACT_ELU : 0;
    a.batch = sh.batch; a.strideA = (int64_t)sh.M * sh.K; a.strideC = (int64_t)sh.M * sh.N;
    double tf[2];
    hipLaunchKernelGGL(split_weights_kernel, dim3(2048), dim3(256), 0, s, W, g_wp, (size_t)sh.N * sh.K);
    for (int which = 0; which < 2; ++which) {
      a.C = which ? C1 : C0;
      auto run = [&]() { return which ? launch_split(&a, s) : ssrhip_gemm(&a, s); };
      if (run()) { printf("launch failed\n"); return 1; }
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float wms = 0.f;
      for (int round = 0; round < 50 && wms < 40.f; ++round) {
        CK(hipEventRecord(e0, s)); for (int i = 0; i < 4; ++i) run(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float m; CK(hipEventElapsedTime(&m, e0, e1)); wms += m;
      }
      CK(hipEventRecord(e0, s)); for (int i = 0; i < 10; ++i) run(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      tf[which] = 2.0 * sh.M * sh.N * sh.K * sh.batch * 10 / (ms * 1e-3) / 1e12;
    }
    printf("%-30s exact fp32 %6.1f TFLOP/s   bf16x3 %6.1f TFLOP/s (fp32-equivalent)   x%.2f\n", sh.name, tf[0], tf[1], tf[1] / tf[0]);
  }
  return 0;
}
