// tools/gemv_bench.hip — micro-benchmark of weight-streaming variants for the decode GEMV (K = 2048 rows, B = 2):
// how fast can 67 MB be streamed by a grid of 4-wave workgroups, and what do nt loads / rows in flight / grid size do.
// Cycles through 16 weight buffers (1 GiB total) so neither L2 nor the 256 MiB Infinity Cache helps.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <vector>
#include "../ssr-speech_amd/csrc/common.h"
void ssrhip_set_error(const char*, ...) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT>
__device__ __forceinline__ float4 ldw(const float* p) { return NT ? ld_nt(p) : ld4(p); }

// RF rows in flight per wave; FMA: do the real dot products + wave reductions, else just touch the data
template <int RF, bool NT, bool FMA>
__global__ __launch_bounds__(256) void stream_kernel(const float* W, const float* x, float* y, int N, int groups) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int G = blockIdx.x * 4 + wave;
  float4 xr[2][8];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 8; ++i) xr[b][i] = ld4(x + b * 2048 + (i * 64 + lane) * 4);
  float4 w[RF][8];
  int n[RF];
#pragma unroll
  for (int r = 0; r < RF; ++r) {
    n[r] = G + r * groups;
    if (n[r] < N) {
#pragma unroll
      for (int i = 0; i < 8; ++i) w[r][i] = ldw<NT>(W + (size_t)n[r] * 2048 + (i * 64 + lane) * 4);
    }
  }
  float dummy = 0.f;
  bool more = true;
  while (more) {
    more = false;
#pragma unroll
    for (int r = 0; r < RF; ++r) {
      if (n[r] < N) {
        if (FMA) {
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) { s0 = dot4(w[r][i], xr[0][i], s0); s1 = dot4(w[r][i], xr[1][i], s1); }
          s0 = wave_sum(s0); s1 = wave_sum(s1);
          if (lane == 0) { y[n[r]] = s0; y[N + n[r]] = s1; }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) dummy += w[r][i].x + w[r][i].w;
        }
        n[r] += RF * groups;
        if (n[r] < N) {
          more = true;
#pragma unroll
          for (int i = 0; i < 8; ++i) w[r][i] = ldw<NT>(W + (size_t)n[r] * 2048 + (i * 64 + lane) * 4);
        }
      }
    }
  }
  if (!FMA && dummy == 123.456f) y[0] = dummy;
}

template <int RF, bool NT, bool FMA>
float run(const std::vector<float*>& Ws, const float* x, float* y, int N, int blocks, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 8; ++i) hipLaunchKernelGGL((stream_kernel<RF, NT, FMA>), dim3(blocks), dim3(256), 0, 0, Ws[i % Ws.size()], x, y, N, blocks * 4);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<RF, NT, FMA>), dim3(blocks), dim3(256), 0, 0, Ws[i % Ws.size()], x, y, N, blocks * 4);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return 1000.f * ms / reps;
}

int main() {
  const int N = 8192, K = 2048, NB = 16;
  std::vector<float*> Ws(NB);
  for (auto& p : Ws) { CK(hipMalloc(&p, (size_t)N * K * 4)); CK(hipMemset(p, 0, (size_t)N * K * 4)); }
  float *x, *y; CK(hipMalloc(&x, 2 * K * 4)); CK(hipMemset(x, 0, 2 * K * 4)); CK(hipMalloc(&y, 2 * N * 4));
  const double mb = (double)N * K * 4 / 1e6;
  printf("N=%d K=%d: %.1f MB per launch; us per launch (back-to-back launches on one stream) and TB/s\n", N, K, mb);
  const int grids[] = {256, 512, 768, 1024, 1536, 2048};
  for (int g : grids) {
    float a = run<2, true, false>(Ws, x, y, N, g, 200), b = run<2, true, true>(Ws, x, y, N, g, 200), c = run<2, false, true>(Ws, x, y, N, g, 200);
    float d = run<4, true, true>(Ws, x, y, N, g, 200), e = run<1, true, true>(Ws, x, y, N, g, 200), f = run<3, true, true>(Ws, x, y, N, g, 200);
    printf("blocks %4d | touch rf2 nt %6.2f (%.2f) | fma rf2 nt %6.2f (%.2f) | fma rf2 plain %6.2f (%.2f) | fma rf4 nt %6.2f (%.2f) | fma rf1 nt %6.2f (%.2f) | fma rf3 nt %6.2f (%.2f)\n",
           g, a, mb / a, b, mb / b, c, mb / c, d, mb / d, e, mb / e, f, mb / f);
  }
  // Infinity-Cache residency: the same 67 MB buffer every launch (fits the 256 MiB MALL) vs 16 different buffers
  {
    std::vector<float*> one(1, Ws[0]), two(2);
    two[0] = Ws[0]; two[1] = Ws[1];
    for (int g : {512, 768}) {
      float a = run<2, true, true>(one, x, y, N, g, 200), b = run<2, false, true>(one, x, y, N, g, 200), c = run<2, true, true>(two, x, y, N, g, 200), d = run<2, true, false>(one, x, y, N, g, 200);
      printf("MALL-resident, blocks %d: fma rf2 nt 1 buffer %6.2f (%.2f TB/s) | plain loads %6.2f (%.2f) | 2 buffers (134 MB) nt %6.2f (%.2f) | touch-only %6.2f (%.2f)\n", g, a, mb / a, b, mb / b, c, mb / c, d, mb / d);
    }
  }
  return 0;
}
