// tools/gemm_bench.hip — TFLOP/s of ssrhip_gemm (fp32 MFMA) on a square problem and on the shapes the product runs
// (prefill rows, codec convolutions as strided-view GEMMs, LSTM input GEMM). Random data; every shape is run for ~40 ms before it is
// timed (clocks settle: a 5-launch measurement right after idle reads 10-15 % low — round 2 quoted 100 TFLOP/s on 4096^3 that way).
// env: GEMM_BENCH_ZERO=1 zero-filled operands, GEMM_BENCH_ONLY=<i> one shape, GEMM_BENCH_REPS=<n>.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/ssrhip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((float)(h & 0xFFFF) / 32768.0f - 1.0f) * scale;
  }
}

int main() {
  struct Shape { const char* name; int M, N, K, batch, act_in; } shapes[] = {
      {"square 4096^3", 4096, 4096, 4096, 1, 0},
      {"prefill qkv 598x6144x2048", 598, 6144, 2048, 1, 0},
      {"prefill ffn2 598x2048x8192", 598, 2048, 8192, 1, 0},
      {"lstm-in 1500x4096x1024 x32", 1500, 4096, 1024, 32, 0},
      {"down2 60000x256x1024 x32", 60000, 256, 1024, 32, 1},
      {"convtr 60000x512x512 x32", 60000, 512, 512, 32, 1},
      {"res128.k3 240000x64x384 x32", 240000, 64, 384, 32, 1},
      {"down1 240000x128x256 x32", 240000, 128, 256, 32, 1},
      {"down3 12000x512x2560 x32", 12000, 512, 2560, 32, 1},
      {"down4 1500x1024x8192 x32", 1500, 1024, 8192, 32, 1},
      {"convtr8 1501x4096x2048 x32", 1501, 4096, 2048, 32, 1},
      {"convtr5 12001x1280x1024 x32", 12001, 1280, 1024, 32, 1},
      {"res512.k3 12000x256x1536 x32", 12000, 256, 1536, 32, 1},
      {"res512.k1 12000x512x256 x32", 12000, 512, 256, 32, 1},
      {"first7 1500x1024x896 x32", 1500, 1024, 896, 32, 0},
  };
  const size_t cap = (size_t)32 * 240000 * 384;   // floats
  float *A, *W, *Cm, *bias;
  CK(hipMalloc(&A, cap * 4)); CK(hipMalloc(&W, (size_t)8192 * 8192 * 4)); CK(hipMalloc(&Cm, cap * 4)); CK(hipMalloc(&bias, 8192 * 4));
  const float zs = getenv("GEMM_BENCH_ZERO") ? 0.f : 1.f;      // zero-filled operands: how much of the gap to peak is clock / power
  hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, A, cap, 1u, 0.5f * zs);
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, W, (size_t)8192 * 8192, 2u, 0.02f * zs);
  hipLaunchKernelGGL(fill_kernel, dim3(8), dim3(256), 0, 0, bias, (size_t)8192, 3u, 0.1f);
  CK(hipDeviceSynchronize());
  hipStream_t s; CK(hipStreamCreate(&s));
  const int only = getenv("GEMM_BENCH_ONLY") ? atoi(getenv("GEMM_BENCH_ONLY")) : -1;   // index of the one shape to run (profiling)
  int idx = -1;
  for (auto& sh : shapes) {
    if (++idx != only && only >= 0) continue;
    ssrhip_gemm_args a; memset(&a, 0, sizeof(a));
    a.A = A; a.W = W; a.bias = bias; a.C = Cm; a.M = sh.M; a.N = sh.N; a.K = sh.K; a.lda = sh.K; a.ldc = sh.N; a.act_in = sh.act_in ? SSRHIP_ACT_ELU : 0;
    a.batch = sh.batch; a.strideA = (int64_t)sh.M * sh.K; a.strideC = (int64_t)sh.M * sh.N;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (ssrhip_gemm(&a, s)) { printf("err: %s\n", ssrhip_last_error()); return 1; }
    {   // warm-up until ~40 ms of this shape have run: the clocks need a few milliseconds of load to settle (a 5-launch measurement
        // right after idle reads 10-15 % low)
      hipEvent_t w0, w1; CK(hipEventCreate(&w0)); CK(hipEventCreate(&w1));
      float wms = 0.f;
      for (int round = 0; round < 50 && wms < 40.f; ++round) {
        CK(hipEventRecord(w0, s));
        for (int i = 0; i < 4; ++i) ssrhip_gemm(&a, s);
        CK(hipEventRecord(w1, s)); CK(hipEventSynchronize(w1));
        float m; CK(hipEventElapsedTime(&m, w0, w1)); wms += m;
      }
    }
    CK(hipEventRecord(e0, s));
    const int reps = getenv("GEMM_BENCH_REPS") ? atoi(getenv("GEMM_BENCH_REPS")) : 10;
    for (int i = 0; i < reps; ++i) ssrhip_gemm(&a, s);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = 2.0 * sh.M * sh.N * sh.K * sh.batch;
    printf("%-32s %8.3f ms  %6.1f TFLOP/s\n", sh.name, ms / reps, fl * reps / (ms * 1e-3) / 1e12);
  }
  return 0;
}
