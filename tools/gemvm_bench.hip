// tools/gemvm_bench.hip — micro-benchmark of the 5..16-row matrix-core GEMV (csrc/gemv_mfma.hip) through the C-ABI on the
// 830M decode-step shapes (LayerNorm prologue on the K = 2048 launches that have one in the product), random data, 16
// rotating weight buffers (no L2 / Infinity-Cache reuse), one hipGraph of 64 launches per shape.
// usage: gemvm_bench [B=16] [tiled=1];  env SSRHIP_GEMVM_V=1|2 (kernel version), SSRHIP_GEMVM_WPC=1..4 (workgroups per CU, v2)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/ssrhip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((float)(h & 0xFFFF) / 32768.0f - 1.0f) * scale;
  }
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 16;
  const int tiled = argc > 2 ? atoi(argv[2]) : 1;
  const int wtiled = argc > 3 ? atoi(argv[3]) : 0;   // weights read in the streaming-order layout (timing only: same random buffer)
  struct Shape { const char* name; int N, K, pro, act, epi, groups; } shapes[] = {
      {"ln1+qkv", 6144, 2048, 1, 0, 0, 1}, {"out_proj", 2048, 2048, 0, 0, 1, 1}, {"ln2+ffn1", 8192, 2048, 1, 1, 0, 1},
      {"ffn2", 2048, 8192, 0, 0, 1, 1}, {"lnf+head1", 4096, 2048, 1, 2, 0, 1}, {"head2", 2056, 1024, 0, 0, 0, 4}};
  const bool nopro = getenv("BENCH_NOPRO") != nullptr, noepi = getenv("BENCH_NOEPI") != nullptr;
  const int NBUF = 16;
  const size_t wsz = (size_t)8224 * 2048;
  float* W; CK(hipMalloc(&W, NBUF * wsz * 4));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, W, NBUF * wsz, 17u, 0.02f);
  float *x, *y, *bias; CK(hipMalloc(&x, 16 * 8192 * 4)); CK(hipMalloc(&y, 16 * 8224 * 4)); CK(hipMalloc(&bias, 8224 * 4));
  hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, x, (size_t)16 * 8192, 3u, 1.0f);
  hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, y, (size_t)16 * 8224, 5u, 1.0f);
  hipLaunchKernelGGL(fill_kernel, dim3(8), dim3(256), 0, 0, bias, (size_t)8224, 7u, 0.1f);
  hipStream_t s; CK(hipStreamCreate(&s));
  CK(hipDeviceSynchronize());
  double tot_us = 0, tot_mb = 0;
  for (auto& sh : shapes) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 64; ++i) {
      ssrhip_gemv_args a; memset(&a, 0, sizeof(a));
      a.W = W + (size_t)(i % NBUF) * wsz; a.x = x; a.y = y; a.bias = bias; a.B = B; a.N = sh.N; a.K = sh.K; a.groups = sh.groups;
      a.x_stride = sh.K * sh.groups; a.y_stride = sh.N * sh.groups; a.x_tiled = tiled; a.y_tiled = tiled; a.w_tiled = wtiled;
      a.pro = nopro ? 0 : sh.pro; a.act = noepi ? 0 : sh.act; a.epi = noepi ? 0 : sh.epi; a.ln_eps = 1e-5f;
      if (noepi) a.bias = nullptr;
      if (ssrhip_gemv(&a, s)) { printf("err: %s\n", ssrhip_last_error()); return 1; }
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ex, s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ex, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / (5 * 64), mb = (double)sh.groups * sh.N * sh.K * 4 / 1e6;
    printf("B=%d tiled=%d wt=%d %-10s N=%d K=%d g=%d: %6.2f us/launch, %.2f TB/s\n", B, tiled, wtiled, sh.name, sh.N, sh.K, sh.groups, us, mb / us);
    const int per_step = (sh.groups == 1 && sh.N != 4096) ? 16 : 1;
    tot_us += us * per_step; tot_mb += mb * per_step;
  }
  printf("one 830M decode step's 66 GEMV launches: %.1f us, %.2f TB/s average (V=%s WPC=%s)\n", tot_us, tot_mb / tot_us,
         getenv("SSRHIP_GEMVM_V") ? getenv("SSRHIP_GEMVM_V") : "2", getenv("SSRHIP_GEMVM_WPC") ? getenv("SSRHIP_GEMVM_WPC") : "1");
  return 0;
}
