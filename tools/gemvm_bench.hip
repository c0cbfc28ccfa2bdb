// tools/gemvm_bench.hip — micro-benchmark of the 5..16-row matrix-core GEMV (csrc/gemv_mfma.hip) through the C-ABI:
// PRO_NONE / STORE launches over 16 rotating weight buffers (no L2 / Infinity-Cache reuse), one hipGraph of 64 launches.
// usage: gemvm_bench <libssrhip.so dir is linked>; env SSRHIP_GEMVM_VAR=0..3 selects the experiment variants.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/ssrhip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 16;
  const int tiled = argc > 2 ? atoi(argv[2]) : 1;
  struct Shape { const char* name; int N, K; } shapes[] = {{"qkv", 6144, 2048}, {"out", 2048, 2048}, {"ffn1", 8192, 2048}, {"ffn2", 2048, 8192}};
  const int NBUF = 16;
  float* W; CK(hipMalloc(&W, (size_t)NBUF * 8192 * 2048 * 4));
  CK(hipMemset(W, 0, (size_t)NBUF * 8192 * 2048 * 4));
  float *x, *y; CK(hipMalloc(&x, 16 * 8192 * 4)); CK(hipMalloc(&y, 16 * 8192 * 4));
  CK(hipMemset(x, 0, 16 * 8192 * 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  for (auto& sh : shapes) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 64; ++i) {
      ssrhip_gemv_args a; memset(&a, 0, sizeof(a));
      a.W = W + (size_t)(i % NBUF) * 8192 * 2048; a.x = x; a.y = y; a.B = B; a.N = sh.N; a.K = sh.K; a.groups = 1;
      a.x_stride = sh.K; a.y_stride = sh.N; a.x_tiled = tiled; a.y_tiled = tiled;
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
    const double us = ms * 1000.0 / (5 * 64), mb = (double)sh.N * sh.K * 4 / 1e6;
    printf("B=%d tiled=%d %-5s N=%d K=%d: %.2f us/launch, %.2f TB/s\n", B, tiled, sh.name, sh.N, sh.K, us, mb / us);
  }
  return 0;
}
