// tools/prefetch_bench.hip — does a concurrent "prefetch the NEXT GEMV's weights into the Infinity Cache" kernel on a
// parallel hipGraph branch speed up the dependent GEMV chain of the decode step?
//   main chain : 64 x ssrhip_gemv (B=2, FFN1 shape 8192x2048 = 67 MB) over 16 rotating weight buffers (no reuse inside 1 GB)
//   prefetch   : P_{j+1} streams the weights of K_{j+1} with discard loads while K_j runs (edge K_{j-1} -> P_{j+1})
// usage: prefetch_bench <mode> [N K]   mode 0 = no prefetch, 1 = prefetch branch, 2 = prefetch branch touching the WRONG buffer
// (mode 2 = same concurrency, no cache benefit: isolates the cost of the side kernel). Run with SSRHIP_GEMV_BLOCKS_PER_CU=2|3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/ssrhip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void prefetch_kernel(const v4f* p, long n4, int nt) {
  // 16 x 16 B in flight per lane; results are only "used" by an empty asm so the loads cannot be dropped
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  while (i < n4) {
    v4f v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long k = i + j * stride;
      const v4f* q = p + (k < n4 ? k : i);
      v[j] = nt ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(v[j].x), "v"(v[j].y), "v"(v[j].z), "v"(v[j].w));
    i += 8 * stride;
  }
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const int N = argc > 2 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 2048;
  const int pf_blocks = argc > 4 ? atoi(argv[4]) : 256, nt = argc > 5 ? atoi(argv[5]) : 0;
  const int NBUF = 16, NL = 64;
  const size_t per = (size_t)8192 * 2048;
  float* W; CK(hipMalloc(&W, (size_t)(NBUF + 1) * per * 4));
  CK(hipMemset(W, 0, (size_t)(NBUF + 1) * per * 4));
  float *x, *y; CK(hipMalloc(&x, 4 * 8192 * 4)); CK(hipMalloc(&y, 4 * 8192 * 4));
  CK(hipMemset(x, 0, 4 * 8192 * 4));
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  hipEvent_t ev[NL + 2], evb;
  for (int i = 0; i < NL + 2; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&evb, hipEventDisableTiming));
  hipGraph_t g; hipGraphExec_t ex;
  CK(hipStreamBeginCapture(a, hipStreamCaptureModeThreadLocal));
  for (int j = 0; j < NL; ++j) {
    if (mode && j + 1 < NL) {
      // fork: P_{j+1} may start once K_{j-1} is done (i.e. it overlaps K_j)
      CK(hipEventRecord(ev[j], a));
      CK(hipStreamWaitEvent(b, ev[j], 0));
      const float* tgt = W + (size_t)(mode == 2 ? NBUF : (j + 1) % NBUF) * per;
      hipLaunchKernelGGL(prefetch_kernel, dim3(pf_blocks), dim3(256), 0, b, (const v4f*)tgt, (long)N * K / 4, nt);
    }
    ssrhip_gemv_args g1; memset(&g1, 0, sizeof(g1));
    g1.W = W + (size_t)(j % NBUF) * per; g1.x = x; g1.y = y; g1.B = 2; g1.N = N; g1.K = K; g1.groups = 1; g1.x_stride = K; g1.y_stride = N;
    if (ssrhip_gemv(&g1, a)) { printf("err: %s\n", ssrhip_last_error()); return 1; }
  }
  if (mode) { CK(hipEventRecord(evb, b)); CK(hipStreamWaitEvent(a, evb, 0)); }   // join
  CK(hipStreamEndCapture(a, &g));
  CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipGraphLaunch(ex, a));
  CK(hipEventRecord(e0, a));
  for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ex, a));
  CK(hipEventRecord(e1, a));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("mode %d N=%d K=%d pf_blocks=%d nt=%d: %.2f us per GEMV launch (%.2f TB/s)\n", mode, N, K, pf_blocks, nt, ms * 1000.0 / (5 * NL),
         (double)N * K * 4 / (ms * 1e-3 / (5 * NL)) / 1e12);
  return 0;
}
