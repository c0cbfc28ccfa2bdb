// tools/launch_floor.hip — what does one kernel node of a dependent hipGraph chain cost on this GPU, before any work?
//  (a) empty kernel, 1 workgroup      (b) empty kernel, 768 x 256 threads (the GEMV grid)
//  (c) 768 x 256, every wave loads ONE 1 KiB line-set from a rotating 64 MB buffer and stores one float (first-byte latency)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void empty_kernel(float* y) { if (y == nullptr) y[0] = 1.f; }
__global__ __launch_bounds__(256) void touch_kernel(const float4* w, float* y, long n4) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 97 % n4;
  const float4 v = w[i];
  if (v.x == 12345.f) y[0] = v.y;
}
__global__ __launch_bounds__(256) void touch1k_kernel(const float4* w, float* y, long n4) {
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const float4 v = w[(wave * 8191 % (n4 / 64)) * 64 + (threadIdx.x & 63)];   // one contiguous KiB per wave
  if (v.x == 12345.f) y[0] = v.y;
}

template <class F>
static double chain(hipStream_t s, int n, F launch) {
  hipGraph_t g; hipGraphExec_t ex;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n; ++i) launch(i);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipGraphLaunch(ex, s));
  CK(hipEventRecord(e0, s));
  for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ex, s));
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.0 / (10.0 * n);
}

int main() {
  const long n4 = 16L * 1024 * 1024 * 16;   // 4 GiB of float4? no: 256 Mi float4 = 4 GiB is too much -> use 64 Mi float4 = 1 GiB
  const long N4 = 64L * 1024 * 1024;
  (void)n4;
  float4* w; float* y;
  CK(hipMalloc(&w, N4 * 16)); CK(hipMemset(w, 0, N4 * 16)); CK(hipMalloc(&y, 4096));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  printf("empty, 1 workgroup          : %.2f us per node\n", chain(s, 64, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, y); }));
  printf("empty, 768 x 256            : %.2f us per node\n", chain(s, 64, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(768), dim3(256), 0, s, y); }));
  printf("one scattered 16 B load/lane: %.2f us per node\n", chain(s, 64, [&](int i) { hipLaunchKernelGGL(touch_kernel, dim3(768), dim3(256), 0, s, w + (long)(i % 16) * (N4 / 16), y, N4 / 16); }));
  printf("one KiB per wave (3 MB)     : %.2f us per node\n", chain(s, 64, [&](int i) { hipLaunchKernelGGL(touch1k_kernel, dim3(768), dim3(256), 0, s, w + (long)(i % 16) * (N4 / 16), y, N4 / 16); }));
  return 0;
}
