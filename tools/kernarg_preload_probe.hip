// kernarg_preload_probe.hip — what does SGPR kernarg preloading (gfx940+: the CP writes the first kernel arguments into user SGPRs,
// -mllvm -amdgpu-kernarg-preload-count=N) save per launch of a DEPENDENT chain? Every wave of a normal kernel starts with an s_load of
// its arguments from the kernarg segment (written by the host / the graph: a cold scalar-cache miss) before it can form its first address.
// The probe replays a hipGraph of 200 dependent launches of a streaming kernel (256 workgroups x 512 threads, each lane one 16-byte
// load whose address needs the arguments) and prints us per launch. Build it twice:
//   hipcc -O3 --offload-arch=gfx950 tools/kernarg_preload_probe.hip -o tools/bin/kp_base
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=8 tools/kernarg_preload_probe.hip -o tools/bin/kp_preload
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(512) void stream_kernel(const float4* __restrict__ w, const float* __restrict__ x, float* __restrict__ y, int n4, int rows) {
  const int i = blockIdx.x * 512 + threadIdx.x;
  float acc = 0.f;
  for (int r = 0; r < rows; ++r) {
    const float4 v = w[(size_t)r * n4 + i];
    acc += v.x + v.y + v.z + v.w;
  }
  y[i] = acc + x[i];
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 1;          // 1: 2 MB per launch (launch-bound), 24: 50 MB
  const int n4 = 256 * 512;
  float4* w; float *x, *y;
  hipMalloc(&w, (size_t)n4 * rows * 16 * 4);
  hipMalloc(&x, n4 * 4); hipMalloc(&y, n4 * 4);
  hipMemset(w, 0, (size_t)n4 * rows * 16 * 4); hipMemset(x, 0, n4 * 4);
  hipStream_t s; hipStreamCreate(&s);
  hipGraph_t g; hipGraphExec_t ex;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < 200; ++i)
    hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(512), 0, s, w + (size_t)(i % 4) * n4 * rows, (i & 1) ? x : y, (i & 1) ? y : x, n4, rows);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, s);
    for (int i = 0; i < 5; ++i) hipGraphLaunch(ex, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("rows %d: %.3f us per launch (%s)\n", rows, ms * 1000 / 1000, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
