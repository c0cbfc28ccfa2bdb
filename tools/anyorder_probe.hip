// anyorder_probe.hip — does hipExtAnyOrderLaunch clear the AQL barrier bit on gfx950 (i.e. may two consecutive launches of ONE stream
// overlap)? hip_ext.h says the flag is "not supported on AMD GFX9xx boards"; this measures it. Two single-workgroup kernels that each
// spin ~50 us on the shader clock are launched back to back: ~50 us total = they overlapped, ~100 us = the runtime serialised them.
// Build: hipcc -O2 --offload-arch=gfx950 tools/anyorder_probe.hip -o tools/bin/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>

__global__ void spin_kernel(long long cycles, int* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (threadIdx.x == 0) out[blockIdx.x] = 1;
}

int main() {
  int* d;
  hipMalloc(&d, 1024);
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const long long cyc = 5000;      // wall_clock64 ticks at 100 MHz: 50 us
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, s);
      for (int i = 0; i < 4; ++i) {
        if (mode == 0) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, cyc, d);
        else hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d);
      }
      hipEventRecord(e1, s);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%s: 4 x 50 us spin kernels on one stream took %.1f us (%s)\n", mode ? "hipExtAnyOrderLaunch" : "plain launch", ms * 1000,
             hipGetErrorString(hipGetLastError()));
    }
  }
  return 0;
}
