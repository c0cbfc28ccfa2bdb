#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../ssr-speech_amd/csrc/common.h"
void ssrhip_set_error(const char*, ...) {}
__global__ void k(const float* x, float* y, float* z, int* w, float* x16, float* x32) {
  float v = x[threadIdx.x];
  y[threadIdx.x] = wave_sum(v);
  z[threadIdx.x] = wave_max(v);
  w[threadIdx.x] = wave_min_i((int)(v * 100));
  x16[threadIdx.x] = xor16_f(v);
  x32[threadIdx.x] = xor32_f(v);
}
int main() {
  float hx[64], hy[64], hz[64], h16[64], h32[64]; int hw[64];
  float s = 0, mx = -1e9; int mn = 1 << 30;
  for (int i = 0; i < 64; ++i) { hx[i] = (float)((i * 37) % 101) - 50.f + 0.25f * i; s += hx[i]; mx = fmaxf(mx, hx[i]); mn = std::min(mn, (int)(hx[i] * 100)); }
  float *dx, *dy, *dz, *d16, *d32; int* dw;
  hipMalloc(&dx, 256); hipMalloc(&dy, 256); hipMalloc(&dz, 256); hipMalloc(&dw, 256); hipMalloc(&d16, 256); hipMalloc(&d32, 256);
  hipMemcpy(dx, hx, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy, dz, dw, d16, d32);
  hipMemcpy(hy, dy, 256, hipMemcpyDeviceToHost); hipMemcpy(hz, dz, 256, hipMemcpyDeviceToHost); hipMemcpy(hw, dw, 256, hipMemcpyDeviceToHost);
  hipMemcpy(h16, d16, 256, hipMemcpyDeviceToHost); hipMemcpy(h32, d32, 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    if (fabsf(hy[i] - s) > 1e-3 || hz[i] != mx || hw[i] != mn || h16[i] != hx[i ^ 16] || h32[i] != hx[i ^ 32]) { ++bad; if (bad < 6) printf("lane %d sum %f (want %f) max %f (%f) min %d (%d) x16 %f (%f) x32 %f (%f)\n", i, hy[i], s, hz[i], mx, hw[i], mn, h16[i], hx[i^16], h32[i], hx[i^32]); }
  }
  printf("bad lanes: %d\n", bad);
  return 0;
}
