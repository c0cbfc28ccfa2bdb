// tools/mfma_peak.hip — ceiling of v_mfma_f32_32x32x2_f32 on this GPU with NO memory traffic in the loop: every wave keeps 4 accumulators
// and 8 operand registers and issues MFMAs back to back. Run with zero, constant-sign and full-range random operands: the difference is
// the power / clock give-back (MI355X_MICROARCH.md "DVFS give-back"), i.e. what a perfect fp32 GEMM could reach on random data.
//   mfma_peak [waves_per_simd=1] [iters=20000]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ in, float* out, int iters) {
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x + 256 * i) & 4095]; b[i] = in[(threadIdx.x * 7 + 256 * i + 13) & 4095]; }
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + 1) & 7], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + 1) & 7], b[i], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + 1) & 7], b[(i + 1) & 7], acc[3], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1, iters = argc > 2 ? atoi(argv[2]) : 20000;
  float *in, *out;
  CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 256 * 8 * 256 * 4));
  float h[4096];
  const char* names[3] = {"zeros", "positive uniform [0,1)", "uniform [-1,1)"};
  for (int mode = 0; mode < 3; ++mode) {
    unsigned st = 12345u;
    for (int i = 0; i < 4096; ++i) { st = st * 1664525u + 1013904223u; const float u = (st >> 8) / 16777216.0f; h[i] = mode == 0 ? 0.f : (mode == 1 ? u : 2 * u - 1) * 1e-3f; }
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * wps;
    hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, in, out, 2000);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, in, out, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = (double)grid * 4 * iters * 32 * (2.0 * 32 * 32 * 2);
    printf("%-26s %d wave(s)/SIMD: %8.3f ms  %6.1f TFLOP/s  (%.3f GHz-equivalent of 64 FLOP/clk/SIMD)\n", names[mode], wps, ms, fl / (ms * 1e-3) / 1e12,
           fl / (ms * 1e-3) / (256.0 * 4 * 64) / 1e9);
  }
  return 0;
}
