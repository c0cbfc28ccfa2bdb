// fetch_calib.hip — calibration of rocprofv3's FETCH_SIZE for the residual-block kernels' access pattern ("calibrate the
// resblock FETCH_SIZE 1.97x with a copy kernel of known traffic"). The guide (MI355X_MICROARCH.md, HBM) says the counter reports HALF the
// bytes of a wide coalesced streaming read (16 B per lane, 1 KiB per wave instruction); the residual blocks read 128-byte row pieces
// (8 lanes x 16 B) of a [T][128]-float array, 64 rows per wave instruction: does the factor hold there?
//   tile_read        reads every element of x[B][T + 2][128] exactly ONCE in the kernel's tile pattern (130 rows x 32 channels per pass)
//   tile_read_twice  the same, then the centre rows again the way the epilogue does (4-byte loads, 32 consecutive channels per row),
//                    ~a tile-time later
// Known traffic: B * T * 512 B (x1, x2). Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (tools/r04_fetch_calib.sh).
// Build: hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/bin/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int CC = 128, BM = 128;

template <bool TWICE>
__global__ __launch_bounds__(256) void tile_read(const float* __restrict__ x, float* __restrict__ out, int T) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * BM;
  const float* xin = x + (size_t)blockIdx.y * (T + 2) * CC;
  float acc = 0.f;
  for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int idx = min(t + 256 * i, 130 * 8 - 1), row = idx >> 3, c4 = idx & 7;
      const int prow = min(m0 + row, T + 1);
      const float4 v = *reinterpret_cast<const float4*>(xin + (size_t)prow * CC + ct * 32 + c4 * 4);
      acc += (v.x + v.y) + (v.z + v.w);
    }
    __syncthreads();
  }
  if (TWICE) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < T) acc += xin[(size_t)(m + 1) * CC + nb * 32 + li];
      }
  }
  out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + t] = acc;
}

int main() {
  const int B = 32, T = 240000;
  float *x, *out;
  const size_t nx = (size_t)B * (T + 2) * CC;
  hipMalloc(&x, nx * 4);
  hipMalloc(&out, (size_t)B * ((T + BM - 1) / BM) * 256 * 4);
  hipMemset(x, 0, nx * 4);
  dim3 grid((T + BM - 1) / BM, B);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(tile_read<false>, grid, dim3(256), 0, 0, x, out, T);
    hipLaunchKernelGGL(tile_read<true>, grid, dim3(256), 0, 0, x, out, T);
  }
  hipDeviceSynchronize();
  printf("known traffic: x1 = %.1f MB, x2 = %.1f MB (B %d, T %d, 128 channels fp32)\n", nx * 4 / 1e6, 2 * nx * 4 / 1e6, B, T);
  return 0;
}
