// gemv_floor_lab.hip — what a 50 MB weight-streaming launch of a DEPENDENT hipGraph chain can cost on this GPU, feature by feature:
// from the bare streaming kernel of tools/kernarg_preload_probe.hip (1.66 us + bytes / 7.1 TB/s on zero-filled memory) towards the
// product GEMV (csrc/gemv.hip: 10.2-10.5 us for LN + QKV, 50.4 MB). Every variant streams N x 2048 fp32 with 256 workgroups x 512
// threads (modes >= 10: 512 workgroups, two per CU) and writes 2 x N floats; launches alternate x / y so each depends on its predecessor.
//   mode 0  lane-strided sweep (all CUs walk the matrix front together), zero data         mode 1  same, random data
//   mode 2  mode 1 with non-temporal loads
//   mode 3  workgroup-contiguous rows (the GEMV's layout: a workgroup owns N/256 consecutive rows; wave w takes units w, w+8, ..),
//           all loads up front, sums only
//   mode 4  mode 3 + the GEMV arithmetic: x slice in registers, dot4, wave all-reduce per unit, LDS park, barrier, y = sum of segments
//   mode 5  mode 4 with at most 4 units (16 loads per lane) in flight: the rest requested as the first ones are consumed
//   mode 6  mode 4 with 2 units in flight
//   mode 10 mode 6 at two workgroups per CU (= the segment kernel's geometry: 1 unit in flight per wave, 16 waves per CU) [uses 1 unit]
//   mode 7 / 11  modes 5 / 10 with x fetched ONCE per workgroup (512 threads x 2 float4 = the 16 KB both rows need) into LDS and read from
//           there by every wave, instead of every wave fetching its 8 KB slice from L2 (64 KB of L1 traffic per workgroup for 16 KB of data)
// Build: hipcc -O3 --offload-arch=gfx950 -I ssr-speech_amd/csrc -I include tools/gemv_floor_lab.hip -o tools/bin/gemv_floor_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "common.h"
void ssrhip_set_error(const char*, ...) {}

constexpr int K = 2048, TH = 512;

// Round 5, modes 20 / 21 / 22 = mode 5 + a PREFETCH of the head of the NEXT launch's matrix (the first units its waves will request, same
// geometry) with default-policy loads, so that they sit in the Infinity Cache when the next launch starts: 20 = requested right behind the
// wave's own first units (shares the stream's bandwidth: no extra HBM bytes over the chain, the next launch's ramp is served by the
// cache), 21 = requested behind the wave's last unit (flies under the reduction / barrier / store tail), 22 = as 20 with half the depth.
template <int MODE>
__global__ __launch_bounds__(TH, (MODE >= 10 && MODE < 20) ? 4 : 2) void k(const float* __restrict__ W, const float* __restrict__ x, float* __restrict__ y, int N, const float* __restrict__ Wn = nullptr) {
  __shared__ float part[64 * 2 * 2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int G = gridDim.x;
  if constexpr (MODE <= 2) {
    const int n4 = G * TH;                       // float4 per sweep line
    const int rows = N * (K / 4) / n4;
    const int i = blockIdx.x * TH + t;
    float acc = 0.f;
#pragma unroll 8
    for (int r = 0; r < rows; ++r) {
      const float* p = W + ((size_t)r * n4 + i) * 4;
      const float4 v = (MODE == 2) ? ld_nt(p) : ld4(p);
      acc += (v.x + v.y) + (v.z + v.w);
    }
    if (i < 2 * N) y[i] = acc + x[i];
  } else {
    const int R = N / G;                         // rows per workgroup
    const int nu = R * 2;                        // (row, 1024-float segment) units
    const int seg = wave & 1;
    const float* Wg = W + (size_t)blockIdx.x * R * K + seg * 1024 + lane * 4;
    constexpr int NUW = (MODE >= 10 && MODE < 20) ? 3 : 6;    // units per wave at N = 6144 (QKV): 48 units / 8 waves; two workgroups per CU: 24 / 8
    constexpr int DEPTH = (MODE == 3 || MODE == 4) ? NUW : ((MODE == 5 || MODE == 7 || MODE >= 20) ? 4 : (MODE == 6 ? 2 : 1));
    constexpr int PFD = (MODE == 20 || MODE == 21) ? 4 : (MODE == 22 ? 2 : 0);      // units of the next launch prefetched per wave
    constexpr bool XLDS = (MODE == 7 || MODE == 11);
    __shared__ __attribute__((aligned(16))) float xs[XLDS ? 2 * K : 4];
    float4 xr[2][4];
    float4 xg[2];
    if constexpr (XLDS) {
#pragma unroll
      for (int b = 0; b < 2; ++b) xg[b] = ld4(x + b * K + t * 4);         // 512 threads x 4 floats = one row of x per pass
    } else if constexpr (MODE >= 4) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) xr[b][i] = ld4(x + b * K + seg * 1024 + (i * 64 + lane) * 4);
    }
    float4 w[DEPTH][4];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) w[j][i] = ld_nt(Wg + (size_t)(min(wave + 8 * j, nu - 1) >> 1) * K + i * 256);
    float4 pf[PFD > 0 ? PFD : 1][4];
    const float* Wng = Wn + (size_t)blockIdx.x * R * K + seg * 1024 + lane * 4;
    if constexpr (MODE == 20 || MODE == 22) {
#pragma unroll
      for (int j = 0; j < PFD; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) pf[j][i] = ld4(Wng + (size_t)(min(wave + 8 * j, nu - 1) >> 1) * K + i * 256);
    }
    if constexpr (XLDS) {
#pragma unroll
      for (int b = 0; b < 2; ++b) *reinterpret_cast<float4*>(xs + b * K + t * 4) = xg[b];
      __syncthreads();
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) xr[b][i] = *reinterpret_cast<const float4*>(xs + b * K + seg * 1024 + (i * 64 + lane) * 4);
    }
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < NUW; ++j) {
      const int u = wave + 8 * j;
      float4 (&wj)[4] = w[j % DEPTH];
      if constexpr (MODE == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) tot += (wj[i].x + wj[i].y) + (wj[i].z + wj[i].w);
      } else {
        float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[b][i & 1] = dot4(wj[i], xr[b][i], acc[b][i & 1]);
          if (j + DEPTH < NUW) {                 // compile-time: re-request this 16-byte piece for unit j + DEPTH
            __builtin_amdgcn_sched_barrier(0);
            wj[i] = ld_nt(Wg + (size_t)(min(wave + 8 * (j + DEPTH), nu - 1) >> 1) * K + i * 256);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        float mine = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const float s = wave_sum(acc[b][0] + acc[b][1]);
          if (lane == b) mine = s;
        }
        if (lane < 2 && u < nu) part[u * 2 + lane] = mine;
      }
    }
    if constexpr (MODE == 21) {
#pragma unroll
      for (int j = 0; j < PFD; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) pf[j][i] = ld4(Wng + (size_t)(min(wave + 8 * j, nu - 1) >> 1) * K + i * 256);
    }
    if constexpr (MODE == 3) {
      y[blockIdx.x * TH + t] = tot + x[t];
    } else {
      __syncthreads();
      if constexpr (PFD > 0) {                   // keep the prefetched registers alive to the end (never true)
        float keep = 0.f;
#pragma unroll
        for (int j = 0; j < PFD; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) keep += (pf[j][i].x + pf[j][i].y) + (pf[j][i].z + pf[j][i].w);
        if (keep == 1234.5678f) y[N * 2 + t] = keep;
      }
      if (t < R * 2) {
        const int r = t >> 1, b = t & 1;
        y[b * N + blockIdx.x * R + r] = part[(r * 2 + 0) * 2 + b] + part[(r * 2 + 1) * 2 + b];
      }
    }
  }
}

// One 1024-thread workgroup per CU (16 waves, 128 VGPRs): unit u of the workgroup = (row u >> 1, segment u & 1); a wave always works on
// segment wave & 1. STATIC: wave w takes units w, w + 16, ..  DYNAMIC: after its first unit a wave draws the next unit of ITS segment from
// an LDS counter (ds_add_rtn, fetched one unit ahead), so no wave idles while another still has units queued: with static hand-out the
// oldest waves win the arbitration for the memory pipeline, finish early, and the kernel ends on a few waves with few loads in flight.
// PRIO: mode 10's geometry (two 8-wave workgroups per CU) with the second-dispatched workgroup at a higher s_setprio.
template <bool DYNAMIC, int DEPTH>
__global__ __launch_bounds__(1024, 4) void k16(const float* __restrict__ W, const float* __restrict__ x, float* __restrict__ y, int N) {
  __shared__ float part[64 * 2 * 2];
  __shared__ int cnt[2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int G = gridDim.x, R = N / G, nu = R * 2, seg = wave & 1;
  const int nseg = R;                                   // units per segment
  const float* Wg = W + (size_t)blockIdx.x * R * K + seg * 1024 + lane * 4;
  if (t < 2) cnt[t] = 8 * DEPTH;                         // the first 8 * DEPTH units of a segment are dealt statically
  float4 xr[2][4];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) xr[b][i] = ld4(x + b * K + seg * 1024 + (i * 64 + lane) * 4);
  float4 w[DEPTH][4];
  int ur[DEPTH];                                         // row (= index inside the segment) of the unit in slot d
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    ur[d] = (wave >> 1) + 8 * d;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[d][i] = ld_nt(Wg + (size_t)min(ur[d], nseg - 1) * K + i * 256);
  }
  __syncthreads();
  auto draw = [&]() {                                    // ONE increment per wave (lane 0), broadcast
    int v = 0;
    if (lane == 0) v = atomicAdd(&cnt[seg], 1);
    return __builtin_amdgcn_readfirstlane(v);
  };
  int nxt = DYNAMIC ? draw() : (wave >> 1) + 8 * DEPTH;
  int slot = 0;
  while (true) {
    // the unit in `slot` is the oldest one in flight
    const int row = ur[0];
    const int nn = nxt;                                  // unit that takes this slot's place
    int nn2 = 0;
    if (DYNAMIC) { nn2 = nseg; if (nn < nseg) nn2 = draw(); }
    else nn2 = nn + 8;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[b][i & 1] = dot4(w[0][i], xr[b][i], acc[b][i & 1]);
      __builtin_amdgcn_sched_barrier(0);
      w[0][i] = ld_nt(Wg + (size_t)min(nn, nseg - 1) * K + i * 256);      // unconditional (clamped): the tail re-reads the last row
      __builtin_amdgcn_sched_barrier(0);
    }
    float mine = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float s = wave_sum(acc[b][0] + acc[b][1]);
      if (lane == b) mine = s;
    }
    if (lane < 2 && row < nseg) part[(row * 2 + seg) * 2 + lane] = mine;
    // rotate the slots: slot 0 <- slot 1 <- .. <- the unit just requested
    if constexpr (DEPTH > 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 tmp = w[0][i];
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; ++d) w[d][i] = w[d + 1][i];
        w[DEPTH - 1][i] = tmp;
      }
#pragma unroll
      for (int d = 0; d + 1 < DEPTH; ++d) ur[d] = ur[d + 1];
    }
    ur[DEPTH - 1] = nn;
    nxt = nn2;
    if (ur[0] >= nseg) break;                            // uniform per wave: nothing valid left in flight (units are handed out in order)
    (void)slot;
  }
  __syncthreads();
  if (t < R * 2) {
    const int r = t >> 1, b = t & 1;
    y[b * N + blockIdx.x * R + r] = part[(r * 2 + 0) * 2 + b] + part[(r * 2 + 1) * 2 + b];
  }
}

template <bool DYNAMIC, int DEPTH>
float run16(const float* W, float* x, float* y, int N, int G, size_t wstride) {
  hipStream_t s; hipStreamCreate(&s);
  hipGraph_t g; hipGraphExec_t ex;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < 128; ++i)
    hipLaunchKernelGGL((k16<DYNAMIC, DEPTH>), dim3(G), dim3(1024), 0, s, W + (size_t)(i % 8) * wstride, (i & 1) ? x : y, (i & 1) ? y : x, N);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, s);
    for (int i = 0; i < 4; ++i) hipGraphLaunch(ex, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  hipGraphExecDestroy(ex); hipGraphDestroy(g); hipStreamDestroy(s);
  return best * 1000.f / 512.f;
}

template <int MODE>
float run(const float* W, float* x, float* y, int N, int G, size_t wstride) {
  hipStream_t s; hipStreamCreate(&s);
  hipGraph_t g; hipGraphExec_t ex;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < 128; ++i)
    hipLaunchKernelGGL(k<MODE>, dim3(G), dim3(TH), 0, s, W + (size_t)(i % 8) * wstride, (i & 1) ? x : y, (i & 1) ? y : x, N, W + (size_t)((i + 1) % 8) * wstride);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, s);
    for (int i = 0; i < 4; ++i) hipGraphLaunch(ex, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  hipGraphExecDestroy(ex); hipGraphDestroy(g); hipStreamDestroy(s);
  return best * 1000.f / 512.f;
}

int main() {
  const int N = 6144;
  const size_t wstride = (size_t)N * K;
  float *W, *Wz, *x, *y;
  hipMalloc(&W, wstride * 8 * 4); hipMalloc(&Wz, wstride * 8 * 4);
  hipMalloc(&x, 4 * N * 4 + 512 * 512 * 4); hipMalloc(&y, 4 * N * 4 + 512 * 512 * 4);
  std::vector<float> h(wstride);
  unsigned st = 12345u;
  for (auto& v : h) { st = st * 1664525u + 1013904223u; v = ((int)(st >> 8) - (1 << 23)) * (1.0f / (1 << 23)) * 0.02f; }
  for (int i = 0; i < 8; ++i) hipMemcpy(W + i * wstride, h.data(), wstride * 4, hipMemcpyHostToDevice);
  hipMemset(Wz, 0, wstride * 8 * 4);
  hipMemset(x, 0, 4 * N * 4 + 512 * 512 * 4); hipMemset(y, 0, 4 * N * 4 + 512 * 512 * 4);
  const double mb = wstride * 4 / 1e6;
  printf("N %d x K %d = %.1f MB per launch, dependent hipGraph chain, us per launch (best of 4 x 512 launches)\n", N, K, mb);
  printf("mode 0  sweep, zeros                          %6.2f\n", run<0>(Wz, x, y, N, 256, wstride));
  printf("mode 1  sweep, random data                    %6.2f\n", run<1>(W, x, y, N, 256, wstride));
  printf("mode 2  sweep, random, nt loads               %6.2f\n", run<2>(W, x, y, N, 256, wstride));
  printf("mode 3  workgroup-contiguous rows, all upfront %6.2f\n", run<3>(W, x, y, N, 256, wstride));
  printf("mode 3z same on zeros                         %6.2f\n", run<3>(Wz, x, y, N, 256, wstride));
  printf("mode 4  + GEMV arithmetic, all 6 units upfront %6.2f\n", run<4>(W, x, y, N, 256, wstride));
  printf("mode 5  4 units in flight                     %6.2f\n", run<5>(W, x, y, N, 256, wstride));
  printf("mode 6  2 units in flight                     %6.2f\n", run<6>(W, x, y, N, 256, wstride));
  printf("mode 7  4 units in flight, x via LDS          %6.2f\n", run<7>(W, x, y, N, 256, wstride));
  printf("mode 20 mode 5 + next launch's head prefetched at entry (4 units/wave) %6.2f\n", run<20>(W, x, y, N, 256, wstride));
  printf("mode 22 same, 2 units/wave                                            %6.2f\n", run<22>(W, x, y, N, 256, wstride));
  printf("mode 21 prefetched behind the last unit (4 units/wave)                %6.2f\n", run<21>(W, x, y, N, 256, wstride));
  printf("mode 5  again                                                         %6.2f\n", run<5>(W, x, y, N, 256, wstride));
  printf("mode 10 two workgroups per CU, 1 unit in flight %6.2f\n", run<10>(W, x, y, N, 512, wstride));
  printf("mode 11 same, x via LDS                       %6.2f\n", run<11>(W, x, y, N, 512, wstride));
  printf("mode 10z same on zeros                        %6.2f\n", run<10>(Wz, x, y, N, 512, wstride));
  printf("k16 static,  1 unit in flight (16 waves/CU)    %6.2f\n", run16<false, 1>(W, x, y, N, 256, wstride));
  printf("k16 dynamic, 1 unit in flight                  %6.2f\n", run16<true, 1>(W, x, y, N, 256, wstride));
  printf("k16 static,  2 units in flight                 %6.2f\n", run16<false, 2>(W, x, y, N, 256, wstride));
  printf("k16 dynamic, 2 units in flight                 %6.2f\n", run16<true, 2>(W, x, y, N, 256, wstride));
  // correctness of the dynamic hand-out against the static one: one launch each from the same x
  {
    std::vector<float> a(2 * N), b(2 * N);
    hipMemset(x, 0, 4 * N * 4);
    hipLaunchKernelGGL((k16<false, 1>), dim3(256), dim3(1024), 0, 0, W, x, y, N);
    hipMemcpy(a.data(), y, 2 * N * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL((k16<true, 2>), dim3(256), dim3(1024), 0, 0, W, x, y, N);
    hipMemcpy(b.data(), y, 2 * N * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 2 * N; ++i) bad += a[i] != b[i];
    hipLaunchKernelGGL((k<10>), dim3(512), dim3(512), 0, 0, W, x, y, N, W);
    hipMemcpy(b.data(), y, 2 * N * 4, hipMemcpyDeviceToHost);
    int bad2 = 0; for (int i = 0; i < 2 * N; ++i) bad2 += a[i] != b[i];
    printf("dynamic vs static: %d of %d outputs differ; mode 10 vs k16: %d differ\n", bad, 2 * N, bad2);
  }
  return 0;
}
