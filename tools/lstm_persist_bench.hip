// tools/lstm_persist_bench.hip — what would ONE persistent kernel per LSTM layer cost per time step on this GPU, against the one
// launch per time step the codec uses now?
//
// The recurrence's per-step structure, without the gate math: 256 workgroups (one per CU), each owning 4 hidden units x 4 gates.
// Per step every workgroup
//   1. reads ALL of h_{t-1}  ([B][C] floats: 128 KB at B = 32, C = 1024) — produced by the other 255 workgroups in the previous step,
//   2. does a token amount of arithmetic on it (the real kernel: 16 rows x C MFMAs against a W_hh slice held in registers),
//   3. writes its 4 hidden units x B items of h_t (double-buffered),
//   4. meets the other workgroups at a grid barrier.
// Variants:  launches  one kernel launch per step in a hipGraph chain (what the product does)
//            flat      persistent kernel, one device-wide counter
//            xcd       persistent kernel, XCD-hierarchical barrier: per-XCD arrival counter (same L2), the XCD's last arriver goes to a
//                      top counter, the last of those publishes the generation; leaders re-publish it per XCD, the others poll their XCD's word
// Memory model: h is written with ordinary stores, made visible by a release fence at agent scope before the arrival (L2 write-back)
// and picked up after an acquire fence at agent scope behind the barrier (L2 invalidate) — the XCDs' L2s are not coherent with each other.
// Every spin is bounded and reports through an error counter: a broken barrier cannot hang the GPU.
// Build: hipcc -O3 --offload-arch=gfx950 tools/lstm_persist_bench.hip -o tools/bin/lstm_persist_bench ; run: lstm_persist_bench [B=32] [T=512]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int WGS = 256, TH = 256, C = 1024, UNITS = C / WGS;   // 4 hidden units per workgroup
constexpr unsigned SPIN_MAX = 1u << 20;

struct Sync {
  unsigned xcd_cnt[8][32];   // one counter per XCD, 128 B apart
  unsigned xcd_gen[8][32];
  unsigned top_cnt[32];
  unsigned top_gen[32];
  unsigned flat_cnt[32];
  unsigned err[32];
};

__device__ __forceinline__ unsigned ld_relaxed(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void spin_until(unsigned* p, unsigned want, unsigned* err) {
  unsigned it = 0;
  while (ld_relaxed(p) < want) {
    __builtin_amdgcn_s_sleep(1);
    if (++it > SPIN_MAX) { atomicAdd(err, 1u); break; }
  }
}

// gen = number of barriers already passed (0-based); returns after every workgroup has arrived at barrier `gen`
__device__ __forceinline__ void barrier_flat(Sync* s, unsigned gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(&s->flat_cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    spin_until(&s->flat_cnt[0], (gen + 1) * WGS, &s->err[0]);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__device__ __forceinline__ void barrier_xcd(Sync* s, unsigned gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const int x = blockIdx.x & 7;                                   // workgroups are dealt round-robin to the 8 XCDs
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned c = __hip_atomic_fetch_add(&s->xcd_cnt[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c == (gen + 1) * (WGS / 8) - 1) {                           // last arriver of this XCD: go to the top level
      const unsigned t = __hip_atomic_fetch_add(&s->top_cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == (gen + 1) * 8 - 1) __hip_atomic_store(&s->top_gen[0], gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else spin_until(&s->top_gen[0], gen + 1, &s->err[0]);
      __hip_atomic_store(&s->xcd_gen[x][0], gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      spin_until(&s->xcd_gen[x][0], gen + 1, &s->err[0]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// one step's work of workgroup `wg`: read all of h_prev, reduce a little, write UNITS x B values of h_next
__device__ __forceinline__ void step_work(const float* __restrict__ hp, float* __restrict__ hn, int B, int wg) {
  const int n4 = B * C / 4;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n4; i += TH) {
    const float4 v = reinterpret_cast<const float4*>(hp)[i];
    acc += (v.x + v.y) + (v.z + v.w);
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
  __shared__ float red[TH / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  const float tot = (red[0] + red[1]) + (red[2] + red[3]);
  if (threadIdx.x < UNITS * B) {
    const int b = threadIdx.x / UNITS, u = threadIdx.x % UNITS;
    hn[b * C + wg * UNITS + u] = 0.5f + 1e-6f * tot / (float)(B * C) + 1e-3f * (float)((wg * UNITS + u) & 7);   // stays O(1)
  }
  __syncthreads();
}

__global__ __launch_bounds__(TH) void step_kernel(const float* hp, float* hn, int B) { step_work(hp, hn, B, blockIdx.x); }

template <int MODE>   // 1 flat, 2 xcd
__global__ __launch_bounds__(TH) void persistent_kernel(float* h0, float* h1, int B, int T, Sync* s, unsigned gen0) {
  for (int t = 0; t < T; ++t) {
    const float* hp = (t & 1) ? h1 : h0;
    float* hn = (t & 1) ? h0 : h1;
    step_work(hp, hn, B, blockIdx.x);
    if (MODE == 1) barrier_flat(s, gen0 + t); else barrier_xcd(s, gen0 + t);
  }
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, T = argc > 2 ? atoi(argv[2]) : 512;
  if (UNITS * B > TH) { printf("B too large for this toy (UNITS*B <= %d)\n", TH); return 1; }
  float *h0, *h1; Sync* s;
  CK(hipMalloc(&h0, (size_t)B * C * 4)); CK(hipMalloc(&h1, (size_t)B * C * 4)); CK(hipMalloc(&s, sizeof(Sync)));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float* host = (float*)malloc((size_t)B * C * 4);
  auto reset = [&]() {
    for (int i = 0; i < B * C; ++i) host[i] = 0.5f;
    CK(hipMemcpy(h0, host, (size_t)B * C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(h1, host, (size_t)B * C * 4, hipMemcpyHostToDevice));
    CK(hipMemset(s, 0, sizeof(Sync)));
  };
  auto checksum = [&]() {
    CK(hipMemcpy(host, (T & 1) ? h1 : h0, (size_t)B * C * 4, hipMemcpyDeviceToHost));
    double c = 0; for (int i = 0; i < B * C; ++i) c += host[i];
    return c;
  };
  printf("B = %d items, C = %d, %d workgroups x %d threads, %d steps; h = %d KB read by every workgroup per step\n", B, C, WGS, TH, T, B * C * 4 / 1024);
  // ---- launches
  {
    reset();
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int t = 0; t < T; ++t) hipLaunchKernelGGL(step_kernel, dim3(WGS), dim3(TH), 0, st, (t & 1) ? h1 : h0, (t & 1) ? h0 : h1, B);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ex, st)); CK(hipStreamSynchronize(st)); reset();
    auto t0 = std::chrono::steady_clock::now();
    CK(hipGraphLaunch(ex, st)); CK(hipStreamSynchronize(st));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("one launch per step (hipGraph chain)      : %6.2f us per step   checksum %.6f\n", us / T, checksum());
  }
  for (int mode = 1; mode <= 2; ++mode) {
    reset();
    auto launch = [&](unsigned gen0) {
      if (mode == 1) hipLaunchKernelGGL(persistent_kernel<1>, dim3(WGS), dim3(TH), 0, st, h0, h1, B, T, s, gen0);
      else hipLaunchKernelGGL(persistent_kernel<2>, dim3(WGS), dim3(TH), 0, st, h0, h1, B, T, s, gen0);
    };
    launch(0); CK(hipStreamSynchronize(st));
    unsigned e0; CK(hipMemcpy(&e0, &s->err[0], 4, hipMemcpyDeviceToHost));
    reset();
    auto t0 = std::chrono::steady_clock::now();
    launch(0); CK(hipStreamSynchronize(st));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    unsigned e1; CK(hipMemcpy(&e1, &s->err[0], 4, hipMemcpyDeviceToHost));
    printf("persistent, %-30s: %6.2f us per step   checksum %.6f   spin time-outs %u / %u\n", mode == 1 ? "flat barrier" : "XCD-hierarchical barrier", us / T, checksum(), e0, e1);
  }
  return 0;
}
