// tools/overlap_bench.hip — can two CONSECUTIVE launches of the dependent decode chain overlap (the next launch streaming its first
// weights while the previous one is in its reduction / store tail) if the dependency is carried by a flag in memory instead of the
// kernel boundary?  `hipExtAnyOrderLaunch` is documented as unsupported on gfx9, so the only HIP-level way is two streams that
// alternate launches with NO event between them and an in-kernel handshake.
//
// The kernel imitates one fused GEMV of the 2-row step: 256 workgroups x 512 threads stream `MB` megabytes of weights (non-temporal
// 16-byte loads, 8 in flight per thread), multiply with a 256-float vector x produced by the PREVIOUS launch (all 256 workgroups of
// it: one value each), reduce, write one value, and (handshake modes) bump the launch's counter.
//
//   mode 0  one stream, ordinary dependent launches (eager)
//   mode 1  the same chain captured in a hipGraph
//   mode 2  two streams alternating, release/acquire handshake at agent scope (fetch_add RELEASE / load ACQUIRE)
//   mode 3  two streams alternating, x written / read with agent-scope relaxed atomics (write-through / L2-bypass), flag relaxed
//
// Every spin is bounded (~50 ms) and reports through an error counter, so a broken handshake cannot hang the GPU.
// Build: hipcc -O3 --offload-arch=gfx950 tools/overlap_bench.hip -o tools/bin/overlap_bench ; run: overlap_bench [MB per launch = 50]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int WGS = 256, THREADS = 512, DEPTH = 8, CHAIN = 66;   // the handshake experiment's shape; `sweep` varies them

__device__ __forceinline__ float4 ld_nt(const float4* p) {
  float4 v;
  v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y);
  v.z = __builtin_nontemporal_load(&p->z); v.w = __builtin_nontemporal_load(&p->w);
  return v;
}

template <int MODE, int THREADS = ::THREADS, int DEPTH = ::DEPTH>   // MODE 0: plain loads/stores, 2: release/acquire, 3: relaxed agent-scope atomics
__global__ __launch_bounds__(THREADS) void gemv_like(const float4* __restrict__ w, int iters, const float* x_in, float* x_out,
                                                     unsigned* flag_in, unsigned* flag_out, unsigned expect, unsigned* err) {
  const int tid = threadIdx.x;
  const float4* p = w + (long)blockIdx.x * iters * THREADS + tid;
  float4 r[DEPTH];
#pragma unroll
  for (int j = 0; j < DEPTH; ++j) r[j] = ld_nt(p + (long)min(j, iters - 1) * THREADS);   // weights do not depend on x: issued before the wait
  __shared__ float red[THREADS / 64];
  if (MODE >= 2 && flag_in != nullptr) {
    if (tid == 0) {
      unsigned it = 0;
      if (MODE == 2) {
        while (__hip_atomic_load(flag_in, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < expect) {
          __builtin_amdgcn_s_sleep(1);
          if (++it > (1u << 20)) { atomicAdd(err, 1u); break; }
        }
      } else {
        while (__hip_atomic_load(flag_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
          __builtin_amdgcn_s_sleep(1);
          if (++it > (1u << 20)) { atomicAdd(err, 1u); break; }
        }
      }
    }
    __syncthreads();
  }
  float xv;
  if (MODE == 3) xv = __hip_atomic_load(x_in + (tid & 255), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (MODE == 2) xv = __builtin_nontemporal_load(x_in + (tid & 255));       // after the acquire: L2 was invalidated
  else xv = x_in[tid & 255];
  float acc = 0.f;
  for (int i = 0; i < iters; i += DEPTH) {
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      const float4 v = r[j];
      if (i + DEPTH + j < iters) r[j] = ld_nt(p + (long)(i + DEPTH + j) * THREADS);
      if (i + j < iters) acc += (v.x + v.y + v.z + v.w) * xv;
    }
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < THREADS / 64; ++k) s += red[k];
    s = s * 1e-3f + 1.0f;                                                          // keep the chain's values O(1)
    if (MODE == 3) {
      __hip_atomic_store(x_out + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_waitcnt(0);                                               // the write-through store has been acknowledged
      __hip_atomic_fetch_add(flag_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 2) {
      x_out[blockIdx.x] = s;
      __hip_atomic_fetch_add(flag_out, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      x_out[blockIdx.x] = s;
    }
  }
}

// ---- sweep: the plain dependent chain (hipGraph) over sizes and launch shapes: where is the fixed cost, where the bandwidth?
template <int TH, int DP>
static void sweep_one(float4* w, long cap_bytes, float* x, unsigned* err, hipStream_t s, int wgs, const int* mbs, int nmb) {
  printf("%4d workgroups x %3d threads, %2d loads in flight:", wgs, TH, DP);
  for (int m = 0; m < nmb; ++m) {
    const int iters = (int)(((long)mbs[m] * 1000000 + (long)wgs * TH * 8) / ((long)wgs * TH * 16));
    if (iters < 1) { printf("  -"); continue; }
    const long n4 = (long)wgs * TH * iters;
    const int nbuf = (int)(cap_bytes / (n4 * 16));
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < CHAIN; ++i)
      hipLaunchKernelGGL((gemv_like<0, TH, DP>), dim3(wgs), dim3(TH), 0, s, (const float4*)((char*)w + (long)(i % nbuf) * n4 * 16), iters, x + ((i + 3) % 4) * 1024,
                         x + (i % 4) * 1024, (unsigned*)nullptr, (unsigned*)nullptr, 0u, err);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ex, s)); CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ex, s));
    CK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (10.0 * CHAIN);
    printf("  %5.1f MB %6.2f us", n4 * 16 / 1e6, us);
    CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
  }
  printf("\n");
}

static int sweep() {
  const long cap = 1600L * 1000 * 1000;
  float4* w; float* x; unsigned* err;
  CK(hipMalloc(&w, cap)); CK(hipMalloc(&x, 4 * 1024 * 4)); CK(hipMalloc(&err, 4));
  {
    float* h = (float*)malloc(cap);
    unsigned sd = 1;
    for (long i = 0; i < cap / 4; ++i) { sd = sd * 1664525u + 1013904223u; h[i] = ((sd >> 8) & 0xffff) * (1.f / 65536.f) - 0.5f; }
    CK(hipMemcpy(w, h, cap, hipMemcpyHostToDevice));
    free(h);
    float one[4096]; for (int i = 0; i < 4096; ++i) one[i] = 1.f;
    CK(hipMemcpy(x, one, sizeof(one), hipMemcpyHostToDevice));
  }
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int mbs[] = {8, 17, 25, 34, 42, 50, 59, 67, 84, 100};
  const int nmb = sizeof(mbs) / sizeof(int);
  for (int wgs : {256, 512, 768}) {
    sweep_one<256, 4>(w, cap, x, err, s, wgs, mbs, nmb);
    sweep_one<256, 8>(w, cap, x, err, s, wgs, mbs, nmb);
    sweep_one<256, 16>(w, cap, x, err, s, wgs, mbs, nmb);
    sweep_one<512, 4>(w, cap, x, err, s, wgs, mbs, nmb);
    sweep_one<512, 8>(w, cap, x, err, s, wgs, mbs, nmb);
    if (wgs <= 512) sweep_one<1024, 4>(w, cap, x, err, s, wgs, mbs, nmb);
    if (wgs <= 512) sweep_one<1024, 8>(w, cap, x, err, s, wgs, mbs, nmb);
  }
  return 0;
}


// ---- xmodes: what does the ACTIVATION vector cost?  The product kernel keeps x[2][2048] (16 KB) in the registers of every wave.
//   XM 0  one float of x per thread (the sweep's kernel)
//   XM 1  every wave loads its 16 KB of x BEFORE its first weight loads      (the product kernel's order)
//   XM 2  every wave loads its 16 KB of x AFTER its first weight loads
//   XM 3  the workgroup loads x once (after the weight loads), parks it in LDS, every wave reads its copy from LDS
//   XM 4  as 3 but x is loaded before the weights
// AM 0: a load instruction of the workgroup covers TH x 16 contiguous bytes (wave w takes the w-th KiB); AM 1: a wave's DP loads are one
// contiguous DP-KiB unit (the segment kernel's order), units dealt round-robin to the waves
template <int XM, int TH, int DP, int AM = 0>
__global__ __launch_bounds__(TH) void gemv_x(const float4* __restrict__ w, int iters, const float* x_in, float* x_out) {
  const int tid = threadIdx.x, lane = tid & 63;
  const float4* p = w + (long)blockIdx.x * iters * TH + tid;
  if (AM == 1) p = w + (long)blockIdx.x * iters * TH + (long)(tid >> 6) * DP * 64 + lane;   // unit = DP*64 float4; then +j*64 inside, + NW*DP*64 per round
  __shared__ __attribute__((aligned(16))) float xs[4096];
  __shared__ float red[TH / 64];
  float4 r[DP];
  float4 xr[2][8];
  float4 xv[4096 / 4 / TH > 0 ? 4096 / 4 / TH : 1];
  const float4* x4 = reinterpret_cast<const float4*>(x_in);
  if (XM == 1) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 8; ++i) xr[b][i] = x4[b * 512 + i * 64 + lane];
  }
  if (XM == 4) {
#pragma unroll
    for (int k = 0; k < 4096 / 4 / TH; ++k) xv[k] = x4[k * TH + tid];
  }
#pragma unroll
  for (int j = 0; j < DP; ++j) r[j] = (AM == 1) ? ld_nt(p + j * 64) : ld_nt(p + (long)min(j, iters - 1) * TH);
  if (XM == 2) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 8; ++i) xr[b][i] = x4[b * 512 + i * 64 + lane];
  }
  if (XM == 3) {
#pragma unroll
    for (int k = 0; k < 4096 / 4 / TH; ++k) xv[k] = x4[k * TH + tid];
  }
  if (XM >= 3) {
#pragma unroll
    for (int k = 0; k < 4096 / 4 / TH; ++k) reinterpret_cast<float4*>(xs)[k * TH + tid] = xv[k];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 8; ++i) xr[b][i] = reinterpret_cast<const float4*>(xs)[b * 512 + i * 64 + lane];
  }
  if (XM == 0) {
    const float v = x_in[tid & 255];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 8; ++i) xr[b][i] = make_float4(v, v, v, v);
  }
  float a0 = 0.f, a1 = 0.f;
  for (int i = 0; i < iters; i += DP) {
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      const float4 v = r[j];
      if (i + DP + j < iters) r[j] = (AM == 1) ? ld_nt(p + (long)((i + DP) / DP) * (TH / 64) * DP * 64 + j * 64) : ld_nt(p + (long)(i + DP + j) * TH);
      if (i + j < iters) {
        const float4 u0 = xr[0][j % 8], u1 = xr[1][j % 8];
        a0 += v.x * u0.x + v.y * u0.y + v.z * u0.z + v.w * u0.w;
        a1 += v.x * u1.x + v.y * u1.y + v.z * u1.z + v.w * u1.w;
      }
    }
  }
  float acc = a0 + a1;
#pragma unroll
  for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) red[tid >> 6] = acc;
  __syncthreads();
  // every workgroup writes 16 floats of the next x (4096 floats = 256 workgroups x 16), bounded values
  if (tid < 16) {
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < TH / 64; ++k) s2 += red[k];
    x_out[(blockIdx.x * 16 + tid) % 4096] = 1.0f + 1e-4f * sinf(s2);
  }
}

template <int XM, int TH, int DP, int AM = 0>
static void xmode_one(float4* w, long cap_bytes, float* x, hipStream_t s, int wgs, const int* mbs, int nmb) {
  printf("x mode %d, %4d x %3d threads, %2d in flight, order %d:", XM, wgs, TH, DP, AM);
  for (int m = 0; m < nmb; ++m) {
    int iters = (int)(((long)mbs[m] * 1000000 + (long)wgs * TH * 8) / ((long)wgs * TH * 16));
    if (AM == 1) iters = (iters + DP - 1) / DP * DP;              // whole units
    const long n4 = (long)wgs * TH * iters;
    const int nbuf = (int)(cap_bytes / (n4 * 16));
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < CHAIN; ++i)
      hipLaunchKernelGGL((gemv_x<XM, TH, DP, AM>), dim3(wgs), dim3(TH), 0, s, (const float4*)((char*)w + (long)(i % nbuf) * n4 * 16), iters, x + ((i + 3) % 4) * 4096, x + (i % 4) * 4096);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ex, s)); CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ex, s));
    CK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (10.0 * CHAIN);
    printf("  %5.1f MB %6.2f us", n4 * 16 / 1e6, us);
    CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
  }
  printf("\n");
}

static int orders(float4* w, long cap, float* x, hipStream_t s) {
  const int mbs[] = {17, 50, 67};
  xmode_one<1, 512, 4, 0>(w, cap, x, s, 512, mbs, 3);
  xmode_one<1, 512, 4, 1>(w, cap, x, s, 512, mbs, 3);
  xmode_one<1, 512, 8, 0>(w, cap, x, s, 512, mbs, 3);
  xmode_one<1, 512, 8, 1>(w, cap, x, s, 512, mbs, 3);
  xmode_one<1, 256, 8, 0>(w, cap, x, s, 768, mbs, 3);
  xmode_one<1, 256, 8, 1>(w, cap, x, s, 768, mbs, 3);
  xmode_one<1, 256, 4, 1>(w, cap, x, s, 1024, mbs, 3);
  xmode_one<1, 256, 4, 0>(w, cap, x, s, 1024, mbs, 3);
  return 0;
}
static int xmodes(int which) {
  const long cap = 1600L * 1000 * 1000;
  float4* w; float* x;
  CK(hipMalloc(&w, cap)); CK(hipMalloc(&x, 4 * 4096 * 4));
  {
    float* h = (float*)malloc(cap);
    unsigned sd = 1;
    for (long i = 0; i < cap / 4; ++i) { sd = sd * 1664525u + 1013904223u; h[i] = ((sd >> 8) & 0xffff) * (1.f / 65536.f) - 0.5f; }
    CK(hipMemcpy(w, h, cap, hipMemcpyHostToDevice));
    for (int i = 0; i < 4 * 4096; ++i) h[i] = 1.f;
    CK(hipMemcpy(x, h, 4 * 4096 * 4, hipMemcpyHostToDevice));
    free(h);
  }
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  if (which == 1) return orders(w, cap, x, s);
  const int mbs[] = {17, 50, 67};
#define XALL(TH, DP, WG) xmode_one<0, TH, DP>(w, cap, x, s, WG, mbs, 3); xmode_one<1, TH, DP>(w, cap, x, s, WG, mbs, 3); xmode_one<2, TH, DP>(w, cap, x, s, WG, mbs, 3); \
                         xmode_one<3, TH, DP>(w, cap, x, s, WG, mbs, 3); xmode_one<4, TH, DP>(w, cap, x, s, WG, mbs, 3);
  XALL(256, 16, 768)
  XALL(256, 8, 768)
  XALL(256, 8, 512)
  XALL(512, 8, 256)
  XALL(512, 4, 512)
  XALL(1024, 4, 256)
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 's') return sweep();
  if (argc > 1 && argv[1][0] == 'x') return xmodes(0);
  if (argc > 1 && argv[1][0] == 'o') return xmodes(1);
  const int MB = argc > 1 ? atoi(argv[1]) : 50;
  const int iters = (int)(((long)MB * 1000000 / (WGS * THREADS * 16) + DEPTH - 1) / DEPTH * DEPTH);
  const long n4 = (long)WGS * THREADS * iters;                                    // float4 per launch
  const int NBUF = 16;
  float4* w; float* x; unsigned *flags, *err;
  CK(hipMalloc(&w, n4 * 16 * NBUF));
  CK(hipMalloc(&x, 4 * 256 * 4)); CK(hipMalloc(&flags, CHAIN * 4)); CK(hipMalloc(&err, 4));
  {
    float* h = (float*)malloc(n4 * 16);
    unsigned sd = 1;
    for (long i = 0; i < n4 * 4; ++i) { sd = sd * 1664525u + 1013904223u; h[i] = ((sd >> 8) & 0xffff) * (1.f / 65536.f) - 0.5f; }
    for (int b = 0; b < NBUF; ++b) CK(hipMemcpy((char*)w + (long)b * n4 * 16, h, n4 * 16, hipMemcpyHostToDevice));
    free(h);
  }
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  printf("%d launches per chain, %.1f MB of weights per launch (%d iterations x %d workgroups x %d threads x 16 B)\n", CHAIN, n4 * 16 / 1e6, iters, WGS, THREADS);

  auto reset = [&]() {
    float one[256]; for (int i = 0; i < 256; ++i) one[i] = 1.f + i * 1e-3f;
    for (int b = 0; b < 4; ++b) CK(hipMemcpy(x + b * 256, one, 1024, hipMemcpyHostToDevice));
    CK(hipMemset(flags, 0, CHAIN * 4)); CK(hipMemset(err, 0, 4));
  };
  auto result = [&](const char* name, double us, int reps) {
    float h[256]; unsigned e;
    CK(hipMemcpy(h, x + ((CHAIN - 1) % 4) * 256, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
    double cs = 0; for (int i = 0; i < 256; ++i) cs += h[i];
    printf("%-58s %7.2f us per launch  (%.2f TB/s)  checksum %.6f  spin time-outs %u\n", name, us / (reps * CHAIN), n4 * 16 / (us / (reps * CHAIN)) / 1e6, cs, e);
  };
  const int REPS = 20;
  // ---- mode 0: one stream, eager
  {
    reset();
    auto run = [&](int rep) {
      for (int i = 0; i < CHAIN; ++i)
        hipLaunchKernelGGL(gemv_like<0>, dim3(WGS), dim3(THREADS), 0, sa, (const float4*)((char*)w + (long)(i % NBUF) * n4 * 16), iters, x + ((i + 3) % 4) * 256,
                           x + (i % 4) * 256, (unsigned*)nullptr, (unsigned*)nullptr, 0u, err);
    };
    run(0); CK(hipDeviceSynchronize()); reset();
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REPS; ++r) run(r);
    CK(hipDeviceSynchronize());
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    result("mode 0: one stream, dependent launches (eager)", us, REPS);
  }
  // ---- mode 1: graph
  {
    reset();
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < CHAIN; ++i)
      hipLaunchKernelGGL(gemv_like<0>, dim3(WGS), dim3(THREADS), 0, sa, (const float4*)((char*)w + (long)(i % NBUF) * n4 * 16), iters, x + ((i + 3) % 4) * 256,
                         x + (i % 4) * 256, (unsigned*)nullptr, (unsigned*)nullptr, 0u, err);
    CK(hipStreamEndCapture(sa, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ex, sa)); CK(hipDeviceSynchronize()); reset();
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ex, sa));
    CK(hipDeviceSynchronize());
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    result("mode 1: the same chain as a hipGraph", us, REPS);
  }
  // ---- modes 2, 3: two streams alternating, flag handshake
  for (int mode = 2; mode <= 3; ++mode) {
    reset();
    unsigned epoch = 0;
    auto run = [&]() {
      // launch i waits for flags[i-1] == WGS * (epoch+1); the first launch of a chain waits for the LAST launch of the previous chain
      for (int i = 0; i < CHAIN; ++i) {
        hipStream_t s = (i & 1) ? sb : sa;
        unsigned* fin = i ? flags + (i - 1) : (epoch ? flags + (CHAIN - 1) : (unsigned*)nullptr);
        const unsigned expect = WGS * (i ? epoch + 1 : epoch);
        const float4* wp = (const float4*)((char*)w + (long)(i % NBUF) * n4 * 16);
        if (mode == 2) hipLaunchKernelGGL(gemv_like<2>, dim3(WGS), dim3(THREADS), 0, s, wp, iters, x + ((i + 3) % 4) * 256, x + (i % 4) * 256, fin, flags + i, expect, err);
        else           hipLaunchKernelGGL(gemv_like<3>, dim3(WGS), dim3(THREADS), 0, s, wp, iters, x + ((i + 3) % 4) * 256, x + (i % 4) * 256, fin, flags + i, expect, err);
      }
      ++epoch;
    };
    run(); CK(hipDeviceSynchronize()); reset(); epoch = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REPS; ++r) run();
    CK(hipDeviceSynchronize());
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    result(mode == 2 ? "mode 2: two streams, release/acquire flag handshake" : "mode 3: two streams, relaxed agent-scope atomics for x and flag", us, REPS);
  }
  return 0;
}
