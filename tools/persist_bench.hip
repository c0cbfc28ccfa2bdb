// tools/persist_bench.hip — feasibility study for a PERSISTENT decode step: 16 layers x 4 dependent weight-streaming
// GEMVs (50 + 17 + 67 + 67 MB, K = 2048, B = 2) either as 64 dependent kernel launches or as ONE launch with grid
// barriers, where every wave requests its first two weight rows of phase p+1 BEFORE waiting at the barrier that ends
// phase p (weights do not depend on activations), so the barrier latency overlaps HBM traffic.
// Every spin is bounded (abort flag) — a deadlock must not hang the GPU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <stdarg.h>
#include <vector>
#include "../ssr-speech_amd/csrc/common.h"
void ssrhip_set_error(const char*, ...) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int K = 2048;
struct Phase { const float* W; const float* in; float* out; int N; int in_stride; };   // out[b][n] = sum_k W[n][k] in[b][(n%slices)...]
struct Plan { Phase ph[64]; int n; unsigned* bar; unsigned* abort_flag; };

__device__ __forceinline__ void load_row(float4 (&w)[8], const float* p, int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = ld_nt(p + (i * 64 + lane) * 4);
}

__device__ __forceinline__ void consume(const float4 (&w)[8], const float4 (&xr)[2][8], float* out, int N, int n, int lane) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { s0 = dot4(w[i], xr[0][i], s0); s1 = dot4(w[i], xr[1][i], s1); }
  s0 = wave_sum(s0); s1 = wave_sum(s1);
  if (lane == 0) { out[n] = s0 * 1e-3f; out[N + n] = s1 * 1e-3f; }
}

// ---- baseline: one launch per phase
__global__ __launch_bounds__(256, 3) void phase_kernel(Phase p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int groups = gridDim.x * 4, G = blockIdx.x * 4 + wave;
  float4 wa[8], wb[8], xr[2][8];
  int na = G, nb = G + groups;
  if (na < p.N) load_row(wa, p.W + (size_t)na * K, lane);
  if (nb < p.N) load_row(wb, p.W + (size_t)nb * K, lane);
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 8; ++i) xr[b][i] = ld4(p.in + b * p.in_stride + (i * 64 + lane) * 4);
  while (na < p.N) {
    consume(wa, xr, p.out, p.N, na, lane);
    na += 2 * groups;
    if (na < p.N) load_row(wa, p.W + (size_t)na * K, lane);
    if (nb >= p.N) break;
    consume(wb, xr, p.out, p.N, nb, lane);
    nb += 2 * groups;
    if (nb < p.N) load_row(wb, p.W + (size_t)nb * K, lane);
  }
}

// ---- grid barriers
constexpr unsigned SPIN_LIMIT = 4000000u;
__device__ __forceinline__ unsigned ld_relaxed(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// flat: one monotonic counter
// wait for this wave's STORES but not for the `npre` prefetch loads issued after them (vmcnt counts both, in order)
__device__ __forceinline__ void drain_stores(int npre) {
  if (npre >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (npre >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void barrier_flat(unsigned* bar, unsigned target, unsigned* abort_flag, int npre = 0) {
  drain_stores(npre);
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (ld_relaxed(bar) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > SPIN_LIMIT || ld_relaxed(abort_flag)) { __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// hierarchical: bar[0] = top counter, bar[16+x] = arrivals of XCD x, bar[32+x] = population of XCD x, bar[48+x] = generation of XCD x
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7u; }

__device__ __forceinline__ void barrier_xcd(unsigned* bar, unsigned epoch, unsigned n_xcd, unsigned xcc, unsigned pop, unsigned* abort_flag, int npre) {
  drain_stores(npre);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(bar + 16 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    if (old + 1 == pop * epoch) {                 // last arriver of this XCD: publish the XCD's L2, meet the other leaders
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (ld_relaxed(bar) < n_xcd * epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT || ld_relaxed(abort_flag)) { __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
      __hip_atomic_store(bar + 48 + xcc, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (ld_relaxed(bar + 48 + xcc) < epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT || ld_relaxed(abort_flag)) { __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <int MODE>   // 0 flat barrier, 1 xcd barrier
__global__ __launch_bounds__(256, 3) void persistent_kernel(Plan pl) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int groups = gridDim.x * 4, G = blockIdx.x * 4 + wave;
  unsigned xcc = 0, pop = 0, n_xcd = 0;
  unsigned epoch = 0;
  if (MODE == 1) {
    xcc = xcc_id();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(pl.bar + 32 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    barrier_flat(pl.bar + 8, gridDim.x, pl.abort_flag);        // census
    pop = ld_relaxed(pl.bar + 32 + xcc);
    for (int i = 0; i < 8; ++i) n_xcd += ld_relaxed(pl.bar + 32 + i) ? 1u : 0u;
  }
  float4 wa[8], wb[8], xr[2][8];
  int na = G, nb = G + groups;
  {
    const Phase& p = pl.ph[0];
    if (na < p.N) load_row(wa, p.W + (size_t)na * K, lane);
    if (nb < p.N) load_row(wb, p.W + (size_t)nb * K, lane);
  }
  for (int ip = 0; ip < pl.n; ++ip) {
    const Phase p = pl.ph[ip];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 8; ++i) xr[b][i] = ld4(p.in + b * p.in_stride + (i * 64 + lane) * 4);
    while (na < p.N) {
      consume(wa, xr, p.out, p.N, na, lane);
      na += 2 * groups;
      if (na < p.N) load_row(wa, p.W + (size_t)na * K, lane);
      if (nb >= p.N) break;
      consume(wb, xr, p.out, p.N, nb, lane);
      nb += 2 * groups;
      if (nb < p.N) load_row(wb, p.W + (size_t)nb * K, lane);
    }
    // request the first rows of the NEXT phase, then wait for everyone's outputs of this one
    na = G; nb = G + groups;
    int npre = 0;
    if (ip + 1 < pl.n) {
      const Phase& q = pl.ph[ip + 1];
      if (na < q.N) { load_row(wa, q.W + (size_t)na * K, lane); npre += 8; }
      if (nb < q.N) { load_row(wb, q.W + (size_t)nb * K, lane); npre += 8; }
    }
    ++epoch;
    if (MODE == 0) barrier_flat(pl.bar, epoch * gridDim.x, pl.abort_flag, npre);
    else barrier_xcd(pl.bar, epoch, n_xcd, xcc, pop, pl.abort_flag, npre);
    if (ld_relaxed(pl.abort_flag)) return;
  }
}

int main(int argc, char** argv) {
  const int L = 16;
  const int Ns[4] = {6144, 2048, 8192, 8192};
  std::vector<float*> Wb;
  Plan pl; memset(&pl, 0, sizeof(pl));
  float *x0, *x1, *xz;
  CK(hipMalloc(&x0, 2 * 8192 * 4)); CK(hipMalloc(&x1, 2 * 8192 * 4)); CK(hipMalloc(&xz, 2 * 8192 * 4));
  std::vector<float> hx(2 * 8192);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 37) % 101) / 101.f - 0.5f;
  for (int l = 0; l < L; ++l)
    for (int j = 0; j < 4; ++j) {
      float* W; size_t n = (size_t)Ns[j] * K;
      CK(hipMalloc(&W, n * 4));
      std::vector<float> hw(1 << 16);
      for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)(((i + l * 7 + j * 3) * 29) % 113) / 113.f - 0.5f;
      for (size_t off = 0; off < n; off += hw.size()) CK(hipMemcpy(W + off, hw.data(), std::min(hw.size(), n - off) * 4, hipMemcpyHostToDevice));
      Wb.push_back(W);
      Phase& p = pl.ph[l * 4 + j];
      p.W = W; p.N = Ns[j]; p.in_stride = 8192;
      p.in = ((l * 4 + j) & 1) ? x1 : x0;
      p.out = ((l * 4 + j) & 1) ? x0 : x1;
    }
  pl.n = L * 4;
  CK(hipMalloc(&pl.bar, 256 * 4)); CK(hipMalloc(&pl.abort_flag, 4));
  double mb = 0; for (int j = 0; j < 4; ++j) mb += (double)Ns[j] * K * 4 * L / 1e6;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ref(2 * 8192), got(2 * 8192);
  for (int blocks : {512, 768}) {
    // baseline
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipMemcpy(x0, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x1, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < pl.n; ++i) hipLaunchKernelGGL(phase_kernel, dim3(blocks), dim3(256), 0, 0, pl.ph[i]);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) best = fminf(best, ms);
    }
    CK(hipMemcpy(ref.data(), x0, ref.size() * 4, hipMemcpyDeviceToHost));
    printf("blocks %d | %d launches: %.1f us (%.2f TB/s, %.2f us/phase)\n", blocks, pl.n, 1000 * best, mb / (1000 * best), 1000 * best / pl.n);
    for (int mode = 0; mode < 2; ++mode) {
      best = 1e9; unsigned ab = 0; double maxdiff = 0;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemcpy(x0, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x1, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(pl.bar, 0, 256 * 4)); CK(hipMemset(pl.abort_flag, 0, 4));
        CK(hipEventRecord(e0, 0));
        if (mode == 0) hipLaunchKernelGGL(persistent_kernel<0>, dim3(blocks), dim3(256), 0, 0, pl);
        else hipLaunchKernelGGL(persistent_kernel<1>, dim3(blocks), dim3(256), 0, 0, pl);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) best = fminf(best, ms);
        CK(hipMemcpy(&ab, pl.abort_flag, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(got.data(), x0, got.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < got.size(); ++i) maxdiff = fmax(maxdiff, fabs((double)got[i] - ref[i]));
        if (ab) break;
      }
      printf("blocks %d | persistent %s barrier: %.1f us (%.2f TB/s, %.2f us/phase) abort=%u maxdiff=%.3g\n", blocks, mode ? "xcd " : "flat", 1000 * best,
             mb / (1000 * best), 1000 * best / pl.n, ab, maxdiff);
    }
  }
  return 0;
}
