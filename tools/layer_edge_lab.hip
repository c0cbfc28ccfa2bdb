// tools/layer_edge_lab.hip — ONE all-to-all edge of the decode layer inside a launch, priced on this machine (tier B:
// "fuse only the two untried 32 KB edges ... or a lab log with per-edge costs that shows why the two-edge fusion loses").
//
// The edge: [FFN2 of layer l  (N = 2048, K = 8192, + bias + residual)]  ->  every CU needs all 2 x 2048 outputs  ->  [LN1 + QKV of layer
// l + 1 (N = 6144, K = 2048)]. Two forms, same arithmetic, outputs compared bit for bit:
//   A. two launches in a dependent hipGraph chain (the product's shape: one 8-wave workgroup per CU, (row, 1024-float segment) units, 4 units
//      in flight per wave, LayerNorm statistics per segment merged through LDS);
//   B. ONE launch of 256 workgroups x 9 waves. Waves 0-7 stream FFN2's units exactly as in A and park the partial sums; behind the barrier
//      they post their first QKV units (the weights do not depend on the edge) and wait; wave 8 (the "edge wave") finishes the workgroup's
//      8 rows x 2 outputs, PUBLISHES them as 8-byte {value, tag} granules with write-through (sc1) stores — the guide's R2 hand-off — and
//      GATHERS all 4096 granules with sc1 loads (64 per lane, swept until every tag is valid; bounded), puts x' into LDS and releases the
//      streaming waves, which normalise and run the QKV units. Tags: two granule buffers alternate between consecutive launches, launch i
//      uses buffer i & 1 and poisons buffer (i + 1) & 1 with plain stores (visible behind the kernel boundary), so a tag of 1 is always
//      this launch's.
// Printed: us per launch (graph-chained, best of 5 x 64) for A's two kernels and for B; B's wave-8 stamps — last FFN2 partial parked,
// granules published, gather complete (min / median / max over CUs), number of sweeps — and the bit-compare.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I ssr-speech_amd/csrc -I include tools/layer_edge_lab.hip -o tools/bin/layer_edge_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "common.h"
void ssrhip_set_error(const char*, ...) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int D = 2048, F = 8192, NQ = 6144, SEGF = 1024;
constexpr int G = 256;                        // workgroups = CUs
constexpr int RA = D / G, RB = NQ / G;        // rows per workgroup: 8 (FFN2), 24 (QKV)
constexpr int NUA = RA * 8 / 8, NUB = RB * 2 / 8;   // units per wave: FFN2 8 rows x 8 segments / 8 waves = 8; QKV 24 x 2 / 8 = 6
constexpr int DEPTH = 4;
typedef float v4f __attribute__((ext_vector_type(4)));

struct Args {
  const float* W2; const float* b2; const float* h;        // FFN2: W2 [D][F], bias, input h [2][F]
  float* x;                                                // residual stream [2][D]: read (residual) and written (x')
  const float* Wq; const float* bq; float* q;              // QKV: Wq [NQ][D] (LayerNorm gamma / beta folded), bias, output [2][NQ]
  unsigned long long* gran;                                // fused: this launch's granule buffer [2 * D] of {value, tag}
  unsigned long long* gran_next;                           // fused: the other buffer, poisoned by this launch
  long long* prof;                                         // fused: [G][4] stamps of wave 8, or NULL
  int* giveup;
};

// ---- the units of one GEMV, as the product's gemv_segu_kernel runs them (S segments of 1024 floats, wave w owns segment w & (S - 1))
template <int S, int NUW, int K, int J0 = 0, int J1 = DEPTH>
__device__ __forceinline__ void first_units(const float* Wg, int wave, float4 (&w)[DEPTH][4]) {
  constexpr int sh = (S == 8) ? 3 : 1;
#pragma unroll
  for (int j = J0; j < J1; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) w[j][i] = ld_nt(Wg + (size_t)((wave + 8 * j) >> sh) * K + i * 256);
}
template <int S, int NUW, int K>
__device__ __forceinline__ void run_units(const float* Wg, int wave, int lane, const float4 (&xr)[2][4], float4 (&w)[DEPTH][4], float* part) {
  constexpr int sh = (S == 8) ? 3 : 1;
#pragma unroll
  for (int j = 0; j < NUW; ++j) {
    float4 (&wj)[4] = w[j % DEPTH];
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[b][i & 1] = dot4(wj[i], xr[b][i], acc[b][i & 1]);
      if (j + DEPTH < NUW) {
        __builtin_amdgcn_sched_barrier(0);
        wj[i] = ld_nt(Wg + (size_t)((wave + 8 * (j + DEPTH)) >> sh) * K + i * 256);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float mine = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float sum = wave_sum(acc[b][0] + acc[b][1]);
      if (lane == b) mine = sum;
    }
    if (lane < 2) part[(wave + 8 * j) * 2 + lane] = mine;
  }
}
// LayerNorm of the wave's segment (two segments per row: K = 2048), statistics exchanged through `aux`
__device__ __forceinline__ void layernorm2(float4 (&xr)[2][4], int wave, int lane, float* aux) {
  float m[2], q[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float s0 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s0 += (xr[b][i].x + xr[b][i].y) + (xr[b][i].z + xr[b][i].w);
    m[b] = wave_sum(s0) * (1.0f / SEGF);
    float q0 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dx = xr[b][i].x - m[b], dy = xr[b][i].y - m[b], dz = xr[b][i].z - m[b], dw = xr[b][i].w - m[b];
      q0 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    q[b] = wave_sum(q0);
    if (wave < 2 && lane == 0) { aux[(wave * 2 + b) * 2] = m[b]; aux[(wave * 2 + b) * 2 + 1] = q[b]; }
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float mean = (aux[(0 * 2 + b) * 2] + aux[(1 * 2 + b) * 2]) / 2.0f;
    const float d0 = aux[(0 * 2 + b) * 2] - mean, d1 = aux[(1 * 2 + b) * 2] - mean;
    const float M2 = aux[(0 * 2 + b) * 2 + 1] + aux[(1 * 2 + b) * 2 + 1];
    const float dev = fmaf(d1, d1, fmaf(d0, d0, 0.f));
    const float rstd = 1.0f / sqrtf((M2 + (float)SEGF * dev) / (float)D + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      xr[b][i] = make_float4((xr[b][i].x - mean) * rstd, (xr[b][i].y - mean) * rstd, (xr[b][i].z - mean) * rstd, (xr[b][i].w - mean) * rstd);
  }
}

// ---- A1: FFN2 + bias + residual (8 waves)
__global__ __launch_bounds__(512, 2) void ffn2_kernel(const Args a) {
  __shared__ float part[RA * 8 * 2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r0 = blockIdx.x * RA;
  const float* Wg = a.W2 + (size_t)r0 * F + wave * SEGF + lane * 4;        // S = 8: the wave's segment is `wave`
  float4 xr[2][4];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) xr[b][i] = ld4(a.h + (size_t)b * F + wave * SEGF + (i * 64 + lane) * 4);
  float4 w[DEPTH][4];
  first_units<8, NUA, F>(Wg, wave, w);
  run_units<8, NUA, F>(Wg, wave, lane, xr, w, part);
  __syncthreads();
  if (t < RA * 2) {
    const int r = t >> 1, b = t & 1;
    float v = 0.f;
    for (int s = 0; s < 8; ++s) v += part[(r * 8 + s) * 2 + b];
    a.x[(size_t)b * D + r0 + r] = a.x[(size_t)b * D + r0 + r] + (v + a.b2[r0 + r]);
  }
}
// ---- A2: LayerNorm + QKV (8 waves)
__global__ __launch_bounds__(512, 2) void qkv_kernel(const Args a) {
  __shared__ float part[RB * 2 * 2];
  __shared__ float aux[8];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r0 = blockIdx.x * RB, seg = wave & 1;
  const float* Wg = a.Wq + (size_t)r0 * D + seg * SEGF + lane * 4;
  float4 xr[2][4];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) xr[b][i] = ld4(a.x + (size_t)b * D + seg * SEGF + (i * 64 + lane) * 4);
  float4 w[DEPTH][4];
  first_units<2, NUB, D>(Wg, wave, w);
  layernorm2(xr, wave, lane, aux);
  run_units<2, NUB, D>(Wg, wave, lane, xr, w, part);
  __syncthreads();
  if (t < RB * 2) {
    const int r = t >> 1, b = t & 1;
    a.q[(size_t)b * NQ + r0 + r] = (part[(r * 2 + 0) * 2 + b] + part[(r * 2 + 1) * 2 + b]) + a.bq[r0 + r];
  }
}

// eight 16-byte write-through loads of the granule sweep, issued together, waited for together (the compiler must not see them):
// 8 x 64 lanes x 2 granules = 1024 granules per call, contiguous KiB per wave-level load
__device__ __forceinline__ void sweep8(const unsigned long long* p, v4f (&g)[8]) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:1024 sc1\n\t"
      "global_load_dwordx4 %2, %8, off offset:2048 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:3072 sc1\n\t"
      "global_load_dwordx4 %4, %9, off sc1\n\tglobal_load_dwordx4 %5, %9, off offset:1024 sc1\n\t"
      "global_load_dwordx4 %6, %9, off offset:2048 sc1\n\tglobal_load_dwordx4 %7, %9, off offset:3072 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7])
      : "v"(p), "v"(p + 512)
      : "memory");
}

// four 16-byte write-through loads: one streaming wave's eighth of the sweep (mode 3)
__device__ __forceinline__ void sweep4(const unsigned long long* p, v4f (&g)[4]) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
      "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:3072 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3])
      : "v"(p)
      : "memory");
}

// ---- B: both GEMVs in one launch, the edge inside it (9 waves). PRE = when the streaming waves request QKV's first units, and who gathers:
//   0  behind barrier (2), i.e. no overlap at all: the launch boundary is replaced by publish + gather, nothing else changes
//   1  behind a barrier that follows the edge wave's publish (the publish does not queue behind 128 KB of requests; the gather does)
//   2  right behind barrier (1) (publish AND gather queue behind the requests)
//   3  as 0, but the eight streaming waves gather an eighth each (the edge wave only publishes)
//   NE = number of edge waves (1 or 4: each gathers a quarter, one round trip per sweep); PF = how many of the 4 first units are requested early
template <int PRE, int NE, int PF>
__global__ __launch_bounds__(512 + 64 * NE, 2) void fused_kernel(const Args a) {
  __shared__ float partA[RA * 8 * 2];
  __shared__ float partB[RB * 2 * 2];
  __shared__ float aux[8];
  __shared__ __attribute__((aligned(16))) float xs[2 * D];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float4 xr[2][4];
  float4 w[DEPTH][4];
  float e_resid = 0.f, e_bias = 0.f;
  if (wave < 8) {
    const int r0 = blockIdx.x * RA;
    const float* Wg = a.W2 + (size_t)r0 * F + wave * SEGF + lane * 4;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) xr[b][i] = ld4(a.h + (size_t)b * F + wave * SEGF + (i * 64 + lane) * 4);
    first_units<8, NUA, F>(Wg, wave, w);
    run_units<8, NUA, F>(Wg, wave, lane, xr, w, partA);
  } else if (wave == 8) {
    // the edge wave: poison the OTHER granule buffer (this workgroup's 16 granules of it), fetch what its finalisation needs
    if (lane < RA * 2) {
      a.gran_next[(size_t)blockIdx.x * (RA * 2) + lane] = 0ull;
      e_resid = a.x[(size_t)(lane & 1) * D + blockIdx.x * RA + (lane >> 1)];
      e_bias = a.b2[blockIdx.x * RA + (lane >> 1)];
    }
  }
  __syncthreads();                                                        // (1) FFN2's partial sums are parked
  const int segB = wave & 1, rB0 = blockIdx.x * RB;
  const float* WgB = a.Wq + (size_t)rB0 * D + segB * SEGF + lane * 4;
  long long t_parked = 0, t_pub = 0, t_ready = 0;
  int sweeps = 0;
  bool done = false;
  if (wave < 8) {
    if (PRE == 2) first_units<2, NUB, D, 0, PF>(WgB, wave, w);            // the next matrix does not depend on the edge
  } else if (wave == 8) {
    if (a.prof && lane == 0) t_parked = wall_clock64();
    // finish this workgroup's 8 rows x 2 outputs: granule index = (row0 + r) * 2 + b, i.e. one workgroup's 16 granules are contiguous
    if (lane < RA * 2) {
      const int r = lane >> 1, b = lane & 1, n = blockIdx.x * RA + r;
      float v = 0.f;
      for (int s = 0; s < 8; ++s) v += partA[(r * 8 + s) * 2 + b];
      const float out = e_resid + (v + e_bias);
      a.x[(size_t)b * D + n] = out;                                       // the residual stream itself (later launches read it the ordinary way)
      const unsigned long long gval = ((unsigned long long)1u << 32) | (unsigned long long)__float_as_uint(out);
      __hip_atomic_store(a.gran + (size_t)n * 2 + b, gval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // 8-byte write-through store: {value, tag = 1}
    }
    if (a.prof && lane == 0) t_pub = wall_clock64();
  }
  if (PRE == 1) {
    __syncthreads();                                                      // (1b) the publish has been issued
    if (wave < 8) first_units<2, NUB, D, 0, PF>(WgB, wave, w);
  }
  if (PRE != 3 && wave >= 8) {
    constexpr int NP = 4 / NE;                                            // passes of 1024 granules per edge wave
    // gather: pass p (0..3) covers granules [1024 p, 1024 p + 1024): 8 loads x 64 lanes x 2 granules, a contiguous KiB per wave-level load
    v4f g[8];
    for (int spin = 0; spin < 4000 && !done; ++spin) {
      bool all = true;
#pragma unroll 1
      for (int pass = (wave - 8) * NP; pass < (wave - 8) * NP + NP; ++pass) {
        sweep8(a.gran + (size_t)pass * 1024 + lane * 2, g);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const unsigned tag0 = __float_as_uint(g[i][1]), tag1 = __float_as_uint(g[i][3]);
          all = all && (tag0 == 1u) && (tag1 == 1u);
          // granule index = pass * 1024 + (i / 4) * 512 + (i % 4) * 128 + lane * 2 + {0, 1}; value -> xs[b][n], n = index >> 1, b = index & 1
          const int gi = pass * 1024 + (i >> 2) * 512 + (i & 3) * 128 + lane * 2;
          xs[(gi >> 1)] = g[i][0];
          xs[D + (gi >> 1)] = g[i][2];
        }
      }
      ++sweeps;
      done = __all(all);
      if (!done) __builtin_amdgcn_s_sleep(2);
    }
    if (!done && lane == 0) *a.giveup = 1;
  }
  if (PRE == 3 && wave < 8) {
    v4f g[4];
    for (int spin = 0; spin < 4000 && !done; ++spin) {
      bool all = true;
      sweep4(a.gran + (size_t)wave * 512 + lane * 2, g);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        all = all && (__float_as_uint(g[i][1]) == 1u) && (__float_as_uint(g[i][3]) == 1u);
        const int gi = wave * 512 + i * 128 + lane * 2;
        xs[(gi >> 1)] = g[i][0];
        xs[D + (gi >> 1)] = g[i][2];
      }
      ++sweeps;
      done = __all(all);
      if (!done) __builtin_amdgcn_s_sleep(1);
    }
    if (!done && lane == 0) *a.giveup = 1;
  }
  __syncthreads();                                                        // (2) x' is in LDS
  if (a.prof && lane == 0 && (wave == 8 || (PRE == 3 && wave == 0))) {
    long long* pr = a.prof + (size_t)blockIdx.x * 4;
    if (wave == 8) { pr[0] = t_parked; pr[1] = t_pub; pr[2] = wall_clock64(); }
    if (PRE != 3 || wave == 0) pr[3] = sweeps;
  }
  if (wave < 8) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) xr[b][i] = *reinterpret_cast<const float4*>(xs + b * D + segB * SEGF + (i * 64 + lane) * 4);
    if (PRE == 0 || PRE == 3) first_units<2, NUB, D>(WgB, wave, w);
    else if (PF < DEPTH) first_units<2, NUB, D, PF, DEPTH>(WgB, wave, w);
  }
  // (3) LayerNorm: all 9 waves take part in its barrier; the edge wave's registers are dummies
  if (wave >= 8) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) xr[b][i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  layernorm2(xr, wave, lane, aux);
  if (wave < 8) run_units<2, NUB, D>(WgB, wave, lane, xr, w, partB);
  __syncthreads();                                                        // (4)
  if (t < RB * 2) {
    const int r = t >> 1, b = t & 1;
    a.q[(size_t)b * NQ + rB0 + r] = (partB[(r * 2 + 0) * 2 + b] + partB[(r * 2 + 1) * 2 + b]) + a.bq[rB0 + r];
  }
}
// ---- C: mode 6 / 7 again with the two ROLES as the two arms of one branch, every barrier written in both arms. In the form above the
// streaming waves' code is a series of `if (wave < 8)` blocks with joins between them; hipcc's wait-count pass merges the "block skipped"
// path into every join, so the first use of a unit requested two blocks earlier is guarded by `s_waitcnt vmcnt(PF * 4 - 1)` instead of
// vmcnt(15) — the QKV phase runs with 8 (4) loads in flight per wave instead of 16 (read off the ISA). One arm per role: straight-line
// code for the streaming waves, exact counts.
__device__ __forceinline__ void ln_stats(const float4 (&xr)[2][4], int wave, int lane, float* aux) {
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float s0 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s0 += (xr[b][i].x + xr[b][i].y) + (xr[b][i].z + xr[b][i].w);
    const float m = wave_sum(s0) * (1.0f / SEGF);
    float q0 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dx = xr[b][i].x - m, dy = xr[b][i].y - m, dz = xr[b][i].z - m, dw = xr[b][i].w - m;
      q0 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    const float q = wave_sum(q0);
    if (wave < 2 && lane == 0) { aux[(wave * 2 + b) * 2] = m; aux[(wave * 2 + b) * 2 + 1] = q; }
  }
}
__device__ __forceinline__ void ln_apply(float4 (&xr)[2][4], const float* aux) {
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float mean = (aux[(0 * 2 + b) * 2] + aux[(1 * 2 + b) * 2]) / 2.0f;
    const float d0 = aux[(0 * 2 + b) * 2] - mean, d1 = aux[(1 * 2 + b) * 2] - mean;
    const float M2 = aux[(0 * 2 + b) * 2 + 1] + aux[(1 * 2 + b) * 2 + 1];
    const float dev = fmaf(d1, d1, fmaf(d0, d0, 0.f));
    const float rstd = 1.0f / sqrtf((M2 + (float)SEGF * dev) / (float)D + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      xr[b][i] = make_float4((xr[b][i].x - mean) * rstd, (xr[b][i].y - mean) * rstd, (xr[b][i].z - mean) * rstd, (xr[b][i].w - mean) * rstd);
  }
}
template <int PF>
__global__ __launch_bounds__(768, 2) void roles_kernel(const Args a) {
  __shared__ float partA[RA * 8 * 2];
  __shared__ float partB[RB * 2 * 2];
  __shared__ float aux[8];
  __shared__ __attribute__((aligned(16))) float xs[2 * D];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  if (wave < 8) {
    float4 xr[2][4];
    float4 w[DEPTH][4];
    const int r0 = blockIdx.x * RA, segB = wave & 1, rB0 = blockIdx.x * RB;
    const float* Wg = a.W2 + (size_t)r0 * F + wave * SEGF + lane * 4;
    const float* WgB = a.Wq + (size_t)rB0 * D + segB * SEGF + lane * 4;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) xr[b][i] = ld4(a.h + (size_t)b * F + wave * SEGF + (i * 64 + lane) * 4);
    first_units<8, NUA, F>(Wg, wave, w);
    run_units<8, NUA, F>(Wg, wave, lane, xr, w, partA);
    __syncthreads();                                                      // (1)
    __syncthreads();                                                      // (1b)
    first_units<2, NUB, D, 0, PF>(WgB, wave, w);
    __syncthreads();                                                      // (2)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) xr[b][i] = *reinterpret_cast<const float4*>(xs + b * D + segB * SEGF + (i * 64 + lane) * 4);
    if (PF < DEPTH) first_units<2, NUB, D, PF, DEPTH>(WgB, wave, w);
    ln_stats(xr, wave, lane, aux);
    __syncthreads();                                                      // (3)
    ln_apply(xr, aux);
    run_units<2, NUB, D>(WgB, wave, lane, xr, w, partB);
    __syncthreads();                                                      // (4)
    if (t < RB * 2) {
      const int r = t >> 1, b = t & 1;
      a.q[(size_t)b * NQ + rB0 + r] = (partB[(r * 2 + 0) * 2 + b] + partB[(r * 2 + 1) * 2 + b]) + a.bq[rB0 + r];
    }
  } else {
    float e_resid = 0.f, e_bias = 0.f;
    long long t_parked = 0, t_pub = 0;
    if (wave == 8 && lane < RA * 2) {
      a.gran_next[(size_t)blockIdx.x * (RA * 2) + lane] = 0ull;
      e_resid = a.x[(size_t)(lane & 1) * D + blockIdx.x * RA + (lane >> 1)];
      e_bias = a.b2[blockIdx.x * RA + (lane >> 1)];
    }
    __syncthreads();                                                      // (1)
    if (wave == 8) {
      if (a.prof && lane == 0) t_parked = wall_clock64();
      if (lane < RA * 2) {
        const int r = lane >> 1, b = lane & 1, n = blockIdx.x * RA + r;
        float v = 0.f;
        for (int s = 0; s < 8; ++s) v += partA[(r * 8 + s) * 2 + b];
        const float out = e_resid + (v + e_bias);
        a.x[(size_t)b * D + n] = out;
        const unsigned long long gval = ((unsigned long long)1u << 32) | (unsigned long long)__float_as_uint(out);
        __hip_atomic_store(a.gran + (size_t)n * 2 + b, gval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (a.prof && lane == 0) t_pub = wall_clock64();
    }
    __syncthreads();                                                      // (1b)
    bool done = false;
    int sweeps = 0;
    v4f g[8];
    for (int spin = 0; spin < 4000 && !done; ++spin) {
      bool all = true;
      sweep8(a.gran + (size_t)(wave - 8) * 1024 + lane * 2, g);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        all = all && (__float_as_uint(g[i][1]) == 1u) && (__float_as_uint(g[i][3]) == 1u);
        const int gi = (wave - 8) * 1024 + (i >> 2) * 512 + (i & 3) * 128 + lane * 2;
        xs[(gi >> 1)] = g[i][0];
        xs[D + (gi >> 1)] = g[i][2];
      }
      ++sweeps;
      done = __all(all);
      if (!done) __builtin_amdgcn_s_sleep(2);
    }
    if (!done && lane == 0) *a.giveup = 1;
    __syncthreads();                                                      // (2)
    if (a.prof && lane == 0 && wave == 8) {
      long long* pr = a.prof + (size_t)blockIdx.x * 4;
      pr[0] = t_parked; pr[1] = t_pub; pr[2] = wall_clock64(); pr[3] = sweeps;
    }
    __syncthreads();                                                      // (3)
    __syncthreads();                                                      // (4)
  }
}
typedef void (*fused_fn)(const Args);
struct Mode { fused_fn fn; int ne; const char* what; };
static const Mode MODES[] = {
    {fused_kernel<0, 1, 4>, 1, "QKV requested behind the gather (no overlap), one edge wave gathers (4 round trips per sweep)"},
    {fused_kernel<1, 1, 4>, 1, "QKV's first 4 units requested behind the publish, one edge wave gathers"},
    {fused_kernel<2, 1, 4>, 1, "QKV's first 4 units requested behind barrier (1), one edge wave gathers"},
    {fused_kernel<3, 1, 4>, 1, "QKV requested behind the gather, the eight streaming waves gather an eighth each"},
    {fused_kernel<0, 4, 4>, 4, "QKV requested behind the gather, four edge waves gather a quarter each (1 round trip per sweep)"},
    {fused_kernel<1, 4, 4>, 4, "QKV's first 4 units requested behind the publish, four edge waves gather"},
    {fused_kernel<1, 4, 2>, 4, "QKV's first 2 units requested behind the publish (the other 2 behind the gather), four edge waves gather"},
    {fused_kernel<1, 4, 1>, 4, "QKV's first unit requested behind the publish (the other 3 behind the gather), four edge waves gather"},
    {roles_kernel<4>, 4, "mode 5 with one branch arm per role (exact wait counts in the QKV phase)"},
    {roles_kernel<2>, 4, "mode 6 with one branch arm per role"},
    {roles_kernel<1>, 4, "mode 7 with one branch arm per role"},
    {roles_kernel<3>, 4, "first 3 units behind the publish, one branch arm per role"},
};

static float time_graph(hipGraphExec_t ex, hipStream_t s, int launches) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  CK(hipGraphLaunch(ex, s));
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, s));
    CK(hipGraphLaunch(ex, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  return best * 1000.f / launches;
}

int main() {
  const int NB = 8;                                                       // rotating weight sets (no cache residency across launches)
  float *W2, *Wq, *b2, *bq, *h, *x, *q, *xref, *qref; unsigned long long* gran; long long* prof; int* giveup;
  CK(hipMalloc(&W2, (size_t)NB * D * F * 4)); CK(hipMalloc(&Wq, (size_t)NB * NQ * D * 4));
  CK(hipMalloc(&b2, D * 4)); CK(hipMalloc(&bq, NQ * 4)); CK(hipMalloc(&h, 2 * F * 4)); CK(hipMalloc(&x, 2 * D * 4)); CK(hipMalloc(&q, 2 * NQ * 4));
  CK(hipMalloc(&xref, 2 * D * 4)); CK(hipMalloc(&qref, 2 * NQ * 4));
  CK(hipMalloc(&gran, 2 * 2 * D * 8)); CK(hipMalloc(&prof, G * 4 * 8)); CK(hipMalloc(&giveup, 4));
  CK(hipMemset(gran, 0, 2 * 2 * D * 8)); CK(hipMemset(giveup, 0, 4));
  std::vector<float> hw((size_t)D * F);
  unsigned st = 777u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((int)(st >> 8) - (1 << 23)) * (1.0f / (1 << 23)); };
  for (auto& v : hw) v = rnd() * 0.01f;
  for (int i = 0; i < NB; ++i) CK(hipMemcpy(W2 + (size_t)i * D * F, hw.data(), (size_t)D * F * 4, hipMemcpyHostToDevice));
  hw.resize((size_t)NQ * D);
  for (auto& v : hw) v = rnd() * 0.02f;
  for (int i = 0; i < NB; ++i) CK(hipMemcpy(Wq + (size_t)i * NQ * D, hw.data(), (size_t)NQ * D * 4, hipMemcpyHostToDevice));
  std::vector<float> hb(NQ), hh(2 * F), hx(2 * D);
  for (auto& v : hb) v = rnd() * 0.1f;
  CK(hipMemcpy(b2, hb.data(), D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bq, hb.data(), NQ * 4, hipMemcpyHostToDevice));
  for (auto& v : hh) v = rnd();
  for (auto& v : hx) v = rnd();
  CK(hipMemcpy(h, hh.data(), 2 * F * 4, hipMemcpyHostToDevice));
  hipStream_t s; CK(hipStreamCreate(&s));
  auto args = [&](int i, float* xbuf, float* qbuf, bool with_prof) {
    Args a; memset(&a, 0, sizeof(a));
    a.W2 = W2 + (size_t)(i % NB) * D * F; a.b2 = b2; a.h = h; a.x = xbuf; a.Wq = Wq + (size_t)(i % NB) * NQ * D; a.bq = bq; a.q = qbuf;
    a.gran = gran + (size_t)(i & 1) * 2 * D; a.gran_next = gran + (size_t)((i + 1) & 1) * 2 * D; a.prof = with_prof ? prof : nullptr; a.giveup = giveup;
    return a;
  };
  hipGraph_t g; hipGraphExec_t exA, exA1, exA2;
  const int NL = 64;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < NL; ++i) { hipLaunchKernelGGL(ffn2_kernel, dim3(G), dim3(512), 0, s, args(i, x, q, false)); hipLaunchKernelGGL(qkv_kernel, dim3(G), dim3(512), 0, s, args(i, x, q, false)); }
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&exA, g, nullptr, nullptr, 0));
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < NL; ++i) hipLaunchKernelGGL(ffn2_kernel, dim3(G), dim3(512), 0, s, args(i, x, q, false));
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&exA1, g, nullptr, nullptr, 0));
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < NL; ++i) hipLaunchKernelGGL(qkv_kernel, dim3(G), dim3(512), 0, s, args(i, x, q, false));
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&exA2, g, nullptr, nullptr, 0));
  for (int round = 0; round < 2; ++round) {
    const float ta = time_graph(exA, s, NL), ta1 = time_graph(exA1, s, NL), ta2 = time_graph(exA2, s, NL);
    printf("round %d: two launches %.2f us per edge (FFN2 alone %.2f + LN+QKV alone %.2f = %.2f)\n", round, ta, ta1, ta2, ta1 + ta2);
  }
  for (int pre = 0; pre < (int)(sizeof(MODES) / sizeof(MODES[0])); ++pre) {
    fused_fn fk = MODES[pre].fn;
    const int nthr = 512 + 64 * MODES[pre].ne;
    // ---- correctness: one pair of launches vs one fused launch, from the same x
    CK(hipMemcpy(xref, hx.data(), 2 * D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), 2 * D * 4, hipMemcpyHostToDevice));
    CK(hipMemset(gran, 0, 2 * 2 * D * 8)); CK(hipMemset(giveup, 0, 4));
    hipLaunchKernelGGL(ffn2_kernel, dim3(G), dim3(512), 0, s, args(0, xref, qref, false));
    hipLaunchKernelGGL(qkv_kernel, dim3(G), dim3(512), 0, s, args(0, xref, qref, false));
    hipLaunchKernelGGL(fk, dim3(G), dim3(nthr), 0, s, args(0, x, q, true));
    CK(hipStreamSynchronize(s));
    std::vector<float> q0(2 * NQ), q1(2 * NQ), x0(2 * D), x1(2 * D);
    CK(hipMemcpy(q0.data(), qref, 2 * NQ * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(q1.data(), q, 2 * NQ * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(x0.data(), xref, 2 * D * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(x1.data(), x, 2 * D * 4, hipMemcpyDeviceToHost));
    int badq = 0, badx = 0, gu = 0;
    for (int i = 0; i < 2 * NQ; ++i) badq += memcmp(&q0[i], &q1[i], 4) != 0;
    for (int i = 0; i < 2 * D; ++i) badx += memcmp(&x0[i], &x1[i], 4) != 0;
    CK(hipMemcpy(&gu, giveup, 4, hipMemcpyDeviceToHost));
    printf("\nmode %d (%s)\n  one fused launch vs two launches: %d of %d q values and %d of %d x' values differ; gather gave up: %d\n", pre, MODES[pre].what, badq, 2 * NQ, badx, 2 * D, gu);
    hipGraphExec_t exB;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < NL; ++i) hipLaunchKernelGGL(fk, dim3(G), dim3(nthr), 0, s, args(i, x, q, i == NL - 1));
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&exB, g, nullptr, nullptr, 0));
    CK(hipMemset(gran, 0, 2 * 2 * D * 8));
    const float tb0 = time_graph(exB, s, NL), tb1 = time_graph(exB, s, NL);
    printf("  ONE fused launch: %.2f / %.2f us per edge\n", tb0, tb1);
    // the protocol over a long chain: 3 replays x 64 edges from the same x, both forms, final x' and q compared (a stale granule — a tag of an
    // earlier launch surviving in some L2 — would show here)
    {
      CK(hipMemcpy(xref, hx.data(), 2 * D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), 2 * D * 4, hipMemcpyHostToDevice));
      for (int rep = 0; rep < 3; ++rep)
        for (int i = 0; i < NL; ++i) {
          hipLaunchKernelGGL(ffn2_kernel, dim3(G), dim3(512), 0, s, args(i, xref, qref, false));
          hipLaunchKernelGGL(qkv_kernel, dim3(G), dim3(512), 0, s, args(i, xref, qref, false));
        }
      for (int rep = 0; rep < 3; ++rep) CK(hipGraphLaunch(exB, s));
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(q0.data(), qref, 2 * NQ * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(q1.data(), q, 2 * NQ * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(x0.data(), xref, 2 * D * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(x1.data(), x, 2 * D * 4, hipMemcpyDeviceToHost));
      badq = badx = 0;
      for (int i = 0; i < 2 * NQ; ++i) badq += memcmp(&q0[i], &q1[i], 4) != 0;
      for (int i = 0; i < 2 * D; ++i) badx += memcmp(&x0[i], &x1[i], 4) != 0;
      printf("  after 192 chained edges: %d q values and %d x' values differ from the two-launch chain\n", badq, badx);
    }
    CK(hipMemcpy(&gu, giveup, 4, hipMemcpyDeviceToHost));
    std::vector<long long> hp(G * 4);
    CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
    long long tmin = hp[0];
    for (int b = 0; b < G; ++b) tmin = std::min(tmin, hp[b * 4]);
    const char* names[3] = {"FFN2 partials parked", "granules published", "x' in LDS (barrier 2)"};
    printf("  last launch of the chain, us after the first workgroup parked its partials (min / median / max over %d CUs); gave up: %d\n", G, gu);
    for (int k = 0; k < 3; ++k) {
      std::vector<double> v;
      for (int b = 0; b < G; ++b) v.push_back((hp[b * 4 + k] - tmin) * 0.01);
      std::sort(v.begin(), v.end());
      printf("    %-24s %6.2f / %6.2f / %6.2f\n", names[k], v[0], v[G / 2], v[G - 1]);
    }
    std::vector<long long> sw;
    for (int b = 0; b < G; ++b) sw.push_back(hp[b * 4 + 3]);
    std::sort(sw.begin(), sw.end());
    printf("    sweeps per CU            %lld / %lld / %lld\n", sw[0], sw[G / 2], sw[G - 1]);
  }
  return 0;
}
