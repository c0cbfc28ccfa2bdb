// tools/edge16_lab.hip — what would the all-to-all edge of a pair launch cost at 16 rows? (round 5, for next round's decision)
// The 2-row pair kernels (csrc/gemv.hip) hand 2 x 2048 values to every CU as 4096 tagged 8-byte granules: 32 KB per CU, gathered by four
// waves in one round trip per sweep, ~2.5 us per edge. At 16 rows the edge carries 16 x 2048 values: 256 KB per CU as 8-byte granules,
// 192 KB as 16-byte {v, v, v, tag} granules. This lab times ONLY the edge, in a dependent hipGraph chain of launches of 256 workgroups:
// every workgroup publishes its 8 outputs x ROWS values (write-through stores), then NG of its waves gather everything (sc1 loads, DEPTH
// 16-byte loads in flight per lane) until every tag is this launch's, and writes a checksum; against an empty kernel in the same chain.
// The streaming waves of a real pair launch would be requesting weights meanwhile: this is the edge's LOWER bound.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/edge16_lab.hip -o tools/bin/edge16_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int G = 256, NOUT = 2048, RA = NOUT / G;      // 8 outputs per workgroup

struct Args {
  float* gran;          // this launch's granules
  float* gran_next;     // reset for the next launch
  float* sink;          // [G] checksums
  int* giveup;
  unsigned tagbits;     // tag value of this launch (nonzero)
};

__device__ __forceinline__ v4f ld_sc1(const float* p) {
  v4f v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// FORM 0: 8-byte granules {value, tag}: a 16-byte load = 2 values. FORM 1: 16-byte granules {v, v, v, tag}: 3 values.
template <int ROWS, int FORM, int NG, int DEPTH>
__global__ __launch_bounds__(768) void edge_kernel(const Args a) {
  __shared__ float red[12];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  constexpr int NVAL = NOUT * ROWS;                                    // values on the edge
  constexpr int NG16 = FORM == 0 ? NVAL / 2 : (NVAL + 2) / 3;          // 16-byte pieces to gather
  constexpr int MINE = RA * ROWS;                                      // values this workgroup publishes
  constexpr int MINE16 = FORM == 0 ? MINE / 2 : (MINE + 2) / 3;
  const float tagf = __uint_as_float(a.tagbits);
  // reset the next launch's granules (this workgroup's share), then publish this launch's
  for (int i = t; i < MINE16; i += 768) {
    v4f z = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<v4f*>(a.gran_next + ((size_t)blockIdx.x * MINE16 + i) * 4) = z;
  }
  if (wave == 8) {
    for (int i = lane; i < MINE16; i += 64) {
      const float v = (float)(blockIdx.x * MINE16 + i) * 1e-3f;
      v4f gq;
      if (FORM == 0) gq = v4f{v, tagf, v + 0.5f, tagf}; else gq = v4f{v, v + 0.25f, v + 0.5f, tagf};
      float* dst = a.gran + ((size_t)blockIdx.x * MINE16 + i) * 4;
      if (FORM == 0) {                                                 // two 8-byte write-through stores
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), ((unsigned long long)a.tagbits << 32) | __float_as_uint(gq[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst) + 1, ((unsigned long long)a.tagbits << 32) | __float_as_uint(gq[2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {                                                         // one 16-byte write-through store (the tag travels in the same 16 bytes)
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(gq) : "memory");
      }
    }
  }
  // gather: waves 12 - NG .. 11 share the NG16 pieces, DEPTH loads in flight per lane
  float acc = 0.f;
  if (wave >= 12 - NG) {
    const int w = wave - (12 - NG);
    constexpr int PER = (NG16 + NG * 64 - 1) / (NG * 64);              // pieces per lane
    bool done = false;
    for (int spin = 0; spin < 20000 && !done; ++spin) {
      bool all = true;
      float s = 0.f;
      for (int i0 = 0; i0 < PER; i0 += DEPTH) {
        v4f g[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const int piece = min(((i0 + d) * NG + w) * 64 + lane, NG16 - 1);
          g[d] = ld_sc1(a.gran + (size_t)piece * 4);
        }
        // the wait names every destination register as in-out: hipcc must not schedule a use of g[d] in front of it
        if constexpr (DEPTH == 8)
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]) :: "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]), "+v"(g[8 % DEPTH]),
                       "+v"(g[9 % DEPTH]), "+v"(g[10 % DEPTH]), "+v"(g[11 % DEPTH]), "+v"(g[12 % DEPTH]), "+v"(g[13 % DEPTH]), "+v"(g[14 % DEPTH]), "+v"(g[15 % DEPTH]) :: "memory");
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          if (FORM == 0) { all = all && __float_as_uint(g[d][1]) == a.tagbits && __float_as_uint(g[d][3]) == a.tagbits; s += g[d][0] + g[d][2]; }
          else { all = all && __float_as_uint(g[d][3]) == a.tagbits; s += g[d][0] + g[d][1] + g[d][2]; }
        }
      }
      done = __all(all);
      acc = s;
      if (!done) __builtin_amdgcn_s_sleep(2);
    }
    if (!done && lane == 0) *a.giveup = 1;
  }
  for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (t == 0) { float s = 0.f; for (int i = 0; i < 12; ++i) s += red[i]; a.sink[blockIdx.x] = s; }
}
__global__ __launch_bounds__(768) void empty_kernel(const Args a) {
  if (threadIdx.x == 0) a.sink[blockIdx.x] = 1.f;
}

template <typename K>
static float chain_us(K kern, float* gran, float* sink, int* giveup, size_t gran_floats, hipStream_t s) {
  const int NL = 64;
  hipGraph_t g; hipGraphExec_t ex;
  CK(hipMemsetAsync(gran, 0, 3 * gran_floats * 4, s));
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < NL; ++i) {
    Args a; a.gran = gran + (size_t)(i % 3) * gran_floats; a.gran_next = gran + (size_t)((i + 1) % 3) * gran_floats; a.sink = sink; a.giveup = giveup;
    a.tagbits = 0x3f800000u + (unsigned)i + 1u;                        // a different tag per launch of the chain (NL % 3 == 1: the wrap would collide otherwise)
    hipLaunchKernelGGL(kern, dim3(G), dim3(768), 0, s, a);
  }
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  CK(hipGraphLaunch(ex, s));
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ex, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  return best * 1000.f / NL;
}

int main() {
  float *gran, *sink; int* giveup;
  const size_t gran_floats = (size_t)NOUT * 16 * 2 + 64;              // the largest form: 16 rows x 8-byte granules
  CK(hipMalloc(&gran, 3 * gran_floats * 4)); CK(hipMalloc(&sink, G * 4)); CK(hipMalloc(&giveup, 4));
  CK(hipMemset(giveup, 0, 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  const float t_empty = chain_us(empty_kernel, gran, sink, giveup, gran_floats, s);
  printf("empty 768-thread kernel in the chain: %.2f us per launch\n", t_empty);
#define RUN(ROWS, FORM, NG, DEPTH) do { \
    const float us = chain_us(edge_kernel<ROWS, FORM, NG, DEPTH>, gran, sink, giveup, gran_floats, s); \
    int gu = 0; CK(hipMemcpy(&gu, giveup, 4, hipMemcpyDeviceToHost)); \
    printf("%2d rows, %s granules (%3d KB per CU), %2d gathering waves, %2d loads in flight per lane: %6.2f us per launch = edge %5.2f us%s\n", ROWS, \
           FORM == 0 ? " 8-byte {v, tag}      " : "16-byte {v, v, v, tag}", (int)((FORM == 0 ? NOUT * ROWS / 2 : (NOUT * ROWS + 2) / 3) * 16 / 1024), NG, DEPTH, us, us - t_empty, gu ? "  GAVE UP" : ""); \
  } while (0)
  RUN(2, 0, 4, 8);
  RUN(2, 1, 4, 8);
  RUN(4, 0, 4, 8);
  RUN(4, 1, 4, 8);
  RUN(16, 0, 4, 8);
  RUN(16, 0, 4, 16);
  RUN(16, 0, 8, 16);
  RUN(16, 0, 12, 16);
  RUN(16, 1, 4, 16);
  RUN(16, 1, 8, 16);
  RUN(16, 1, 12, 16);
  RUN(16, 1, 12, 8);
  return 0;
}
