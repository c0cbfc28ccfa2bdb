// common.h — shared host/device helpers for libssrhip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ssrhip.h"

#define SSR_WAVE 64

void ssrhip_set_error(const char* fmt, ...);
const char* ssrhip_gemv_pair_why();      // gemv.hip: why the last pair-applicability question on this thread was answered with no

#define SSR_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      ssrhip_set_error(__VA_ARGS__);      \
      return -1;                          \
    }                                     \
  } while (0)

#define SSR_HIP(call)                                                         \
  do {                                                                        \
    hipError_t _e = (call);                                                   \
    if (_e != hipSuccess) {                                                   \
      ssrhip_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      return -2;                                                              \
    }                                                                         \
  } while (0)

// hipFuncSetAttribute (the dynamic-LDS limit of a kernel) is per DEVICE: one flag per device id, so that a process which drives several
// GPUs raises the limit on each of them (a lost race sets the attribute twice, which is harmless)
struct ssr_once_per_device {
  unsigned long long mask[2] = {0ull, 0ull};
  bool need() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) return true;
    d &= 127;
    const bool n = !((mask[d >> 6] >> (d & 63)) & 1ull);
    mask[d >> 6] |= 1ull << (d & 63);
    return n;
  }
};

// profiling aid (tools/prof_summary.py --gemm-log): one line per GEMM launch — M N K batch, the launch grid, split (1) or fp32 chain (0) —
// so that a kernel trace's GEMM rows get their problem size and a TFLOP/s column. Off unless SSRHIP_GEMM_LOG names a file.
#include <stdio.h>
#include <stdlib.h>
static inline void ssr_gemm_log(const ssrhip_gemm_args* a, unsigned gx, unsigned gy, unsigned gz, int split) {
  static FILE* glog = getenv("SSRHIP_GEMM_LOG") ? fopen(getenv("SSRHIP_GEMM_LOG"), "a") : nullptr;
  if (glog) {
    fprintf(glog, "%d %d %d %d %u %u %u %d\n", a->M, a->N, a->K, a->batch > 1 ? a->batch : 1, gx, gy, gz, split);
    fflush(glog);
  }
}

#define SSR_LAUNCH_CHECK()                                                    \
  do {                                                                        \
    hipError_t _e = hipGetLastError();                                        \
    if (_e != hipSuccess) {                                                   \
      ssrhip_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return -3;                                                              \
    }                                                                         \
  } while (0)

#ifdef __HIPCC__
// ---- wave-wide all-reduce without LDS traffic: 4 DPP row rotations (within each 16-lane row) + the
// gfx950 v_permlane16_swap / v_permlane32_swap for the cross-row steps. 8 VALU instructions, every lane
// ends with the total; ~10x lower latency than six ds_bpermute round trips. The association order is
// fixed, so results are bit-reproducible.
// NOTE: the two operands of v_permlane{16,32}_swap must live in DIFFERENT registers (the instruction
// swaps lanes between its two registers in place); feeding the builtin the same value twice lets hipcc
// allocate one register for both and the swap degenerates to a no-op (seen on ROCm 7.2). `opaque_copy`
// hides the equality from the compiler.
__device__ __forceinline__ unsigned opaque_copy(unsigned v) {
  asm volatile("; permlane operand copy" : "+v"(v));
  return v;
}
struct U2 { unsigned a, b; };
__device__ __forceinline__ U2 swap16(unsigned u) {
  auto r = __builtin_amdgcn_permlane16_swap(u, opaque_copy(u), false, false);
  return U2{r[0], r[1]};   // a = {row0,row0,row2,row2}, b = {row1,row1,row3,row3}
}
__device__ __forceinline__ U2 swap32(unsigned u) {
  auto r = __builtin_amdgcn_permlane32_swap(u, opaque_copy(u), false, false);
  return U2{r[0], r[1]};   // a = {lo,lo}, b = {hi,hi}
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
struct OpAdd { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };
struct OpMax { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };

template <class Op>
__device__ __forceinline__ float wave_allreduce(float v, Op op) {
  v = op(v, dpp_f<0x128>(v));   // row_ror:8
  v = op(v, dpp_f<0x124>(v));   // row_ror:4
  v = op(v, dpp_f<0x122>(v));   // row_ror:2
  v = op(v, dpp_f<0x121>(v));   // row_ror:1
  const U2 r = swap16(__builtin_bit_cast(unsigned, v));
  v = op(__builtin_bit_cast(float, r.a), __builtin_bit_cast(float, r.b));
  const U2 s = swap32(__builtin_bit_cast(unsigned, v));
  return op(__builtin_bit_cast(float, s.a), __builtin_bit_cast(float, s.b));
}
__device__ __forceinline__ float wave_sum(float v) { return wave_allreduce(v, OpAdd()); }
__device__ __forceinline__ float wave_max(float v) { return wave_allreduce(v, OpMax()); }
__device__ __forceinline__ int wave_min_i(int v) {
  v = min(v, dpp_i<0x128>(v));
  v = min(v, dpp_i<0x124>(v));
  v = min(v, dpp_i<0x122>(v));
  v = min(v, dpp_i<0x121>(v));
  const U2 r = swap16((unsigned)v);
  v = min((int)r.a, (int)r.b);
  const U2 s = swap32((unsigned)v);
  return min((int)s.a, (int)s.b);
}
// sum over groups of 32 / 16 lanes only (attention: one key row per group)
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0x128>(v);
  v += dpp_f<0x124>(v);
  v += dpp_f<0x122>(v);
  v += dpp_f<0x121>(v);
  return v;
}
__device__ __forceinline__ float half32_sum(float v) {
  v = row16_sum(v);
  const U2 r = swap16(__builtin_bit_cast(unsigned, v));
  return __builtin_bit_cast(float, r.a) + __builtin_bit_cast(float, r.b);
}
// exchange with lane^16 / lane^32 (one VALU op each)
__device__ __forceinline__ float xor16_f(float v) {
  const U2 r = swap16(__builtin_bit_cast(unsigned, v));
  const float a = __builtin_bit_cast(float, r.a), b = __builtin_bit_cast(float, r.b);
  return ((threadIdx.x >> 4) & 1) ? a : b;
}
__device__ __forceinline__ float xor32_f(float v) {
  const U2 r = swap32(__builtin_bit_cast(unsigned, v));
  const float a = __builtin_bit_cast(float, r.a), b = __builtin_bit_cast(float, r.b);
  return ((threadIdx.x >> 5) & 1) ? a : b;
}

__device__ __forceinline__ float dot4(const float4 a, const float4 b, float acc) {
  acc = fmaf(a.x, b.x, acc);
  acc = fmaf(a.y, b.y, acc);
  acc = fmaf(a.z, b.z, acc);
  acc = fmaf(a.w, b.w, acc);
  return acc;
}
// streamed-once weights: non-temporal 16-byte load (global_load_dwordx4 ... nt)
__device__ __forceinline__ float4 ld_nt(const float* p) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// nn.ELU(alpha = 1): v > 0 ? v : exp(v) - 1 with the hardware exponential (v_exp_f32). Absolute error <= 1.2e-7 everywhere (for |v| below
// 6e-8 the result is 0 instead of v). Rounds 1-3 switched to a degree-6 Taylor polynomial above -0.25 for RELATIVE accuracy near zero;
// round 4's counters showed what that costs — the fused residual block was VALU-bound, half of its 4,230 VALU instructions per wave and
// tile were ELUs at ~13 instructions each (profiles/r04_microbench/resblock_pmc.log) — and nothing downstream needs it: every consumer is a
// convolution whose result is compared in absolute terms (5e-5 per layer, 2e-5 end to end against the reference fixtures). ONE definition
// for every codec kernel, so ELU-on-store stays bit-identical to ELU-on-load (tests/test_gpu_codec.py).
__device__ __forceinline__ float elu1(float v) {
  const float e = __expf(v) - 1.0f;
  return v > 0.f ? v : e;
}

// address of element (pos, d) of head h, k(0)/v(1), layer, in sequence `seq`'s paged cache
__device__ __forceinline__ float* kv_addr(const ssrhip_kv& kv, int seq, int layer, int which, int h, int pos) {
  const int page = kv.table[(size_t)seq * kv.max_pages + (pos / SSRHIP_PAGE)];
  const size_t off = ((((size_t)page * kv.n_layer + layer) * 2 + which) * kv.n_head + h) * SSRHIP_PAGE + (pos % SSRHIP_PAGE);
  return kv.pool + off * kv.head_dim;
}
#endif
