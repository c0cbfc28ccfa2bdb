// common.h — shared host/device helpers for libssrhip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ssrhip.h"

#define SSR_WAVE 64

void ssrhip_set_error(const char* fmt, ...);

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

#define SSR_LAUNCH_CHECK()                                                    \
  do {                                                                        \
    hipError_t _e = hipGetLastError();                                        \
    if (_e != hipSuccess) {                                                   \
      ssrhip_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return -3;                                                              \
    }                                                                         \
  } while (0)

#ifdef __HIPCC__
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
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

// address of element (pos, d) of head h, k(0)/v(1), layer, in sequence `seq`'s paged cache
__device__ __forceinline__ float* kv_addr(const ssrhip_kv& kv, int seq, int layer, int which, int h, int pos) {
  const int page = kv.table[(size_t)seq * kv.max_pages + (pos / SSRHIP_PAGE)];
  const size_t off = ((((size_t)page * kv.n_layer + layer) * 2 + which) * kv.n_head + h) * SSRHIP_PAGE + (pos % SSRHIP_PAGE);
  return kv.pool + off * kv.head_dim;
}
#endif
