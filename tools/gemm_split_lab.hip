// tools/gemm_split_lab.hip — EXPERIMENT (round 3): fp32-accurate GEMM on the bf16 matrix cores ("bf16x3": every fp32 operand is split
// exactly into three bf16 pieces a = a0 + a1 + a2, and C accumulates the six largest of the nine cross products in fp32:
// a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0; the three dropped ones are <= 2^-24 relative, the size of one fp32 rounding).
// v_mfma_f32_32x32x16_bf16 is 16x the rate of v_mfma_f32_32x32x2_f32, so six of them per 16 k-values cost 192 matrix-pipe cycles
// against 512 for the exact fp32 chain. Measured against the library's ssrhip_gemm (exact fp32 FMA chain) for speed and against an
// fp64 reference for the error of both.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -Issr-speech_amd/csrc tools/gemm_split_lab.hip \
//          -o tools/bin/gemm_split_lab -Lssr-speech_amd/csrc -lssrhip -Wl,-rpath,'$ORIGIN/../../ssr-speech_amd/csrc'
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <type_traits>
#include "common.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
void ssrhip_set_error(const char*, ...) {}

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_fn(float v, int act) {
  if (act == SSRHIP_ACT_RELU) return fmaxf(v, 0.f);
  if (act == SSRHIP_ACT_GELU_ERF) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  return v;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bfx2 __attribute__((ext_vector_type(2)));
// exact three-way split of 4 consecutive k-values (v_cvt_pk_bf16_f32 = round to nearest even, two elements per instruction;
// bf16 -> fp32 is a 16-bit shift; the residuals a - hi and (a - hi) - mid are exact in fp32): piece p of element e in out[p][e]
__device__ __forceinline__ void split4(const float4 v, bf16x4 (&out)[3]) {
  f32x2 r[2] = {{v.x, v.y}, {v.z, v.w}};
#pragma unroll
  for (int p = 0; p < 3; ++p) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bfx2 b = __builtin_convertvector(r[h], bfx2);
      const unsigned bits = __builtin_bit_cast(unsigned, b);
      out[p][2 * h] = (short)(bits & 0xFFFFu);
      out[p][2 * h + 1] = (short)(bits >> 16);
      if (p < 2) {
        const f32x2 back = {__builtin_bit_cast(float, bits << 16), __builtin_bit_cast(float, bits & 0xFFFF0000u)};
        r[h] = r[h] - back;
      }
    }
  }
}

// one-time weight preparation: W fp32 [N][K] -> three bf16 planes [3][N][K]
__global__ void split_weights_kernel(const float* __restrict__ W, short* __restrict__ out, size_t n_elems) {
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n_elems; i += (size_t)gridDim.x * blockDim.x * 4) {
    bf16x4 p[3];
    split4(ld4(W + i), p);
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(out + (size_t)q * n_elems + i) = p[q];
  }
}

// Block tile 128 x 128 x 32, 4 waves (2 x 2), each 64 x 64 = 2 x 2 accumulators of 32 x 32. LDS per piece: rows of 32 bf16 (64 B)
// padded to 80 B: a lane's ds_read_b128 (8 consecutive k of one row) then hits 16 distinct 16-byte slots per 16-lane group.
constexpr int BM = 128, BN = 128, BK = 32, PITCH = 40;              // pitch in bf16 elements (80 B)
// ABL: timing ablations (results are WRONG for ABL > 0; they only tell where the time goes): 1 no global loads after the first tile,
// 2 no operand split of A (one conversion, pieces duplicated), 3 no W loads / W LDS stores after the first tile, 4 no barriers,
// 5 MFMA + ds_read loop only (1 + 3 + 4 + no A stores), 6 one product of the six (the staging path alone)
template <bool ELU, int ABL = 0>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const ssrhip_gemm_args a0, const short* __restrict__ Wp, const size_t plane) {
  __shared__ __attribute__((aligned(16))) short As[3][BM * PITCH];
  __shared__ __attribute__((aligned(16))) short Ws[3][BN * PITCH];
  ssrhip_gemm_args a = a0;
  {
    const size_t z = blockIdx.z;
    a.A += z * (size_t)a.strideA;
    a.C += z * (size_t)a.strideC;
    if (a.R) a.R += z * (size_t)a.strideR;
  }
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int lr = t >> 3, lc = (t & 7) * 4;                          // loader: 8 threads per row (32 k), 32 rows per pass
  const int M = a.M, N = a.N, K = a.K;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 ra[4];
  bf16x8 rwp[3][2];                                                 // W: 2 chunks of 8 k per piece (row lw + 64 * i, k-chunk cw)
  const int lw = t >> 2, cw = (t & 3) * 8;
  auto gload = [&](int k0) {
    const bool kin = (k0 + lc) < K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + lr + 32 * i;
      ra[i] = (kin && m < M) ? ld4(a.A + (size_t)m * a.lda + k0 + lc) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (ELU) { ra[i].x = elu1(ra[i].x); ra[i].y = elu1(ra[i].y); ra[i].z = elu1(ra[i].z); ra[i].w = elu1(ra[i].w); }
    }
    const bool kinw = (k0 + cw) < K;                                 // K % 8 == 0 required for the pre-split planes
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n = n0 + lw + 64 * i;
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        rwp[q][i] = (kinw && n < N) ? *reinterpret_cast<const bf16x8*>(Wp + (size_t)q * plane + (size_t)n * K + k0 + cw) : z;
      }
  };
  auto lds_store = [&](bool first) {
    if (ABL != 5 || first) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf16x4 p[3];
        if (ABL == 2) {
          const bfx2 b0 = __builtin_convertvector((f32x2){ra[i].x, ra[i].y}, bfx2), b1 = __builtin_convertvector((f32x2){ra[i].z, ra[i].w}, bfx2);
          const unsigned u0 = __builtin_bit_cast(unsigned, b0), u1 = __builtin_bit_cast(unsigned, b1);
          p[0] = bf16x4{(short)(u0 & 0xFFFF), (short)(u0 >> 16), (short)(u1 & 0xFFFF), (short)(u1 >> 16)};
          p[1] = p[0]; p[2] = p[0];
        } else {
          split4(ra[i], p);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(&As[q][(lr + 32 * i) * PITCH + lc]) = p[q];
      }
    }
    if ((ABL != 3 && ABL != 5) || first) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<bf16x8*>(&Ws[q][(lw + 64 * i) * PITCH + cw]) = rwp[q][i];
    }
  };
  auto mma_tile = [&]() {
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      bf16x8 fa[3][2], fb[3][2];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[q][i] = *reinterpret_cast<const bf16x8*>(&As[q][((wm * 2 + i) * 32 + li) * PITCH + kk + lh * 8]);
          fb[q][i] = *reinterpret_cast<const bf16x8*>(&Ws[q][((wn * 2 + i) * 32 + li) * PITCH + kk + lh * 8]);
        }
      // smallest terms first
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int pq = (ABL == 6 ? 5 : 0); pq < 6; ++pq)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pq]][i], fb[PB[pq]][j], acc[i][j], 0, 0, 0);
    }
  };
  auto gload_a = [&](int k0) {                                       // ABL 3: the A half of gload only
    const bool kin = (k0 + lc) < K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + lr + 32 * i;
      ra[i] = (kin && m < M) ? ld4(a.A + (size_t)m * a.lda + k0 + lc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  gload(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    if (ABL != 4 && ABL != 5) __syncthreads();
    lds_store(k0 == 0);
    if ((ABL != 4 && ABL != 5) || k0 == 0) __syncthreads();
    if (k0 + BK < K) {
      if (ABL == 3) gload_a(k0 + BK);
      else if (ABL != 1 && ABL != 5) gload(k0 + BK);
    }
    mma_tile();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + (wn * 2 + j) * 32 + li;
    if (n >= N) continue;
    const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * 2 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M) {
          float v = act_fn(acc[mt][j][r] + bias, a.act);
          float* c = a.C + (size_t)m * a.ldc + n;
          if (a.residual) v += *c;
          if (a.R) v += a.R[(size_t)m * a.ldr + n];
          *c = v;
        }
      }
    }
  }
}


// ---- round 3, second design: ONE workgroup of 8 waves per CU, 128 x 128 x 32 tiles in TWO LDS stages (122,880 B, dynamic), one barrier per
// k-tile. Waves w and w + 4 share a SIMD and run the two halves of an interval in opposite order — w: [split + store tile k+1, MFMAs of tile
// k], w + 4: [MFMAs of tile k, split + store tile k+1] — so one wave's staging runs under the other's matrix work. Global loads run two
// tiles ahead (two register sets). Wave tile 64 x 32 (2 x 1 accumulators). Per output element the arithmetic is the same as
// gemm_split_kernel's (k blocks of 16 in order, the six products in the same order): results are bit-identical.
// MODE: 0 as described, 1 both wave groups stage first (no phase shift), timing ablations as above: 5 MFMA + ds_read only
template <bool ELU, int MODE>
__global__ __launch_bounds__(512, 1) void gemm_split_db_kernel(const ssrhip_gemm_args a0, const short* __restrict__ Wp, const size_t plane) {
  extern __shared__ __attribute__((aligned(16))) short lds[];
  constexpr int PLANE_E = 128 * PITCH, STAGE_E = 6 * PLANE_E;      // elements (bf16): one stage = 3 A planes + 3 W planes = 61,440 B
  ssrhip_gemm_args a = a0;
  {
    const size_t z = blockIdx.z;
    a.A += z * (size_t)a.strideA;
    a.C += z * (size_t)a.strideC;
    if (a.R) a.R += z * (size_t)a.strideR;
  }
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = (wave >> 2) & 1, wn = wave & 3;
  const int n0 = blockIdx.x * 128, m0 = blockIdx.y * 128;
  const int lr = t >> 3, lc = (t & 7) * 4;                           // A loader: 8 threads per row (32 k), 64 rows per pass, 2 passes
  const int lw = t >> 2, cw = (t & 3) * 8;                           // W loader: 4 threads per row (8 k each), 128 rows, one pass per piece
  const int M = a.M, N = a.N, K = a.K;
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float4 ra[2][2];
  bf16x8 rw[2][3];
  // branch-free loads: buffer descriptors whose extent ends with the tile's last valid row (rows past M / N read as zero), and a k past K
  // sends the lane's offset out of range (a plain load + select is turned back into a branch around the load by the compiler)
  const int rows_a = min(128, M - m0), rows_w = min(128, N - n0);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A + (size_t)m0 * a.lda), 0,
                                                                        (int)(((size_t)(rows_a - 1) * a.lda + K) * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rsW[3];
#pragma unroll
  for (int q = 0; q < 3; ++q)
    rsW[q] = __builtin_amdgcn_make_buffer_rsrc(const_cast<short*>(Wp + (size_t)q * plane + (size_t)n0 * K), 0, (int)((size_t)rows_w * K * 2), 0x00020000);
  unsigned offA[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) offA[i] = (unsigned)(((size_t)(lr + 64 * i) * a.lda + lc) * 4);
  const unsigned offW = (unsigned)(((size_t)lw * K + cw) * 2);
  constexpr unsigned OOB = 0x80000000u;
  auto gload = [&](auto SET, int k0) {
    constexpr int set = decltype(SET)::value;
    const bool kin = (k0 + lc) < K;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsA, kin ? offA[i] + (unsigned)k0 * 4 : OOB, 0, 0);
      ra[set][i] = __builtin_bit_cast(float4, v);
    }
    const bool kinw = (k0 + cw) < K;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsW[q], kinw ? offW + (unsigned)k0 * 2 : OOB, 0, 0);
      rw[set][q] = __builtin_bit_cast(bf16x8, v);
    }
  };
  auto stage = [&](auto SET, int buf) {
    constexpr int set = decltype(SET)::value;
    short* base = lds + buf * STAGE_E;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 v = ra[set][i];
      if (ELU) { v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w); }
      bf16x4 p[3];
      split4(v, p);
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(base + q * PLANE_E + (lr + 64 * i) * PITCH + lc) = p[q];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8*>(base + (3 + q) * PLANE_E + lw * PITCH + cw) = rw[set][q];
  };
  auto mma = [&](int buf) {
    const short* base = lds + buf * STAGE_E;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      bf16x8 fa[3][2], fb[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[q][i] = *reinterpret_cast<const bf16x8*>(base + q * PLANE_E + ((wm * 2 + i) * 32 + li) * PITCH + kk + lh * 8);
        fb[q] = *reinterpret_cast<const bf16x8*>(base + (3 + q) * PLANE_E + (wn * 32 + li) * PITCH + kk + lh * 8);
      }
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int pq = 0; pq < 6; ++pq)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pq]][i], fb[PB[pq]], acc[i], 0, 0, 0);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  const int nk = (K + BK - 1) / BK;
  // every load / store below is unconditional (a tile past K reads as zeros through the descriptors): the number of loads in flight is
  // then the same on every path and the compiler can wait for the OLDER register set only (vmcnt(5)), not for everything
  gload(S0{}, 0);
  gload(S1{}, BK);
  stage(S0{}, 0);
  gload(S0{}, 2 * BK);
  __syncthreads();
  // two separate loops (not one loop with a branch inside: the compiler merges the common halves of the two orders and then waits for
  // ALL loads in flight)
  if (MODE == 5) {
    for (int kt = 0; kt < nk; kt += 2) { mma(0); mma(1); }
  } else if (MODE == 1 || wave < 4) {                                  // stage first, then the MFMAs
    auto body = [&](auto PAR, int kt) {                                // tile kt in stage PAR; tile kt + 1 waits in register set PAR ^ 1
      constexpr int par = decltype(PAR)::value;
      using NXT = std::integral_constant<int, par ^ 1>;
      stage(NXT{}, par ^ 1);
      gload(NXT{}, (kt + 3) * BK);
      if (kt < nk) mma(par);
      __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) { body(S0{}, kt); body(S1{}, kt + 1); }
  } else {                                                             // MFMAs first
    auto body = [&](auto PAR, int kt) {
      constexpr int par = decltype(PAR)::value;
      using NXT = std::integral_constant<int, par ^ 1>;
      if (kt < nk) mma(par);
      stage(NXT{}, par ^ 1);
      gload(NXT{}, (kt + 3) * BK);
      __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) { body(S0{}, kt); body(S1{}, kt + 1); }
  }
  {
    const int n = n0 + wn * 32 + li;
    if (n < N) {
      const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * 2 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < M) {
            float v = act_fn(acc[mt][r] + bias, a.act);
            float* c = a.C + (size_t)m * a.ldc + n;
            if (a.residual) v += *c;
            if (a.R) v += a.R[(size_t)m * a.ldr + n];
            *c = v;
          }
        }
      }
    }
  }
}

// ---- third design: the product kernel's structure (one LDS stage of 61,440 B, two barriers per k-tile, two workgroups per CU) with 8
// waves per workgroup (wave tile 64 x 32): 16 waves per CU = 4 per SIMD at <= 128 VGPRs, branch-free buffer loads
template <bool ELU>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_split_w8_kernel(const ssrhip_gemm_args a0, const short* __restrict__ Wp, const size_t plane) {
  __shared__ __attribute__((aligned(16))) short As[3][BM * PITCH];
  __shared__ __attribute__((aligned(16))) short Ws[3][BN * PITCH];
  ssrhip_gemm_args a = a0;
  {
    const size_t z = blockIdx.z;
    a.A += z * (size_t)a.strideA;
    a.C += z * (size_t)a.strideC;
    if (a.R) a.R += z * (size_t)a.strideR;
  }
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = (wave >> 2) & 1, wn = wave & 3;
  const int n0 = blockIdx.x * 128, m0 = blockIdx.y * 128;
  const int lr = t >> 3, lc = (t & 7) * 4;
  const int lw = t >> 2, cw = (t & 3) * 8;
  const int M = a.M, N = a.N, K = a.K;
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float4 ra[2];
  bf16x8 rw[3];
  const int rows_a = min(128, M - m0), rows_w = min(128, N - n0);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A + (size_t)m0 * a.lda), 0,
                                                                        (int)(((size_t)(rows_a - 1) * a.lda + K) * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rsW[3];
#pragma unroll
  for (int q = 0; q < 3; ++q)
    rsW[q] = __builtin_amdgcn_make_buffer_rsrc(const_cast<short*>(Wp + (size_t)q * plane + (size_t)n0 * K), 0, (int)((size_t)rows_w * K * 2), 0x00020000);
  unsigned offA[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) offA[i] = (unsigned)(((size_t)(lr + 64 * i) * a.lda + lc) * 4);
  const unsigned offW = (unsigned)(((size_t)lw * K + cw) * 2);
  constexpr unsigned OOB = 0x80000000u;
  auto gload = [&](int k0) {
    const bool kin = (k0 + lc) < K;
#pragma unroll
    for (int i = 0; i < 2; ++i) ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, kin ? offA[i] + (unsigned)k0 * 4 : OOB, 0, 0));
    const bool kinw = (k0 + cw) < K;
#pragma unroll
    for (int q = 0; q < 3; ++q) rw[q] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW[q], kinw ? offW + (unsigned)k0 * 2 : OOB, 0, 0));
  };
  auto lds_store = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 v = ra[i];
      if (ELU) { v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w); }
      bf16x4 p[3];
      split4(v, p);
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(&As[q][(lr + 64 * i) * PITCH + lc]) = p[q];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8*>(&Ws[q][lw * PITCH + cw]) = rw[q];
  };
  auto mma_tile = [&]() {
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      bf16x8 fa[3][2], fb[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[q][i] = *reinterpret_cast<const bf16x8*>(&As[q][((wm * 2 + i) * 32 + li) * PITCH + kk + lh * 8]);
        fb[q] = *reinterpret_cast<const bf16x8*>(&Ws[q][(wn * 32 + li) * PITCH + kk + lh * 8]);
      }
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int pq = 0; pq < 6; ++pq)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pq]][i], fb[PB[pq]], acc[i], 0, 0, 0);
    }
  };
  gload(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
    lds_store();
    __syncthreads();
    gload(k0 + BK);                                                  // past K: reads as zeros, never stored
    __builtin_amdgcn_sched_barrier(0);                               // keep the loads here: the scheduler sinks them to the end of the MFMAs
    mma_tile();
  }
  {
    const int n = n0 + wn * 32 + li;
    if (n < N) {
      const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * 2 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < M) {
            float v = act_fn(acc[mt][r] + bias, a.act);
            float* c = a.C + (size_t)m * a.ldc + n;
            if (a.residual) v += *c;
            if (a.R) v += a.R[(size_t)m * a.ldr + n];
            *c = v;
          }
        }
      }
    }
  }
}

// ---- fourth design: as w8, with W brought straight into LDS by the DMA path (buffer_load ... lds: no staging registers, no ds_write) one
// tile ahead into a second W stage, and XOR-swizzled 64-byte rows instead of padded ones (73,728 B per workgroup: A 3 x 8 KB, W 2 x 3 x 8 KB;
// two workgroups per CU). Row r keeps its 16-byte chunk c in slot c ^ ((r >> 2) & 3): a lane group of ds_read_b128 (rows {0-3, 12-15,
// 20-27} or {4-11, 16-19, 28-31}, one chunk index) then touches 16 different slots of the 256-byte bank row, and the 64-byte rows make the
// ds_write_b64 of A conflict-free too (two rows per 16-lane group = 32 banks).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
// MODE bits: 1 s_setprio(1) around the MFMAs, 2 all 18 fragment reads of a k-tile issued before its MFMAs, 4 XCD-aware tile order
template <bool ELU, int MODE = 0>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_split_w8g_kernel(const ssrhip_gemm_args a0, const short* __restrict__ Wp, const size_t plane) {
  extern __shared__ __attribute__((aligned(1024))) char ldsb[];
  constexpr int PL = 128 * 64;                                       // bytes per plane
  char* const As = ldsb;                                             // [3][128][64 B]
  char* const Wsb = ldsb + 3 * PL;                                   // [2][3][128][64 B]
  ssrhip_gemm_args a = a0;
  {
    const size_t z = blockIdx.z;
    a.A += z * (size_t)a.strideA;
    a.C += z * (size_t)a.strideC;
    if (a.R) a.R += z * (size_t)a.strideR;
  }
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = (wave >> 2) & 1, wn = wave & 3;
  int bx = blockIdx.x, by = blockIdx.y;
  if (MODE & 4) {   // workgroups are dealt to the 8 XCDs round-robin: give each XCD a contiguous run of tiles (x fastest), so the tiles that share an A row block meet in one L2
    const unsigned nwg = gridDim.x * gridDim.y, orig = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const unsigned id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    bx = id % gridDim.x; by = id / gridDim.x;
  }
  const int n0 = bx * 128, m0 = by * 128;
  const int lr = t >> 3, lc = (t & 7) * 4;
  const int M = a.M, N = a.N, K = a.K;
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float4 ra[2];
  const int rows_a = min(128, M - m0), rows_w = min(128, N - n0);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A + (size_t)m0 * a.lda), 0,
                                                                        (int)(((size_t)(rows_a - 1) * a.lda + K) * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rsW[3];
#pragma unroll
  for (int q = 0; q < 3; ++q)
    rsW[q] = __builtin_amdgcn_make_buffer_rsrc(const_cast<short*>(Wp + (size_t)q * plane + (size_t)n0 * K), 0, (int)((size_t)rows_w * K * 2), 0x00020000);
  unsigned offA[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) offA[i] = (unsigned)(((size_t)(lr + 64 * i) * a.lda + lc) * 4);
  constexpr unsigned OOB = 0x80000000u;
  // W by DMA: wave w brings rows 16w .. 16w+15 of a plane with one instruction; lane l lands in LDS slot l of the wave's KiB, which is
  // (row 16w + l/4, slot l%4) and has to hold chunk (l%4) ^ ((row >> 2) & 3)
  const int wrow = 16 * wave + (lane >> 2), wchunk = (lane & 3) ^ ((lane >> 4) & 3);
  const unsigned offW = (unsigned)(((size_t)wrow * K + wchunk * 8) * 2);
  auto gload_a = [&](int k0) {
    const bool kin = (k0 + lc) < K;
#pragma unroll
    for (int i = 0; i < 2; ++i) ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, kin ? offA[i] + (unsigned)k0 * 4 : OOB, 0, 0));
  };
  auto dma_w = [&](int k0, int stage) {
    const bool kin = (k0 + wchunk * 8) < K;
    const unsigned off = kin ? offW + (unsigned)k0 * 2 : OOB;
#pragma unroll
    for (int q = 0; q < 3; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW[q], (lds_ptr_t)(Wsb + (stage * 3 + q) * PL + wave * 1024), 16, off, 0, 0, 0);
  };
  const int aswz = (((lc >> 3) ^ ((lr >> 2) & 3)) << 4) + ((lc >> 2) & 1) * 8;   // row lr and row lr + 64 share (row >> 2) & 3
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 v = ra[i];
      if (ELU) { v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w); }
      bf16x4 p[3];
      split4(v, p);
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(As + q * PL + (lr + 64 * i) * 64 + aswz) = p[q];
    }
  };
  const int fsw = (li >> 2) & 3;
  auto mma_tile = [&](int stage) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    if (MODE & 2) {
      bf16x8 fa[2][3][2], fb[2][3];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int coff = (((ks * 2) + lh) ^ fsw) << 4;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[ks][q][i] = *reinterpret_cast<const bf16x8*>(As + q * PL + ((wm * 2 + i) * 32 + li) * 64 + coff);
          fb[ks][q] = *reinterpret_cast<const bf16x8*>(Wsb + (stage * 3 + q) * PL + (wn * 32 + li) * 64 + coff);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pq = 0; pq < 6; ++pq)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][PA[pq]][i], fb[ks][PB[pq]], acc[i], 0, 0, 0);
      if (MODE & 1) __builtin_amdgcn_s_setprio(0);
      return;
    }
    if (MODE & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      const int coff = (((kk >> 3) + lh) ^ fsw) << 4;
      bf16x8 fa[3][2], fb[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[q][i] = *reinterpret_cast<const bf16x8*>(As + q * PL + ((wm * 2 + i) * 32 + li) * 64 + coff);
        fb[q] = *reinterpret_cast<const bf16x8*>(Wsb + (stage * 3 + q) * PL + (wn * 32 + li) * 64 + coff);
      }
#pragma unroll
      for (int pq = 0; pq < 6; ++pq)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pq]][i], fb[PB[pq]], acc[i], 0, 0, 0);
    }
    if (MODE & 1) __builtin_amdgcn_s_setprio(0);
  };
  gload_a(0);
  dma_w(0, 0);
  int stage = 0;
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();                                                 // tile k0 - BK fully consumed (A stage, W stage ^ 1)
    store_a();
    __syncthreads();                                                 // A(k0) stored, W(k0) landed (the compiler waits for the DMA here)
    gload_a(k0 + BK);                                                // past K: zeros, never used
    dma_w(k0 + BK, stage ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    mma_tile(stage);
    stage ^= 1;
  }
  {
    const int n = n0 + wn * 32 + li;
    if (n < N) {
      const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * 2 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < M) {
            float v = act_fn(acc[mt][r] + bias, a.act);
            float* c = a.C + (size_t)m * a.ldc + n;
            if (a.residual) v += *c;
            if (a.R) v += a.R[(size_t)m * a.ldr + n];
            *c = v;
          }
        }
      }
    }
  }
}

short* g_wp = nullptr;
int g_abl = 0;
int g_mode = 0;
int g_db = -1;                                                       // >= 0: the double-buffered 8-wave kernel, MODE = g_db
int launch_split(const ssrhip_gemm_args* a, hipStream_t s) {
  dim3 grid((a->N + BN - 1) / BN, (a->M + BM - 1) / BM, a->batch > 1 ? a->batch : 1);
  const size_t plane = (size_t)a->N * a->K;
  if (g_db == 9) {
    constexpr int LDS = 9 * 128 * 64;
#define W8G(E, MD) do { static bool once_ = false; if (!once_) { once_ = true; CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_w8g_kernel<E, MD>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); } \
      hipLaunchKernelGGL((gemm_split_w8g_kernel<E, MD>), grid, dim3(512), LDS, s, *a, g_wp, plane); } while (0)
    if (a->act_in == SSRHIP_ACT_ELU) { if (g_mode == 4) W8G(true, 4); else W8G(true, 0); }
    else switch (g_mode) {
      case 1: W8G(false, 1); break;
      case 2: W8G(false, 2); break;
      case 3: W8G(false, 3); break;
      case 4: W8G(false, 4); break;
      default: W8G(false, 0);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  if (g_db == 8) {
    if (a->act_in == SSRHIP_ACT_ELU) hipLaunchKernelGGL((gemm_split_w8_kernel<true>), grid, dim3(512), 0, s, *a, g_wp, plane);
    else hipLaunchKernelGGL((gemm_split_w8_kernel<false>), grid, dim3(512), 0, s, *a, g_wp, plane);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  if (g_db >= 0) {
    constexpr int LDS = 2 * 6 * 128 * PITCH * 2;
    static bool once = false;
    if (!once) {
      once = true;
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_db_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_db_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_db_kernel<false, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_db_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    if (a->act_in == SSRHIP_ACT_ELU) hipLaunchKernelGGL((gemm_split_db_kernel<true, 0>), grid, dim3(512), LDS, s, *a, g_wp, plane);
    else if (g_db == 1) hipLaunchKernelGGL((gemm_split_db_kernel<false, 1>), grid, dim3(512), LDS, s, *a, g_wp, plane);
    else if (g_db == 5) hipLaunchKernelGGL((gemm_split_db_kernel<false, 5>), grid, dim3(512), LDS, s, *a, g_wp, plane);
    else hipLaunchKernelGGL((gemm_split_db_kernel<false, 0>), grid, dim3(512), LDS, s, *a, g_wp, plane);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  if (a->act_in == SSRHIP_ACT_ELU) hipLaunchKernelGGL(gemm_split_kernel<true>, grid, dim3(256), 0, s, *a, g_wp, plane);
  else switch (g_abl) {
    case 1: hipLaunchKernelGGL((gemm_split_kernel<false, 1>), grid, dim3(256), 0, s, *a, g_wp, plane); break;
    case 2: hipLaunchKernelGGL((gemm_split_kernel<false, 2>), grid, dim3(256), 0, s, *a, g_wp, plane); break;
    case 3: hipLaunchKernelGGL((gemm_split_kernel<false, 3>), grid, dim3(256), 0, s, *a, g_wp, plane); break;
    case 4: hipLaunchKernelGGL((gemm_split_kernel<false, 4>), grid, dim3(256), 0, s, *a, g_wp, plane); break;
    case 5: hipLaunchKernelGGL((gemm_split_kernel<false, 5>), grid, dim3(256), 0, s, *a, g_wp, plane); break;
    case 6: hipLaunchKernelGGL((gemm_split_kernel<false, 6>), grid, dim3(256), 0, s, *a, g_wp, plane); break;
    default: hipLaunchKernelGGL((gemm_split_kernel<false, 0>), grid, dim3(256), 0, s, *a, g_wp, plane);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) * scale;      // 24 random mantissa bits: every split piece is exercised
  }
}
}  // namespace

int main(int argc, char** argv) {
  const bool ablate = argc > 1 && !strcmp(argv[1], "ablate");
  struct Shape { const char* name; int M, N, K, batch, act_in; } shapes[] = {
      {"square 4096^3", 4096, 4096, 4096, 1, 0},       {"prefill qkv 598x6144x2048", 598, 6144, 2048, 1, 0},
      {"lstm-in 1500x4096x1024 x32", 1500, 4096, 1024, 32, 0}, {"down2 60000x256x1024 x32", 60000, 256, 1024, 32, 1},
      {"down1 240000x128x256 x32", 240000, 128, 256, 32, 1},   {"down4 1500x1024x8192 x32", 1500, 1024, 8192, 32, 1},
      {"convtr 60000x512x512 x32", 60000, 512, 512, 32, 1},  {"ragged 3000x2056x1000", 3000, 2056, 1000, 1, 0},
  };
  const size_t cap = (size_t)32 * 240000 * 256;
  float *A, *W, *C0, *C1, *bias;
  CK(hipMalloc(&A, cap * 4)); CK(hipMalloc(&W, (size_t)8192 * 8192 * 4)); CK(hipMalloc(&C0, cap * 4)); CK(hipMalloc(&C1, cap * 4));
  CK(hipMalloc(&bias, 8192 * 4));
  CK(hipMalloc(&g_wp, (size_t)3 * 8192 * 8192 * 2));
  hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, A, cap, 1u, 0.5f);
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, W, (size_t)8192 * 8192, 2u, 0.02f);
  hipLaunchKernelGGL(fill_kernel, dim3(8), dim3(256), 0, 0, bias, (size_t)8192, 3u, 0.1f);
  CK(hipDeviceSynchronize());
  hipStream_t s; CK(hipStreamCreate(&s));
  {   // error of both kernels against fp64 on a 256 x 256 x 4096 problem (no ELU)
    const int M = 256, N = 256, K = 4096;
    ssrhip_gemm_args a; memset(&a, 0, sizeof(a));
    a.A = A; a.W = W; a.bias = bias; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = N;
    a.C = C0; ssrhip_gemm(&a, s);
    hipLaunchKernelGGL(split_weights_kernel, dim3(2048), dim3(256), 0, s, W, g_wp, (size_t)N * K);
    a.C = C1; launch_split(&a, s);
    CK(hipStreamSynchronize(s));
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N), h0((size_t)M * N), h1((size_t)M * N);
    CK(hipMemcpy(hA.data(), A, hA.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hW.data(), W, hW.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), bias, N * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
    double e0 = 0, e1 = 0, s0 = 0, s1 = 0, scale = 0;
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) {
        double ref = hb[n], mag = 0;
        for (int k = 0; k < K; ++k) { const double p = (double)hA[(size_t)m * K + k] * hW[(size_t)n * K + k]; ref += p; mag += fabs(p); }
        const double d0 = fabs(h0[(size_t)m * N + n] - ref), d1 = fabs(h1[(size_t)m * N + n] - ref);
        e0 = fmax(e0, d0 / mag); e1 = fmax(e1, d1 / mag); s0 += d0 / mag; s1 += d1 / mag; scale = fmax(scale, mag);
      }
    printf("error vs fp64 (256x256x4096, relative to sum|a.b|): exact fp32 MFMA chain max %.3e mean %.3e | bf16x3 six products max %.3e mean %.3e\n",
           e0, s0 / (M * N), e1, s1 / (M * N));
  }
  if (argc > 1 && !strcmp(argv[1], "prof")) {   // for rocprofv3 --pmc: 4096^3, three launches of the product-shaped kernel, three of the 8-wave one
    ssrhip_gemm_args a; memset(&a, 0, sizeof(a));
    a.A = A; a.W = W; a.bias = bias; a.M = 4096; a.N = 4096; a.K = 4096; a.lda = 4096; a.ldc = 4096; a.C = C1;
    hipLaunchKernelGGL(split_weights_kernel, dim3(2048), dim3(256), 0, s, W, g_wp, (size_t)4096 * 4096);
    const bool dma = argc > 2 && !strcmp(argv[2], "dma");               // prof dma: the 8-wave DMA kernel instead of the double-buffered one
    for (int v = 0; v < 2; ++v) { g_db = v ? (dma ? 9 : 0) : -1; g_mode = dma ? 1 : 0; for (int i = 0; i < 3; ++i) launch_split(&a, s); }
    CK(hipStreamSynchronize(s));
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "modes")) {   // the DMA kernel's scheduling variants
    for (auto& sh : shapes) {
      ssrhip_gemm_args a; memset(&a, 0, sizeof(a));
      a.A = A; a.W = W; a.bias = bias; a.M = sh.M; a.N = sh.N; a.K = sh.K; a.lda = sh.K; a.ldc = sh.N; a.act_in = sh.act_in ? SSRHIP_ACT_ELU : 0; a.C = C1;
      a.batch = sh.batch; a.strideA = (int64_t)sh.M * sh.K; a.strideC = (int64_t)sh.M * sh.N;
      hipLaunchKernelGGL(split_weights_kernel, dim3(2048), dim3(256), 0, s, W, g_wp, (size_t)sh.N * sh.K);
      g_db = 9;
      printf("%-30s", sh.name);
      for (g_mode = 0; g_mode < 5; ++g_mode) {
        if (sh.act_in && g_mode != 0 && g_mode != 4) { printf("  mode %d      -", g_mode); continue; }
        if (launch_split(&a, s)) { printf("launch failed\n"); return 1; }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float wms = 0.f;
        for (int round = 0; round < 50 && wms < 40.f; ++round) {
          CK(hipEventRecord(e0, s)); for (int i = 0; i < 4; ++i) launch_split(&a, s); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
          float m; CK(hipEventElapsedTime(&m, e0, e1)); wms += m;
        }
        CK(hipEventRecord(e0, s)); for (int i = 0; i < 10; ++i) launch_split(&a, s); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  mode %d %6.1f", g_mode, 2.0 * sh.M * sh.N * sh.K * sh.batch * 10 / (ms * 1e-3) / 1e12);
      }
      printf("\n");
      g_db = -1; g_mode = 0;
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "db")) {   // the double-buffered 8-wave kernel against the product's split kernel: time and bit identity
    for (auto& sh : shapes) {
      ssrhip_gemm_args a; memset(&a, 0, sizeof(a));
      a.A = A; a.W = W; a.bias = bias; a.M = sh.M; a.N = sh.N; a.K = sh.K; a.lda = sh.K; a.ldc = sh.N; a.act_in = sh.act_in ? SSRHIP_ACT_ELU : 0;
      a.batch = sh.batch; a.strideA = (int64_t)sh.M * sh.K; a.strideC = (int64_t)sh.M * sh.N;
      hipLaunchKernelGGL(split_weights_kernel, dim3(2048), dim3(256), 0, s, W, g_wp, (size_t)sh.N * sh.K);
      const int modes[4] = {-1, 9, 8, 5};
      double tf[4];
      for (int v = 0; v < 4; ++v) {
        if (sh.act_in && v >= 3) { tf[v] = 0; continue; }
        g_db = modes[v];
        a.C = v == 0 ? C0 : C1;
        if (launch_split(&a, s)) { printf("launch failed\n"); return 1; }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float wms = 0.f;
        for (int round = 0; round < 50 && wms < 40.f; ++round) {
          CK(hipEventRecord(e0, s)); for (int i = 0; i < 4; ++i) launch_split(&a, s); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
          float m; CK(hipEventElapsedTime(&m, e0, e1)); wms += m;
        }
        CK(hipEventRecord(e0, s)); for (int i = 0; i < 10; ++i) launch_split(&a, s); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        tf[v] = 2.0 * sh.M * sh.N * sh.K * sh.batch * 10 / (ms * 1e-3) / 1e12;
        if (v == 1 || v == 2) {   // bit identity on a sample of the output
          const size_t n = (size_t)sh.M * sh.N * sh.batch, take = n < (1u << 22) ? n : (1u << 22);
          std::vector<float> h0(take), h1(take);
          CK(hipMemcpy(h0.data(), C0 + (n - take), take * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), C1 + (n - take), take * 4, hipMemcpyDeviceToHost));
          size_t bad = 0; for (size_t i = 0; i < take; ++i) bad += memcmp(&h0[i], &h1[i], 4) != 0;
          CK(hipMemcpy(h0.data(), C0, take * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), C1, take * 4, hipMemcpyDeviceToHost));
          for (size_t i = 0; i < take; ++i) bad += memcmp(&h0[i], &h1[i], 4) != 0;
          printf("%-30s differing values in the first / last %zu: %zu\n", sh.name, take, bad);
        }
      }
      g_db = -1;
      printf("%-30s split (2 WG/CU) %6.1f | 8 waves + W by DMA + swizzle %6.1f | 8 waves, one stage, 2 WG/CU %6.1f | its MFMA + ds_read only %6.1f  TFLOP/s fp32-equivalent\n",
             sh.name, tf[0], tf[1], tf[2], tf[3]);
    }
    return 0;
  }
  if (ablate) {   // where the split kernel's time goes: the same launch with one part removed at a time (no ELU shapes)
    const char* what[7] = {"full kernel", "no global loads in the loop", "no split of A (1 cvt)", "no W loads / W LDS stores", "no barriers",
                           "MFMA + ds_read only", "1 of 6 products"};
    for (auto& sh : shapes) {
      if (sh.act_in && strncmp(sh.name, "convtr", 6)) continue;
      ssrhip_gemm_args a; memset(&a, 0, sizeof(a));
      a.A = A; a.W = W; a.bias = bias; a.M = sh.M; a.N = sh.N; a.K = sh.K; a.lda = sh.K; a.ldc = sh.N; a.C = C1;
      a.batch = sh.batch; a.strideA = (int64_t)sh.M * sh.K; a.strideC = (int64_t)sh.M * sh.N;
      hipLaunchKernelGGL(split_weights_kernel, dim3(2048), dim3(256), 0, s, W, g_wp, (size_t)sh.N * sh.K);
      printf("%s\n", sh.name);
      for (g_abl = 0; g_abl < 7; ++g_abl) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float wms = 0.f;
        for (int round = 0; round < 50 && wms < 40.f; ++round) {
          CK(hipEventRecord(e0, s)); for (int i = 0; i < 4; ++i) launch_split(&a, s); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
          float m; CK(hipEventElapsedTime(&m, e0, e1)); wms += m;
        }
        CK(hipEventRecord(e0, s)); for (int i = 0; i < 10; ++i) launch_split(&a, s); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %-30s %8.3f ms/launch  %6.1f TFLOP/s (fp32-equivalent of the FULL problem)\n", what[g_abl], ms / 10,
               2.0 * sh.M * sh.N * sh.K * sh.batch * 10 / (ms * 1e-3) / 1e12);
      }
    }
    g_abl = 0;
    return 0;
  }
  for (auto& sh : shapes) {
    ssrhip_gemm_args a; memset(&a, 0, sizeof(a));
    a.A = A; a.W = W; a.bias = bias; a.M = sh.M; a.N = sh.N; a.K = sh.K; a.lda = sh.K; a.ldc = sh.N; a.act_in = sh.act_in ? SSRHIP_ACT_ELU : 0;
    a.batch = sh.batch; a.strideA = (int64_t)sh.M * sh.K; a.strideC = (int64_t)sh.M * sh.N;
    double tf[2];
    hipLaunchKernelGGL(split_weights_kernel, dim3(2048), dim3(256), 0, s, W, g_wp, (size_t)sh.N * sh.K);
    for (int which = 0; which < 2; ++which) {
      a.C = which ? C1 : C0;
      auto run = [&]() { return which ? launch_split(&a, s) : ssrhip_gemm(&a, s); };
      if (run()) { printf("launch failed\n"); return 1; }
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float wms = 0.f;
      for (int round = 0; round < 50 && wms < 40.f; ++round) {
        CK(hipEventRecord(e0, s)); for (int i = 0; i < 4; ++i) run(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float m; CK(hipEventElapsedTime(&m, e0, e1)); wms += m;
      }
      CK(hipEventRecord(e0, s)); for (int i = 0; i < 10; ++i) run(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      tf[which] = 2.0 * sh.M * sh.N * sh.K * sh.batch * 10 / (ms * 1e-3) / 1e12;
    }
    printf("%-30s exact fp32 %6.1f TFLOP/s   bf16x3 %6.1f TFLOP/s (fp32-equivalent)   x%.2f\n", sh.name, tf[0], tf[1], tf[1] / tf[0]);
  }
  return 0;
}
