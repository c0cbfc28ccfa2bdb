// lstm_split.hip — one LSTM time step for a LARGE batch on the bf16 matrix cores with exactly split fp32 operands (written blind at the end
// of round 4; first run in round 5: kernel test vs fp64 and every codec fixture green, 21.0 us per step alone / 35.6 us with both layers'
// chains sharing the GPU against 31.0 / 56.8 for lstm_step_wide_kernel at 256 items — the codec's default from 128 items up).
//
//     gates[b][g C + j] = gin[b][t][g C + j] + sum_k h_{t-1}[b][k] W_hh[g C + j][k]         (torch.nn.LSTM, gates i f g o; modules/lstm.py:10-25)
//
// Why: lstm_step_wide_kernel (csrc/codec.hip) keeps its 256 KB W_hh slice in registers and is MATRIX-bound on the fp32 pipe — four batch
// tiles x 256 v_mfma_f32_16x16x4_f32 x 32 cycles = 13.7 us of a step's 29 us (one wave per SIMD), 21 % of a config-5 codec pass. The
// arithmetic of csrc/gemm_split.hip (every fp32 operand the exact sum of three bf16 pieces, the six largest cross products accumulated in
// fp32: error against fp64 no larger than the fp32 chain's) runs the same step in 4 waves x 384 v_mfma_f32_32x32x16_bf16 = 5.1 us.
//
// Work split: a workgroup (4 waves, one per CU: grid C/16 x ceil(B/64)) owns 16 hidden units = 64 gate rows (m = 16 g + u) and 64 batch
// rows; wave w multiplies the K range [w C/4, (w + 1) C/4) for the whole 64 x 64 tile (4 accumulator blocks), so no operand is read twice
// inside a workgroup; the four partial tiles meet in LDS and every thread finishes 4 (unit, batch) pairs: gates, cell update, h.
// Operands come from global memory (L2) in FRAGMENT ORDER — one wave-level dwordx4 load = one contiguous KiB = one MFMA operand:
//     w_split [C/16 unit blocks][4 waves][KS k-steps][2 row blocks][3 planes][64 lanes][8 bf16]      (host, once: wmencodec._Lstm)
//         lane (li = l % 32, lh = l / 32), element e: piece q of W_hh[g C + 16 ub + u][w C/4 + 16 s + 8 lh + e],  32 mb + li = 16 g + u
//     hsplit  [2 (t parity)][ceil(B/64)][4 waves][KS][2 batch blocks][3 planes][64 lanes][8 bf16]        (written by the previous step)
//         piece q of h[64 bg + 32 nb + li][w C/4 + 16 s + 8 lh + e]
// with KS = C/64 k-steps of 16 per wave. The producing thread of h[b][j] splits it and stores the three 2-byte pieces at their fragment
// positions, so the next step reads its B operand with no staging at all. fp32 h is not kept (`hbuf` is unused on this path).
#include <stdlib.h>
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bfx2 __attribute__((ext_vector_type(2)));

constexpr int LS_TH = 256, LS_BN = 64, LS_UN = 16, LS_PS = 65;         // threads, batch rows / hidden units per workgroup, LDS row stride (floats)
constexpr int ls_lds() { return 4 * LS_BN * LS_PS * 4; }               // four partial tiles [n][m], rows padded to 65: conflict-free stores

__device__ __forceinline__ float sigmoid_(float v) { return 1.0f / (1.0f + expf(-v)); }

// index (in uint16 units) of the KiB block (64 lanes x 8 bf16) of `hsplit` / `w_split`: [outer][wave][kstep][block][plane]
__device__ __forceinline__ size_t frag_block(int outer, int wave, int ks, int s, int blk, int q) {
  return (((((size_t)outer * 4 + wave) * ks + s) * 2 + blk) * 3 + q) * 512;
}

// exact three-way split of one value (gemm_split.hip's split4, one lane wide)
__device__ __forceinline__ void split1(float v, unsigned short (&p)[3]) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const f32x2 r = {v, 0.f};
    const bfx2 b = __builtin_convertvector(r, bfx2);                  // v_cvt_pk_bf16_f32 (RNE)
    const unsigned bits = __builtin_bit_cast(unsigned, b) & 0xFFFFu;
    p[q] = (unsigned short)bits;
    v -= __builtin_bit_cast(float, bits << 16);
  }
}

template <int DEPTH>
__global__ __launch_bounds__(LS_TH, 1) void lstm_step_split_kernel(const ssrhip_lstm_args a, const int t, const unsigned short* __restrict__ hp,
                                                                     unsigned short* __restrict__ hn, const int KS) {
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // w2h0, w0h2, w1h1, w1h0, w0h1, w0h0 (smallest first, as gemm_split.hip)
  extern __shared__ __attribute__((aligned(16))) float part[];           // [4 waves][64 n][LS_PS]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int ub = blockIdx.x, bg = blockIdx.y, C = a.C;

  // ---- what the finishing phase needs, requested first: thread p = tid + 256 i finishes unit u = p % 16 of batch row bl = p / 16
  // (16 consecutive lanes = 16 consecutive hidden units of one item: 64-byte runs of gin / c / skip / out)
  float pg[4][4], pc[4], ps[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = tid + LS_TH * i, u = p & 15, bl = p >> 4;
    const int b = min(bg * LS_BN + bl, a.B - 1), j = ub * LS_UN + u;
    const float* gin = a.gin + (size_t)b * a.gin_bstride + (size_t)t * 4 * C + j;
#pragma unroll
    for (int g = 0; g < 4; ++g) pg[i][g] = gin[(size_t)g * C];
    pc[i] = (t == 0) ? 0.f : a.cbuf[(size_t)b * C + j];
    ps[i] = a.skip ? a.skip[(size_t)b * a.skip_bstride + (size_t)t * C + j] : 0.f;
  }

  // ---- the K loop: DEPTH k-steps of operands in flight (12 KiB-loads each), 24 MFMAs per k-step
  const unsigned short* wsrc = reinterpret_cast<const unsigned short*>(a.w_split) + frag_block(ub, wave, KS, 0, 0, 0) + lane * 8;
  const unsigned short* hsrc = hp + frag_block(bg, wave, KS, 0, 0, 0) + lane * 8;
  bf16x8 fa[DEPTH][2][3], fb[DEPTH][2][3];
  auto load = [&](int slot, int s) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const size_t off = ((size_t)(s * 2 + blk) * 3 + q) * 512;
        fa[slot][blk][q] = *reinterpret_cast<const bf16x8*>(wsrc + off);
        fb[slot][blk][q] = *reinterpret_cast<const bf16x8*>(hsrc + off);
      }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    load(d, d);
    __builtin_amdgcn_sched_barrier(0);           // slot by slot, as the loop refills them: the loop head then waits for slot 0 only
  }
  for (int s = 0; s < KS; s += DEPTH) {                                 // KS % DEPTH == 0 (checked by the launcher)
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int pq = 0; pq < 6; ++pq)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[d][mb][PA[pq]], fb[d][nb][PB[pq]], acc[mb][nb], 0, 0, 0);
      load(d, min(s + d + DEPTH, KS - 1));       // unconditional (past the end: the last k-step again, never used): a branch around the
                                                 // loads makes every wait of the loop a vmcnt(0) and the prefetch distance is gone
      __builtin_amdgcn_sched_barrier(0);         // the refill stays HERE, DEPTH - 1 MFMA blocks (768 cycles each) ahead of its use: one
                                                 // wave per SIMD has nobody else to hide an L2 round trip behind
    }
  }

  // ---- the four partial tiles meet in LDS: part[wave][n][m]; accumulator register r of block (mb, nb) = D[32 mb + (r&3) + 8(r>>2) + 4 lh][32 nb + li]
  float* mine = part + (size_t)wave * LS_BN * LS_PS;
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[(nb * 32 + li) * LS_PS + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = acc[mb][nb][r];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = tid + LS_TH * i, u = p & 15, bl = p >> 4;
    const int b = bg * LS_BN + bl, j = ub * LS_UN + u;
    float gt[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float* src = part + (size_t)bl * LS_PS + g * 16 + u;
      gt[g] = ((src[0] + src[LS_BN * LS_PS]) + (src[2 * LS_BN * LS_PS] + src[3 * LS_BN * LS_PS])) + pg[i][g];     // fixed order: deterministic
    }
    if (b < a.B) {
      const float cn = sigmoid_(gt[1]) * pc[i] + sigmoid_(gt[0]) * tanhf(gt[2]);
      const float hv = sigmoid_(gt[3]) * tanhf(cn);
      a.cbuf[(size_t)b * C + j] = cn;
      float o = hv + ps[i];                                            // y = lstm(x) + x on the last layer (lstm.py:21-23); ps == 0 otherwise
      if (a.out_act == SSRHIP_ACT_ELU) o = elu1(o);
      a.out[(size_t)b * a.out_bstride + (size_t)t * C + j] = o;
      // h_t[b][j] as the next step's B operand: k = j -> wave j / (C/4), k-step (j % (C/4)) / 16, lane half (j % 16) / 8, element j % 8
      unsigned short pcs[3];
      split1(hv, pcs);
      const int kq = C >> 2, w2 = j / kq, s2 = (j % kq) >> 4, lh2 = (j >> 3) & 1, e2 = j & 7;
      unsigned short* dst = hn + frag_block(bg, w2, KS, s2, bl >> 5, 0) + (lh2 * 32 + (bl & 31)) * 8 + e2;
#pragma unroll
      for (int q = 0; q < 3; ++q) dst[q * 512] = pcs[q];
    }
  }
}

__global__ __launch_bounds__(256) void zero_u32_kernel(unsigned* p, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = 0u;
}

}  // namespace

// bf16 elements of ONE h buffer of the split path (the caller allocates two: ssrhip_lstm_args.hsplit)
size_t ssrhip_lstm_split_hplane_elems(int B, int C) { return (size_t)((B + LS_BN - 1) / LS_BN) * LS_BN * C * 3; }

bool ssrhip_lstm_split_eligible(const ssrhip_lstm_args* a) {
  return a->w_split && a->hsplit && a->C % 128 == 0 && a->C <= 4096 && a->B >= 32;
}

// steps [t_lo, t_hi) of the layer (called by ssrhip_lstm_layer in csrc/codec.hip)
int ssrhip_lstm_split_steps(const ssrhip_lstm_args* a, int t_lo, int t_hi, hipStream_t s) {
  const int KS = a->C / 64;                                            // even: C % 128 == 0
  const size_t he = ssrhip_lstm_split_hplane_elems(a->B, a->C);
  unsigned short* hs = reinterpret_cast<unsigned short*>(a->hsplit);
  if (t_lo == 0) hipLaunchKernelGGL(zero_u32_kernel, dim3(1024), dim3(256), 0, s, reinterpret_cast<unsigned*>(hs), (long)he);   // both buffers: h_0 = 0
  static ssr_once_per_device once;
  if (once.need()) {
    SSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_step_split_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, ls_lds()));
    SSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_step_split_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, ls_lds()));
  }
  dim3 grid(a->C / LS_UN, (a->B + LS_BN - 1) / LS_BN);
  for (int t = t_lo; t < t_hi; ++t) {
    const unsigned short* hp = hs + (size_t)(t & 1) * he;
    unsigned short* hn = hs + (size_t)((t + 1) & 1) * he;
    if (KS % 4 == 0) hipLaunchKernelGGL((lstm_step_split_kernel<4>), grid, dim3(LS_TH), ls_lds(), s, *a, t, hp, hn, KS);   // three blocks of prefetch distance
    else hipLaunchKernelGGL((lstm_step_split_kernel<2>), grid, dim3(LS_TH), ls_lds(), s, *a, t, hp, hn, KS);
  }
  SSR_LAUNCH_CHECK();
  return 0;
}
