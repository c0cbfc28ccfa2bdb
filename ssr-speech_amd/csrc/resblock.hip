// resblock.hip — SEANetResnetBlock (audiocraft/modules/seanet.py:16-60, true_skip, dilation 1) as ONE kernel for the
// 64-channel, full-rate layers (16 kHz: T = 480,000 per 30 s clip), SURVEY §8 row B3 / K10:
//
//     y[t] = x[t] + b1 + W1 . ELU( b3 + W3 . ELU( x[t-1 : t+2] ) )          W3: [32][3*64], W1: [64][32]
//
// As two GEMM launches the 32-channel intermediate makes a round trip through HBM and the activation is read three times
// (once per tap) through L2. Here a wave owns 32 consecutive time steps:
//   * the 34 input rows (halo included) are fetched once, coalesced (one tile ahead), ELU'd once and parked in the wave's
//     LDS region;
//   * the weights stay on chip for the whole kernel, which is persistent over time tiles: W3 (24 KB) in LDS shared by the 8
//     waves of the workgroup, W1 and b3 in registers in the order stage 2 consumes them;
//   * stage 1  D1[h][t] = sum_k W3[h][k] ELU(x)[t][k]  on v_mfma_f32_32x32x2_f32 with the time steps as the N dimension:
//     the accumulator layout then leaves lane (t, half) with 16 hidden channels of ITS time step, which is exactly the
//     B-operand layout stage 2 needs (k = hidden channel, n = time step) — no shuffle, no LDS round trip for the
//     intermediate (the k order of stage 2 follows the accumulator's row order; W1 is preloaded in that order);
//   * stage 2  D2[c][t] = sum_h W1[c][h] ELU(D1 + b3)[h][t], then the tile goes back through LDS so that the residual add
//     and the store to HBM are coalesced 256-byte rows.
// Waves never synchronise with each other (wave-scope fences only). HBM traffic: one read + one write of the activation.
#include <stdlib.h>
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C = 64, H = 32, K3 = 3 * C, XS = C + 4;     // XS: padded LDS row stride (floats)

__device__ __forceinline__ float elu_fast(float v) { return elu1(v); }      // the codec's one ELU (common.h)

__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

constexpr int NWAVE = 8, W3S = K3 + 4;                    // waves per workgroup; padded LDS row stride of W3

__global__ __launch_bounds__(NWAVE * 64) void resblock64_kernel(const ssrhip_resblock_args a) {
  __shared__ __attribute__((aligned(16))) float W3s[H * W3S];        // 25 KB, shared by the 8 waves
  __shared__ __attribute__((aligned(16))) float Xs[NWAVE][34 * XS];  // per wave: ELU(x) rows t0-1 .. t0+32, later the output tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y;
  const float* xin = a.x + (size_t)b * a.x_bstride;          // row 0 = the halo row in front of t = 0
  float* yout = a.y + (size_t)b * a.y_bstride;
  float* xs = Xs[wave];

  // W3 -> LDS once per workgroup (A operand of stage 1, read as float4 per 4 MFMAs); W1 / b3 -> registers
  for (int i = threadIdx.x; i < H * K3 / 4; i += NWAVE * 64) {
    const int r = i / (K3 / 4), c4 = i % (K3 / 4);
    *reinterpret_cast<float4*>(W3s + r * W3S + c4 * 4) = ld4(a.w3 + (size_t)r * K3 + c4 * 4);
  }
  float w1r[2][16];                                          // A operand of stage 2: W1[c = 32mb + li][rho(s)], rho = accumulator row order
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int s = 0; s < 16; ++s) w1r[mb][s] = a.w1[(size_t)(mb * 32 + li) * H + (s & 3) + 8 * (s >> 2) + 4 * lh];
  float b3r[16];                                             // bias of the hidden channel held in accumulator register s
#pragma unroll
  for (int s = 0; s < 16; ++s) b3r[s] = a.b3[(s & 3) + 8 * (s >> 2) + 4 * lh];
  __syncthreads();                                           // the only workgroup barrier: W3s is ready

  const int T = a.T;
  const int ntile = (T + 31) / 32;
  const int stride = gridDim.x * NWAVE;
  // software pipeline: the NEXT tile's 34 input rows are requested (9 float4 per lane) before the current tile is computed
  float4 xn[9];
  auto fetch = [&](int tile) {
    const int t0 = min(tile, ntile - 1) * 32;              // past the end: re-read the last tile (never used)
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int idx = min(i * 64 + lane, 34 * 16 - 1);
      const int row = idx >> 4, c4 = idx & 15;
      xn[i] = ld4(xin + (size_t)min(t0 + row, T + 1) * C + c4 * 4);   // rows t0-1 .. t0+32 = buffer rows t0 .. t0+33, clamped to the right halo
    }
  };
  fetch(blockIdx.x * NWAVE + wave);
  for (int tile = blockIdx.x * NWAVE + wave; tile < ntile; tile += stride) {
    const int t0 = tile * 32;
    wave_lds_fence();                                        // previous tile's LDS reads are done
#pragma unroll
    for (int i = 0; i < 9; ++i) {                            // ELU once per element (it is used by 3 taps)
      const int idx = min(i * 64 + lane, 34 * 16 - 1);
      *reinterpret_cast<float4*>(xs + (idx >> 4) * XS + (idx & 15) * 4) =
          make_float4(elu_fast(xn[i].x), elu_fast(xn[i].y), elu_fast(xn[i].z), elu_fast(xn[i].w));
    }
    wave_lds_fence();
    fetch(tile + stride);
    // ---- stage 1: 24 x 4 MFMAs, both operands from LDS
    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
    for (int j = 0; j < K3 / 8; ++j) {
      const float4 wv = *reinterpret_cast<const float4*>(W3s + li * W3S + 8 * j + 4 * lh);
      const float4 xv = *reinterpret_cast<const float4*>(xs + (li + (j >> 3)) * XS + 8 * (j & 7) + 4 * lh);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, xv.x, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, xv.y, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, xv.z, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, xv.w, acc1, 0, 0, 0);
    }
    // ---- stage 2: hidden = ELU(acc1 + b3) straight from the accumulator registers
    f32x16 acc2[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc2[0][r] = 0.f; acc2[1][r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float hv = elu_fast(acc1[s] + b3r[s]);
      acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1r[0][s], hv, acc2[0], 0, 0, 0);
      acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1r[1][s], hv, acc2[1], 0, 0, 0);
    }
    // ---- transpose through LDS (the x tile is dead now): lane (t = li) holds channels 32mb + (r&3) + 8(r>>2) + 4lh
    wave_lds_fence();
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(xs + li * XS + mb * 32 + 8 * g + 4 * lh) =
            make_float4(acc2[mb][4 * g], acc2[mb][4 * g + 1], acc2[mb][4 * g + 2], acc2[mb][4 * g + 3]);
    wave_lds_fence();
    // ---- residual (raw x re-read: an L2 hit, coalesced) + bias, coalesced 256-byte rows out
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = i * 64 + lane;
      const int row = idx >> 4, c4 = idx & 15;
      const int tt = min(t0 + row, T - 1);
      const float4 xr = ld4(xin + (size_t)(tt + 1) * C + c4 * 4);
      const float4 v = *reinterpret_cast<const float4*>(xs + row * XS + c4 * 4);
      const float4 bb = ld4(a.b1 + c4 * 4);
      float4 o = make_float4(xr.x + (v.x + bb.x), xr.y + (v.y + bb.y), xr.z + (v.z + bb.z), xr.w + (v.w + bb.w));
      if (a.out_act == SSRHIP_ACT_ELU) { o.x = elu_fast(o.x); o.y = elu_fast(o.y); o.z = elu_fast(o.z); o.w = elu_fast(o.w); }
      if (t0 + row < T) *reinterpret_cast<float4*>(yout + (size_t)tt * C + c4 * 4) = o;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same block for C = 128 / 256 / 512 channels (the 8 kHz / 2 kHz / 400 Hz stages of SEANet) as TWO CHAINED GEMMs in one
// kernel. W3 (98 KB .. 1.5 MB) no longer fits next to the activation tiles, so the weights stream through LDS in 16-wide k-tiles
// like in gemm.hip — what is kept on chip is the C/2-channel INTERMEDIATE: stage 1 leaves ELU(b3 + W3 . ELU(x-window)) for a
// block of BM time steps in LDS (64 KB whatever C is: BM = 256 / 128 / 64 rows), stage 2 reads it as its A operand straight
// from there. As two GEMM launches the intermediate made a round trip through HBM and the 1x1 convolution ran on a thin
// K = C/2 GEMM that was bound by its three activation streams (C = 128 at 32 x 30 s: 5.6 + 4.6 ms for the two launches).
//   4 waves; both stages give every wave a 64-row slab: stage 1  (BM x C/2):  64 x 64  per wave = 4 accumulators,
//                                                       stage 2  (BM x C)  :  64 x 128 per wave = 8 accumulators.
//   v_mfma_f32_32x32x2_f32, operands via conflict-free ds_read_b128, next k-tile prefetched into registers under the MFMAs.
template <int CC, int BM>
__global__ __launch_bounds__(256) void resblock_chain_kernel(const ssrhip_resblock_args a) {
  constexpr int HH = CC / 2, MW = BM / 64, NW = 4 / MW;
  constexpr int NT1 = HH / (NW * 32), NT2 = CC / (NW * 32);        // 32-column accumulator blocks per wave in stage 1 / 2
  static_assert(MW * NW == 4 && NT1 >= 1 && NT2 >= 1 && NT1 * NW * 32 == HH && NT2 * NW * 32 == CC, "bad tile");
  constexpr int BKc = 16, LDT = BKc + 4, LDH = HH + 4;
  constexpr int LA = BM / 64, LW3 = HH / 64, LW1 = CC / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hs = smem;                          // [BM][LDH]  intermediate (A operand of stage 2)
  float* As = Hs + BM * LDH;                 // [BM][LDT]  stage-1 A tile
  float* Ws = As + BM * LDT;                 // [CC][LDT]  W3 tile (HH rows) / W1 tile (CC rows)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / NW, wn = wave % NW;
  const int lr = t >> 2, lc = (t & 3) * 4;
  const int T = a.T, m0 = blockIdx.x * BM;
  const float* xin = a.x + (size_t)blockIdx.y * a.x_bstride;      // row 0 = the halo row in front of t = 0
  float* yout = a.y + (size_t)blockIdx.y * a.y_bstride;
  constexpr int K3c = 3 * CC;

  // ---------------- stage 1: Hm[BM][HH] = ELU(x-window)[BM][3C] . W3^T
  f32x16 acc1[2][NT1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT1; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.f;
  float4 ra[LA], rw[LW1];
  auto gload1 = [&](int k0) {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int m = min(m0 + lr + 64 * i, T - 1);                  // rows past T: clamped (their outputs are not stored)
      float4 v = ld4(xin + (size_t)m * CC + k0 + lc);              // window of time m = padded rows m, m+1, m+2 = 3C contiguous floats
      ra[i] = make_float4(elu_fast(v.x), elu_fast(v.y), elu_fast(v.z), elu_fast(v.w));
    }
#pragma unroll
    for (int i = 0; i < LW3; ++i) rw[i] = ld4(a.w3 + (size_t)(lr + 64 * i) * K3c + k0 + lc);
  };
  gload1(0);
  for (int k0 = 0; k0 < K3c; k0 += BKc) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < LA; ++i) *reinterpret_cast<float4*>(&As[(lr + 64 * i) * LDT + lc]) = ra[i];
#pragma unroll
    for (int i = 0; i < LW3; ++i) *reinterpret_cast<float4*>(&Ws[(lr + 64 * i) * LDT + lc]) = rw[i];
    __syncthreads();
    if (k0 + BKc < K3c) gload1(k0 + BKc);
#pragma unroll
    for (int kk = 0; kk < BKc; kk += 8) {
      float4 a4[2], b4[NT1];
#pragma unroll
      for (int i = 0; i < 2; ++i) a4[i] = *reinterpret_cast<const float4*>(&As[(wm * 64 + i * 32 + li) * LDT + kk + lh * 4]);
#pragma unroll
      for (int j = 0; j < NT1; ++j) b4[j] = *reinterpret_cast<const float4*>(&Ws[((wn * NT1 + j) * 32 + li) * LDT + kk + lh * 4]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT1; ++j) {
          acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, b4[j].x, acc1[i][j], 0, 0, 0);
          acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, b4[j].y, acc1[i][j], 0, 0, 0);
          acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, b4[j].z, acc1[i][j], 0, 0, 0);
          acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, b4[j].w, acc1[i][j], 0, 0, 0);
        }
    }
  }
  // intermediate -> LDS with bias + ELU (accumulator element r: row (r&3) + 8(r>>2) + 4lh, column li of its 32x32 block)
#pragma unroll
  for (int j = 0; j < NT1; ++j) {
    const int hc = (wn * NT1 + j) * 32 + li;
    const float b3 = a.b3[hc];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Hs[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDH + hc] = elu_fast(acc1[i][j][r] + b3);
  }
  // ---------------- stage 2: Y[BM][C] = Hm[BM][HH] . W1^T  (+ b1 + x)
  f32x16 acc2[2][NT2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
  auto gload2 = [&](int k0) {
#pragma unroll
    for (int i = 0; i < LW1; ++i) rw[i] = ld4(a.w1 + (size_t)(lr + 64 * i) * HH + k0 + lc);
  };
  gload2(0);
  for (int k0 = 0; k0 < HH; k0 += BKc) {
    __syncthreads();                                               // first pass: Hs complete and the W3 tile consumed
#pragma unroll
    for (int i = 0; i < LW1; ++i) *reinterpret_cast<float4*>(&Ws[(lr + 64 * i) * LDT + lc]) = rw[i];
    __syncthreads();
    if (k0 + BKc < HH) gload2(k0 + BKc);
#pragma unroll
    for (int kk = 0; kk < BKc; kk += 8) {
      float4 a4[2], b4[NT2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a4[i] = *reinterpret_cast<const float4*>(&Hs[(wm * 64 + i * 32 + li) * LDH + k0 + kk + lh * 4]);
#pragma unroll
      for (int j = 0; j < NT2; ++j) b4[j] = *reinterpret_cast<const float4*>(&Ws[((wn * NT2 + j) * 32 + li) * LDT + kk + lh * 4]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT2; ++j) {
          acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, b4[j].x, acc2[i][j], 0, 0, 0);
          acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, b4[j].y, acc2[i][j], 0, 0, 0);
          acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, b4[j].z, acc2[i][j], 0, 0, 0);
          acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, b4[j].w, acc2[i][j], 0, 0, 0);
        }
    }
  }
  // ---------------- epilogue: + b1 + x (raw, the centre tap's row), 128-byte runs per accumulator row
#pragma unroll
  for (int j = 0; j < NT2; ++j) {
    const int n = (wn * NT2 + j) * 32 + li;
    const float b1 = a.b1[n];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < T) {
          float o = xin[(size_t)(m + 1) * CC + n] + (acc2[i][j][r] + b1);
          if (a.out_act == SSRHIP_ACT_ELU) o = elu_fast(o);
          yout[(size_t)m * CC + n] = o;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The chained block once more with both GEMMs on the bf16 matrix cores and EXACTLY split fp32 operands (round 3; the arithmetic
// of csrc/gemm_split.hip: a = a0 + a1 + a2 in bf16 pieces, the six largest cross products accumulated in fp32, error against fp64
// no larger than the fp32 FMA chain's). Built for C = 128 (the 8 kHz stage: 11 % of the codec's time at 256 clips x 30 s, 65 TFLOP/s
// on the fp32 pipe). Same structure as the kernel above; what changes:
//   * A (ELU of the x window) and the W3 / W1 tiles are split between the global load and the LDS store (three bf16 planes per tile,
//     16 k-values per tile, rows at a 48-byte pitch: a lane's ds_read_b128 = its whole 8-k MFMA operand, conflict-free);
//   * the C/2-channel intermediate stays in LDS as fp32 (35 KB: three bf16 planes of it would push the workgroup past half a CU's
//     LDS) and is split by the reading lane into its stage-2 A operand;
//   * per 16 k-values of a 32 x 32 block: 6 x v_mfma_f32_32x32x16_bf16 = 192 matrix-pipe cycles instead of 8 x 64 = 512.
typedef short bf16x8_ __attribute__((ext_vector_type(8)));
typedef short bf16x4_ __attribute__((ext_vector_type(4)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bfx2_ __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split4_(const float4 v, bf16x4_ (&out)[3]) {       // exact three-way split (see gemm_split.hip)
  f32x2_ r[2] = {{v.x, v.y}, {v.z, v.w}};
#pragma unroll
  for (int p = 0; p < 3; ++p) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bfx2_ b = __builtin_convertvector(r[h], bfx2_);
      const unsigned bits = __builtin_bit_cast(unsigned, b);
      out[p][2 * h] = (short)(bits & 0xFFFFu);
      out[p][2 * h + 1] = (short)(bits >> 16);
      if (p < 2) {
        const f32x2_ back = {__builtin_bit_cast(float, bits << 16), __builtin_bit_cast(float, bits & 0xFFFF0000u)};
        r[h] = r[h] - back;
      }
    }
  }
}

template <int CC, int BM>
__global__ __launch_bounds__(256, 2) void resblock_chain_split_kernel(const ssrhip_resblock_args a) {
  constexpr int HH = CC / 2, MW = BM / 64, NW = 4 / MW;
  constexpr int NT1 = HH / (NW * 32), NT2 = CC / (NW * 32);
  static_assert(MW * NW == 4 && NT1 >= 1 && NT2 >= 1 && NT1 * NW * 32 == HH && NT2 * NW * 32 == CC, "bad tile");
  constexpr int BKc = 16, PT = 24, LDH = HH + 4;                   // PT: bf16 row pitch (48 B)
  constexpr int LA = BM / 64, LW3 = HH / 64, LW1 = CC / 64;
  static_assert(LW3 >= 1, "C >= 128");
  extern __shared__ __attribute__((aligned(16))) float smem[];    // 70 KB: above the static limit, requested at launch
  float* Hs = smem;                                                // [BM][LDH] fp32 intermediate
  short (*As)[BM * PT] = reinterpret_cast<short (*)[BM * PT]>(Hs + BM * LDH);            // [3][BM * PT]
  short (*Ws)[CC * PT] = reinterpret_cast<short (*)[CC * PT]>(&As[3][0]);                // [3][CC * PT]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / NW, wn = wave % NW;
  const int lr = t >> 2, lc = (t & 3) * 4;
  const int T = a.T, m0 = blockIdx.x * BM;
  const float* xin = a.x + (size_t)blockIdx.y * a.x_bstride;      // row 0 = the halo row in front of t = 0
  float* yout = a.y + (size_t)blockIdx.y * a.y_bstride;
  constexpr int K3c = 3 * CC;
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // a2w0, a0w2, a1w1, a1w0, a0w1, a0w0 (smallest first)

  // ---------------- stage 1: Hm[BM][HH] = ELU(x-window)[BM][3C] . W3^T
  f32x16 acc1[2][NT1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT1; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.f;
  float4 ra[LA], rw[LW1];
  auto gload1 = [&](int k0) {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int m = min(m0 + lr + 64 * i, T - 1);                  // rows past T: clamped (their outputs are not stored)
      const float4 v = ld4(xin + (size_t)m * CC + k0 + lc);        // window of time m = padded rows m, m+1, m+2 = 3C contiguous floats
      ra[i] = make_float4(elu_fast(v.x), elu_fast(v.y), elu_fast(v.z), elu_fast(v.w));
    }
#pragma unroll
    for (int i = 0; i < LW3; ++i) rw[i] = ld4(a.w3 + (size_t)(lr + 64 * i) * K3c + k0 + lc);
  };
  gload1(0);
  for (int k0 = 0; k0 < K3c; k0 += BKc) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      bf16x4_ p[3];
      split4_(ra[i], p);
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4_*>(&As[q][(lr + 64 * i) * PT + lc]) = p[q];
    }
#pragma unroll
    for (int i = 0; i < LW3; ++i) {
      bf16x4_ p[3];
      split4_(rw[i], p);
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4_*>(&Ws[q][(lr + 64 * i) * PT + lc]) = p[q];
    }
    __syncthreads();
    if (k0 + BKc < K3c) gload1(k0 + BKc);
    bf16x8_ fa[3][2], fb[3][NT1];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[q][i] = *reinterpret_cast<const bf16x8_*>(&As[q][(wm * 64 + i * 32 + li) * PT + lh * 8]);
#pragma unroll
      for (int j = 0; j < NT1; ++j) fb[q][j] = *reinterpret_cast<const bf16x8_*>(&Ws[q][((wn * NT1 + j) * 32 + li) * PT + lh * 8]);
    }
#pragma unroll
    for (int pq = 0; pq < 6; ++pq)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT1; ++j) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pq]][i], fb[PB[pq]][j], acc1[i][j], 0, 0, 0);
  }
  // intermediate -> LDS (fp32) with bias + ELU
#pragma unroll
  for (int j = 0; j < NT1; ++j) {
    const int hc = (wn * NT1 + j) * 32 + li;
    const float b3 = a.b3[hc];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Hs[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDH + hc] = elu_fast(acc1[i][j][r] + b3);
  }
  // ---------------- stage 2: Y[BM][C] = Hm[BM][HH] . W1^T  (+ b1 + x)
  f32x16 acc2[2][NT2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
  auto gload2 = [&](int k0) {
#pragma unroll
    for (int i = 0; i < LW1; ++i) rw[i] = ld4(a.w1 + (size_t)(lr + 64 * i) * HH + k0 + lc);
  };
  gload2(0);
  for (int k0 = 0; k0 < HH; k0 += BKc) {
    __syncthreads();                                               // first pass: Hs complete and the W3 tile consumed
#pragma unroll
    for (int i = 0; i < LW1; ++i) {
      bf16x4_ p[3];
      split4_(rw[i], p);
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4_*>(&Ws[q][(lr + 64 * i) * PT + lc]) = p[q];
    }
    __syncthreads();
    if (k0 + BKc < HH) gload2(k0 + BKc);
    bf16x8_ fa[3][2], fb[3][NT2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                  // this lane's 8 k-values of the intermediate, split here
      const float* hp = &Hs[(wm * 64 + i * 32 + li) * LDH + k0 + lh * 8];
      bf16x4_ p0[3], p1[3];
      split4_(*reinterpret_cast<const float4*>(hp), p0);
      split4_(*reinterpret_cast<const float4*>(hp + 4), p1);
#pragma unroll
      for (int q = 0; q < 3; ++q) fa[q][i] = __builtin_shufflevector(p0[q], p1[q], 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int j = 0; j < NT2; ++j) fb[q][j] = *reinterpret_cast<const bf16x8_*>(&Ws[q][((wn * NT2 + j) * 32 + li) * PT + lh * 8]);
#pragma unroll
    for (int pq = 0; pq < 6; ++pq)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT2; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pq]][i], fb[PB[pq]][j], acc2[i][j], 0, 0, 0);
  }
  // ---------------- epilogue: + b1 + x (raw, the centre tap's row)
#pragma unroll
  for (int j = 0; j < NT2; ++j) {
    const int n = (wn * NT2 + j) * 32 + li;
    const float b1 = a.b1[n];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < T) {
          float o = xin[(size_t)(m + 1) * CC + n] + (acc2[i][j][r] + b1);
          if (a.out_act == SSRHIP_ACT_ELU) o = elu_fast(o);
          yout[(size_t)m * CC + n] = o;
        }
      }
  }
}

template <int CC, int BM>
int launch_resblock_chain(const ssrhip_resblock_args* a, hipStream_t s) {
  constexpr int HH = CC / 2;
  const size_t smem = ((size_t)BM * (HH + 4) + (size_t)BM * 20 + (size_t)CC * 20) * sizeof(float);
  static ssr_once_per_device once;
  if (once.need()) SSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_chain_kernel<CC, BM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL((resblock_chain_kernel<CC, BM>), dim3((a->T + BM - 1) / BM, a->B), dim3(256), smem, s, *a);
  return 0;
}

}  // namespace

int ssrhip_resblock_split_launch(const ssrhip_resblock_args* a, hipStream_t s);   // resblock_split.hip

extern "C" int ssrhip_resblock(const ssrhip_resblock_args* a, ssrhip_stream_t stream) {
  SSR_REQUIRE(a && a->x && a->y && a->w3 && a->b3 && a->w1 && a->b1, "ssrhip_resblock: null argument");
  SSR_REQUIRE(a->B > 0 && a->B <= 65535 && a->T > 0, "ssrhip_resblock: bad B / T");
  SSR_REQUIRE(!a->w3_split == !a->w1_split, "ssrhip_resblock: w3_split and w1_split come together");
  if (a->w3_split && (a->C == 64 || a->C == 128)) {                // the caller prepared the bf16 planes: the DMA kernel (round 4)
    static const bool split_off2 = (getenv("SSRHIP_GEMM_SPLIT") && getenv("SSRHIP_GEMM_SPLIT")[0] == '0') ||
                                   (getenv("SSRHIP_RESBLOCK_DMA") && getenv("SSRHIP_RESBLOCK_DMA")[0] == '0');   // A/B knobs
    if (!split_off2) return ssrhip_resblock_split_launch(a, (hipStream_t)stream);
  }
  if (a->C == 128 || a->C == 256 || a->C == 512) {
    static const int big = getenv("SSRHIP_RESCHAIN_BIG") ? atoi(getenv("SSRHIP_RESCHAIN_BIG")) : 0;   // tuning knob: the larger row block
    hipStream_t s = (hipStream_t)stream;
    int rc;
    static const bool split_off = getenv("SSRHIP_GEMM_SPLIT") && getenv("SSRHIP_GEMM_SPLIT")[0] == '0';   // A/B knob: the fp32 FMA chain everywhere
    if (a->C == 128 && !split_off && !big) {
      constexpr int SPLIT_SMEM = 128 * 68 * 4 + 3 * 128 * 24 * 2 + 3 * 128 * 24 * 2;     // Hs + As planes + Ws planes = 71,680 B
      static ssr_once_per_device once;
      if (once.need()) SSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_chain_split_kernel<128, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, SPLIT_SMEM));
      hipLaunchKernelGGL((resblock_chain_split_kernel<128, 128>), dim3((a->T + 127) / 128, a->B), dim3(256), SPLIT_SMEM, s, *a);
      SSR_LAUNCH_CHECK();
      return 0;
    }
    if (a->C == 128) rc = big ? launch_resblock_chain<128, 256>(a, s) : launch_resblock_chain<128, 128>(a, s);
    else if (a->C == 256) rc = big ? launch_resblock_chain<256, 128>(a, s) : launch_resblock_chain<256, 64>(a, s);
    else rc = launch_resblock_chain<512, 64>(a, s);
    if (rc) return rc;
    SSR_LAUNCH_CHECK();
    return 0;
  }
  SSR_REQUIRE(a->C == 64, "ssrhip_resblock: fused for C in {64, 128, 256, 512} (C=%d)", a->C);
  int gx = (256 + a->B - 1) / a->B;                       // ~1 eight-wave workgroup per CU in total; every wave then walks many tiles
  const int need = (a->T + 32 * NWAVE - 1) / (32 * NWAVE);
  if (gx > need) gx = need;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(resblock64_kernel, dim3(gx, a->B), dim3(NWAVE * 64), 0, (hipStream_t)stream, *a);
  SSR_LAUNCH_CHECK();
  return 0;
}
