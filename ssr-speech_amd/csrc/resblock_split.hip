// resblock_split.hip — SEANetResnetBlock (audiocraft/modules/seanet.py:16-60; true_skip, dilation 1, kernels 3 and 1) for C = 64 and
// C = 128 channels as ONE kernel on the bf16 matrix cores with exactly split fp32 operands (round 4).
//
//     y[t] = x[t] + b1 + W1 . ELU( b3 + W3 . ELU( x[t-1 : t+2] ) )          W3: [C/2][3][C], W1: [C][C/2]
//
// The arithmetic is csrc/gemm_split.hip's: every fp32 operand is the exact sum of three bf16 pieces, the six largest cross products are
// accumulated in fp32 (error against fp64 no larger than the fp32 FMA chain's). What round 3's resblock_chain_split_kernel<128> left on
// the table (55.6 ms per call at 256 clips x 30 s against 9.6 ms of matrix time and 12.6 ms of HBM time) and what is done
// about it here:
//   * W3 / W1 were split from fp32 again by every workgroup at every k-step: the host splits them ONCE (ssrhip_resblock_args.w3_split /
//     w1_split, three bf16 planes each) and the kernel brings the tiles global -> LDS by DMA, one step ahead, no registers, no ds_write;
//   * the x window was loaded, ELU'd and split once PER TAP (three times per element): a tile of ELU(x) — 130 rows x 32 channels, three
//     bf16 planes, XOR-swizzled 64-byte rows — is built once per channel tile and the three taps read it at row offsets 0 / 1 / 2;
//   * the C/2-channel intermediate made a round trip through LDS (fp32, split again by every reading wave): stage 1 runs TRANSPOSED
//     (H^T[h][t] = W3 . ELU(x)^T, the time steps as the N dimension), so that a lane's accumulator registers are 16 hidden channels of
//     ITS time step — exactly the A operand stage 2 wants (row = time step, k = hidden channel) once W1's k order follows the
//     accumulator's row order (the host permutes W1's columns; resblock64_kernel's trick, carried over to the bf16 pipe). A wave owns
//     32 time steps through both stages; waves exchange nothing but the shared weight / input tiles;
//   * two barriers per 16-wide k-step with 12 MFMAs between them: one barrier per 32-wide step with 24 MFMAs per wave between them
//     (two at the channel-tile boundaries), 4 waves per workgroup;
//   * a 32-wide step is ~0.3 us of MFMAs, an L2 -> LDS DMA ~1 us: the weight tiles go through a ring of FOUR LDS slots, three steps ahead
//     of their use (first version: one ahead, 36.6 ms per call at C = 128 — every step waited for its tile). The DMA is issued as inline
//     asm and counted by hand (vmcnt): through the builtin hipcc makes every ds_read wait for every LDS-DMA issued before it.
// Replaces the two strided-view GEMMs (or the round-3 fused kernels) for `SEANetResnetBlock` at C in {64, 128}; results are NOT
// bit-identical to them (same class of error; tests: G7 per-module fixtures and the end-to-end codec fixtures, 2e-4 / 2e-5).
#include <stdlib.h>
#include <algorithm>
using std::min;
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bfx2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ float elu_r(float v) { return elu1(v); }         // the codec's one ELU (common.h: hardware exponential, |error| <= 1.2e-7)

// exact three-way split of 4 consecutive values: piece p of element e in out[p][e]   (gemm_split.hip)
__device__ __forceinline__ void split4r(const float4 v, bf16x4 (&out)[3]) {
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

constexpr int RB_BM = 128, RB_TH = 256, RB_NW = 4;        // time steps per workgroup, threads, waves
constexpr int RB_ER = RB_BM + 2 + 2;                      // rows of the ELU(x) tile (130 used; padded to a multiple of 4 for the swizzle period)
constexpr int RB_EP = RB_ER * 64;                         // bytes per bf16 plane of the tile (64-byte rows = 32 channels)
constexpr int rb_wtile(int CC) { return 96 * CC; }        // bytes of one weight tile: W3 [3][C/2][32 k] = W1 [3][C][16 k] = 96 C
constexpr int NS1X(int CC) { return 3 * (CC / 32); }
// weight tiles in LDS = RING: the one in use + RING - 1 in flight (a 32-wide step is ~0.3 us of MFMAs, an L2 -> LDS DMA ~1 us)
constexpr int rb_lds(int CC, int RING) { return 3 * RB_EP + RING * rb_wtile(CC); }
typedef int i32x4r __attribute__((ext_vector_type(4)));

// LDS-DMA as inline asm: with the builtin hipcc applies its conservative alias rule — every ds_read behind an LDS-DMA first waits for
// the DMA (seen in the ISA: s_waitcnt vmcnt right behind the request), which makes a prefetch distance impossible. The asm form is
// invisible to that pass; its completion is counted by hand (wait_vm below). M0 = LDS byte address of the wave's KiB, restored afterwards.
__device__ __forceinline__ void dma_kib(const i32x4r rsrc, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(rsrc) : "memory");
}
// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;   // n >= 12: a stricter wait is always correct
  }
}

// Cost attribution hooks for tools/resblock_lab.hip (KO = 0 in the product: every `if constexpr` below folds away): knock one component
// out and time the rest, or (RB_PROF) leave 100 MHz timestamps of wave 0's phases in LDS and dump them for a sample of the workgroups.
enum { RB_KO_WAIT = 1, RB_KO_ESTORE = 2, RB_KO_RESID = 4, RB_KO_MFMA = 8, RB_KO_DMA = 16, RB_KO_ELOAD = 32, RB_KO_STORE = 64, RB_PROF = 128 };
constexpr int RB_NSTAMP = 64;
constexpr int RB_RDEPTH = 1;                             // residual blocks (16 registers each) requested ahead of the one being stored
__device__ unsigned* g_rb_prof = nullptr;                 // [sampled workgroup][RB_NSTAMP]

template <int CC, int RB_RING, int KO = 0>
__global__ __launch_bounds__(RB_TH, 2) void resblock_split_dma_kernel(const ssrhip_resblock_args a, const int wide) {
  constexpr int HH = CC / 2, NHB = HH / 32, NCT = CC / 32, NNB = CC / 32, NJ = HH / 16;
  constexpr int NS1 = NCT * 3;                             // weight tiles: NS1 of W3 (channel tile, tap), then NJ of W1
  constexpr int WT = rb_wtile(CC), NIT = 3 * HH / 16;      // DMA instructions per tile (1 KiB each): 3 planes x HH/16 (= 3 x CC/32)
  constexpr int CNT = (NIT + RB_NW - 1) / RB_NW;           // ... per wave (C = 64: 6 over 4 waves -> 2 each, the surplus repeats the last KiB)
  constexpr int NU = NS1X(CC) + HH / 16;                   // weight tiles in all
  constexpr int K3 = 3 * CC;
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // a2b0, a0b2, a1b1, a1b0, a0b1, a0b0 (smallest first)
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  char* const Es = lds;                                    // [3][RB_ER][64 B]
  char* const Wb = lds + 3 * RB_EP;                        // [RB_RING][WT]
  unsigned* const stamps = reinterpret_cast<unsigned*>(lds + rb_lds(CC, RB_RING));      // RB_PROF only (the lab adds the bytes)
  auto stamp = [&](int i) {
    if constexpr ((KO & RB_PROF) != 0) {
      if (threadIdx.x == 0) stamps[i] = (unsigned)wall_clock64();
    }
  };
  stamp(0);
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int T = a.T, m0 = blockIdx.x * RB_BM;
  const float* xin = a.x + (size_t)blockIdx.y * a.x_bstride;      // row 0 = the halo row in front of t = 0; T + 2 rows per item
  float* yout = a.y + (size_t)blockIdx.y * a.y_bstride;
  const short* w3p = reinterpret_cast<const short*>(a.w3_split);  // [3][HH][3 CC]
  const short* w1p = reinterpret_cast<const short*>(a.w1_split);  // [3][CC][HH], k in accumulator order (include/ssrhip.h)
  auto make_rsrc = [](const void* p, int bytes) {          // raw buffer descriptor: base, stride 0, extent, the flags make_buffer_rsrc uses
    const unsigned long long ad = (unsigned long long)(uintptr_t)p;
    i32x4r r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ad);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((ad >> 32) & 0xFFFFu));
    r[2] = bytes;
    r[3] = 0x00020000;
    return r;
  };
  const i32x4r rs3 = make_rsrc(w3p, 3 * HH * K3 * 2), rs1 = make_rsrc(w1p, 3 * CC * HH * 2);
  const unsigned wb_addr = (unsigned)(uintptr_t)(lds_ptr_t)Wb;          // LDS byte address of the ring

  // ---- weight tile u -> ring slot u % RB_RING by DMA (a wave-level instruction = one lane-linear KiB; CNT per wave and tile)
  auto dma_tile = [&](int u) {
    if constexpr ((KO & RB_KO_DMA) != 0) return;
    const unsigned dst = wb_addr + (unsigned)(u % RB_RING) * WT;
    if (u < NS1) {
      // W3 tile (channel tile ct, tap): rows h, 32 k = 64 B per row and plane. KiB = 16 rows; lane l lands in (row 16g + l/4, slot l%4),
      // which has to hold chunk (l%4) ^ ((row >> 2) & 3)
      const int ct = u / 3, tap = u % 3;
#pragma unroll
      for (int i = 0; i < CNT; ++i) {
        const int it = min(wave + RB_NW * i, NIT - 1);
        const int q = it / (HH / 16), g = it % (HH / 16);
        const int row = 16 * g + (lane >> 2), chunk = (lane & 3) ^ ((row >> 2) & 3);
        const unsigned off = (unsigned)((((size_t)q * HH + row) * K3 + tap * CC + ct * 32 + chunk * 8) * 2);
        dma_kib(rs3, off, __builtin_amdgcn_readfirstlane(dst + q * HH * 64 + g * 1024));
      }
    } else {
      // W1 tile j: rows n, 16 k' = 32 B per row and plane. KiB = 32 rows; lane l lands in (row 32g + l/2, slot l%2)
      const int j = u - NS1;
#pragma unroll
      for (int i = 0; i < CNT; ++i) {
        const int it = min(wave + RB_NW * i, NIT - 1);
        const int q = it / (CC / 32), g = it % (CC / 32);
        const int row = 32 * g + (lane >> 1);
        const int half = (lane & 1) ^ ((row >> 3) & 1);        // 32-byte rows: the two halves swap places every 8 rows (dense rows read 2-way)
        const unsigned off = (unsigned)((((size_t)q * CC + row) * HH + j * 16 + half * 8) * 2);
        dma_kib(rs1, off, __builtin_amdgcn_readfirstlane(dst + q * CC * 32 + g * 1024));
      }
    }
  };
  // tile u is complete when at most `younger` of this wave's vector-memory operations are still outstanding: the DMA of the tiles
  // requested after it (CNT each) and — while they are in flight — the NEL loads of the next ELU(x) tile, which were issued behind
  // DMA(v + 3) of their step v = 3 ct
  auto tiles_after = [&](int u) { return min(u + RB_RING - 2, NU - 1) - u; };        // tiles u+1 .. u+RING-2 that exist (u + RING - 1 is requested after the wait)
  // ---- ELU(x) tile of channel tile ct: 130 rows (times m0 - 1 .. m0 + 128) x 32 channels; 1040 float4 over 256 threads
  constexpr int NEL = (130 * 8 + RB_TH - 1) / RB_TH;       // 5
  float4 er[NEL];
  auto e_load = [&](int ct) {
#pragma unroll
    for (int i = 0; i < NEL; ++i) {
      const int idx = min(t + RB_TH * i, 130 * 8 - 1);      // the surplus threads of the last pass repeat the last piece (same value, same slot)
      const int row = idx >> 3, c4 = idx & 7;
      const int prow = min(m0 + row, T + 1);               // padded row; tiles that run past the item: clamped (those outputs are not stored)
      if constexpr ((KO & RB_KO_ELOAD) != 0) er[i] = make_float4(a.T * 1e-9f, 0.1f, -0.2f, 0.3f);
      else er[i] = ld4(xin + (size_t)prow * CC + ct * 32 + c4 * 4);
    }
  };
  auto e_store = [&]() {
    if constexpr ((KO & RB_KO_ESTORE) != 0) return;
#pragma unroll
    for (int i = 0; i < NEL; ++i) {
      const int idx = min(t + RB_TH * i, 130 * 8 - 1);
      const int row = idx >> 3, c4 = idx & 7;
      const float4 v = make_float4(elu_r(er[i].x), elu_r(er[i].y), elu_r(er[i].z), elu_r(er[i].w));
      bf16x4 p[3];
      split4r(v, p);
      const int off = row * 64 + (((c4 >> 1) ^ ((row >> 2) & 3)) << 4) + (c4 & 1) * 8;
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(Es + q * RB_EP + off) = p[q];
    }
  };

  f32x16 acc1[NHB];
#pragma unroll
  for (int hb = 0; hb < NHB; ++hb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[hb][r] = 0.f;
  f32x16 acc2[NNB];
  bf16x8 hf[3][NJ];                                        // stage-2 A operand: ELU(H + b3) of this lane's time step, split, in k' order

  e_load(0);
  // b3 of the hidden channels this lane's accumulator registers hold, requested NOW: a load between the stages would make the compiler
  // wait with vmcnt(0) there — it cannot see the DMA in flight — and drain the ring once per tile
  float b3r[NHB][16];
#pragma unroll
  for (int hb = 0; hb < NHB; ++hb)
#pragma unroll
    for (int r = 0; r < 16; ++r) b3r[hb][r] = a.b3[hb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
  for (int u0 = 0; u0 < RB_RING - 1; ++u0) dma_tile(u0);
  // ---- stage 1, transposed: acc1[hb][h][m] += W3[h][k] . E[m + tap][k]; a runtime loop over the channel tiles (a fully unrolled
  // 16-step loop was refused by the compiler and left the register arrays in scratch), the three taps of a tile unrolled
  for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
      const int u = ct * 3 + tap;
      if (tap == 0) {
        if (ct > 0) __syncthreads();                       // every wave has finished reading the previous channel tile
        e_store();                                         // (the compiler waits for the x loads here; it cannot see the DMA: vmcnt(0))
      }
      // x loads of the next channel tile are in flight behind DMA(3 ct + 3) during the steps of tap 1 and 2
      // (they are YOUNGER than tile u — and count — only while u <= 3 ct + RING - 1)
      if (tap == 0) stamp(40 + ct);
      if constexpr ((KO & (RB_KO_WAIT | RB_KO_DMA)) == 0) wait_vm(CNT * tiles_after(u) + ((tap != 0 && tap <= RB_RING - 1 && ct + 1 < NCT) ? NEL : 0));
      __syncthreads();                                     // tile u (and a new ELU(x) tile) visible; everyone is done with tile u - 1
      stamp(2 + 2 * u);
      if (u + RB_RING - 1 < NU) dma_tile(u + RB_RING - 1); // into the slot tile u - 1 has just left
      if (tap == 0 && ct + 1 < NCT) e_load(ct + 1);
      __builtin_amdgcn_sched_barrier(0);                   // keep the requests HERE, in front of the MFMA block
      const char* Wt = Wb + (u % RB_RING) * WT;
      const int erow = 32 * wave + li + tap;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 32; kk += 16) {
        const int cidx = (kk >> 3) + lh;
        bf16x8 fb[3], fa[3][NHB];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          fb[q] = *reinterpret_cast<const bf16x8*>(Es + q * RB_EP + erow * 64 + ((cidx ^ ((erow >> 2) & 3)) << 4));
#pragma unroll
          for (int hb = 0; hb < NHB; ++hb) {
            const int hrow = hb * 32 + li;
            fa[q][hb] = *reinterpret_cast<const bf16x8*>(Wt + q * HH * 64 + hrow * 64 + ((cidx ^ ((hrow >> 2) & 3)) << 4));
          }
        }
#pragma unroll
        for (int pq = 0; pq < 6; ++pq)
#pragma unroll
          for (int hb = 0; hb < NHB; ++hb)
            if constexpr ((KO & RB_KO_MFMA) == 0) acc1[hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pq]][hb], fb[PB[pq]], acc1[hb], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      stamp(3 + 2 * u);
    }
  }
  // ---- between the stages, in registers: accumulator register r of block hb = hidden channel hb*32 + (r&3) + 8(r>>2) + 4lh of time
  // step 32 wave + li; ELU(. + b3), exact three-way split, packed as the A operand of stage 2: k16-step j = 2 hb + (r >> 3), element r & 7
#pragma unroll
  for (int hb = 0; hb < NHB; ++hb)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = half * 8 + i;
        v[i] = elu_r(acc1[hb][r] + b3r[hb][r]);
      }
      bf16x4 p0[3], p1[3];
      split4r(make_float4(v[0], v[1], v[2], v[3]), p0);
      split4r(make_float4(v[4], v[5], v[6], v[7]), p1);
#pragma unroll
      for (int q = 0; q < 3; ++q) hf[q][2 * hb + half] = __builtin_shufflevector(p0[q], p1[q], 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
  for (int nb = 0; nb < NNB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[nb][r] = 0.f;
  // ---- the residual: x at the accumulator's own (time step, channel) positions, 16 values per lane and output block. Requested two
  // blocks ahead of their use and ALL 16 at once: the first form of this epilogue (load -> add -> store per output, under a per-row
  // predicate) compiled to 64 dependent HBM round trips per lane, 30 of the workgroup's 49 us (tools/resblock_lab.hip,
  // profiles/r04_microbench/resblock_lab.log). Rows past the end of the item are clamped (their outputs are not stored).
  const bool whole = m0 + RB_BM <= T;                      // uniform; the one ragged tile of an item takes the plain loop at the end
  const int mrow = m0 + 32 * wave + 4 * lh;
  // The epilogue works on 16-byte pieces: an accumulator block (lane = one channel, 16 time steps) is turned through a wave-private
  // 4 KB of the (by then dead) ELU(x) tile into rows — lane l holds channels 4 (l % 8) .. + 3 of time steps l / 8 + 8 p, p = 0..3 — so
  // that the residual comes in and the result goes out as dwordx4 (128-byte row pieces per 8 lanes, the x tile's own pattern) instead
  // of 16 + 16 dword instructions per block (store-issue-bound: the guide's `attention epilogue store tail` row). 32-dword rows are
  // conflict-free for both the ds_write_b32 (32 lanes = one row) and the ds_read_b128 lane groups.
  const int trow = lane >> 3, tc4 = lane & 7;
  const unsigned lane_el4 = (unsigned)(trow * CC + tc4 * 4);   // lane part of the element index; the rest is wave-uniform (scalar base + immediate)
  auto res_load4 = [&](int nb, float4 (&xr)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float* src = xin + ((size_t)(m0 + 32 * wave + 8 * p + 1) * CC + nb * 32);      // + 1: the centre tap's row
      if constexpr ((KO & RB_KO_RESID) != 0) xr[p] = make_float4(0.25f, 0.25f, 0.25f, 0.25f);
      else xr[p] = ld4(src + lane_el4);
    }
  };
  float4 xr4[2][4];
  // the dword form (SSRHIP_EPILOGUE_WIDE=0): the accumulator's own positions, 16 values per lane and block
  const unsigned lane_el = (unsigned)(4 * lh * CC + li);
  auto res_load = [&](int nb, float (&xr)[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float* src = xin + ((size_t)(m0 + 32 * wave + 8 * g + 1) * CC + nb * 32);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr ((KO & RB_KO_RESID) != 0) xr[4 * g + i] = 0.25f;
        else xr[4 * g + i] = src[lane_el + (unsigned)(i * CC)];
      }
    }
  };
  float xr[2][16];
  // ---- stage 2: acc2[nb][m][n] += H[m][k'] . W1[n][k'] for the 16 hidden channels of step j (unrolled: hf[.][j] stays in registers)
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int u = NS1 + j;
    if constexpr ((KO & (RB_KO_WAIT | RB_KO_DMA)) == 0) wait_vm(CNT * tiles_after(u));
    __syncthreads();
    stamp(2 + 2 * u);
    if (u + RB_RING - 1 < NU) dma_tile(u + RB_RING - 1);
    if (j == NJ - 1 && whole) {                              // behind the last DMA wait: from here on the compiler's own vmcnt bookkeeping is complete
      if (wide) {
        res_load4(0, xr4[0]);
        if constexpr (RB_RDEPTH > 1) res_load4(1, xr4[1]);
      } else {
        res_load(0, xr[0]);
        if constexpr (RB_RDEPTH > 1) res_load(1, xr[1]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const char* Wt = Wb + (u % RB_RING) * WT;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int nb = 0; nb < NNB; ++nb) {                       // one output block at a time: 12 operand registers instead of 12 x NNB
      bf16x8 fb[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) fb[q] = *reinterpret_cast<const bf16x8*>(Wt + q * CC * 32 + (nb * 32 + li) * 32 + ((lh ^ ((li >> 3) & 1)) << 4));
#pragma unroll
      for (int pq = 0; pq < 6; ++pq)
        if constexpr ((KO & RB_KO_MFMA) == 0) acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[PA[pq]][j], fb[PB[pq]], acc2[nb], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    stamp(3 + 2 * u);
  }
  // ---- epilogue: + b1 + x, 128-byte runs per accumulator row; the loads of the next block(s) go out before this one is stored
  float b1r[NNB];
#pragma unroll
  for (int nb = 0; nb < NNB; ++nb) b1r[nb] = a.b1[nb * 32 + li];
  const bool act = a.out_act == SSRHIP_ACT_ELU;
  if (whole && wide) {
    float* const tr = reinterpret_cast<float*>(Es) + wave * (32 * 32);
#pragma unroll
    for (int nb = 0; nb < NNB; ++nb) {
      if constexpr (RB_RDEPTH == 1) { if (nb + 1 < NNB) res_load4(nb + 1, xr4[(nb + 1) & 1]); }
#pragma unroll
      for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc2[nb][r] + b1r[nb];
      __builtin_amdgcn_wave_barrier();                       // the wave's own LDS accesses complete in order; this keeps the compiler from moving them
      float4 o[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float4 v = *reinterpret_cast<const float4*>(tr + (trow + 8 * p) * 32 + tc4 * 4);
        const float4 x4 = xr4[nb & 1][p];
        o[p] = make_float4(x4.x + v.x, x4.y + v.y, x4.z + v.z, x4.w + v.w);
      }
      if (act) {                                             // what a consumer would compute on load (common.h): ELU-on-store stays bit-identical to ELU-on-load
#pragma unroll
        for (int p = 0; p < 4; ++p) o[p] = make_float4(elu1(o[p].x), elu1(o[p].y), elu1(o[p].z), elu1(o[p].w));
      }
      __builtin_amdgcn_wave_barrier();
      if constexpr (RB_RDEPTH > 1) { if (nb + 2 < NNB) res_load4(nb + 2, xr4[nb & 1]); }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float* dst = yout + ((size_t)(m0 + 32 * wave + 8 * p) * CC + nb * 32);
        if ((KO & RB_KO_STORE) == 0 || o[p].x == 1.2345e-33f) *reinterpret_cast<float4*>(dst + lane_el4) = o[p];
      }
    }
  } else if (whole) {
#pragma unroll
    for (int nb = 0; nb < NNB; ++nb) {
      if constexpr (RB_RDEPTH == 1) { if (nb + 1 < NNB) res_load(nb + 1, xr[(nb + 1) & 1]); }
      float o[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = xr[nb & 1][r] + (acc2[nb][r] + b1r[nb]);
        o[r] = act ? elu1(v) : v;
      }
      if constexpr (RB_RDEPTH > 1) { if (nb + 2 < NNB) res_load(nb + 2, xr[nb & 1]); }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float* dst = yout + ((size_t)(m0 + 32 * wave + 8 * g) * CC + nb * 32);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if ((KO & RB_KO_STORE) == 0 || o[4 * g + i] == 1.2345e-33f) dst[lane_el + (unsigned)(i * CC)] = o[4 * g + i];
      }
    }
  } else {
#pragma unroll
    for (int nb = 0; nb < NNB; ++nb) {
      const size_t base = (size_t)mrow * CC + nb * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dm = (r & 3) + 8 * (r >> 2);
        if (mrow + dm < T) {
          float o = xin[base + (size_t)(dm + 1) * CC] + (acc2[nb][r] + b1r[nb]);
          if (a.out_act == SSRHIP_ACT_ELU) o = elu1(o);
          yout[base + (size_t)dm * CC] = o;
        }
      }
    }
  }
  if constexpr ((KO & RB_PROF) != 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(2 + 2 * NU);
    __syncthreads();
    if (blockIdx.x % 61 == 7 && blockIdx.y % 8 == 3 && threadIdx.x < RB_NSTAMP) {
      const int slot = (blockIdx.y / 8) * ((gridDim.x + 53) / 61) + blockIdx.x / 61;
      g_rb_prof[(size_t)slot * RB_NSTAMP + threadIdx.x] = stamps[threadIdx.x];
    }
  }
}

}  // namespace

// launched by ssrhip_resblock (resblock.hip) when the caller supplied the weight planes
int ssrhip_resblock_split_launch(const ssrhip_resblock_args* a, hipStream_t s) {
  SSR_REQUIRE(a->C == 64 || a->C == 128, "ssrhip_resblock (split planes): C=%d not in {64, 128}", a->C);
  SSR_REQUIRE((size_t)(a->T + 2) * a->C * 4 < 0x7FFFFFF0ull, "ssrhip_resblock (split planes): item too long");
  dim3 grid((a->T + RB_BM - 1) / RB_BM, a->B);
  // ring of 2 (one tile ahead, three workgroups per CU) measured FASTER than a ring of 4 (three ahead, two workgroups per CU): 4.76 vs 5.46 ms
  // at 32 clips, C = 128 (profiles/r04_microbench/resblock_pmc.log): a third workgroup hides more than the deeper prefetch does
  static const int ring = getenv("SSRHIP_RESBLOCK_RING") ? atoi(getenv("SSRHIP_RESBLOCK_RING")) : 2;      // A/B knob
  // the 16-byte epilogue (accumulator blocks turned through LDS) measured 1-3 % SLOWER here than the dword form (3.29 vs 3.21 ms at C = 128,
  // 2.33 vs 2.29 at C = 64, 32 clips, alternating runs, bit-identical outputs: profiles/r04_microbench/resblock_lab_ab.log) — unlike in the GEMM,
  // whose default it is; kept behind the same knob for A/B runs
  static const int wide = getenv("SSRHIP_EPILOGUE_WIDE") ? getenv("SSRHIP_EPILOGUE_WIDE")[0] != '0' : 0;
  if (ring == 2) {
    if (a->C == 128) hipLaunchKernelGGL((resblock_split_dma_kernel<128, 2>), grid, dim3(RB_TH), rb_lds(128, 2), s, *a, wide);
    else hipLaunchKernelGGL((resblock_split_dma_kernel<64, 2>), grid, dim3(RB_TH), rb_lds(64, 2), s, *a, wide);
  } else if (a->C == 128) {
    static ssr_once_per_device once;
    if (once.need()) SSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_split_dma_kernel<128, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, rb_lds(128, 4)));
    hipLaunchKernelGGL((resblock_split_dma_kernel<128, 4>), grid, dim3(RB_TH), rb_lds(128, 4), s, *a, wide);
  } else {
    hipLaunchKernelGGL((resblock_split_dma_kernel<64, 4>), grid, dim3(RB_TH), rb_lds(64, 4), s, *a, wide);
  }
  SSR_LAUNCH_CHECK();
  return 0;
}
