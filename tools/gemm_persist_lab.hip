// tools/gemm_persist_lab.hip — EXPERIMENT (round 3, negative result): a persistent fp32 MFMA GEMM with ONE wave per SIMD, measured
// against the library's ssrhip_gemm (csrc/gemm.hip: 64 x 128 x 16 tiles, several workgroups per CU). Built because
// tools/mfma_peak.hip shows the bare v_mfma_f32_32x32x2_f32 stream at 155 TFLOP/s with one wave per SIMD and at 103 with two; the
// kernel below hides its own latencies instead of relying on a second workgroup (design notes in the kernel comment). Outcome on
// one MI355X, warm clocks (profiles/r03_microbench/gemm_persist.log): 4096^3 120.7 TFLOP/s vs 120.8 for the library kernel, and
// 50..97 vs 92..116 on the codec's batched shapes (K = 256..8192, M = 1500..240000 per item): the per-tile epilogue and the first
// tile's prologue are not covered by anything when a CU holds one workgroup, and the ELU / staging work in the MFMA shadows is
// not free. Ablations of the same kernel on 4096^3 (5-launch runs, cooler clocks): 109.7 as is, 122 without the global loads, 125
// without the LDS stores, 116 without the fragment reads, 131 with the MFMAs alone. Kept for the record; not linked into the library.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -Issr-speech_amd/csrc tools/gemm_persist_lab.hip \
//          -o tools/bin/gemm_persist_lab -Lssr-speech_amd/csrc -lssrhip -Wl,-rpath,'$ORIGIN/../../ssr-speech_amd/csrc'
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <type_traits>
#include "common.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
void ssrhip_set_error(const char*, ...) {}

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float act_fn(float v, int act) {
  if (act == SSRHIP_ACT_RELU) return fmaxf(v, 0.f);
  if (act == SSRHIP_ACT_GELU_ERF) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  return v;
}

// ---- round 3: ONE wave per SIMD, software-pipelined, persistent ------------------------------------------------------------------
// tools/mfma_peak.hip: v_mfma_f32_32x32x2_f32 issued back to back by ONE wave per SIMD runs at 155 TFLOP/s; the same loop with TWO
// waves per SIMD at 103 (two streams of the 64-cycle fp32 MFMA do not interleave for free). Every tile shape that relies on a second
// workgroup per CU to cover its barrier / staging gaps therefore tops out at ~105 TFLOP/s (measured: 64x128x16, 128x128x32 in two
// buffers, 64x64x32: all 92..110 on 4096^3). So: one 4-wave workgroup per CU (110 KB of LDS keeps a second one out) and the wave
// hides its own latencies:
//   * three LDS buffers, ONE barrier per k-tile: tile t+2 is written while tile t is computed, so tile t+1 is already complete and
//     visible — its first fragments are read BEFORE the barrier that ends step t, and the MFMAs continue behind it at once;
//   * the global loads of k-tile t+3 are issued during step t and stored to LDS during step t+1 (a whole k-tile of MFMAs, ~2 us,
//     covers the memory latency; one sub-step was measured too short), row by row in the shadow of an MFMA each, every wave in its
//     own slot (the four waves of a CU share one LDS store path);
//   * inside the k-tile the fragments of sub-step s+1 are requested in the first slots of sub-step s (two fragment register sets);
//   * PERSISTENT: a workgroup walks over several output tiles and the (tile, k-tile) steps form ONE stream — the staging runs three
//     steps ahead across tile boundaries, so only the very first tile of a workgroup pays an exposed prologue (with one workgroup per
//     CU nothing else would cover it: K = 256 is 8 steps of 2 us against ~6 us of exposed loads). The epilogue of a tile is the only
//     bubble left.
//   * the loop body is ONE basic block: no branch, every select is data. The issue order below is the source order (sched_barrier)
//     and hipcc counts the outstanding loads exactly (vmcnt(7) in front of each LDS store).
// Same arithmetic as the other tiles (per output element a k-ordered fp32 FMA chain).
typedef float v4f __attribute__((ext_vector_type(4)));
// order fence for the hand-written schedule below: NOTHING moves across it. (Letting plain ALU work move — mask 0x7 — made hipcc
// hoist the zero-masking of a staged float4 to right behind its global load, with an s_waitcnt vmcnt(0) in front: a full memory
// round trip per k-tile in the middle of the MFMA stream.)
#define SSR_PIN() __builtin_amdgcn_sched_barrier(0)

template <int MT, int NT, bool ELU>
__global__ __launch_bounds__(256, 1) void gemm_persist_kernel(const ssrhip_gemm_args a, const int nbx, const int nby, const int ntiles) {
  constexpr int MW = 2, NW = 2;                                   // 4 waves = one per SIMD
  constexpr int BK_ = 32;
  constexpr int THREADS = 256;
  constexpr int BM = MW * MT * 32, BN = NW * NT * 32;
  constexpr int LDSW_ = BK_ + 4;                                  // row pitch 36 floats: 16 consecutive rows -> 16 distinct 16-byte slots
  constexpr int TPR_ = BK_ / 4, RPP_ = THREADS / TPR_;            // loader: 8 threads per tile row, 32 rows per pass
  static_assert(BM % RPP_ == 0 && BN % RPP_ == 0, "tile rows must be a multiple of the loader's rows per pass");
  constexpr int LA = BM / RPP_, LW = BN / RPP_;
  constexpr int BUF = (BM + BN) * LDSW_;                          // floats per LDS buffer: A rows then W rows
  constexpr int Q = 4 * MT * NT;                                  // MFMAs per sub-step (8 k-values)
  constexpr int SPA = Q / LA, SPW = Q / LW;                       // MFMA slots per A row / W row
  static_assert(SPA >= 1 && SPW >= 1 && MT + NT <= Q, "a staging piece per MFMA slot at most");
  constexpr int EPS = (4 * LA + Q - 1) / Q;                       // ELU elements per slot
  extern __shared__ __attribute__((aligned(16))) float lds[];     // 3 buffers

  // ---- which tiles are mine: every XCD gets a contiguous run of the tile order, its workgroups take that run round-robin ---------
  const int nwg = gridDim.x, id = blockIdx.x;
  const int xcd = id & 7, loc = id >> 3, wpx = nwg >> 3;          // host guarantees nwg % 8 == 0
  const int tq = ntiles >> 3, tr = ntiles & 7;
  const int t_start = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int t_cnt = tq + (xcd < tr ? 1 : 0);
  if (loc >= t_cnt) return;
  const int n_my = (t_cnt - loc + wpx - 1) / wpx;

  const int t = threadIdx.x, wave = t >> 6;
  const int li_ = t & 31, lh_ = (t & 63) >> 5;
  const int wm = wave / NW, wn = wave % NW;
  const int lr = t / TPR_, lc = (t % TPR_) * 4;
  const int M = a.M, N = a.N, K = a.K;
  const int nk = (K + BK_ - 1) / BK_;                             // host guarantees nk >= 3

  // tile order inside an item: groups of GM tile rows, column-major inside a group, so that the ~32 tiles an XCD works on at one time
  // form a GM x (32 / GM) block: per k-step they fetch GM activation stripes + 32 / GM weight stripes instead of 1 + 32 (row-major)
  constexpr int GM = 4;
  struct Coord { int bx, by, bz; };
  auto coords = [&](int tile) {
    Coord c;
    c.bz = tile / (nbx * nby);
    const int tz = tile - c.bz * (nbx * nby);
    const int grp = tz / (GM * nbx), tg = tz - grp * (GM * nbx);
    const int rows_g = min(GM, nby - grp * GM);
    c.by = grp * GM + tg % rows_g;
    c.bx = tg / rows_g;
    return c;
  };
  // this thread's rows of a tile's A / W panels (clamped: loads are unconditional; rows beyond M / N only feed outputs never stored)
  auto panel_ptrs = [&](const Coord& c, const float** pa_, const float** pw_) {
    const float* Ab = a.A + (size_t)c.bz * (size_t)a.strideA;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int m = c.by * BM + lr + RPP_ * i;
      pa_[i] = Ab + (size_t)(m < M ? m : 0) * a.lda + lc;
    }
#pragma unroll
    for (int i = 0; i < LW; ++i) {
      const int n = c.bx * BN + lr + RPP_ * i;
      pw_[i] = a.W + (size_t)(n < N ? n : 0) * K + lc;
    }
  };

  f32x16 acc[MT][NT];
  const float* pa[LA];
  const float* pw[LW];
  const float* pan[LA];
  const float* pwn[LW];
  v4f ra[LA], rw[LW];
  v4f fa[2][MT], fb[2][NT];
  const int st_off = lr * LDSW_ + lc;                             // this thread's float4 slot inside a panel (row lr + RPP_ * i)
  const int a_off = ((wm * MT) * 32 + li_) * LDSW_ + lh_ * 4;
  const int b_off = BM * LDSW_ + ((wn * NT) * 32 + li_) * LDSW_ + lh_ * 4;
  const v4f zero4 = {0.f, 0.f, 0.f, 0.f};

  auto frag_piece = [&](int set, int buf, int sub, int r) {
    const float* base = lds + buf * BUF + sub * 8;
    if (r < NT) fb[set][r] = *reinterpret_cast<const v4f*>(base + b_off + r * 32 * LDSW_);
    else if (r < NT + MT) fa[set][r - NT] = *reinterpret_cast<const v4f*>(base + a_off + (r - NT) * 32 * LDSW_);
  };
  auto store_a_row = [&](int i, int buf) { *reinterpret_cast<v4f*>(lds + buf * BUF + st_off + i * RPP_ * LDSW_) = ra[i]; };
  auto store_w_row = [&](int i, bool kin, int buf) {             // beyond K (k-tail of a panel's last k-tile) the W side is zeroed: that
    v4f v = rw[i];                                                // cancels whatever finite A values sit there
    if (!kin) v = zero4;
    *reinterpret_cast<v4f*>(lds + buf * BUF + BM * LDSW_ + st_off + i * RPP_ * LDSW_) = v;
  };
  auto elu_row = [&](int i) { ra[i][0] = elu1(ra[i][0]); ra[i][1] = elu1(ra[i][1]); ra[i][2] = elu1(ra[i][2]); ra[i][3] = elu1(ra[i][3]); };
  // One sub-step = the Q MFMAs of 8 k-values; `filler(p)` is issued right behind MFMA p.
  auto mma_sub = [&](int set, auto&& filler) {
#pragma unroll
    for (int p = 0; p < Q; ++p) {
      const int c = p / (MT * NT), i = (p / NT) % MT, j = p % NT;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i][c], fb[set][j][c], acc[i][j], 0, 0, 0);
      filler(p);
      SSR_PIN();
    }
  };

  // ---- prologue of the workgroup's FIRST tile: k-tiles 0 and 1 into buffers 0 and 1, k-tile 2 into the registers --------------------
  Coord cur = coords(t_start + loc);
  panel_ptrs(cur, pa, pw);
#pragma unroll
  for (int kt0 = 0; kt0 < 3; ++kt0) {
    const bool kin = (kt0 * BK_ + lc) < K;
    const int ko = kin ? kt0 * BK_ : 0;
#pragma unroll
    for (int i = 0; i < LA; ++i) ra[i] = *reinterpret_cast<const v4f*>(pa[i] + ko);
#pragma unroll
    for (int i = 0; i < LW; ++i) rw[i] = *reinterpret_cast<const v4f*>(pw[i] + ko);
    if (ELU && kt0 < 2) {
#pragma unroll
      for (int i = 0; i < LA; ++i) elu_row(i);
    }
    if (kt0 < 2) {
#pragma unroll
      for (int i = 0; i < LA; ++i) store_a_row(i, kt0);
#pragma unroll
      for (int i = 0; i < LW; ++i) store_w_row(i, kin, kt0);
    }
  }
  if (ELU) {
    // k-tile 2 waits in the registers: the stream applies ELU to a row's components in the slots before its store, some of which
    // belong to the END of the previous step — for the very first step those never ran: apply exactly those components here
    const int wsa0 = (wave * SPA) / 4;
#pragma unroll
    for (int i = 0; i < LA; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (SPA * i + wsa0 - (3 - c) < 0) ra[i][c] = elu1(ra[i][c]);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < MT + NT; ++r) frag_piece(0, 0, 0, r);
  int b0 = 0, b1 = 1, b2 = 2;                                     // buffers of step s, s+1, s+2

  // One step (= one k-tile of the current output tile). wsa / wsw: this wave's slot inside a row's window (compile-time constants).
  auto step = [&](const int kt, const int wsa, const int wsw) {
    // the k-tile requested now is three steps ahead: k-tile kt+3 of this output tile, or k-tile kt+3-nk of the NEXT one
    const bool nxt = kt + 3 >= nk;
    const int k3 = (nxt ? kt + 3 - nk : kt + 3) * BK_;
    const int ko = (k3 + lc) < K ? k3 : 0;
    // the k-tile stored now (two steps ahead) may be a panel's last one: its K tail is zeroed on the W side
    const int k2 = (kt + 2 >= nk ? kt + 2 - nk : kt + 2) * BK_;
    const bool kin_s = (k2 + lc) < K;
    // ELU of a staged A row: its four components in the four slots that end with the row's store slot (slots count through the whole
    // step, wrapping: row 0's first components sit at the end of the PREVIOUS step's stream) — a row's data then has ~60 MFMAs to arrive
    auto elu_slots = [&](int g) {
      if (!ELU) return;
#pragma unroll
      for (int i = 0; i < LA; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (g == (SPA * i + wsa - (3 - c) + 4 * Q) % (4 * Q)) ra[i][c] = elu1(ra[i][c]);
    };
    // sub-step 0: A rows of step s+2 to LDS, A rows of step s+3 requested
    mma_sub(0, [&](int p) {
      if (p < MT + NT) frag_piece(1, b0, 1, p);
      elu_slots(p);
      if (wsa == p % SPA && p / SPA < LA) {
        store_a_row(p / SPA, b2);
        ra[p / SPA] = *reinterpret_cast<const v4f*>((nxt ? pan[p / SPA] : pa[p / SPA]) + ko);
      }
    });
    // sub-step 1: the same for the W rows
    mma_sub(1, [&](int p) {
      if (p < MT + NT) frag_piece(0, b0, 2, p);
      elu_slots(Q + p);
      if (wsw == p % SPW && p / SPW < LW) {
        store_w_row(p / SPW, kin_s, b2);
        rw[p / SPW] = *reinterpret_cast<const v4f*>((nxt ? pwn[p / SPW] : pw[p / SPW]) + ko);
      }
    });
    mma_sub(0, [&](int p) {
      if (p < MT + NT) frag_piece(1, b0, 3, p);
      elu_slots(2 * Q + p);
    });
    // sub-step 3: first fragments of step s+1 (complete and visible since the previous barrier)
    mma_sub(1, [&](int p) {
      if (p < MT + NT) frag_piece(0, b1, 0, p);
      elu_slots(3 * Q + p);
    });
    __syncthreads();
    const int tmp = b0; b0 = b1; b1 = b2; b2 = tmp;
  };

  for (int j = 0; j < n_my; ++j) {
    // staging pointers of the next tile (the last tile re-requests its own first k-tiles: never used, but the stream stays uniform)
    const Coord nxc = j + 1 < n_my ? coords(t_start + loc + (j + 1) * wpx) : cur;
    panel_ptrs(nxc, pan, pwn);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int jj = 0; jj < NT; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
    // the k-loop once per wave of the workgroup, with that wave's store slots as constants (no branch in the loop body)
    switch (wave) {
      case 0: for (int kt = 0; kt < nk; ++kt) step(kt, 0, 0); break;
      case 1: for (int kt = 0; kt < nk; ++kt) step(kt, (1 * SPA) / 4, (1 * SPW) / 4); break;
      case 2: for (int kt = 0; kt < nk; ++kt) step(kt, (2 * SPA) / 4, (2 * SPW) / 4); break;
      default: for (int kt = 0; kt < nk; ++kt) step(kt, (3 * SPA) / 4, (3 * SPW) / 4); break;
    }
    // ---- epilogue of tile `cur` (C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) ----------
    {
      const size_t z = cur.bz;
      float* Cb = a.C + z * (size_t)a.strideC;
      const float* Rb = a.R ? a.R + z * (size_t)a.strideR : nullptr;
      const int32_t* rcls = a.rclass ? a.rclass + z * (size_t)a.rclass_stride : nullptr;
      const int m0 = cur.by * BM, n0 = cur.bx * BN;
#pragma unroll
      for (int jj = 0; jj < NT; ++jj) {
        const int n = n0 + (wn * NT + jj) * 32 + li_;
        if (n >= N) continue;
        const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh_;
            if (m < M) {
              if (a.tm_c > 0) {   // transposed-conv trimming: rows of the full output outside [tm_lo, tm_hi) are not stored
                const long u = ((long)m * N + n) / a.tm_c;
                if (u < a.tm_lo || u >= a.tm_hi) continue;
              }
              float v = act_fn(acc[mt][jj][r] + bias, a.act);
              float* c = Cb + (size_t)m * a.ldc + n;
              if (a.residual) v += *c;
              if (Rb) v += Rb[(size_t)m * a.ldr + n];
              if (a.rbias) v += a.rbias[(size_t)rcls[m / a.rrep] * N + n];
              *c = v;
            }
          }
        }
      }
    }
    cur = nxc;
#pragma unroll
    for (int i = 0; i < LA; ++i) pa[i] = pan[i];
#pragma unroll
    for (int i = 0; i < LW; ++i) pw[i] = pwn[i];
  }
}

template <int MT, int NT>
static int launch_persist(const ssrhip_gemm_args* a, hipStream_t s) {
  constexpr int BM = 2 * MT * 32, BN = 2 * NT * 32;
  constexpr int TILE_BYTES = 3 * (BM + BN) * 36 * 4;
  static_assert(TILE_BYTES <= 160 * 1024, "three k-tiles must fit the CU's LDS");
  // more than half of the CU's LDS even for small tiles: a second workgroup on the CU would put a second MFMA stream on every SIMD
  constexpr int LDS_BYTES = TILE_BYTES > 84 * 1024 ? TILE_BYTES : 84 * 1024;
  const long nbx = (a->N + BN - 1) / BN, nby = (a->M + BM - 1) / BM, nbz = a->batch > 1 ? a->batch : 1;
  const long ntiles = nbx * nby * nbz;
  SSR_REQUIRE(ntiles < (1L << 31), "ssrhip_gemm: grid too large (M=%d N=%d batch=%d)", a->M, a->N, a->batch);
  static bool attr_set = false;                                   // > 64 KB of dynamic LDS needs the opt-in (once per kernel)
  if (!attr_set) {
    SSR_HIP(hipFuncSetAttribute((const void*)gemm_persist_kernel<MT, NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    SSR_HIP(hipFuncSetAttribute((const void*)gemm_persist_kernel<MT, NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_set = true;
  }
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    SSR_HIP(hipGetDevice(&dev));
    SSR_HIP(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  // one workgroup per CU at most; a multiple of 8 so that every XCD gets the same number
  long per_xcd = (ntiles + 7) / 8;
  if (per_xcd > n_cu / 8) per_xcd = n_cu / 8;
  const dim3 grid((unsigned)(8 * per_xcd)), block(256);
  if (a->act_in == SSRHIP_ACT_ELU) hipLaunchKernelGGL((gemm_persist_kernel<MT, NT, true>), grid, block, LDS_BYTES, s, *a, (int)nbx, (int)nby, (int)ntiles);
  else hipLaunchKernelGGL((gemm_persist_kernel<MT, NT, false>), grid, block, LDS_BYTES, s, *a, (int)nbx, (int)nby, (int)ntiles);
  SSR_LAUNCH_CHECK();
  return 0;
}


__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((float)(h & 0xFFFF) / 32768.0f - 1.0f) * scale;
  }
}
__global__ void maxdiff_kernel(const float* a, const float* b, size_t n, float* out) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(a[i] - b[i]));
  atomicMax(reinterpret_cast<int*>(out), __float_as_int(m));      // non-negative floats order like their bit patterns
}
}  // namespace

int main() {
  struct Shape { const char* name; int M, N, K, batch, act_in; } shapes[] = {
      {"square 4096^3", 4096, 4096, 4096, 1, 0},       {"prefill qkv 598x6144x2048", 598, 6144, 2048, 1, 0},
      {"lstm-in 1500x4096x1024 x32", 1500, 4096, 1024, 32, 0}, {"down2 60000x256x1024 x32", 60000, 256, 1024, 32, 1},
      {"down1 240000x128x256 x32", 240000, 128, 256, 32, 1},   {"down4 1500x1024x8192 x32", 1500, 1024, 8192, 32, 1},
      {"ragged 3000x2056x1000", 3000, 2056, 1000, 1, 1},
  };
  const size_t cap = (size_t)32 * 240000 * 256;
  float *A, *W, *C0, *C1, *bias, *md;
  CK(hipMalloc(&A, cap * 4)); CK(hipMalloc(&W, (size_t)8192 * 8192 * 4)); CK(hipMalloc(&C0, cap * 4)); CK(hipMalloc(&C1, cap * 4));
  CK(hipMalloc(&bias, 8192 * 4)); CK(hipMalloc(&md, 4));
  hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, A, cap, 1u, 0.5f);
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, W, (size_t)8192 * 8192, 2u, 0.02f);
  hipLaunchKernelGGL(fill_kernel, dim3(8), dim3(256), 0, 0, bias, (size_t)8192, 3u, 0.1f);
  CK(hipDeviceSynchronize());
  hipStream_t s; CK(hipStreamCreate(&s));
  for (auto& sh : shapes) {
    ssrhip_gemm_args a; memset(&a, 0, sizeof(a));
    a.A = A; a.W = W; a.bias = bias; a.M = sh.M; a.N = sh.N; a.K = sh.K; a.lda = sh.K; a.ldc = sh.N; a.act_in = sh.act_in ? SSRHIP_ACT_ELU : 0;
    a.batch = sh.batch; a.strideA = (int64_t)sh.M * sh.K; a.strideC = (int64_t)sh.M * sh.N;
    double tf[2];
    for (int which = 0; which < 2; ++which) {
      a.C = which ? C1 : C0;
      auto run = [&]() { return which ? launch_persist<2, 2>(&a, s) : ssrhip_gemm(&a, s); };
      if (run()) { printf("launch failed\n"); return 1; }
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float wms = 0.f;
      for (int round = 0; round < 50 && wms < 40.f; ++round) {     // warm clocks
        CK(hipEventRecord(e0, s)); for (int i = 0; i < 4; ++i) run(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float m; CK(hipEventElapsedTime(&m, e0, e1)); wms += m;
      }
      CK(hipEventRecord(e0, s)); for (int i = 0; i < 10; ++i) run(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      tf[which] = 2.0 * sh.M * sh.N * sh.K * sh.batch * 10 / (ms * 1e-3) / 1e12;
    }
    CK(hipMemsetAsync(md, 0, 4, s));
    hipLaunchKernelGGL(maxdiff_kernel, dim3(4096), dim3(256), 0, s, C0, C1, (size_t)sh.M * sh.N * sh.batch, md);
    float h; CK(hipMemcpyAsync(&h, md, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
    printf("%-30s library %6.1f TFLOP/s   persistent 1-wave/SIMD %6.1f TFLOP/s   max |diff| %.3g\n", sh.name, tf[0], tf[1], h);
  }
  return 0;
}
