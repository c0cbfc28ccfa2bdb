// gemm_split.hip — fp32 GEMM on the bf16 matrix cores with EXACT operand splitting (round 3).
//
//   C[M][N] = epi( A[M][K] . W[N][K]^T + bias[N] )          same contract and epilogues as gemm.hip's ssrhip_gemm
//
// Why: v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (157 TFLOP/s peak; gemm.hip reaches 115-121 on 4096^3 and a
// persistent one-wave-per-SIMD kernel does not beat it: DESIGN §4b), v_mfma_f32_32x32x16_bf16 at 16x that. Every fp32 value is
// the exact sum of three bf16 values (a = a0 + a1 + a2: a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1); the
// subtractions are exact in fp32, 3 x 8 significand bits cover the 24), and a bf16 x bf16 product is exact in the fp32
// accumulator, so a . w = sum over the nine cross terms a_i w_j EXACTLY (magnitudes below ~1e-33, whose third piece would be an fp32
// subnormal, lose that piece: subnormals are flushed by the conversion — no activation or weight lives there). The kernel accumulates the six largest
//   a0w0 + a0w1 + a1w0 + a1w1 + a0w2 + a2w0          (smallest first inside each 16-wide k block)
// and drops a1w2, a2w1 (each <= 2^-24 |a||w|, the size of ONE fp32 rounding of the product) and a2w2 (2^-32). What remains is the
// rounding of the fp32 accumulations, as in the fp32 FMA chain. Measured against an fp64 reference (tools/gemm_split_lab, K = 4096,
// operands with 24 random mantissa bits, error relative to sum |a w|): this kernel max 5.2e-7 / mean 1.5e-8, the exact fp32 MFMA
// chain of gemm.hip max 7.0e-7 / mean 1.8e-8 — the split form is no less accurate than the fp32 chain it replaces (its products are
// exact, the chain rounds every one). It is NOT bit-identical to it: callers that need the k-ordered fp32 FMA chain (the LM
// prefill, whose greedy tokens are compared bit for bit with the reference) keep gemm.hip; the codec, whose parity bar is a
// waveform tolerance (2e-4) and RVQ codes up to fp32 near-ties, takes this kernel when the caller supplies the weight planes.
//
// Cost model per 16 k-values of a 32 x 32 block: 6 x 32 = 192 matrix-pipe cycles against 8 x 64 = 512 for the fp32 chain. The
// operands are split while they are staged: W once per checkpoint (`ssrhip_split_weights`: three bf16 planes [3][N][K]), A on the
// fly between the global load and the LDS store (v_cvt_pk_bf16_f32 = round to nearest even, two elements per instruction;
// bf16 -> fp32 is a shift). Block tile 128 x 128 x 32, 4 waves (2 x 2) of 64 x 64 (2 x 2 accumulators of 32 x 32), two workgroups
// per CU. LDS: per piece rows of 32 bf16 (64 B) at a pitch of 80 B — a lane's ds_read_b128 (8 consecutive k of one row = its
// whole A / B operand of one MFMA) then hits 16 distinct 16-byte slots per 16-lane group.
// Measured (tools/gemm_split_lab, one MI355X, warm clocks, fp32-equivalent TFLOP/s, exact kernel -> this one): 4096^3 121 -> 187,
// LSTM input GEMM 117 -> 168, the codec's batched strided views 91..107 -> 99..141 (the ELU-on-load layers gain least: the
// staging VALU work, not the matrix pipe, is what they wait for).
#include <stdlib.h>
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bfx2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float act_fn(float v, int act) {
  if (act == SSRHIP_ACT_RELU) return fmaxf(v, 0.f);
  if (act == SSRHIP_ACT_GELU_ERF) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  return v;
}

// exact three-way split of 4 consecutive k-values: piece p of element e in out[p][e]
__device__ __forceinline__ void split4(const float4 v, bf16x4 (&out)[3]) {
  f32x2 r[2] = {{v.x, v.y}, {v.z, v.w}};
#pragma unroll
  for (int p = 0; p < 3; ++p) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bfx2 b = __builtin_convertvector(r[h], bfx2);                   // v_cvt_pk_bf16_f32 (RNE)
      const unsigned bits = __builtin_bit_cast(unsigned, b);
      out[p][2 * h] = (short)(bits & 0xFFFFu);
      out[p][2 * h + 1] = (short)(bits >> 16);
      if (p < 2) {
        const f32x2 back = {__builtin_bit_cast(float, bits << 16), __builtin_bit_cast(float, bits & 0xFFFF0000u)};
        r[h] = r[h] - back;                                                 // exact: the residual has fewer significant bits than fp32 holds
      }
    }
  }
}

__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ W, short* __restrict__ out, size_t n_elems) {
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n_elems; i += (size_t)gridDim.x * blockDim.x * 4) {
    bf16x4 p[3];
    split4(ld4(W + i), p);
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(out + (size_t)q * n_elems + i) = p[q];
  }
}

constexpr int BN = 128, BK = 32, PITCH = 40;                                // pitch in bf16 elements (80 B)

// BM = 128: 2 x 2 waves of 64 x 64 (2 x 2 accumulators); BM = 64: 1 x 4 waves of 64 x 32 (2 x 1) for grids that 128-row tiles would not
// fill. Per output element both do the same arithmetic (k blocks of 16 in order, the six products in the same order), so a result does
// not depend on which tile shape — i.e. on which batch size — computed it.
template <int BM, bool ELU>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const ssrhip_gemm_args a0) {
  constexpr int MT = 2, NT = BM == 128 ? 2 : 1;
  constexpr int LA = BM / 32;
  __shared__ __attribute__((aligned(16))) short As[3][BM * PITCH];
  __shared__ __attribute__((aligned(16))) short Ws[3][BN * PITCH];
  ssrhip_gemm_args a = a0;
  {   // batched problems: grid.z
    const size_t z = blockIdx.z;
    a.A += z * (size_t)a.strideA;
    a.C += z * (size_t)a.strideC;
    if (a.R) a.R += z * (size_t)a.strideR;
    if (a.rclass) a.rclass += z * (size_t)a.rclass_stride;
  }
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = BM == 128 ? wave >> 1 : 0, wn = BM == 128 ? wave & 1 : wave;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int lr = t >> 3, lc = (t & 7) * 4;                                  // A loader: 8 threads per row (32 k), 32 rows per pass
  const int lw = t >> 2, cw = (t & 3) * 8;                                  // W loader: 4 threads per row (8 k each), 64 rows per pass
  const int M = a.M, N = a.N, K = a.K;
  const short* Wp = reinterpret_cast<const short*>(a.W_split);
  const size_t plane = (size_t)N * K;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[LA];
  bf16x8 rwp[3][2];
  auto gload = [&](int k0) {
    const bool kin = (k0 + lc) < K;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int m = m0 + lr + 32 * i;
      ra[i] = (kin && m < M) ? ld4(a.A + (size_t)m * a.lda + k0 + lc) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (ELU) { ra[i].x = elu1(ra[i].x); ra[i].y = elu1(ra[i].y); ra[i].z = elu1(ra[i].z); ra[i].w = elu1(ra[i].w); }
    }
    const bool kinw = (k0 + cw) < K;                                        // K % 8 == 0 (checked by the host)
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n = n0 + lw + 64 * i;
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        rwp[q][i] = (kinw && n < N) ? *reinterpret_cast<const bf16x8*>(Wp + (size_t)q * plane + (size_t)n * K + k0 + cw) : z;
      }
  };
  auto lds_store = [&]() {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      bf16x4 p[3];
      split4(ra[i], p);
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(&As[q][(lr + 32 * i) * PITCH + lc]) = p[q];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<bf16x8*>(&Ws[q][(lw + 64 * i) * PITCH + cw]) = rwp[q][i];
  };
  auto mma_tile = [&]() {
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      bf16x8 fa[3][MT], fb[3][NT];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[q][i] = *reinterpret_cast<const bf16x8*>(&As[q][((wm * MT + i) * 32 + li) * PITCH + kk + lh * 8]);
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[q][j] = *reinterpret_cast<const bf16x8*>(&Ws[q][((wn * NT + j) * 32 + li) * PITCH + kk + lh * 8]);
      }
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // a2w0, a0w2, a1w1, a1w0, a0w1, a0w0
#pragma unroll
      for (int pq = 0; pq < 6; ++pq)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pq]][i], fb[PB[pq]][j], acc[i][j], 0, 0, 0);
    }
  };
  gload(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();                                                       // previous tile fully consumed
    lds_store();
    __syncthreads();
    if (k0 + BK < K) gload(k0 + BK);                                        // prefetch next tile under the MFMAs
    mma_tile();
  }
  // epilogue: C/D layout of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) — identical to gemm.hip's
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + (wn * NT + j) * 32 + li;
    if (n >= N) continue;
    const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M) {
          if (a.tm_c > 0) {
            const long u = ((long)m * N + n) / a.tm_c;
            if (u < a.tm_lo || u >= a.tm_hi) continue;
          }
          float v = act_fn(acc[mt][j][r] + bias, a.act);
          float* c = a.C + (size_t)m * a.ldc + n;
          if (a.residual) v += *c;
          if (a.R) v += a.R[(size_t)m * a.ldr + n];
          if (a.rbias) v += a.rbias[(size_t)a.rclass[m / a.rrep] * N + n];
          if (a.act_out == SSRHIP_ACT_ELU) v = elu1(v);          // the consumer's ELU-on-load, done once here
          *c = v;
        }
      }
    }
  }
}

// ---- the 128-row tile, second generation (round 3, after the counters in profiles/r03_split_gemm_pmc.md): same tile, same arithmetic
// per output element (bit-identical results), restructured for latency hiding:
//   * 8 waves per workgroup (2 x 4, wave tile 64 x 32 = 2 x 1 accumulators), <= 128 VGPRs: two workgroups per CU = FOUR waves per SIMD
//     (the 4-wave kernel above keeps two per SIMD waiting on s_waitcnt half of the time: SQ_WAIT_INST_ANY 59 %, MfmaUtil 45 %);
//   * W goes global -> LDS by the DMA path (buffer_load_dwordx4 ... lds: no staging registers, no ds_write), ONE TILE AHEAD into a second
//     W stage, so its latency runs under the previous tile's MFMAs; A (which has to pass through the VALU for the split) keeps one stage;
//   * XOR-swizzled 64-byte rows instead of padded 80-byte ones: row r keeps its 16-byte chunk c in slot c ^ ((r >> 2) & 3). A lane group of
//     ds_read_b128 (rows {0-3, 12-15, 20-27} or {4-11, 16-19, 28-31} of a 32-row block, one chunk index) touches 16 different slots of
//     the 256-byte bank row; two rows per 16-lane group of ds_write_b64 = all 32 banks once (the padded rows were 2-way on every store:
//     SQ_LDS_BANK_CONFLICT = 29 % of the LDS cycles); and the DMA's lane-linear KiB is exactly 16 rows. 73,728 B per workgroup (dynamic);
//   * branch-free loads: buffer descriptors that end with the tile's last valid row (rows past M / N read as zero, the DMA writes zeros),
//     a k past K sends the lane's offset out of range (a plain load + select is turned back into a branch around the load, and the branch
//     makes every wait a vmcnt(0)); the loads are pinned behind the second barrier (sched_barrier: the scheduler sinks them to the end of
//     the MFMA block otherwise and the next iteration waits for their whole latency).
// Measured (tools/gemm_split_lab db, one MI355X, fp32-equivalent TFLOP/s, 4-wave kernel -> this one): 4096^3 186 -> 207, LSTM input GEMM
// 167 -> 196, the codec's strided views 99..137 -> 124..158.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int DMA_PLANE = 128 * 64;                                   // bytes: one bf16 plane of a 128 x 32 tile
constexpr int dma_lds(int bm) { return 3 * bm * 64 + 6 * DMA_PLANE; }  // A 3 planes of BM rows + W 2 stages x 3 planes

// BM = 128: waves 2 x 4 of 64 x 32; BM = 64 (grids that 128-row tiles would not fill, half-empty tiles): waves 2 x 4 of 32 x 32. Same
// arithmetic per output element in both (and in the 4-wave kernels above).
// Cost attribution hooks for tools/gemm_dma_lab.hip (KO = 0 in the library: every `if constexpr` below folds away): knock one component out and
// time the rest, or (GD_PROF) leave 100 MHz time stamps of wave 0's first k-steps in LDS and dump them for a sample of the workgroups.
enum { GD_KO_MFMA = 1, GD_KO_ALOAD = 2, GD_KO_ASTORE = 4, GD_KO_DMA = 8, GD_KO_STORE = 16, GD_PROF = 32 };
constexpr int GD_NSTAMP = 128, GD_STEPS = 40;              // stamps 3 s + {0, 1, 2} of step s < GD_STEPS: loop top, tile ready, MFMA block issued; 124 / 125: entry / end
__device__ unsigned* g_gd_prof = nullptr;                  // [sampled workgroup][GD_NSTAMP]

// TMF (default since round 5; SSRHIP_EPILOGUE_TM=0 = the general loop. Written blind at the end of round 4; round 5's first GPU call: GEMM tests
// and all codec fixtures green with it, config-5 decode 383.6 -> 373.6 ms, profiles/r05_microbench/codec256_ab.log): whole tiles of a TRANSPOSED convolution's
// GEMM — time mask tm_c > 0 with N % tm_c == 0 and tm_c % 4 == 0, nothing added behind the activation (the launcher checks) — take the
// 16-byte epilogue with the mask as a per-row predicate: element (m, n) belongs to time row (m N + n) / tm_c = m (N / tm_c) + n / tm_c, the
// second term one 32-bit division per LANE. The general loop those launches take today pays a 64-bit division per OUTPUT (32 per lane).
//
// Round 6 — which tile a workgroup takes (flags bit 1, default on; SSRHIP_GEMM_XCD=0 = the plain blockIdx order): the hardware deals
// workgroups to the 8 XCDs round-robin in launch order (x fastest), so the N-tiles that share an A row-tile — neighbours in x — ran on
// DIFFERENT XCDs and every one of their L2s pulled its own copy of the tile across the fabric: rocprofv3 FETCH_SIZE 134.6 GB for a 31.5 GB
// operand at `60001 x 512 x 512 x 256` (4 N-tiles), 197 GB for 12.6 GB at `12001 x 1280 x 1024 x 256` (10 N-tiles)
// (profiles/r05_codec_b256_pmc_fetch_size.md). Now workgroup w (linear launch index) runs on XCD w % 8 as the
// (w / 8)-th workgroup of that XCD and takes logical tile start(w % 8) + w / 8, where XCD c owns the contiguous range of logical tiles
// [start(c), start(c + 1)) and logical tiles are numbered x fastest: the N-tiles of one row-tile run on ONE XCD, back to back, and
// its L2 fetches the A tile once. A bijection of the grid onto itself (tests/test_split_gemm_addressing_model.py), results unchanged.
// Second measurement of the round (profiles/r06_microbench/codec256_fetch_xcd_*.md): with ALL N-tiles of a row-tile on one XCD the A
// traffic fell (60001 x 512 x 512: 91 -> 40 GB for a 31.5 GB operand) but launches whose W planes do not fit an XCD's 4 MB L2 got WORSE
// (12001 x 1280 x 1024 stayed at 161-197 GB, two K >= 1024 shapes tripled): in blockIdx order an XCD only ever touched nx / gcd(nx, 8) of
// the N-tiles, now it cycles through all of them and re-fetches every 0.4-0.8 MB W tile per workgroup. So the logical order is N-tile
// GROUPS: GN = flags >> 8 consecutive N-tiles (host: as many as keep GN W tiles within ~2.5 MB) are swept over every row-tile and item
// before the next group starts; inside a group x is fastest. One fabric read of an A tile per group, W tiles of the group resident.
__device__ __forceinline__ void xcd_tile(const int flags, int& bx, int& by, int& bz) {
  bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
  if (!(flags & 2)) return;
  const unsigned nx = gridDim.x, ny = gridDim.y, nyz = ny * gridDim.z, total = nx * nyz;
  const unsigned w = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
  const unsigned c = w & 7u, j = w >> 3, q = total >> 3, r = total & 7u;
  const unsigned L = c * q + (c < r ? c : r) + j;                    // XCD c owns q + (c < r) logical tiles, a contiguous range
  unsigned gn = (unsigned)flags >> 8;
  if (gn == 0 || gn > nx) gn = nx;
  const unsigned per_group = gn * nyz, g = L / per_group;            // full groups first; the last one may be narrower
  const unsigned idx = L - g * per_group;
  const unsigned width = (g * gn + gn <= nx) ? gn : nx - g * gn;
  bx = (int)(g * gn + idx % width);
  const unsigned t2 = idx / width;
  by = (int)(t2 % ny);
  bz = (int)(t2 / ny);
}

template <int BM, bool ELU, int KO = 0, bool TMF = false>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_split_dma_kernel(const ssrhip_gemm_args a0, const int flags) {
  const int wide = flags & 1;
  int bx, by, bz;
  xcd_tile(flags, bx, by, bz);
  constexpr int MT = BM / 64, LA = BM / 64, APL = BM * 64;           // accumulators per wave, A loader passes, bytes per A plane
  extern __shared__ __attribute__((aligned(1024))) char ldsb[];
  char* const As = ldsb;                                             // [3][BM][64 B]
  char* const Wsb = ldsb + 3 * APL;                                  // [2][3][128][64 B]
  unsigned* const stamps = reinterpret_cast<unsigned*>(ldsb + dma_lds(BM));   // GD_PROF only (the lab adds the bytes)
  auto stamp = [&](int i) {
    if constexpr ((KO & GD_PROF) != 0) {
      if (threadIdx.x == 0 && i < GD_NSTAMP) stamps[i] = (unsigned)wall_clock64();
    }
  };
  stamp(124);
  // lab only (GD_PROF): at every exit of the kernel — N a multiple of 128 in the lab, so no lane leaves early — wave 0 marks the end of
  // its epilogue and a sample of the workgroups copies the stamps out
  struct Dump {
    unsigned* st;
    int bx_, by_, bz_;
    __device__ ~Dump() {
      if constexpr ((KO & GD_PROF) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) st[125] = (unsigned)wall_clock64();
        __syncthreads();
        if (bx_ == 0 && by_ % 37 == 5 && bz_ % 8 == 3 && threadIdx.x < GD_NSTAMP)
          g_gd_prof[((size_t)(bz_ / 8) * ((gridDim.y + 31) / 37) + by_ / 37) * GD_NSTAMP + threadIdx.x] = st[threadIdx.x];
      }
    }
  } dump{stamps, bx, by, bz};
  ssrhip_gemm_args a = a0;
  {   // batched problems: grid.z
    const size_t z = (size_t)bz;
    a.A += z * (size_t)a.strideA;
    a.C += z * (size_t)a.strideC;
    if (a.R) a.R += z * (size_t)a.strideR;
    if (a.rclass) a.rclass += z * (size_t)a.rclass_stride;
  }
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = (wave >> 2) & 1, wn = wave & 3;
  const int n0 = bx * BN, m0 = by * BM;
  const int lr = t >> 3, lc = (t & 7) * 4;                           // A loader: 8 threads per row (32 k), 64 rows per pass, 2 passes
  const int M = a.M, N = a.N, K = a.K;
  const short* Wp = reinterpret_cast<const short*>(a.W_split);
  const size_t plane = (size_t)N * K;

  f32x16 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  float4 ra[LA];
  const int rows_a = min(BM, M - m0), rows_w = min(128, N - n0);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A + (size_t)m0 * a.lda), 0,
                                                                        (int)(((size_t)(rows_a - 1) * a.lda + K) * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rsW[3];
#pragma unroll
  for (int q = 0; q < 3; ++q)
    rsW[q] = __builtin_amdgcn_make_buffer_rsrc(const_cast<short*>(Wp + (size_t)q * plane + (size_t)n0 * K), 0, (int)((size_t)rows_w * K * 2), 0x00020000);
  unsigned offA[LA];
#pragma unroll
  for (int i = 0; i < LA; ++i) offA[i] = (unsigned)(((size_t)(lr + 64 * i) * a.lda + lc) * 4);
  constexpr unsigned OOB = 0x80000000u;                              // >= every descriptor's extent (checked by the host): reads as zero
  // W by DMA: wave w brings rows 16w .. 16w+15 of a plane with one instruction; lane l lands in slot l of the wave's KiB = (row 16w + l/4,
  // slot l%4), which has to hold chunk (l%4) ^ ((row >> 2) & 3)
  const int wrow = 16 * wave + (lane >> 2), wchunk = (lane & 3) ^ ((lane >> 4) & 3);
  const unsigned offW = (unsigned)(((size_t)wrow * K + wchunk * 8) * 2);
  auto gload_a = [&](int k0) {
    const bool kin = (k0 + lc) < K;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      if constexpr ((KO & GD_KO_ALOAD) != 0) ra[i] = make_float4(a.M * 1e-9f, 0.25f, -0.5f, 0.125f);
      else ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, kin ? offA[i] + (unsigned)k0 * 4 : OOB, 0, 0));
    }
  };
  auto dma_w = [&](int k0, int stage) {
    if constexpr ((KO & GD_KO_DMA) != 0) return;
    const bool kin = (k0 + wchunk * 8) < K;                          // K % 8 == 0 (checked by the host)
    const unsigned off = kin ? offW + (unsigned)k0 * 2 : OOB;
#pragma unroll
    for (int q = 0; q < 3; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW[q], (lds_ptr_t)(Wsb + (stage * 3 + q) * DMA_PLANE + wave * 1024), 16, off, 0, 0, 0);
  };
  const int aswz = (((lc >> 3) ^ ((lr >> 2) & 3)) << 4) + ((lc >> 2) & 1) * 8;   // rows lr and lr + 64 share (row >> 2) & 3
  auto store_a = [&]() {
    if constexpr ((KO & GD_KO_ASTORE) != 0) return;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      float4 v = ra[i];
      if (ELU) { v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w); }
      bf16x4 p[3];
      split4(v, p);
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x4*>(As + q * APL + (lr + 64 * i) * 64 + aswz) = p[q];
    }
  };
  const int fsw = (li >> 2) & 3;
  auto mma_tile = [&](int stage) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // a2w0, a0w2, a1w1, a1w0, a0w1, a0w0
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      const int coff = (((kk >> 3) + lh) ^ fsw) << 4;
      bf16x8 fa[3][MT], fb[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[q][i] = *reinterpret_cast<const bf16x8*>(As + q * APL + ((wm * MT + i) * 32 + li) * 64 + coff);
        fb[q] = *reinterpret_cast<const bf16x8*>(Wsb + (stage * 3 + q) * DMA_PLANE + (wn * 32 + li) * 64 + coff);
      }
#pragma unroll
      for (int pq = 0; pq < 6; ++pq)
#pragma unroll
        for (int i = 0; i < MT; ++i)
          if constexpr ((KO & GD_KO_MFMA) == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[pq]][i], fb[PB[pq]], acc[i], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  gload_a(0);
  dma_w(0, 0);
  int stage = 0;
  for (int k0 = 0; k0 < K; k0 += BK) {
    if constexpr ((KO & GD_PROF) != 0) { if (k0 < GD_STEPS * BK) stamp(3 * (k0 / BK)); }
    __syncthreads();                                                 // the previous tile is consumed (the A stage, W stage ^ 1)
    store_a();
    // W(k0) was written into LDS by OTHER waves' DMA: every wave drains its own DMA before the barrier publishes the tile. hipcc happens to
    // place a vmcnt(0) at the loop head today (it merges the prologue's issue order with its conservative LDS-DMA alias rule); a compiler that
    // tracks the DMA more precisely could legally move that wait behind the barrier — stale W rows, no error. Explicit, and
    // guarded in tests/test_isa_guards.py. Free: the wait is already there.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                 // A(k0) stored, W(k0) landed
    if constexpr ((KO & GD_PROF) != 0) { if (k0 < GD_STEPS * BK) stamp(3 * (k0 / BK) + 1); }
    gload_a(k0 + BK);                                                // past K: zeros, never used
    dma_w(k0 + BK, stage ^ 1);
    __builtin_amdgcn_sched_barrier(0);                               // keep the loads HERE
    mma_tile(stage);
    if constexpr ((KO & GD_PROF) != 0) { if (k0 < GD_STEPS * BK) stamp(3 * (k0 / BK) + 2); }
    stage ^= 1;
  }
  stamp(123);
  // epilogue: C/D layout of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  if (wide) {                                                        // the 16-byte epilogue below turns blocks through the tiles' LDS:
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the last step's W DMA (zeros past K) has landed,
    __syncthreads();                                                 // and every wave has read its last tile
  }
  const int n = n0 + wn * 32 + li;
  if (n >= N) return;
  const float bias = a.bias ? a.bias[n] : 0.f;
  if constexpr (TMF) {
    if (m0 + BM <= M && n0 + wn * 32 + 32 <= N) {                    // uniform per wave; ragged tiles fall through to the general loop
      float* const tr = reinterpret_cast<float*>(ldsb) + wave * 1024;
      const int trow = lane >> 3, tc4 = (lane & 7) * 4, nu = n0 + wn * 32;
      const unsigned cld = (unsigned)a.ldc;
      const int ratio = N / a.tm_c, nq = (nu + tc4) / a.tm_c;        // the four columns of a lane share n / tm_c (tm_c % 4 == 0)
      const bool elu_out = a.act_out == SSRHIP_ACT_ELU;
      const int act = a.act;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int mu = m0 + (wm * MT + mt) * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = act_fn(acc[mt][r] + bias, act);
        __builtin_amdgcn_wave_barrier();
        float4 o[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) o[p] = *reinterpret_cast<const float4*>(tr + (trow + 8 * p) * 32 + tc4);
        __builtin_amdgcn_wave_barrier();
        if (elu_out) {
#pragma unroll
          for (int p = 0; p < 4; ++p) o[p] = make_float4(elu1(o[p].x), elu1(o[p].y), elu1(o[p].z), elu1(o[p].w));
        }
        float* dst = a.C + ((size_t)mu * cld + nu);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const long u = (long)(mu + trow + 8 * p) * ratio + nq;
          if (u >= a.tm_lo && u < a.tm_hi) *reinterpret_cast<float4*>(dst + (unsigned)(trow + 8 * p) * cld + (unsigned)tc4) = o[p];
        }
      }
      return;
    }
  }
  // Whole tiles without a time mask or class bias, with at most ONE operand added behind the activation (C itself or R: a residual block's
  // second convolution): its 16 values per accumulator block are requested together and the 16 stores follow together. The general loop
  // below (load -> add -> store per element under a row predicate) compiles to one dependent HBM round trip per output — 32 per lane, the
  // reason `60000 x 256 x 128 + R` ran at 56 TFLOP/s next to 185 for its neighbours (profiles/r04_codec_b256_kernel_trace_summary.md;
  // found with tools/resblock_lab.hip on the residual-block kernel, which had the same epilogue). Same arithmetic per element.
  if (m0 + BM <= M && a.tm_c <= 0 && !a.rbias && !(a.residual && a.R) && a.ldc < (1 << 24) && a.ldr < (1 << 24)) {      // uniform
    const float* ap = a.residual ? a.C : a.R;
    const unsigned ald = (unsigned)(a.residual ? a.ldc : a.ldr), cld = (unsigned)a.ldc;
    const bool elu_out = a.act_out == SSRHIP_ACT_ELU;
    const int nu = n0 + wn * 32;
    // 16-byte form (default; SSRHIP_EPILOGUE_WIDE=0 = the dword form below): an accumulator block (lane = one column, 16 rows) goes through
    // a wave-private 4 KB of LDS and comes back as rows — lane l holds columns 4 (l % 8) .. + 3 of rows l / 8 + 8 p — so that the added
    // operand is read and C is written as dwordx4, 128-byte row pieces per 8 lanes: 4 + 4 vector-memory instructions per block instead of
    // 16 + 16 (the stores of a dword epilogue are issue-bound, MI355X_MICROARCH.md `epilogue store tail`). Same arithmetic per element.
    if (wide && n0 + wn * 32 + 32 <= N && ((ald | cld) & 3u) == 0 && (((uintptr_t)a.C | (uintptr_t)ap) & 15) == 0) {
      float* const tr = reinterpret_cast<float*>(ldsb) + wave * 1024;
      const int trow = lane >> 3, tc4 = (lane & 7) * 4;
      auto wblocks = [&](const int act) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int mu = m0 + (wm * MT + mt) * 32;
          float4 av[4];
          if (ap) {
            const float* src = ap + ((size_t)mu * ald + nu);                                                             // wave-uniform base
#pragma unroll
            for (int p = 0; p < 4; ++p) av[p] = *reinterpret_cast<const float4*>(src + (unsigned)(trow + 8 * p) * ald + (unsigned)tc4);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = act_fn(acc[mt][r] + bias, act);
          __builtin_amdgcn_wave_barrier();                           // a wave's own LDS accesses complete in order; this pins the compiler's order
          float4 o[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            o[p] = *reinterpret_cast<const float4*>(tr + (trow + 8 * p) * 32 + tc4);
            if (ap) o[p] = make_float4(o[p].x + av[p].x, o[p].y + av[p].y, o[p].z + av[p].z, o[p].w + av[p].w);
          }
          __builtin_amdgcn_wave_barrier();
          if (elu_out) {
#pragma unroll
            for (int p = 0; p < 4; ++p) o[p] = make_float4(elu1(o[p].x), elu1(o[p].y), elu1(o[p].z), elu1(o[p].w));
          }
          float* dst = a.C + ((size_t)mu * cld + nu);
#pragma unroll
          for (int p = 0; p < 4; ++p)
            if ((KO & GD_KO_STORE) == 0 || o[p].x == 1.2345e-33f) *reinterpret_cast<float4*>(dst + (unsigned)(trow + 8 * p) * cld + (unsigned)tc4) = o[p];
        }
      };
      if (a.act == SSRHIP_ACT_NONE) wblocks(SSRHIP_ACT_NONE);
      else wblocks(a.act);
      return;
    }
    auto blocks = [&](const int act) __attribute__((always_inline)) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int mu = m0 + (wm * MT + mt) * 32;
        float av[16];
        if (ap) {
          const float* src = ap + ((size_t)mu * ald + nu);                                                               // wave-uniform base
#pragma unroll
          for (int r = 0; r < 16; ++r) av[r] = src[(unsigned)((r & 3) + 8 * (r >> 2) + 4 * lh) * ald + (unsigned)li];
        }
        float* dst = a.C + ((size_t)mu * cld + nu);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = act_fn(acc[mt][r] + bias, act);
          if (ap) v += av[r];
          if (elu_out) v = elu1(v);
          dst[(unsigned)((r & 3) + 8 * (r >> 2) + 4 * lh) * cld + (unsigned)li] = v;
        }
      }
    };
    if (a.act == SSRHIP_ACT_NONE) blocks(SSRHIP_ACT_NONE);             // the codec's case: no per-element dispatch
    else blocks(a.act);
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < M) {
        if (a.tm_c > 0) {
          const long u = ((long)m * N + n) / a.tm_c;
          if (u < a.tm_lo || u >= a.tm_hi) continue;
        }
        float v = act_fn(acc[mt][r] + bias, a.act);
        float* c = a.C + (size_t)m * a.ldc + n;
        if (a.residual) v += *c;
        if (a.R) v += a.R[(size_t)m * a.ldr + n];
        if (a.rbias) v += a.rbias[(size_t)a.rclass[m / a.rrep] * N + n];
        if (a.act_out == SSRHIP_ACT_ELU) v = elu1(v);
        *c = v;
      }
    }
  }
}

}  // namespace

// true when the split kernel applies to this call (decided by ssrhip_gemm). It depends on the matrix (N, K) and on what the caller
// supplied — not on M or the batch as long as they fit the launch grid (M <= 65535 x 64 rows = 4.19 M rows per item, batch <= 65535:
// 87 s of 48 kHz audio per item at the codec's finest layer; beyond that the call takes the fp32 chain, which is NOT bit-identical):
// an item's result must not change with the batch it is computed in (codec batch lanes, decode_ragged == batch-1 decode).
// gfx950 only, like the whole library (the Makefile's ARCH): the DMA kernels use `buffer_load ... lds` with 16-byte pieces and 73,728 B
// of LDS per workgroup at two workgroups per CU; ssrhip_gemm_split_launch raises the dynamic-LDS limit per device and returns an error
// (never a silent fallback) when the runtime refuses it.
bool ssrhip_gemm_split_eligible(const ssrhip_gemm_args* a) {
  if (!a->W_split || a->N <= 64 || a->K % 8 != 0 || a->lda % 4 != 0) return false;
  static const int off = getenv("SSRHIP_GEMM_SPLIT") && getenv("SSRHIP_GEMM_SPLIT")[0] == '0';   // A/B knob: always the exact fp32 chain
  return !off && (a->M + 63) / 64 <= 65535 && a->batch <= 65535;
}

int ssrhip_gemm_split_launch(const ssrhip_gemm_args* a, hipStream_t s) {
  const long nb = a->batch > 1 ? a->batch : 1;
  const long tiles128 = (long)((a->N + BN - 1) / BN) * ((a->M + 127) / 128) * nb;
  const bool elu = a->act_in == SSRHIP_ACT_ELU;
  // 128-row tiles when there are enough of them for two workgroups on most CUs — unless they would be half empty: the LSTM's second
  // layer projects 64-step chunks (M = 64 per item: a 128-row tile wastes half of its MFMAs; seen in the codec trace: 96 launches of
  // 212 us at 32 clips)
  const int waste128 = (a->M + 127) / 128 * 128 - a->M, waste64 = (a->M + 63) / 64 * 64 - a->M;
  const bool half_empty = waste128 - waste64 >= 64 && 8 * (waste128 - waste64) >= a->M;
  // the DMA kernels address a tile through 32-bit buffer offsets: 128 rows of A (and of a W plane) have to stay below 2 GiB
  static const bool dma_off = getenv("SSRHIP_GEMM_SPLIT_DMA") && getenv("SSRHIP_GEMM_SPLIT_DMA")[0] == '0';   // A/B knob: the 4-wave kernels
  const bool dma = !dma_off && ((size_t)127 * a->lda + a->K) * 4 < 0x7FFFFFF0ull && (size_t)128 * a->K * 2 < 0x7FFFFFF0ull;
  static const int wide1 = !(getenv("SSRHIP_EPILOGUE_WIDE") && getenv("SSRHIP_EPILOGUE_WIDE")[0] == '0');     // A/B knob: 0 = dword epilogue
  // XCD-aware tile order of the DMA kernels (read at every launch: tests flip it inside one process); total tiles must fit 32 bits
  const char* xe = getenv("SSRHIP_GEMM_XCD");
  // N-tile group width of the XCD order: W tiles of one group (3 bf16 planes of 128 rows x K) within ~2.5 MB of an XCD's 4 MB L2
  const long wtile = 128L * a->K * 6;
  long gn = wtile > 0 ? (2560L * 1024) / wtile : 1;
  if (gn < 1) gn = 1;
  if (gn > 255) gn = 255;
  if (xe && xe[0] >= '1' && xe[0] <= '9' && xe[1] == ':') gn = atoi(xe + 2) > 0 ? atoi(xe + 2) : gn;   // "1:<GN>": force the group width (lab)
  const int flags0 = wide1 | ((xe && xe[0] == '0') ? 0 : 2) | (int)(gn << 8);
  // the transposed convolutions' time mask as a row predicate of the 16-byte epilogue (SSRHIP_EPILOGUE_TM=0: the general per-element loop)
  static const bool tm_knob = !(getenv("SSRHIP_EPILOGUE_TM") && getenv("SSRHIP_EPILOGUE_TM")[0] == '0');
  const bool tmf = tm_knob && wide1 && a->tm_c > 0 && a->N % a->tm_c == 0 && a->tm_c % 4 == 0 && !a->R && !a->residual && !a->rbias && a->ldc % 4 == 0 &&
                   a->strideC % 4 == 0 && ((uintptr_t)a->C & 15) == 0 && (long)a->M * (a->N / a->tm_c) < 0x7FFFFFFFL;
  if (dma) {
    static ssr_once_per_device once;
    if (once.need()) {
      SSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_dma_kernel<128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, dma_lds(128)));
      SSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_dma_kernel<128, false>), hipFuncAttributeMaxDynamicSharedMemorySize, dma_lds(128)));
      SSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_dma_kernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, dma_lds(64)));
      SSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_dma_kernel<64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, dma_lds(64)));
      if (tm_knob) {
        SSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_dma_kernel<128, true, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, dma_lds(128)));
        SSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_dma_kernel<128, false, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, dma_lds(128)));
      }
    }
  }
  if (tiles128 >= 384 && !half_empty) {
    dim3 grid((a->N + BN - 1) / BN, (a->M + 127) / 128, (unsigned)nb);
    ssr_gemm_log(a, grid.x, grid.y, grid.z, 1);
    const int wide = ((long)grid.x * grid.y * grid.z > 0x7FFFFFFFL) ? (flags0 & 1) : flags0;      // the remap counts tiles in 32 bits
    if (dma && tmf) {
      if (elu) hipLaunchKernelGGL((gemm_split_dma_kernel<128, true, 0, true>), grid, dim3(512), dma_lds(128), s, *a, wide);
      else hipLaunchKernelGGL((gemm_split_dma_kernel<128, false, 0, true>), grid, dim3(512), dma_lds(128), s, *a, wide);
    } else if (dma) {
      if (elu) hipLaunchKernelGGL((gemm_split_dma_kernel<128, true>), grid, dim3(512), dma_lds(128), s, *a, wide);
      else hipLaunchKernelGGL((gemm_split_dma_kernel<128, false>), grid, dim3(512), dma_lds(128), s, *a, wide);
    } else if (elu) hipLaunchKernelGGL((gemm_split_kernel<128, true>), grid, dim3(256), 0, s, *a);
    else hipLaunchKernelGGL((gemm_split_kernel<128, false>), grid, dim3(256), 0, s, *a);
  } else {
    dim3 grid((a->N + BN - 1) / BN, (a->M + 63) / 64, (unsigned)nb);
    ssr_gemm_log(a, grid.x, grid.y, grid.z, 1);
    const int wide = ((long)grid.x * grid.y * grid.z > 0x7FFFFFFFL) ? (flags0 & 1) : flags0;
    if (dma) {
      if (elu) hipLaunchKernelGGL((gemm_split_dma_kernel<64, true>), grid, dim3(512), dma_lds(64), s, *a, wide);
      else hipLaunchKernelGGL((gemm_split_dma_kernel<64, false>), grid, dim3(512), dma_lds(64), s, *a, wide);
    } else if (elu) hipLaunchKernelGGL((gemm_split_kernel<64, true>), grid, dim3(256), 0, s, *a);
    else hipLaunchKernelGGL((gemm_split_kernel<64, false>), grid, dim3(256), 0, s, *a);
  }
  SSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int ssrhip_split_weights(const float* W, uint16_t* out, int64_t n_elems, ssrhip_stream_t stream) {
  SSR_REQUIRE(W && out && n_elems > 0 && n_elems % 4 == 0, "ssrhip_split_weights: bad argument (n_elems must be a positive multiple of 4)");
  long blocks = (n_elems / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, W, reinterpret_cast<short*>(out), (size_t)n_elems);
  SSR_LAUNCH_CHECK();
  return 0;
}
