// gemv.hip — weight-streaming fused GEMV for the AR decode step (B <= 4 rows), gfx950.
//
//   y[b][n] = epi( sum_k pro(x)[b][k] * W[n][k] + bias[n] )
//
// Roofline: HBM-bound. Every weight byte is read exactly once per step (2*B FLOP per 4 bytes), so the
// kernel is organised around keeping 16-byte non-temporal loads in flight from the first cycle:
//   * a wave owns whole weight rows (or one <=2048-float slice of them when K > 2048); its slice of
//     x (<= 8 float4 per lane per batch row) lives in VGPRs for the whole kernel;
//   * a row slice is 8 x global_load_dwordx4 per lane (1 KiB per wave-instruction, fully coalesced);
//     rows are software-pipelined two deep (ping-pong register sets): the loads of row i+1 are issued
//     before the FMAs of row i, and the FIRST row is requested before the prologue runs, so the
//     LayerNorm / split-KV-combine prologue overlaps HBM latency instead of preceding it;
//   * the grid is sized to be fully resident (<= 3 workgroups of 4 waves per CU) and rows are dealt
//     round-robin, so every wave streams a similar number of bytes and there is no second "wave" of
//     workgroups paying the prologue again;
//   * no LDS on the weight path (each weight element is used once: staging would be pure overhead);
//     LDS only shares the prologue result between the 4 waves and adds the K-slices of one row;
//   * wave reductions are DPP/permlane (no ds_bpermute).
// The paragraph above describes the generic row-per-wave kernel `gemv_kernel`. Since the second half of round 2 the shapes of the
// 830M step (K = 1024 * {1,2,4,8}, LayerNorm folded) run on `gemv_seg_kernel` further down — same fusion, a different split of the
// work (contiguous rows per 512-thread workgroup, (row, 1024-float segment) units, 4 loads in flight per lane); the generic kernel
// serves every other shape (the narrow test models, an unfolded LayerNorm) and `SSRHIP_GEMV_SEG=0`. (Round 3 removed the
// intermediate generation `gemv_fast_kernel` — the unconditional-load specialisation of the row-per-wave kernel for exactly the
// shapes the segment kernel took over.)
// Replaces F.linear (+LayerNorm / ReLU / GELU / residual) of the reference: see include/ssrhip.h.
#include <stdlib.h>
#include "common.h"

namespace {

struct GemvK {
  ssrhip_gemv_args a;
  int nslice;     // waves cooperating on one row (K split), 1|2|4
  int slice_len;  // floats per slice (multiple of 4)
  int nch;        // float4 chunks per lane per slice (<= 8)
  int groups_x;   // wave-groups along N (= gridDim.x * 4/nslice)
  int hd;         // head_dim (QKV epilogue / combine prologue)
  int seg_shift;  // segment kernel: log2(K / 1024)
  int rows_max;   // segment kernel: most rows a workgroup owns (sizes the LDS partials)
  long long* prof;          // -DSSR_GEMV_PROFILE builds only (ssrhip_debug_gemv_prof): 8 wall_clock64 stamps per workgroup, or NULL
  int rows_per, rows_rem;   // segment / front kernels: N / groups_x and N % groups_x (workgroup b owns rows_per + (b < rows_rem) rows)
};

// Measured and NOT kept at 1..4 rows (round 5): posting a wave's first weight requests only when its activations (x / attention partials)
// have arrived. The 16-row kernels gain from it (csrc/gemv_mfma.hip: 128 KB of x per CU); here 16 KB per workgroup is no obstacle and the
// later weight requests only cost: 0.8273 ms/step as is, 0.8392 with the wait in the x-operand launches, 0.8269 in the merge launch only,
// 0.8393 in both; a bare s_barrier between the two request phases 0.8052 -> 0.8105 (profiles/r05_microbench/decode_ab_xwait.log,
// decode_ab_xfirst_barrier.log). The experiment's `asm volatile("s_waitcnt vmcnt(0)" ::: "memory")` also taught a lesson by staying in the
// source behind a runtime flag: its memory clobber made every later uniform load "possibly clobbered", the kv_pos -> page-table chain of the
// QKV launch left the scalar path again (vector loads behind vmcnt(0)) and the step lost 2 % with the flag OFF — caught by the ISA guard
// tests/test_isa_guards.py::test_round5_decode_kernels_keep_their_request_order_in_the_isa. Removed.

long long* g_gemv_prof = nullptr;   // debug: ssrhip_debug_gemv_prof
constexpr int PRO_LN_REGS = 3;   // internal: LayerNorm with gamma/beta folded into W/bias, whole row per wave (no LDS)
constexpr int MAXCH = 8;
constexpr int MAX_IT = 8;   // max rows per wave-group when K is split (LDS partials)

constexpr int NJ = 4;   // float4 per thread per row in the LayerNorm prologue (K <= 4096)

// LayerNorm prologue, split so that the x loads are issued BEFORE the first weight row and consumed
// after it (loads return in order per wave: x first, then the weight row keeps flying during the math).
template <int B>
__device__ __forceinline__ void ln_issue(const GemvK& p, int g, float4 (&xv)[B][NJ]) {
  const int K = p.a.K, t = threadIdx.x;
#pragma unroll
  for (int b = 0; b < B; ++b) {
    const float* xb = p.a.x + (size_t)b * p.a.x_stride + (size_t)g * K;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int k = t * 4 + j * 1024;
      xv[b][j] = (k < K) ? ld4(xb + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

template <int B>
__device__ __forceinline__ void ln_finish(const GemvK& p, float4 (&xv)[B][NJ], float* xs, float* red) {
  // biased variance, eps inside the sqrt (F.layer_norm, models/modules/transformer.py:58-75);
  // two-pass (mean, then centered squares), values stay in registers between the passes.
  const int K = p.a.K, t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
  for (int b = 0; b < B; ++b) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) s += (xv[b][j].x + xv[b][j].y) + (xv[b][j].z + xv[b][j].w);   // zero beyond K
    s = wave_sum(s);
    if (lane == 0) red[b * 4 + wave] = s;
  }
  __syncthreads();
  float mean[B];
#pragma unroll
  for (int b = 0; b < B; ++b) mean[b] = ((red[b * 4 + 0] + red[b * 4 + 1]) + (red[b * 4 + 2] + red[b * 4 + 3])) / (float)K;
  __syncthreads();
#pragma unroll
  for (int b = 0; b < B; ++b) {
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (t * 4 + j * 1024 < K) {
        const float dx = xv[b][j].x - mean[b], dy = xv[b][j].y - mean[b], dz = xv[b][j].z - mean[b], dw = xv[b][j].w - mean[b];
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
    }
    q = wave_sum(q);
    if (lane == 0) red[b * 4 + wave] = q;
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < B; ++b) {
    const float var = ((red[b * 4 + 0] + red[b * 4 + 1]) + (red[b * 4 + 2] + red[b * 4 + 3])) / (float)K;
    const float rstd = 1.0f / sqrtf(var + p.a.ln_eps);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int k = t * 4 + j * 1024;
      if (k < K) {
        float4 v = xv[b][j];
        const float4 w = ld4(p.a.ln_w + k), bb = ld4(p.a.ln_b + k);
        v.x = (v.x - mean[b]) * rstd * w.x + bb.x;
        v.y = (v.y - mean[b]) * rstd * w.y + bb.y;
        v.z = (v.z - mean[b]) * rstd * w.z + bb.z;
        v.w = (v.w - mean[b]) * rstd * w.w + bb.w;
        *reinterpret_cast<float4*>(xs + b * K + k) = v;
      }
    }
  }
  __syncthreads();
}

// splits (pages) whose partial outputs are prefetched ahead of the weight row (register budget: B*2*CS float4)
template <int B> struct CSof { static constexpr int v = (B <= 2) ? 8 : 2; };

// Split-KV combine prologue (merge of ssrhip_attn_decode's per-page partials):
//   out[b][h][:] = sum_s w_s * o_s,  w_s = e^{m_s-M} l-normalised:  e^{m_s-M} / sum_t e^{m_t-M} l_t
// Split like the LayerNorm prologue: ALL loads (the (m,l) pairs of every (row, head) and the first CS
// partial outputs of this thread's two float4 columns) are issued before the first weight row, the math
// runs while that row is in flight. Threads 0..B*H-1 turn (m,l) into the weights w_s once per block (LDS).
template <int B>
struct CombineRegs {
  static constexpr int CS = CSof<B>::v;
  float4 o[B][2][CS];
  float4 ml[CS / 2];      // CS (m,l) pairs of the (row, head) this thread owns in the weight table
};

template <int B>
__device__ __forceinline__ void combine_issue(const GemvK& p, CombineRegs<B>& r, const int (&ns)[B]) {
  constexpr int CS = CSof<B>::v;
  const int K = p.a.K, hd = p.hd, H = K / hd, MS = p.a.max_splits, t = threadIdx.x;
  if (t < B * H) {
    const float* ml = p.a.part_ml + (size_t)t * MS * 2;
    const int n = ns[t / H];
#pragma unroll
    for (int i = 0; i < CS / 2; ++i) {
      if (2 * i + 1 < MS) {
        r.ml[i] = (2 * i < n) ? ld4(ml + 4 * i) : make_float4(-INFINITY, 0.f, -INFINITY, 0.f);
      } else {                                                       // odd max_splits: the last pair has no partner (never read past the block)
        const float2 v = (2 * i < n) ? *reinterpret_cast<const float2*>(ml + 4 * i) : make_float2(-INFINITY, 0.f);
        r.ml[i] = make_float4(v.x, v.y, -INFINITY, 0.f);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int e = t * 4 + j * 1024;
    if (e < K) {
      const int h = e / hd, d = e % hd;
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const float* po = p.a.part_o + (((size_t)b * H + h) * MS) * hd + d;
#pragma unroll
        for (int s2 = 0; s2 < CS; ++s2) r.o[b][j][s2] = (s2 < ns[b]) ? ld4(po + (size_t)s2 * hd) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
}

template <int B>
__device__ __forceinline__ void combine_finish(const GemvK& p, CombineRegs<B>& r, const int (&ns)[B], float* xs, float* wtab) {
  constexpr int CS = CSof<B>::v;
  const int K = p.a.K, hd = p.hd, H = K / hd, MS = p.a.max_splits, t = threadIdx.x;
  if (t < B * H) {          // weights of (row, head) = t
    const int n = ns[t / H];
    const float* ml = p.a.part_ml + (size_t)t * MS * 2;
    float M = -INFINITY;
#pragma unroll
    for (int i = 0; i < CS / 2; ++i) M = fmaxf(M, fmaxf(r.ml[i].x, r.ml[i].z));
    for (int s2 = CS; s2 < n; ++s2) M = fmaxf(M, ml[2 * s2]);
    float den = 0.f;
#pragma unroll
    for (int i = 0; i < CS / 2; ++i) {
      if (2 * i < n) den = fmaf(expf(r.ml[i].x - M), r.ml[i].y, den);
      if (2 * i + 1 < n) den = fmaf(expf(r.ml[i].z - M), r.ml[i].w, den);
    }
    for (int s2 = CS; s2 < n; ++s2) den = fmaf(expf(ml[2 * s2] - M), ml[2 * s2 + 1], den);
    const float inv = 1.0f / den;
#pragma unroll
    for (int i = 0; i < CS / 2; ++i) {
      if (2 * i < n) wtab[t * MS + 2 * i] = expf(r.ml[i].x - M) * inv;
      if (2 * i + 1 < n) wtab[t * MS + 2 * i + 1] = expf(r.ml[i].z - M) * inv;
    }
    for (int s2 = CS; s2 < n; ++s2) wtab[t * MS + s2] = expf(ml[2 * s2] - M) * inv;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int e = t * 4 + j * 1024;
    if (e < K) {
      const int h = e / hd, d = e % hd;
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const float* w = wtab + (b * H + h) * MS;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s2 = 0; s2 < CS; ++s2) {
          if (s2 < ns[b]) {
            const float ws = w[s2];
            acc.x = fmaf(ws, r.o[b][j][s2].x, acc.x);
            acc.y = fmaf(ws, r.o[b][j][s2].y, acc.y);
            acc.z = fmaf(ws, r.o[b][j][s2].z, acc.z);
            acc.w = fmaf(ws, r.o[b][j][s2].w, acc.w);
          }
        }
        const float* po = p.a.part_o + (((size_t)b * H + h) * MS) * hd + d;
        for (int s2 = CS; s2 < ns[b]; ++s2) {      // long contexts: the remaining pages, not prefetched
          const float ws = w[s2];
          const float4 o = ld4(po + (size_t)s2 * hd);
          acc.x = fmaf(ws, o.x, acc.x);
          acc.y = fmaf(ws, o.y, acc.y);
          acc.z = fmaf(ws, o.z, acc.z);
          acc.w = fmaf(ws, o.w, acc.w);
        }
        *reinterpret_cast<float4*>(xs + b * K + e) = acc;
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == SSRHIP_ACT_RELU) return fmaxf(v, 0.f);
  if (act == SSRHIP_ACT_GELU_ERF) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  return v;
}

// per-row epilogue operands requested together with the weight row (they sit on the tail of every row otherwise)
struct RowEpi { float bias, resid; };

__device__ __forceinline__ RowEpi load_epi(const GemvK& p, int g, int n, int lane, int B) {
  const ssrhip_gemv_args& a = p.a;
  RowEpi e;
  e.bias = a.bias ? a.bias[(size_t)g * a.N + n] : 0.f;
  e.resid = (a.epi == SSRHIP_EPI_RESIDUAL && lane < B) ? a.y[(size_t)lane * a.y_stride + (size_t)g * a.N + n] : 0.f;
  return e;
}

__device__ __forceinline__ void finalize(const GemvK& p, int g, int n, int b, float v, const RowEpi& e, float* const (&kvb)[2]) {
  const ssrhip_gemv_args& a = p.a;
  v += e.bias;
  v = apply_act(v, a.act);
  if (a.epi == SSRHIP_EPI_STORE) {
    a.y[(size_t)b * a.y_stride + (size_t)g * a.N + n] = v;
  } else if (a.epi == SSRHIP_EPI_RESIDUAL) {
    a.y[(size_t)b * a.y_stride + (size_t)g * a.N + n] = e.resid + v;
  } else {  // QKV append: rows [0,D) -> q, [D,2D) -> k cache, [2D,3D) -> v cache (page bases resolved once per wave)
    const int D = a.K;
    const int which = n / D, c = n % D;
    if (which == 0) a.y[(size_t)b * a.y_stride + c] = v;
    else kvb[which - 1][(size_t)(c / p.hd) * SSRHIP_PAGE * p.hd + (c % p.hd)] = v;
  }
}

// K / V base addresses (head 0) of the cache position this step appends, for the batch row `bsel` of the calling thread. The chain
// kv_pos -> page table -> pool stays on the SCALAR path for all B rows side by side: the B position words, one wait, the B table words,
// one wait — two scalar round trips whatever B — and the thread's own row is picked with selects (a branch per row made hipcc walk the
// rows' chains one after the other inside exec-masked blocks: 2 B dependent round trips).
template <int B>
__device__ __forceinline__ void kv_append_bases(const ssrhip_gemv_args& a, int bsel, float* (&kvb)[2]) {
  int pos[B], page[B];
#pragma unroll
  for (int b = 0; b < B; ++b) pos[b] = a.kv_pos[b];
#pragma unroll
  for (int b = 0; b < B; ++b) page[b] = a.kv.table[(size_t)b * a.kv.max_pages + (pos[b] / SSRHIP_PAGE)];
  size_t ok = 0, ov = 0;
#pragma unroll
  for (int b = 0; b < B; ++b) {
    const size_t k0 = ((((size_t)page[b] * a.kv.n_layer + a.layer) * 2 + 0) * a.kv.n_head) * SSRHIP_PAGE + (pos[b] % SSRHIP_PAGE);
    const size_t v0 = ((((size_t)page[b] * a.kv.n_layer + a.layer) * 2 + 1) * a.kv.n_head) * SSRHIP_PAGE + (pos[b] % SSRHIP_PAGE);
    ok = (bsel == b) ? k0 : ok;
    ov = (bsel == b) ? v0 : ov;
  }
  kvb[0] = a.kv.pool + ok * a.kv.head_dim;
  kvb[1] = a.kv.pool + ov * a.kv.head_dim;
}

template <bool FULL>
__device__ __forceinline__ void load_row(float4 (&w)[MAXCH], const float* wrow, int lane, int nch, int len) {
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    if (i < nch) {
      const int k = (i * 64 + lane) * 4;
      if (FULL || k < len) w[i] = ld_nt(wrow + k);
      else w[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// one row slice: FMAs against the register-resident x, wave all-reduce, lanes 0..B-1 hand the result on
template <int B>
__device__ __forceinline__ void consume_row(const GemvK& p, const float4 (&w)[MAXCH], const float4 (&xr)[B][MAXCH], int g, int n,
                                            int it, int lane, int wave, float* part, const RowEpi& e, float* const (&kvb)[2]) {
  float mine = 0.f;
#pragma unroll
  for (int b = 0; b < B; ++b) {
    float s0 = 0.f, s1 = 0.f;   // two chains for ILP
#pragma unroll
    for (int i = 0; i < MAXCH; i += 2) {
      if (i < p.nch) s0 = dot4(w[i], xr[b][i], s0);
      if (i + 1 < p.nch) s1 = dot4(w[i + 1], xr[b][i + 1], s1);
    }
    const float s = wave_sum(s0 + s1);
    if (lane == b) mine = s;
  }
  if (lane < B) {
    if (p.nslice == 1) finalize(p, g, n, lane, mine, e, kvb);
    else part[(wave * MAX_IT + it) * B + lane] = mine;
  }
}

template <int B, int PRO>
__global__ __launch_bounds__(256, (B <= 2 && PRO != SSRHIP_PRO_ATTN_COMBINE) ? 3 : 2) void gemv_kernel(const GemvK p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const ssrhip_gemv_args& a = p.a;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.y;
  const int K = a.K, N = a.N;
  const int slice = wave % p.nslice, rg = wave / p.nslice, n_rg = 4 / p.nslice;
  const int k0 = slice * p.slice_len;
  const int len = min(p.slice_len, K - k0);   // floats in this wave's slice (may be <= 0 for tiny K)
  const bool full = (len == p.nch * 256);
  const int G = blockIdx.x * n_rg + rg;        // this wave-group's first row; then += groups_x
  const float* Wg = a.W + (size_t)g * N * K + k0;
  float* part = smem;                          // [4 waves][MAX_IT][B] cross-slice partials
  float* xs = smem + 4 * MAX_IT * B + 16;      // staged x (LayerNorm / combine prologues)

  // LayerNorm with gamma/beta folded into W/bias by the caller (ln_w == NULL): only (x - mean) * rstd is left, and a wave
  // that owns whole rows (nslice == 1) holds the complete x row in its own registers -> no LDS, no block barrier.
  constexpr bool ln_regs = (PRO == PRO_LN_REGS);

  // ---- prologue loads (L2-resident activations) go out first ...
  float4 xr[B][MAXCH];
  float4 xv[(PRO == SSRHIP_PRO_LAYERNORM) ? B : 1][NJ];
  CombineRegs<(PRO == SSRHIP_PRO_ATTN_COMBINE) ? B : 1> cr;
  int ns[(PRO == SSRHIP_PRO_ATTN_COMBINE) ? B : 1];
  if constexpr (ln_regs) {
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int i = 0; i < MAXCH; ++i) {
        const int k = (i * 64 + lane) * 4;
        xr[b][i] = (i < p.nch && k < len) ? ld4(a.x + (size_t)b * a.x_stride + (size_t)g * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
  }
  if constexpr (PRO == SSRHIP_PRO_LAYERNORM) ln_issue<B>(p, g, reinterpret_cast<float4 (&)[B][NJ]>(xv));
  if constexpr (PRO == SSRHIP_PRO_ATTN_COMBINE) {
#pragma unroll
    for (int b = 0; b < B; ++b) ns[b] = (a.row_len[b] + SSRHIP_PAGE - 1) / SSRHIP_PAGE;
    combine_issue<B>(p, reinterpret_cast<CombineRegs<B>&>(cr), reinterpret_cast<int (&)[B]>(ns));
  }
  // KV-append destinations of this lane's batch row, resolved once (kv_pos -> page table -> pool address)
  float* kvb[2] = {nullptr, nullptr};
  if (a.epi == SSRHIP_EPI_QKV_APPEND && lane < B) {
    const int pos = a.kv_pos[lane];
    kvb[0] = kv_addr(a.kv, lane, a.layer, 0, 0, pos);
    kvb[1] = kv_addr(a.kv, lane, a.layer, 1, 0, pos);
  }
  // ---- ... then the first TWO weight rows of this wave are requested, before any prologue math
  // (loads return in order per wave: the prologue data arrives first, the rows keep flying under the math)
  float4 wa[MAXCH], wb[MAXCH];
  RowEpi ea = {0.f, 0.f}, eb = {0.f, 0.f};
  int na = G, nb = G + p.groups_x;
  if (na < N) {
    if (full) load_row<true>(wa, Wg + (size_t)na * K, lane, p.nch, len);
    else load_row<false>(wa, Wg + (size_t)na * K, lane, p.nch, len);
    if (p.nslice == 1) ea = load_epi(p, g, na, lane, B);
  }
  if (nb < N) {
    if (full) load_row<true>(wb, Wg + (size_t)nb * K, lane, p.nch, len);
    else load_row<false>(wb, Wg + (size_t)nb * K, lane, p.nch, len);
    if (p.nslice == 1) eb = load_epi(p, g, nb, lane, B);
  }

  // ---- prologue math -> this wave's slice of x in registers
  if constexpr (ln_regs) {
    {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        float s0 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) s0 += (xr[b][i].x + xr[b][i].y) + (xr[b][i].z + xr[b][i].w);     // zero beyond K
        const float mean = wave_sum(s0) / (float)K;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
          const bool ok = (i < p.nch) && ((i * 64 + lane) * 4 < len);
          const float dx = xr[b][i].x - mean, dy = xr[b][i].y - mean, dz = xr[b][i].z - mean, dw = xr[b][i].w - mean;
          q += ok ? ((dx * dx + dy * dy) + (dz * dz + dw * dw)) : 0.f;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + a.ln_eps);
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
          const bool ok = (i < p.nch) && ((i * 64 + lane) * 4 < len);
          xr[b][i] = ok ? make_float4((xr[b][i].x - mean) * rstd, (xr[b][i].y - mean) * rstd, (xr[b][i].z - mean) * rstd, (xr[b][i].w - mean) * rstd)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
  }
  if constexpr (PRO == SSRHIP_PRO_LAYERNORM) ln_finish<B>(p, reinterpret_cast<float4 (&)[B][NJ]>(xv), xs, smem);   // `red` aliases `part`
  if constexpr (PRO == SSRHIP_PRO_ATTN_COMBINE) combine_finish<B>(p, reinterpret_cast<CombineRegs<B>&>(cr), reinterpret_cast<int (&)[B]>(ns), xs, xs + B * K);
  if constexpr (!ln_regs) {
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const float* xb = (PRO == SSRHIP_PRO_NONE) ? (a.x + (size_t)b * a.x_stride + (size_t)g * K) : (xs + b * K);
#pragma unroll
      for (int i = 0; i < MAXCH; ++i) {
        const int k = (i * 64 + lane) * 4;
        xr[b][i] = (i < p.nch && k < len) ? ld4(xb + k0 + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }

  // ---- main loop: two rows in flight per wave, ping-pong register sets
  int it = 0;
  const int step2 = 2 * p.groups_x;
  while (na < N) {
    consume_row<B>(p, wa, xr, g, na, it, lane, wave, part, ea, kvb);
    ++it;
    na += step2;
    if (na < N) {
      if (full) load_row<true>(wa, Wg + (size_t)na * K, lane, p.nch, len);
      else load_row<false>(wa, Wg + (size_t)na * K, lane, p.nch, len);
      if (p.nslice == 1) ea = load_epi(p, g, na, lane, B);
    }
    if (nb >= N) break;
    consume_row<B>(p, wb, xr, g, nb, it, lane, wave, part, eb, kvb);
    ++it;
    nb += step2;
    if (nb < N) {
      if (full) load_row<true>(wb, Wg + (size_t)nb * K, lane, p.nch, len);
      else load_row<false>(wb, Wg + (size_t)nb * K, lane, p.nch, len);
      if (p.nslice == 1) eb = load_epi(p, g, nb, lane, B);
    }
  }
  if (p.nslice > 1) {
    __syncthreads();
    // thread t -> (rg, it, b); sum the slices in a fixed order
    const int t = threadIdx.x;
    if (t < n_rg * MAX_IT * B) {
      const int b = t % B, i2 = (t / B) % MAX_IT, rg2 = t / (B * MAX_IT);
      const int nn = (blockIdx.x * n_rg + rg2) + i2 * p.groups_x;
      if (nn < N) {
        float v = 0.f;
        for (int s = 0; s < p.nslice; ++s) v += part[((rg2 * p.nslice + s) * MAX_IT + i2) * B + b];
        RowEpi e;
        e.bias = a.bias ? a.bias[(size_t)g * N + nn] : 0.f;
        e.resid = (a.epi == SSRHIP_EPI_RESIDUAL) ? a.y[(size_t)b * a.y_stride + (size_t)g * N + nn] : 0.f;
        float* kv2[2] = {nullptr, nullptr};
        if (a.epi == SSRHIP_EPI_QKV_APPEND) {
          kv2[0] = kv_addr(a.kv, b, a.layer, 0, 0, a.kv_pos[b]);
          kv2[1] = kv_addr(a.kv, b, a.layer, 1, 0, a.kv_pos[b]);
        }
        finalize(p, g, nn, b, v, e, kv2);
      }
    }
  }
}

template <int B>
void launch_b(const GemvK& p, dim3 grid, size_t smem, hipStream_t s) {
  switch (p.a.pro) {
    case SSRHIP_PRO_LAYERNORM:
      if (p.a.ln_w == nullptr && p.nslice == 1) hipLaunchKernelGGL((gemv_kernel<B, PRO_LN_REGS>), grid, dim3(256), smem, s, p);
      else hipLaunchKernelGGL((gemv_kernel<B, SSRHIP_PRO_LAYERNORM>), grid, dim3(256), smem, s, p);
      break;
    case SSRHIP_PRO_ATTN_COMBINE: hipLaunchKernelGGL((gemv_kernel<B, SSRHIP_PRO_ATTN_COMBINE>), grid, dim3(256), smem, s, p); break;
    default: hipLaunchKernelGGL((gemv_kernel<B, SSRHIP_PRO_NONE>), grid, dim3(256), smem, s, p); break;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Segment kernel (round 2, second half): the same fused GEMV organised the way `tools/overlap_bench xmodes/sweep` found
// fastest for a DEPENDENT chain of launches on this GPU:
//   * 512-thread workgroups, 2 per CU, each owning a CONTIGUOUS block of rows (balanced split of N over the grid);
//   * the work unit is a (row, 1024-float segment of K): 4 x global_load_dwordx4 per lane; a wave keeps two units = 8 loads
//     in flight (the row-per-wave kernel above keeps 16 in 12 waves per CU: 192 KB per CU in flight where ~60 KB cover the
//     HBM latency-bandwidth product; the surplus only lengthens the ramp: 15.7 MB took 6.5 us with 16 in flight, 4.8 with 4);
//   * a wave always works on the same segment (8 waves, S = K/1024 in {1,2,4,8}), so its x slice is B x 4 float4 registers
//     (was B x 8) and every workgroup fetches each x element once per pair of waves instead of once per wave;
//   * per unit a wave all-reduce, lanes 0..B-1 park the partial in LDS; after ONE barrier threads (row, b) add the S segments
//     in k order and apply bias / activation / residual / KV append (operands fetched at kernel entry);
//   * LayerNorm prologue without staging x: a wave normalises its own segment in registers — per-segment two-pass statistics,
//     exchanged through LDS (one barrier), merged exactly (Chan) — gamma / beta folded into W / bias by the host;
//   * split-KV combine prologue: one float4 column per thread and row (512 threads x 4 = K = 2048), 6 pages prefetched;
//   * 16 waves per CU (two workgroups): 128 VGPRs per lane (HIP's second launch-bound is waves per SIMD).
// Loads of the merge prologue's COLD path (contexts beyond the SEG_CS prefetched pages), hidden from hipcc's wait-count bookkeeping: load
// and wait in one asm statement. A compiler-visible load inside those loops made hipcc put `s_waitcnt vmcnt(0)` on the HOT path as well
// (the loops' pre-headers and the first use of a prefetched partial behind them): every out-projection drained its whole weight slice
// before the merge arithmetic, a barrier and an LDS round trip instead of under them (read off the ISA, round 5). The asm wait drains the
// queue too — but only when a late page exists.
__device__ __forceinline__ float4 ld4_late(const float* p) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f v;
  asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float2 ld2_late(const float* p) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f v;
  asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return make_float2(v.x, v.y);
}
constexpr int SEG = 1024;            // floats per unit
constexpr int SEG_TH = 512, SEG_NW = 8;
template <int B> struct SegCS { static constexpr int v = (B <= 2) ? 6 : 2; };   // pages prefetched by the combine prologue (register budget: 128)

// TWO: every wave owns at most two units (rows_max * S <= 16: the out-projection at one workgroup per CU, head-MLP1): BOTH are requested at
// kernel entry instead of the second one being re-requested in place after the first was used. For the out-projection the first use comes
// only after the split-KV merge prologue (a dependent L2 round trip + exp + two barriers), so the in-place form started its second HBM
// round trip ~2 us into the kernel (round 4's time stamps of the same prologue: tools/attn_fused_lab); with both in flight from the
// start the whole 64 KB slice of the workgroup lands under the prologue. Same arithmetic per unit: bit-identical results.
// (the merge-prologue variant with two units at entry runs at ONE workgroup per CU — try_seg — so it may take up to 256 VGPRs: at 128 it spilled)
template <int B, int PRO, bool TWO>
__global__ __launch_bounds__(SEG_TH, (TWO && PRO == SSRHIP_PRO_ATTN_COMBINE) ? 2 : 4) void gemv_seg_kernel(const GemvK p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // Time stamps exist only in a -DSSR_GEMV_PROFILE build (tools/gemv_prof.py). Round 4 shipped them as a runtime `if (p.prof ..)`: the
  // exec-mask branch at kernel entry split the kernel-argument fetch into two dependent scalar round trips, and — worse — its global store
  // made every later uniform load "possibly clobbered", so the kv_pos -> page table chain of the QKV launch left the scalar path and became
  // vector loads behind `s_waitcnt vmcnt(0)`: four drains of the first unit's weight loads per wave (read off the ISA, round 5).
#ifdef SSR_GEMV_PROFILE
#define GSTAMP(i) do { if (p.prof && threadIdx.x == 0) p.prof[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define GSTAMP(i) do { } while (0)
#endif
  GSTAMP(0);
  constexpr int SEG_CS = SegCS<B>::v;
  const ssrhip_gemv_args& a = p.a;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int g = blockIdx.y;
  const int K = a.K, N = a.N, S = p.nslice;                        // S segments per row (power of two <= 8)
  // balanced contiguous split of the N rows over the grid, WITHOUT a division: floor(N * bid / G) is a 64-bit division, which gfx9 emulates
  // with ~150 dependent scalar / vector instructions — twice, at the head of every wave, in front of the first weight request (seen in the
  // ISA: 397 instructions before the first global_load; round 4). The host passes N / G and N % G instead.
  const int r0 = (int)blockIdx.x * p.rows_per + min((int)blockIdx.x, p.rows_rem);
  const int nrows = p.rows_per + ((int)blockIdx.x < p.rows_rem ? 1 : 0), nu = nrows * S;                       // host guarantees N >= G: nrows >= 1
  const int seg = wave & (S - 1), sh = p.seg_shift;                // sh = log2(S)
  float* part = smem;                                              // [rows_max][S][B]
  float* aux = smem + p.rows_max * S * B;                          // prologue scratch
  const float* Wg = a.W + ((size_t)g * N + r0) * K + seg * SEG + lane * 4;

  // ---- 0. epilogue operands of the (row, b) this thread finalises: the OLDEST loads of the wave. They used to be requested behind the first
  // units; hipcc packs the merge / dot arithmetic into v_pk_fma_f32 and one of those register PAIRS held the residual in its unused half, so
  // the first use of a prefetched partial waited for the youngest load of the wave — `s_waitcnt vmcnt(0)`, the whole weight slice drained
  // before the merge arithmetic (read off the ISA, round 5). As the oldest loads such a false dependency costs nothing.
  RowEpi efin = {0.f, 0.f};
  const int bfin = t % B, rfin = min(t / B, nrows - 1), nfin = r0 + rfin;
  efin.bias = a.bias ? a.bias[(size_t)g * N + nfin] : 0.f;
  efin.resid = (a.epi == SSRHIP_EPI_RESIDUAL) ? a.y[(size_t)bfin * a.y_stride + (size_t)g * N + nfin] : 0.f;
  // ---- 1. activations (L2) — issued first, they return first
  float4 xr[B][4];
  float4 co[(PRO == SSRHIP_PRO_ATTN_COMBINE) ? B : 1][SEG_CS];
  float2 cml[SEG_CS];                                              // (m, l) of the first SEG_CS pages of this thread's (row, head)
  int ns[(PRO == SSRHIP_PRO_ATTN_COMBINE) ? B : 1];
  if constexpr (PRO != SSRHIP_PRO_ATTN_COMBINE) {
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) xr[b][i] = ld4(a.x + (size_t)b * a.x_stride + (size_t)g * K + seg * SEG + (i * 64 + lane) * 4);
  } else {                                                         // K == 2048: thread t owns float4 column t*4 of every row
    const int hd = p.hd, H = K / hd, MS = a.max_splits;
#pragma unroll
    for (int b = 0; b < B; ++b) ns[b] = (a.row_len[b] + SSRHIP_PAGE - 1) / SSRHIP_PAGE;
    const int tt = t % (B * H);                                    // every thread loads an (m,l) block; only t < B*H uses it
    const float* ml = a.part_ml + (size_t)tt * MS * 2;
#pragma unroll
    for (int i = 0; i < SEG_CS; ++i) cml[i] = *reinterpret_cast<const float2*>(ml + 2 * min(i, MS - 1));   // pair by pair: any max_splits >= 1, odd ones too
    const int e = t * 4, h = e / hd, d = e % hd;
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const float* po = a.part_o + (((size_t)b * H + h) * MS) * hd + d;
#pragma unroll
      for (int s2 = 0; s2 < SEG_CS; ++s2) co[b][s2] = ld4(po + (size_t)min(s2, MS - 1) * hd);   // address independent of row_len (no scalar-load
                                                                                                  // round trip ahead of these); pages beyond the row are dropped below
    }
  }
  // ---- 2. the wave's first unit, unconditional (clamped to the workgroup's last unit). FOUR loads in flight per lane, 16 waves per
  // CU: 64 KB per CU cover the HBM latency-bandwidth product; `tools/overlap_bench orders`: 8 in flight cost +1.5 us per launch.
  float4 wa[4];
  float4 wb[TWO ? 4 : 1];
  int ua = wave;
  {
    const int ca = min(ua, nu - 1) >> sh;                           // local row (the segment is the wave's own)
#pragma unroll
    for (int i = 0; i < 4; ++i) wa[i] = ld_nt(Wg + (size_t)ca * K + i * 256);
    if constexpr (TWO) {
      const int cb = min(ua + SEG_NW, nu - 1) >> sh;
#pragma unroll
      for (int i = 0; i < 4; ++i) wb[i] = ld_nt(Wg + (size_t)cb * K + i * 256);
    }
  }
  GSTAMP(1);
  float* kvb[2] = {nullptr, nullptr};
  if (a.epi == SSRHIP_EPI_QKV_APPEND) kv_append_bases<B>(a, bfin, kvb);   // scalar-path address chain (kv_pos -> page table -> pool)
  // ---- 3. prologue math, under the latency of the first units
  if constexpr (PRO == SSRHIP_PRO_LAYERNORM) {
    float m[B], q[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
      float s0 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) s0 += (xr[b][i].x + xr[b][i].y) + (xr[b][i].z + xr[b][i].w);
      m[b] = wave_sum(s0) * (1.0f / SEG);
      float q0 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dx = xr[b][i].x - m[b], dy = xr[b][i].y - m[b], dz = xr[b][i].z - m[b], dw = xr[b][i].w - m[b];
        q0 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
      q[b] = wave_sum(q0);
      if (wave < S && lane == 0) { aux[(wave * B + b) * 2] = m[b]; aux[(wave * B + b) * 2 + 1] = q[b]; }   // wave w < S holds segment w
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < B; ++b) {
      float mean = 0.f, M2 = 0.f, dev = 0.f;
      for (int s2 = 0; s2 < S; ++s2) mean += aux[(s2 * B + b) * 2];
      mean /= (float)S;
      for (int s2 = 0; s2 < S; ++s2) { const float dm = aux[(s2 * B + b) * 2] - mean; M2 += aux[(s2 * B + b) * 2 + 1]; dev = fmaf(dm, dm, dev); }
      const float var = (M2 + (float)SEG * dev) / (float)K;
      const float rstd = 1.0f / sqrtf(var + a.ln_eps);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        xr[b][i] = make_float4((xr[b][i].x - mean) * rstd, (xr[b][i].y - mean) * rstd, (xr[b][i].z - mean) * rstd, (xr[b][i].w - mean) * rstd);
    }
  }
  if constexpr (PRO == SSRHIP_PRO_ATTN_COMBINE) {
    const int hd = p.hd, H = K / hd, MS = a.max_splits;
    float* xs = aux;                                               // [B][K]
    float* wtab = aux + B * K;                                     // [B*H][MS]
    if (t < B * H) {                                               // softmax-merge weights of (row, head) = t
      const int n = ns[t / H];
      const float* ml = a.part_ml + (size_t)t * MS * 2;
      float M = -INFINITY;
#pragma unroll
      for (int i = 0; i < SEG_CS; ++i)
        if (i < n) M = fmaxf(M, cml[i].x);
      for (int s2 = SEG_CS; s2 < n; ++s2) M = fmaxf(M, ld2_late(ml + 2 * s2).x);
      float den = 0.f;
#pragma unroll
      for (int i = 0; i < SEG_CS; ++i)
        if (i < n) den = fmaf(expf(cml[i].x - M), cml[i].y, den);
      for (int s2 = SEG_CS; s2 < n; ++s2) { const float2 v = ld2_late(ml + 2 * s2); den = fmaf(expf(v.x - M), v.y, den); }
      const float inv = 1.0f / den;
#pragma unroll
      for (int i = 0; i < SEG_CS; ++i)
        if (i < n) wtab[t * MS + i] = expf(cml[i].x - M) * inv;
      for (int s2 = SEG_CS; s2 < n; ++s2) wtab[t * MS + s2] = expf(ld2_late(ml + 2 * s2).x - M) * inv;
    }
    __syncthreads();
    const int e = t * 4, h = e / hd, d = e % hd;
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const float* w = wtab + (b * H + h) * MS;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int s2 = 0; s2 < SEG_CS; ++s2) {
        const bool in = s2 < ns[b];                                   // a page beyond the row: whatever the buffer holds there must not count (0 * NaN)
        const float ws = in ? w[s2] : 0.f;
        acc.x = fmaf(ws, in ? co[b][s2].x : 0.f, acc.x);
        acc.y = fmaf(ws, in ? co[b][s2].y : 0.f, acc.y);
        acc.z = fmaf(ws, in ? co[b][s2].z : 0.f, acc.z);
        acc.w = fmaf(ws, in ? co[b][s2].w : 0.f, acc.w);
      }
      const float* po = a.part_o + (((size_t)b * H + h) * MS) * hd + d;
      for (int s2 = SEG_CS; s2 < ns[b]; ++s2) {                    // contexts beyond the prefetched pages: the rest, loaded late
        const float ws = w[s2];
        const float4 o = ld4_late(po + (size_t)s2 * hd);
        acc.x = fmaf(ws, o.x, acc.x);
        acc.y = fmaf(ws, o.y, acc.y);
        acc.z = fmaf(ws, o.z, acc.z);
        acc.w = fmaf(ws, o.w, acc.w);
      }
      *reinterpret_cast<float4*>(xs + b * K + e) = acc;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) xr[b][i] = *reinterpret_cast<const float4*>(xs + b * K + seg * SEG + (i * 64 + lane) * 4);
  }
  GSTAMP(2);
  // ---- 4. stream the units: every 16-byte piece is re-requested for the next unit as soon as it has been used
  auto reduce_park = [&](float (&acc)[B][2], int u) {
    float mine = 0.f;
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const float sum = wave_sum(acc[b][0] + acc[b][1]);
      if (lane == b) mine = sum;
    }
    if (lane < B) part[u * B + lane] = mine;                       // u = local_row * S + seg
  };
  if constexpr (TWO) {
    auto unit = [&](const float4 (&w)[4], int u) {
      float acc[B][2];
#pragma unroll
      for (int b = 0; b < B; ++b) acc[b][0] = acc[b][1] = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < B; ++b) acc[b][i & 1] = dot4(w[i], xr[b][i], acc[b][i & 1]);
      reduce_park(acc, u);
    };
    if (ua < nu) unit(wa, ua);
    if (ua + SEG_NW < nu) unit(reinterpret_cast<const float4 (&)[4]>(wb), ua + SEG_NW);
  } else if (ua < nu) {
    while (ua + SEG_NW < nu) {                                      // not the wave's last unit: re-request in place, UNCONDITIONALLY
      const int un = ua + SEG_NW;                                   // (a conditional re-request makes hipcc drain the queue every unit)
      const float* wn = Wg + (size_t)(un >> sh) * K;
      float acc[B][2];
#pragma unroll
      for (int b = 0; b < B; ++b) acc[b][0] = acc[b][1] = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int b = 0; b < B; ++b) acc[b][i & 1] = dot4(wa[i], xr[b][i], acc[b][i & 1]);
        __builtin_amdgcn_sched_barrier(0);                          // use, THEN overwrite in place: no second register set, no copies
        wa[i] = ld_nt(wn + i * 256);
        __builtin_amdgcn_sched_barrier(0);
      }
      reduce_park(acc, ua);
      ua = un;
    }
    float acc[B][2];                                                // the last unit: nothing left to request
#pragma unroll
    for (int b = 0; b < B; ++b) acc[b][0] = acc[b][1] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int b = 0; b < B; ++b) acc[b][i & 1] = dot4(wa[i], xr[b][i], acc[b][i & 1]);
    reduce_park(acc, ua);
  }
  GSTAMP(3);
  __syncthreads();
  GSTAMP(4);
  if (t < nrows * B) {
    float v = 0.f;
    for (int s2 = 0; s2 < S; ++s2) v += part[(rfin * S + s2) * B + bfin];
    finalize(p, g, nfin, bfin, v, efin, kvb);
  }
  GSTAMP(5);
#undef GSTAMP
}

// Round 5: the segment kernel in the form tools/gemv_floor_lab.hip measured fastest for a dependent chain (modes 5 / 6: 9.16-9.27 us for
// 50.3 MB against 9.44-9.49 for the geometry above): ONE 8-wave workgroup per CU, every wave keeps DEPTH units (DEPTH x 4 loads per lane)
// in flight and re-requests a 16-byte piece for unit j + DEPTH right after unit j used it. The number of units per wave NUW is a template
// parameter — the 830M step's shapes divide evenly (QKV 24 rows x 2 segments / 8 waves = 6, FFN1 32 x 2 / 8 = 8, FFN2 8 x 8 / 8 = 8,
// head-MLP1 16 x 2 / 8 = 4) — so the unit loop is straight-line code: every re-request is unconditional AND none is wasted (no peeled
// tail, no clamping, exact `s_waitcnt vmcnt(n)` everywhere). Half the workgroups of the form above: half the x traffic from L2 (8 waves
// fetch their slices instead of 16), half the LayerNorm statistics, one barrier pair per CU instead of two, 256 dispatches instead of 512.
// Per unit, per segment and per output the SAME operations in the same order as gemv_seg_kernel: bit-identical results (tests compare).
// Measured and NOT kept (round 5): requesting only the wave's FIRST unit at entry and units 1 .. DEPTH-1 behind the LayerNorm — the 16-row
// kernels gain from getting their LayerNorm out of the way of the weight requests (csrc/gemv_mfma.hip), here 16 KB of x per workgroup is no
// obstacle and the later requests only cost: 0.8196 -> 0.8340 ms/step, 655.1 -> 674.6 us per step's GEMVs
// (profiles/r05_microbench/decode_ab_ramp.log, gemvm_bench_2_ramp.log). Everything is requested at entry.
// Also measured and NOT kept (round 5): a LayerNorm without the workgroup barrier — every wave fetches the other segment's x slice too and
// computes both segments' statistics itself (same bits, one barrier left in the kernel, 149 VGPRs): 0.8047 -> 0.8098 ms/step, 657.6 ->
// 664.2 us per step's GEMVs (profiles/r05_microbench/decode_ab_lnlocal.log): the second 8 KB per wave and the doubled statistics cost more
// than the barrier they remove.
template <int B, int PRO, int NUW, int DEPTH>
__global__ __launch_bounds__(SEG_TH, 2) void gemv_segu_kernel(const GemvK p) {
  static_assert(PRO == SSRHIP_PRO_NONE || PRO == SSRHIP_PRO_LAYERNORM, "the split-KV merge prologue stays on gemv_seg_kernel");
  static_assert(DEPTH >= 1 && DEPTH <= NUW, "units in flight");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const ssrhip_gemv_args& a = p.a;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int g = blockIdx.y;
  const int K = a.K, N = a.N, S = p.nslice, sh = p.seg_shift;
  const int nrows = p.rows_per;                                    // exact: the host takes this kernel only when N % G == 0
  const int r0 = (int)blockIdx.x * nrows;
  const int seg = wave & (S - 1);
  float* part = smem;                                              // [nrows][S][B]
  float* aux = smem + nrows * S * B;                               // LayerNorm statistics of the S segments
  const float* Wg = a.W + ((size_t)g * N + r0) * K + seg * SEG + lane * 4;

  // ---- 0. epilogue operands of the (row, b) this thread finalises: the wave's oldest loads (see gemv_seg_kernel)
  RowEpi efin = {0.f, 0.f};
  const int bfin = t % B, rfin = min(t / B, nrows - 1), nfin = r0 + rfin;
  efin.bias = a.bias ? a.bias[(size_t)g * N + nfin] : 0.f;
  efin.resid = (a.epi == SSRHIP_EPI_RESIDUAL) ? a.y[(size_t)bfin * a.y_stride + (size_t)g * N + nfin] : 0.f;
  // ---- 1. the wave's x slice (L2), then its first DEPTH units (HBM, non-temporal)
  float4 xr[B][4];
#pragma unroll
  for (int b = 0; b < B; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) xr[b][i] = ld4(a.x + (size_t)b * a.x_stride + (size_t)g * K + seg * SEG + (i * 64 + lane) * 4);
  float4 w[DEPTH][4];
#pragma unroll
  for (int j = 0; j < DEPTH; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) w[j][i] = ld_nt(Wg + (size_t)((wave + SEG_NW * j) >> sh) * K + i * 256);
  float* kvb[2] = {nullptr, nullptr};
  if (a.epi == SSRHIP_EPI_QKV_APPEND) kv_append_bases<B>(a, bfin, kvb);
  // ---- 2. LayerNorm statistics under the latency of the first units (the arithmetic of gemv_seg_kernel, operation for operation)
  if constexpr (PRO == SSRHIP_PRO_LAYERNORM) {
    float m[B], q[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
      float s0 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) s0 += (xr[b][i].x + xr[b][i].y) + (xr[b][i].z + xr[b][i].w);
      m[b] = wave_sum(s0) * (1.0f / SEG);
      float q0 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dx = xr[b][i].x - m[b], dy = xr[b][i].y - m[b], dz = xr[b][i].z - m[b], dw = xr[b][i].w - m[b];
        q0 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
      q[b] = wave_sum(q0);
      if (wave < S && lane == 0) { aux[(wave * B + b) * 2] = m[b]; aux[(wave * B + b) * 2 + 1] = q[b]; }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < B; ++b) {
      float mean = 0.f, M2 = 0.f, dev = 0.f;
      for (int s2 = 0; s2 < S; ++s2) mean += aux[(s2 * B + b) * 2];
      mean /= (float)S;
      for (int s2 = 0; s2 < S; ++s2) { const float dm = aux[(s2 * B + b) * 2] - mean; M2 += aux[(s2 * B + b) * 2 + 1]; dev = fmaf(dm, dm, dev); }
      const float var = (M2 + (float)SEG * dev) / (float)K;
      const float rstd = 1.0f / sqrtf(var + a.ln_eps);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        xr[b][i] = make_float4((xr[b][i].x - mean) * rstd, (xr[b][i].y - mean) * rstd, (xr[b][i].z - mean) * rstd, (xr[b][i].w - mean) * rstd);
    }
  }
  // ---- 3. the units, straight-line: use a piece, then re-request it in place for unit j + DEPTH
#pragma unroll
  for (int j = 0; j < NUW; ++j) {
    float4 (&wj)[4] = w[j % DEPTH];
    float acc[B][2];
#pragma unroll
    for (int b = 0; b < B; ++b) acc[b][0] = acc[b][1] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int b = 0; b < B; ++b) acc[b][i & 1] = dot4(wj[i], xr[b][i], acc[b][i & 1]);
      if (j + DEPTH < NUW) {
        __builtin_amdgcn_sched_barrier(0);
        wj[i] = ld_nt(Wg + (size_t)((wave + SEG_NW * (j + DEPTH)) >> sh) * K + i * 256);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float mine = 0.f;
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const float sum = wave_sum(acc[b][0] + acc[b][1]);
      if (lane == b) mine = sum;
    }
    if (lane < B) part[(wave + SEG_NW * j) * B + lane] = mine;     // unit u = local_row * S + seg
  }
  __syncthreads();
  if (t < nrows * B) {
    float v = 0.f;
    for (int s2 = 0; s2 < S; ++s2) v += part[(rfin * S + s2) * B + bfin];
    finalize(p, g, nfin, bfin, v, efin, kvb);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 5: TWO consecutive GEMVs of the 2-row step in ONE launch — `A` = a GEMV with the residual epilogue (N = D = 2048) and `B` = the
// launch that consumes its output through a LayerNorm — with the all-to-all edge between them (every CU needs all B x D outputs of A)
// inside the launch. Two forms: A = FFN2 (K = 8192) with B = QKV of the next layer or the head MLP after the last layer
// (gemv_pair_kernel), and A = split-KV merge + out-projection (K = 2048) with B = FFN1 (gemv_pair_merge_kernel); the step is then QKV of
// layer 0 and, per layer, attention + two pair launches: 51 launches instead of 83, 0.7926 -> 0.7338 ms/step in a same-box A/B
// (profiles/r05_microbench/decode_ab_pair2.log). tools/layer_edge_lab.hip priced 12 forms of the edge on this machine
// (profiles/r05_microbench/layer_edge_lab.log): two launches 21.3 us; one edge wave gathering alone 22.8; requests of B posted before the
// publish 23.6 (the publish queues behind 128 KB of requests); ... ; this form 19.7-19.9:
//   * 256 workgroups (one per CU) x 12 waves. Waves 0-7 stream A's units exactly as gemv_segu_kernel does and park the partial sums.
//   * barrier; wave 8 finishes the workgroup's 8 rows x 2 outputs, writes them to the residual stream AND publishes them as 8-byte
//     {value, tag = 1} granules with agent-scope (write-through, sc1) stores; barrier; only now waves 0-7 post their first THREE units of B
//     (the weights do not depend on the edge; not all four: 128 KB per CU of requests in front of the gather's loads delay it by ~3 us).
//   * waves 8-11 gather a quarter of the 4096 granules each — 8 x 16-byte sc1 loads per lane, ONE round trip per sweep — until every tag
//     is valid, and put x' into LDS; barrier; waves 0-7 read their slice from LDS, post the fourth unit, LayerNorm, B's units, B's epilogue.
//   * tags: three granule buffers; consecutive pair launches of a step use different ones and every launch resets (plain stores, visible
//     behind the kernel boundary) the buffer the NEXT pair launch will use, so a tag of 1 is always this launch's (the host assigns the
//     buffers, cyclically closed over the step so that graph replays stay consistent).
// Every partial sum, every statistic and every epilogue is computed by the same operations in the same order as in the two launches
// (tests compare bit for bit; tools/layer_edge_lab.hip compared 192 chained edges).
// The spin is bounded (1 s of the 100 MHz clock): a workgroup that gives up sets ws->gave_up and results are garbage from there on — the host checks the
// flag (ssrhip_gemv_pair_status) and raises. The launch needs all its 256 workgroups resident at the same time: true on an otherwise
// idle or ordinarily busy GPU (other kernels finish and make room), NOT when a second pair launch of another stream / process holds
// half the CUs at the same moment — one decode chain per device, or SSRHIP_GEMV_PAIR=0.
constexpr int PAIR_D = 2048, PAIR_NE = 4, PAIR_TH = SEG_TH + 64 * PAIR_NE, PAIR_GRAN = 2 * PAIR_D, PAIR_PF = 3, PAIR_DEPTH = 4, PAIR_NUWA = 8;
constexpr int PAIR_SPINS = 4000000;                       // a second bound only; the first is PAIR_WAIT_TICKS
constexpr long long PAIR_WAIT_TICKS = 100000000;           // 1 s of wall_clock64 (100 MHz)
struct PairK {
  GemvK a, b;
  unsigned long long* gran;        // this launch's granules [PAIR_GRAN]: index n * 2 + row
  unsigned long long* gran_next;   // the next pair launch's buffer: reset here
  int* gave_up;
};
typedef float pair_v4f __attribute__((ext_vector_type(4)));
// eight 16-byte agent-scope loads of the sweep, issued together and waited for together; hipcc must not see them (its wait-count
// bookkeeping would otherwise serialise them against the streaming waves' loads at the join)
__device__ __forceinline__ void pair_sweep8(const unsigned long long* p, pair_v4f (&g)[8]) {
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

// ---- the streaming role from barrier (1) on: B = LayerNorm + Linear on x' (gemv_segu_kernel<2, PRO_LAYERNORM, NUWB, 4>, operation for operation)
// EARLY (round 6, the merge form only): B's units 0 .. EARLY - 1 were requested at kernel ENTRY into the ring slots A does not use
// (unit j lives in w[(j + EARLY) % DEPTH]); what is requested here starts behind them. Same arithmetic, same order of operations.
// PFX = units in flight behind barrier (1b), i.e. in front of the gather's loads (PAIR_PF = 3 for the forms without early units).
template <int NUWB, int EARLY = 0, int PFX = PAIR_PF>
__device__ __forceinline__ void pair_stream_b(const PairK& p, int t, int lane, int wave, float4 (&xr)[2][4], float4 (&w)[PAIR_DEPTH][4],
                                              const RowEpi& efinB, float* partB, float* aux, const float* xs, const size_t* kvoff) {
  constexpr int B = 2, DEPTH = PAIR_DEPTH, PF = PFX, SB = 2, SHB = 1, RB = NUWB * SEG_NW / SB;
  static_assert(PFX >= EARLY && PFX <= PAIR_DEPTH, "units requested at (1b) start behind the early ones and fit the ring");
  static_assert(EARLY == 0 || EARLY == 2, "ring slots 2 and 3 are the ones the merge form's A phase leaves free");
  const ssrhip_gemv_args& bb = p.b.a;
  const int rB0 = (int)blockIdx.x * RB, segB = wave & (SB - 1);
  const int bfin = t % B, rfinB = min(t / B, RB - 1), nfinB = rB0 + rfinB;
  const float* WgB = bb.W + (size_t)rB0 * bb.K + segB * SEG + lane * 4;
  __syncthreads();                                                  // (1b) wave 8 has issued the publish
  // B's first units: PF of them now, the rest behind the gather
#pragma unroll
  for (int j = EARLY; j < PF; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) w[(j + EARLY) % DEPTH][i] = ld_nt(WgB + (size_t)((wave + SEG_NW * j) >> SHB) * bb.K + i * 256);
  __syncthreads();                                                  // (2) x' is in LDS
#pragma unroll
  for (int b = 0; b < B; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) xr[b][i] = *reinterpret_cast<const float4*>(xs + b * PAIR_D + segB * SEG + (i * 64 + lane) * 4);
#pragma unroll
  for (int j = PF; j < DEPTH; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) w[(j + EARLY) % DEPTH][i] = ld_nt(WgB + (size_t)((wave + SEG_NW * j) >> SHB) * bb.K + i * 256);
  // LayerNorm
  {
    float m[B], q[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
      float s0 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) s0 += (xr[b][i].x + xr[b][i].y) + (xr[b][i].z + xr[b][i].w);
      m[b] = wave_sum(s0) * (1.0f / SEG);
      float q0 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dx = xr[b][i].x - m[b], dy = xr[b][i].y - m[b], dz = xr[b][i].z - m[b], dw = xr[b][i].w - m[b];
        q0 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
      q[b] = wave_sum(q0);
      if (wave < SB && lane == 0) { aux[(wave * B + b) * 2] = m[b]; aux[(wave * B + b) * 2 + 1] = q[b]; }
    }
    __syncthreads();                                                // (3)
#pragma unroll
    for (int b = 0; b < B; ++b) {
      float mean = 0.f, M2 = 0.f, dev = 0.f;
      for (int s2 = 0; s2 < SB; ++s2) mean += aux[(s2 * B + b) * 2];
      mean /= (float)SB;
      for (int s2 = 0; s2 < SB; ++s2) { const float dm = aux[(s2 * B + b) * 2] - mean; M2 += aux[(s2 * B + b) * 2 + 1]; dev = fmaf(dm, dm, dev); }
      const float var = (M2 + (float)SEG * dev) / (float)bb.K;
      const float rstd = 1.0f / sqrtf(var + bb.ln_eps);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        xr[b][i] = make_float4((xr[b][i].x - mean) * rstd, (xr[b][i].y - mean) * rstd, (xr[b][i].z - mean) * rstd, (xr[b][i].w - mean) * rstd);
    }
  }
  // units
#pragma unroll
  for (int j = 0; j < NUWB; ++j) {
    float4 (&wj)[4] = w[(j + EARLY) % DEPTH];
    float acc[B][2];
#pragma unroll
    for (int b = 0; b < B; ++b) acc[b][0] = acc[b][1] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int b = 0; b < B; ++b) acc[b][i & 1] = dot4(wj[i], xr[b][i], acc[b][i & 1]);
      if (j + DEPTH < NUWB) {
        __builtin_amdgcn_sched_barrier(0);
        wj[i] = ld_nt(WgB + (size_t)((wave + SEG_NW * (j + DEPTH)) >> SHB) * bb.K + i * 256);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float mine = 0.f;
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const float sum = wave_sum(acc[b][0] + acc[b][1]);
      if (lane == b) mine = sum;
    }
    if (lane < B) partB[(wave + SEG_NW * j) * B + lane] = mine;
  }
  __syncthreads();                                                  // (4)
  if (t < RB * B) {
    float v = 0.f;
    for (int s2 = 0; s2 < SB; ++s2) v += partB[(rfinB * SB + s2) * B + bfin];
    // K / V append addresses: resolved by the edge role (wave 9) while this role streamed — the kv_pos -> page table -> pool chain costs
    // the streaming waves nothing here (in gemv_segu_kernel it is two scalar round trips per wave under the first units' latency)
    float* kvb[2] = {bb.kv.pool + kvoff[bfin * 2], bb.kv.pool + kvoff[bfin * 2 + 1]};
    finalize(p.b, 0, nfinB, bfin, v, efinB, kvb);
  }
}

// ---- the edge role (waves 8-11). SA = segments per row of A (partial sums to add), NPRE = barriers of A's prologue to keep company
template <int SA, int NPRE>
__device__ __forceinline__ void pair_edge_role(const PairK& p, int e, int lane, const float* partA, float* xs, size_t* kvoff) {
  constexpr int B = 2, RA = 8;
  const ssrhip_gemv_args& a = p.a.a;
  const ssrhip_gemv_args& bb = p.b.a;
  const int rA0 = (int)blockIdx.x * RA;
  RowEpi efinA = {0.f, 0.f};
  if (e == 0 && lane < RA * B) {
    efinA.bias = a.bias ? a.bias[rA0 + (lane >> 1)] : 0.f;
    efinA.resid = a.y[(size_t)(lane & 1) * a.y_stride + rA0 + (lane >> 1)];
    p.gran_next[(size_t)blockIdx.x * (RA * B) + lane] = 0ull;       // reset the NEXT pair launch's granules (this workgroup's 16 of them)
  }
  if (bb.epi == SSRHIP_EPI_QKV_APPEND && e == 1 && lane < B) {      // kv_append_bases()'s arithmetic for batch row `lane`, as element offsets
    const int pos = bb.kv_pos[lane];
    const int page = bb.kv.table[(size_t)lane * bb.kv.max_pages + (pos / SSRHIP_PAGE)];
    const size_t k0 = ((((size_t)page * bb.kv.n_layer + bb.layer) * 2 + 0) * bb.kv.n_head) * SSRHIP_PAGE + (pos % SSRHIP_PAGE);
    const size_t v0 = ((((size_t)page * bb.kv.n_layer + bb.layer) * 2 + 1) * bb.kv.n_head) * SSRHIP_PAGE + (pos % SSRHIP_PAGE);
    kvoff[lane * 2 + 0] = k0 * bb.kv.head_dim;
    kvoff[lane * 2 + 1] = v0 * bb.kv.head_dim;
  }
#pragma unroll
  for (int i = 0; i < NPRE; ++i) __syncthreads();
  __syncthreads();                                                  // (1)
  if (e == 0 && lane < RA * B) {
    // A's epilogue — finalize()'s RESIDUAL arm — into the residual stream and, tagged, into this launch's granules
    const int r = lane >> 1, b = lane & 1, n = rA0 + r;
    float v = 0.f;
    for (int s2 = 0; s2 < SA; ++s2) v += partA[(r * SA + s2) * B + b];
    v += efinA.bias;
    const float out = efinA.resid + v;
    a.y[(size_t)b * a.y_stride + n] = out;
    const unsigned long long gval = ((unsigned long long)1u << 32) | (unsigned long long)__float_as_uint(out);
    __hip_atomic_store(p.gran + (size_t)n * 2 + b, gval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();                                                  // (1b)
  // gather this wave's quarter of the granules: one round trip per sweep, until every tag is this launch's. Once ANY launch on this
  // workspace has given up (results are garbage from there on and the host will raise at its next poll) the later launches do not wait
  // ~half a second each any more: a chain that cannot get its workgroups resident together costs one long stall, not one per launch
  pair_v4f g[8];
  bool done = false;
  const int max_spins = (*p.gave_up != 0) ? 64 : PAIR_SPINS;
  // the bound is TIME (round 6): ~1 s of the constant 100 MHz clock, looked at every 256 sweeps. The sweep count alone (rounds 5) meant
  // anything from 1 s on an idle GPU to well over 9 s when a hundred CUs' worth of edge waves poll the same lines (tests/test_gpu_pair_guard.py)
  const long long t_begin = wall_clock64();
  for (int spin = 0; spin < max_spins && !done; ++spin) {
    bool all = true;
    pair_sweep8(p.gran + (size_t)e * 1024 + lane * 2, g);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      all = all && (__float_as_uint(g[i][1]) == 1u) && (__float_as_uint(g[i][3]) == 1u);
      const int gi = e * 1024 + (i >> 2) * 512 + (i & 3) * 128 + lane * 2;   // granule index of g[i].xy (row 0 of output gi / 2); .zw: row 1
      xs[(gi >> 1)] = g[i][0];
      xs[PAIR_D + (gi >> 1)] = g[i][2];
    }
    done = __all(all);
    if (!done) {
      __builtin_amdgcn_s_sleep(2);
      if ((spin & 255) == 255 && wall_clock64() - t_begin > PAIR_WAIT_TICKS) break;
    }
  }
  if (!done && lane == 0) *p.gave_up = 1;
  __syncthreads();                                                  // (2)
  __syncthreads();                                                  // (3)
  __syncthreads();                                                  // (4)
}

// The two ROLES are the two arms of ONE (wave-uniform) branch and every barrier is written in both arms. A first form — a series of
// `if (wave < 8)` blocks with joins in between — made hipcc's wait-count pass merge the "block skipped" path into every join: the first use
// of a unit requested two blocks earlier was guarded by `s_waitcnt vmcnt(4 * PF - 1)` instead of vmcnt(15) (B's phase ran with 8 loads in
// flight per wave instead of 16), the registers of wave 8's epilogue operands were zeroed on the streaming path behind a vmcnt(0), and the
// kernel took 150 VGPRs instead of 112 (read off the ISA; tools/layer_edge_lab.hip modes 6 / 7 against 9 / 10).
// A = FFN2 + residual (K = 8192): gemv_segu_kernel<2, PRO_NONE, 8, 4>
template <int NUWB>
__global__ __launch_bounds__(PAIR_TH, 2) void gemv_pair_kernel(const PairK p) {
  constexpr int B = 2, DEPTH = PAIR_DEPTH, NUWA = PAIR_NUWA;
  constexpr int RA = 8, SA = 8, SHA = 3;                           // A: 8 rows per workgroup, K = 8192 = 8 segments, wave w owns segment w
  constexpr int RB = NUWB * SEG_NW / 2;
  __shared__ float partA[RA * SA * B];
  __shared__ float partB[RB * 2 * B];
  __shared__ float aux[2 * B * 2];
  __shared__ __attribute__((aligned(16))) float xs[B * PAIR_D];
  __shared__ size_t kvoff[B * 2];
  const ssrhip_gemv_args& a = p.a.a;
  const ssrhip_gemv_args& bb = p.b.a;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  if (wave < SEG_NW) {
    const int rA0 = (int)blockIdx.x * RA;
    const float* WgA = a.W + (size_t)rA0 * a.K + wave * SEG + lane * 4;
    // ---- 0. B's epilogue operand of the (row, b) this thread finalises (wave 0's threads do): the wave's oldest load
    RowEpi efinB = {0.f, 0.f};
    efinB.bias = bb.bias ? bb.bias[(int)blockIdx.x * RB + min(t / B, RB - 1)] : 0.f;
    // ---- 1. the wave's slice of A's input (L2), then its first DEPTH units (HBM, non-temporal)
    float4 xr[B][4];
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) xr[b][i] = ld4(a.x + (size_t)b * a.x_stride + wave * SEG + (i * 64 + lane) * 4);
    float4 w[DEPTH][4];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) w[j][i] = ld_nt(WgA + (size_t)((wave + SEG_NW * j) >> SHA) * a.K + i * 256);
    // ---- 2. A's units
#pragma unroll
    for (int j = 0; j < NUWA; ++j) {
      float4 (&wj)[4] = w[j % DEPTH];
      float acc[B][2];
#pragma unroll
      for (int b = 0; b < B; ++b) acc[b][0] = acc[b][1] = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int b = 0; b < B; ++b) acc[b][i & 1] = dot4(wj[i], xr[b][i], acc[b][i & 1]);
        if (j + DEPTH < NUWA) {
          __builtin_amdgcn_sched_barrier(0);
          wj[i] = ld_nt(WgA + (size_t)((wave + SEG_NW * (j + DEPTH)) >> SHA) * a.K + i * 256);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      float mine = 0.f;
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const float sum = wave_sum(acc[b][0] + acc[b][1]);
        if (lane == b) mine = sum;
      }
      if (lane < B) partA[(wave + SEG_NW * j) * B + lane] = mine;
    }
    __syncthreads();                                                // (1) A's partial sums are parked
    pair_stream_b<NUWB>(p, t, lane, wave, xr, w, efinB, partB, aux, xs, kvoff);
  } else {
    pair_edge_role<SA, 0>(p, wave - SEG_NW, lane, partA, xs, kvoff);
  }
}

// A = split-KV merge + out-projection + residual (K = 2048): gemv_seg_kernel<2, PRO_ATTN_COMBINE, TWO = true> at one workgroup per CU,
// operation for operation — thread t owns float4 column t * 4 of both rows in the merge, wave w the units w and w + 8 (both requested at
// entry). The merged rows pass through the LDS buffer that later receives x' (its A-phase use ends before barrier (1)).
// EARLY = 2 (round 6, SSRHIP_GEMV_PAIR_EARLY, default on): A streams only 64 KB per CU and the edge follows — for ~4 us of this launch HBM
// has next to nothing to do while B's 67 MB wait for barrier (1b). B's weights depend on nothing: its first two units are requested at
// kernel entry, right behind A's two, into the ring slots A leaves free (no extra registers), and stream while the merge, A and the edge run.
template <int NUWB, int EARLY = 0, int PFX = PAIR_PF>
__global__ __launch_bounds__(PAIR_TH, 2) void gemv_pair_merge_kernel(const PairK p) {
  constexpr int B = 2, DEPTH = PAIR_DEPTH, SEG_CS = SegCS<B>::v;
  constexpr int RA = 8, SA = 2, SHA = 1;                           // A: 8 rows per workgroup, K = 2048 = 2 segments, 16 units = 2 per wave
  constexpr int RB = NUWB * SEG_NW / 2;
  extern __shared__ __attribute__((aligned(16))) float wtab[];     // [B * H][max_splits] merge weights
  __shared__ float partA[RA * SA * B];
  __shared__ float partB[RB * 2 * B];
  __shared__ float aux[2 * B * 2];
  __shared__ __attribute__((aligned(16))) float xs[B * PAIR_D];
  __shared__ size_t kvoff[B * 2];
  const ssrhip_gemv_args& a = p.a.a;
  const ssrhip_gemv_args& bb = p.b.a;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  if (wave >= SEG_NW) {
    pair_edge_role<SA, 2>(p, wave - SEG_NW, lane, partA, xs, kvoff);
  } else {
    const int K = a.K, rA0 = (int)blockIdx.x * RA, seg = wave & (SA - 1);
    const float* WgA = a.W + (size_t)rA0 * K + seg * SEG + lane * 4;
    RowEpi efinB = {0.f, 0.f};
    efinB.bias = bb.bias ? bb.bias[(int)blockIdx.x * RB + min(t / B, RB - 1)] : 0.f;
    // ---- 1. the attention partials this thread merges (L2): (m, l) of its (row, head), the first SEG_CS pages of its column
    const int hd = p.a.hd, H = K / hd, MS = a.max_splits;
    float4 co[B][SEG_CS];
    float2 cml[SEG_CS];
    int ns[B];
#pragma unroll
    for (int b = 0; b < B; ++b) ns[b] = (a.row_len[b] + SSRHIP_PAGE - 1) / SSRHIP_PAGE;
    {
      const int tt = t % (B * H);
      const float* ml = a.part_ml + (size_t)tt * MS * 2;
#pragma unroll
      for (int i = 0; i < SEG_CS; ++i) cml[i] = *reinterpret_cast<const float2*>(ml + 2 * min(i, MS - 1));
      const int e = t * 4, h = e / hd, d = e % hd;
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const float* po = a.part_o + (((size_t)b * H + h) * MS) * hd + d;
#pragma unroll
        for (int s2 = 0; s2 < SEG_CS; ++s2) co[b][s2] = ld4(po + (size_t)min(s2, MS - 1) * hd);
      }
    }
    // ---- 2. the wave's two units of A, both now
    float4 w[DEPTH][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) w[j][i] = ld_nt(WgA + (size_t)((wave + SEG_NW * j) >> SHA) * K + i * 256);
    if constexpr (EARLY == 2) {
      // B's units 0 and 1 (pair_stream_b's addresses: SB = 2 segments per row, unit u = row (wave + 8 u) / 2 of this workgroup's RB rows)
      const float* WgB = bb.W + (size_t)((int)blockIdx.x * RB) * bb.K + (wave & 1) * SEG + lane * 4;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) w[2 + j][i] = ld_nt(WgB + (size_t)((wave + SEG_NW * j) >> 1) * bb.K + i * 256);
    }
    // ---- 3. merge, under the latency of the units. EVERY thread computes the softmax-merge weights of (row, head) = t % (B * H), threads
    // 0 .. B * H - 1 store them: inside an `if (t < B * H)` hipcc sinks the (m, l) loads into the branch, behind the weight requests, and
    // their first use drains the whole queue (read off the ISA; in gemv_seg_kernel the same source keeps them in front)
    {
      const int tt = t % (B * H);
      const bool keep = t < B * H;
      const int n = ns[tt / H];
      const float* ml = a.part_ml + (size_t)tt * MS * 2;
      float M = -INFINITY;
#pragma unroll
      for (int i = 0; i < SEG_CS; ++i)
        if (i < n) M = fmaxf(M, cml[i].x);
      for (int s2 = SEG_CS; s2 < n; ++s2) M = fmaxf(M, ld2_late(ml + 2 * s2).x);
      float den = 0.f;
#pragma unroll
      for (int i = 0; i < SEG_CS; ++i)
        if (i < n) den = fmaf(expf(cml[i].x - M), cml[i].y, den);
      for (int s2 = SEG_CS; s2 < n; ++s2) { const float2 v = ld2_late(ml + 2 * s2); den = fmaf(expf(v.x - M), v.y, den); }
      const float inv = 1.0f / den;
#pragma unroll
      for (int i = 0; i < SEG_CS; ++i)
        if (keep && i < n) wtab[tt * MS + i] = expf(cml[i].x - M) * inv;
      for (int s2 = SEG_CS; s2 < n; ++s2) { const float wv = expf(ld2_late(ml + 2 * s2).x - M) * inv; if (keep) wtab[tt * MS + s2] = wv; }
    }
    __syncthreads();                                                // (A1)
    {
      const int e = t * 4, h = e / hd, d = e % hd;
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const float* wt = wtab + (b * H + h) * MS;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s2 = 0; s2 < SEG_CS; ++s2) {
          const bool in = s2 < ns[b];
          const float ws = in ? wt[s2] : 0.f;
          acc.x = fmaf(ws, in ? co[b][s2].x : 0.f, acc.x);
          acc.y = fmaf(ws, in ? co[b][s2].y : 0.f, acc.y);
          acc.z = fmaf(ws, in ? co[b][s2].z : 0.f, acc.z);
          acc.w = fmaf(ws, in ? co[b][s2].w : 0.f, acc.w);
        }
        const float* po = a.part_o + (((size_t)b * H + h) * MS) * hd + d;
        for (int s2 = SEG_CS; s2 < ns[b]; ++s2) {
          const float ws = wt[s2];
          const float4 o = ld4_late(po + (size_t)s2 * hd);
          acc.x = fmaf(ws, o.x, acc.x);
          acc.y = fmaf(ws, o.y, acc.y);
          acc.z = fmaf(ws, o.z, acc.z);
          acc.w = fmaf(ws, o.w, acc.w);
        }
        *reinterpret_cast<float4*>(xs + b * K + e) = acc;
      }
    }
    __syncthreads();                                                // (A2)
    float4 xr[B][4];
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) xr[b][i] = *reinterpret_cast<const float4*>(xs + b * K + seg * SEG + (i * 64 + lane) * 4);
    // ---- 4. A's two units
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float acc[B][2];
#pragma unroll
      for (int b = 0; b < B; ++b) acc[b][0] = acc[b][1] = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < B; ++b) acc[b][i & 1] = dot4(w[j][i], xr[b][i], acc[b][i & 1]);
      float mine = 0.f;
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const float sum = wave_sum(acc[b][0] + acc[b][1]);
        if (lane == b) mine = sum;
      }
      if (lane < B) partA[(wave + SEG_NW * j) * B + lane] = mine;
    }
    __syncthreads();                                                // (1) A's partial sums are parked
    pair_stream_b<NUWB, EARLY, PFX>(p, t, lane, wave, xr, w, efinB, partB, aux, xs, kvoff);
  }
}

// Round 4 measured the opposite organisation too — ALL of a wave's units requested at kernel entry, one 8-wave workgroup per CU (256 VGPRs):
// bit-identical results, 12.4 us per launch in the step against 10.5 here (profiles/r04_microbench/decode_ab.log; the lab form of the same
// idea is tools/gemv_floor_lab.hip mode 4: 10.0 us against 9.2-9.5 for 1-4 units in flight). Removed again.
int g_seg_mode = -1;   // SSRHIP_GEMV_SEG: 1 (default) = take the segment kernel where its conditions hold, 0 = never

// true if the segment kernel was launched
template <int B>
bool try_seg(const ssrhip_gemv_args* a, int num_cu, hipStream_t s) {
  if (a->K % SEG != 0) return false;
  const int S = a->K / SEG;
  if (S != 1 && S != 2 && S != 4 && S != 8) return false;
  if (a->pro == SSRHIP_PRO_LAYERNORM && a->ln_w != nullptr) return false;
  const int H = a->kv.head_dim > 0 ? a->K / a->kv.head_dim : 0;
  if (a->pro == SSRHIP_PRO_ATTN_COMBINE && (a->K != 2048 || a->max_splits < 1 || a->groups != 1 || B * H > SEG_TH || a->kv.head_dim % 4 != 0)) return false;
  int G = (2 * num_cu) / a->groups;                                // two resident workgroups per CU over all groups
  // the combine prologue makes EVERY workgroup read all the attention partials (~100 KB at 6 pages): with two workgroups per CU that is
  // 3x the CU's share of the weights through its 64 B/clk L2 port; one workgroup per CU halves it
  if (a->pro == SSRHIP_PRO_ATTN_COMBINE && !getenv("SSRHIP_GEMV_SEG_COMBINE_2")) G = num_cu;
  if (G > a->N) G = a->N;                                          // fewer rows than workgroups: one row each
  if (G < 1) G = 1;
  const int rows_max = (a->N + G - 1) / G;
  if (rows_max * B > SEG_TH) return false;
  GemvK p;
  p.a = *a;
  p.nslice = S;
  p.slice_len = SEG;
  p.nch = 4;
  p.seg_shift = (S == 1) ? 0 : (S == 2) ? 1 : (S == 4) ? 2 : 3;
  p.rows_max = rows_max;
  p.rows_per = a->N / G;
  p.rows_rem = a->N % G;
  p.prof = g_gemv_prof;
  p.groups_x = G;
  p.hd = (a->kv.head_dim > 0) ? a->kv.head_dim : 1;
  size_t smem = (size_t)rows_max * S * B * sizeof(float);
  if (a->pro == SSRHIP_PRO_LAYERNORM) smem += (size_t)S * B * 2 * sizeof(float);
  if (a->pro == SSRHIP_PRO_ATTN_COMBINE) smem += ((size_t)B * a->K + (size_t)B * H * a->max_splits) * sizeof(float);
  smem = (smem + 15) / 16 * 16;
  // round 5: one workgroup per CU, NUW units per wave straight-line, DEPTH in flight (gemv_segu_kernel) when the shape divides evenly
  int segu_depth = 4;                                               // SSRHIP_GEMV_SEGU = 0 (off) | 2 | 4 (default): units in flight per wave; read at
  if (const char* e = getenv("SSRHIP_GEMV_SEGU")) { segu_depth = atoi(e); if (segu_depth != 0 && segu_depth != 2) segu_depth = 4; }   // every call
  if constexpr (B == 2) {
    const int G1 = num_cu / a->groups;
    if (segu_depth && a->pro != SSRHIP_PRO_ATTN_COMBINE && G1 >= 1 && a->N % G1 == 0 && ((a->N / G1) * S) % SEG_NW == 0) {
      const int nuw = (a->N / G1) * S / SEG_NW;
      if (nuw == 4 || nuw == 6 || nuw == 8) {
        p.rows_max = p.rows_per = a->N / G1;
        p.rows_rem = 0;
        p.groups_x = G1;
        size_t sm = (size_t)p.rows_per * S * B * sizeof(float) + (size_t)S * B * 2 * sizeof(float);
        sm = (sm + 15) / 16 * 16;
        const dim3 g1(G1, a->groups);
#define SEGU_LAUNCH(PRO_, NUW_)                                                                                                       \
        do {                                                                                                                           \
          if (segu_depth == 2) hipLaunchKernelGGL((gemv_segu_kernel<B, PRO_, NUW_, 2>), g1, dim3(SEG_TH), sm, s, p);                    \
          else hipLaunchKernelGGL((gemv_segu_kernel<B, PRO_, NUW_, 4>), g1, dim3(SEG_TH), sm, s, p);                                    \
        } while (0)
        if (a->pro == SSRHIP_PRO_LAYERNORM) {
          if (nuw == 4) SEGU_LAUNCH(SSRHIP_PRO_LAYERNORM, 4); else if (nuw == 6) SEGU_LAUNCH(SSRHIP_PRO_LAYERNORM, 6); else SEGU_LAUNCH(SSRHIP_PRO_LAYERNORM, 8);
        } else {
          if (nuw == 4) SEGU_LAUNCH(SSRHIP_PRO_NONE, 4); else if (nuw == 6) SEGU_LAUNCH(SSRHIP_PRO_NONE, 6); else SEGU_LAUNCH(SSRHIP_PRO_NONE, 8);
        }
#undef SEGU_LAUNCH
        return true;
      }
    }
  }
  dim3 grid(G, a->groups);
  static int two_mode = -1;                                         // SSRHIP_GEMV_SEG_TWO=0: A/B knob (the in-place form for every shape)
  if (two_mode < 0) { const char* e = getenv("SSRHIP_GEMV_SEG_TWO"); two_mode = (e && e[0] == '0') ? 0 : 1; }
  const bool two = B <= 2 && two_mode && rows_max * S <= 2 * SEG_NW && rows_max * S > SEG_NW &&   // more than one and at most two units per wave; 4 rows: no registers left
                   (a->pro != SSRHIP_PRO_ATTN_COMBINE || G <= num_cu);                // (its merge variant is built for one workgroup per CU)
  switch (a->pro) {
    case SSRHIP_PRO_LAYERNORM:
      if constexpr (B <= 2) { if (two) { hipLaunchKernelGGL((gemv_seg_kernel<B, SSRHIP_PRO_LAYERNORM, true>), grid, dim3(SEG_TH), smem, s, p); break; } }
      hipLaunchKernelGGL((gemv_seg_kernel<B, SSRHIP_PRO_LAYERNORM, false>), grid, dim3(SEG_TH), smem, s, p);
      break;
    case SSRHIP_PRO_ATTN_COMBINE:
      if constexpr (B <= 2) { if (two) { hipLaunchKernelGGL((gemv_seg_kernel<B, SSRHIP_PRO_ATTN_COMBINE, true>), grid, dim3(SEG_TH), smem, s, p); break; } }
      hipLaunchKernelGGL((gemv_seg_kernel<B, SSRHIP_PRO_ATTN_COMBINE, false>), grid, dim3(SEG_TH), smem, s, p);
      break;
    default:
      if constexpr (B <= 2) { if (two) { hipLaunchKernelGGL((gemv_seg_kernel<B, SSRHIP_PRO_NONE, true>), grid, dim3(SEG_TH), smem, s, p); break; } }
      hipLaunchKernelGGL((gemv_seg_kernel<B, SSRHIP_PRO_NONE, false>), grid, dim3(SEG_TH), smem, s, p);
      break;
  }
  return true;
}

int g_num_cu = 0;
int g_blocks_per_cu = 3;   // A/B in the real (dependent-launch) decode step: 1 -> 1.536, 2 -> 1.109, 3 -> 1.065 ms/step. (Independent
                          // back-to-back launches, tools/gemv_bench.hip, prefer 2: 12.5 us vs 12.6 for 67 MB; the chain wants the faster ramp.)

}  // namespace

int ssrhip_gemv_mfma_launch(const ssrhip_gemv_args* a, hipStream_t s);   // gemv_mfma.hip: 5..16 rows on the matrix core

#ifdef SSR_GEMV_PROFILE
// debug hook of a profiling build (not part of the ABI; tools/gemv_prof.py): per-workgroup time stamps of every later gemv_seg_kernel launch
extern "C" void ssrhip_debug_gemv_prof(void* dev_ptr) { g_gemv_prof = (long long*)dev_ptr; }
#endif

// Units per wave of B if (a, b) can run as one pair launch, 0 otherwise; *merge = A is the out-projection with the split-KV merge prologue
// SSRHIP_GEMV_PAIR (read at every call: A/B inside one process): 0 = never, 1 = only FFN2 -> {QKV, head MLP}, unset / other = both forms
static thread_local char g_pair_why[200] = "";
const char* ssrhip_gemv_pair_why() { return g_pair_why; }          // why the last pair_nuwb on this thread said no (engine.hip reports it)

// Can a pair kernel keep 256 workgroups resident on THIS device at all? (the CU count alone ignores CU masks and occupancy.)
// Asked once per process: the occupancy calculator must place at least one workgroup of every pair instantiation on a CU (12 waves,
// their registers and LDS), the device must report >= 256 CUs and no CU mask may be set in the environment (the attribute counts
// masked CUs too). Partition modes show up in the CU count.
static const char* pair_device_refusal(int num_cu) {
  static const char* verdict = nullptr;
  static bool asked = false;
  if (asked) return verdict;
  asked = true;
  static char buf[200];
  if (num_cu < 256) { snprintf(buf, sizeof(buf), "the device reports %d CUs (a pair launch needs its 256 workgroups resident together)", num_cu); return verdict = buf; }
  for (const char* name : {"ROC_GLOBAL_CU_MASK", "HSA_CU_MASK", "HSA_CU_MASK_SKIP_INIT"}) {
    const char* e = getenv(name);
    if (e && e[0]) { snprintf(buf, sizeof(buf), "%s is set: fewer CUs than the device reports may be usable", name); return verdict = buf; }
  }
  int nb[4] = {0, 0, 0, 0};
  hipError_t e0 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb[0], gemv_pair_kernel<4>, PAIR_TH, 0);
  hipError_t e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb[1], gemv_pair_kernel<6>, PAIR_TH, 0);
  hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb[2], gemv_pair_kernel<8>, PAIR_TH, 0);
  hipError_t e3 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb[3], gemv_pair_merge_kernel<8, 2>, PAIR_TH, 16 * 1024);
  if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || nb[0] < 1 || nb[1] < 1 || nb[2] < 1 || nb[3] < 1) {
    snprintf(buf, sizeof(buf), "the occupancy calculator places %d / %d / %d / %d workgroups of the pair kernels on a CU (need >= 1 each)", nb[0], nb[1], nb[2], nb[3]);
    return verdict = buf;
  }
  return verdict = nullptr;
}

static int pair_nuwb(const ssrhip_gemv_args* a, const ssrhip_gemv_args* b, int num_cu, bool* merge) {
  *merge = false;
  auto no = [](const char* why) { snprintf(g_pair_why, sizeof(g_pair_why), "%s", why); return 0; };
  int mode = 2;
  if (const char* e = getenv("SSRHIP_GEMV_PAIR")) mode = (e[0] == '0') ? 0 : (e[0] == '1') ? 1 : 2;
  if (mode == 0) return no("SSRHIP_GEMV_PAIR=0");
  if (const char* why = pair_device_refusal(num_cu)) return no(why);
  if (a->B != 2 || b->B != 2 || a->groups != 1 || b->groups != 1) return no("not a 2-row, single-group launch");
  if (a->x_tiled || a->y_tiled || a->w_tiled || b->x_tiled || b->y_tiled || b->w_tiled) return no("tiled layouts");
  if (a->act != SSRHIP_ACT_NONE || a->epi != SSRHIP_EPI_RESIDUAL || a->N != PAIR_D || !a->y || !a->W) return no("the first launch is not a residual GEMV with N = 2048 (the pair kernels are built for d_model 2048)");
  if (a->pro == SSRHIP_PRO_NONE) {
    if (a->K != 8192 || !a->x) return no("FFN2 form needs K = 8192");
  } else if (a->pro == SSRHIP_PRO_ATTN_COMBINE) {
    if (mode < 2) return no("SSRHIP_GEMV_PAIR=1: the merge form is off");
    const int hd = a->kv.head_dim;
    if (a->K != 2048 || !a->part_o || !a->part_ml || !a->row_len || a->max_splits < 1 || hd <= 0 || hd % 4 != 0 || a->K % hd != 0 || 2 * (a->K / hd) > SEG_TH) return no("merge form needs K = 2048 and split-KV partials");
    if ((size_t)2 * (a->K / hd) * a->max_splits * sizeof(float) > 16 * 1024) return no("merge weights exceed 16 KB of LDS (context too long for the merge form)");   // merge weights in LDS next to the 17 KB of static buffers
    *merge = true;
  } else {
    return no("the first launch has a LayerNorm prologue");
  }
  if (b->pro != SSRHIP_PRO_LAYERNORM || b->ln_w || b->ln_b || b->K != PAIR_D || b->x != a->y || b->x_stride != a->y_stride || !b->W || !b->y) return no("the second launch is not LayerNorm (folded) + Linear on the first one's output");
  if (b->epi != SSRHIP_EPI_STORE && b->epi != SSRHIP_EPI_QKV_APPEND) return no("second launch: epilogue");
  if (b->epi == SSRHIP_EPI_QKV_APPEND && !(b->N == 3 * b->K && b->kv.pool && b->kv.table && b->kv_pos && b->kv.head_dim > 0)) return no("second launch: QKV append without a cache");
  if (b->N % 256 != 0) return no("second launch: N not a multiple of 256");
  const int nuwb = (b->N / 256) * 2 / SEG_NW;
  if ((b->N / 256) * 2 % SEG_NW != 0 || (nuwb != 4 && nuwb != 6 && nuwb != 8)) return no("second launch: N not in {4096, 6144, 8192}");
  if (*merge && nuwb != 8) return no("the merge form is instantiated for FFN1 only");                                // the merge form is instantiated for FFN1 only
  g_pair_why[0] = 0;
  return nuwb;
}

static void ensure_num_cu() {
  if (g_num_cu == 0) {
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cu > 0) g_num_cu = cu;
    else g_num_cu = 256;
    if (const char* e = getenv("SSRHIP_GEMV_BLOCKS_PER_CU")) { const int v = atoi(e); if (v >= 1 && v <= 3) g_blocks_per_cu = v; }   // tuning knob
  }
}

// Granule buffer of the i-th of n pair launches that follow each other cyclically (a decode step's pairs, replayed step after step):
// i % 3, except that the last launch takes buffer 1 when n % 3 == 1 (it would otherwise share buffer 0 with launch 0 of the next cycle).
// Any two cyclically consecutive launches use different buffers for every n >= 2 (tests/test_capi.py checks 2..200).
extern "C" int ssrhip_pair_buffer(int32_t i, int32_t n) {
  if (n < 2 || i < 0 || i >= n) return -1;
  return (i == n - 1 && n % 3 == 1) ? 1 : i % 3;
}

extern "C" int ssrhip_gemv_pair_applicable(const ssrhip_gemv_args* a, const ssrhip_gemv_args* b) {
  if (!a || !b) return 0;
  ensure_num_cu();
  bool merge;
  return pair_nuwb(a, b, g_num_cu, &merge) != 0;
}

extern "C" int ssrhip_gemv_pair(const ssrhip_gemv_args* a, const ssrhip_gemv_args* b, void* ws, int32_t buf, int32_t buf_next, ssrhip_stream_t stream) {
  SSR_REQUIRE(a && b && ws, "ssrhip_gemv_pair: null argument");
  SSR_REQUIRE(buf >= 0 && buf < 3 && buf_next >= 0 && buf_next < 3 && buf != buf_next, "ssrhip_gemv_pair: granule buffers %d -> %d (0..2, different)", buf, buf_next);
  ensure_num_cu();
  bool merge = false;
  const int nuwb = pair_nuwb(a, b, g_num_cu, &merge);
  if (!nuwb) return 1;
  PairK p;
  auto fill = [](GemvK& k, const ssrhip_gemv_args* g, int S) {
    k.a = *g; k.nslice = S; k.slice_len = SEG; k.nch = 4; k.groups_x = 256; k.hd = (g->kv.head_dim > 0) ? g->kv.head_dim : 1;
    k.seg_shift = (S == 8) ? 3 : 1; k.rows_max = k.rows_per = g->N / 256; k.rows_rem = 0; k.prof = nullptr;
  };
  fill(p.a, a, merge ? 2 : 8);
  fill(p.b, b, 2);
  unsigned long long* gran = (unsigned long long*)ws;
  p.gran = gran + (size_t)buf * PAIR_GRAN;
  p.gran_next = gran + (size_t)buf_next * PAIR_GRAN;
  p.gave_up = (int*)(gran + 3 * (size_t)PAIR_GRAN);
  hipStream_t s = (hipStream_t)stream;
  if (merge) {
    const size_t sm = ((size_t)2 * (a->K / a->kv.head_dim) * a->max_splits * sizeof(float) + 15) / 16 * 16;
    const char* ee = getenv("SSRHIP_GEMV_PAIR_EARLY");             // read at every call (graph capture): A/B inside one process
    // 0 = round 5's order. (With the early units in place, FOUR units in flight behind (1b) measured 0.759 ms/step and TWO — nothing between
    // the publish and the gather — 0.767 against 0.745-0.748 for three: PFX stays PAIR_PF; profiles/r06_microbench/decode_ab_pair_early_pf*.log.)
    if (ee && ee[0] == '0') hipLaunchKernelGGL((gemv_pair_merge_kernel<8, 0>), dim3(256), dim3(PAIR_TH), sm, s, p);
    else hipLaunchKernelGGL((gemv_pair_merge_kernel<8, 2>), dim3(256), dim3(PAIR_TH), sm, s, p);
  } else if (nuwb == 4) hipLaunchKernelGGL((gemv_pair_kernel<4>), dim3(256), dim3(PAIR_TH), 0, s, p);
  else if (nuwb == 6) hipLaunchKernelGGL((gemv_pair_kernel<6>), dim3(256), dim3(PAIR_TH), 0, s, p);
  else hipLaunchKernelGGL((gemv_pair_kernel<8>), dim3(256), dim3(PAIR_TH), 0, s, p);
  SSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int ssrhip_gemv_pair_status(const void* ws, ssrhip_stream_t stream) {
  SSR_REQUIRE(ws, "ssrhip_gemv_pair_status: null workspace");
  int flag = 0;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { ssrhip_set_error("ssrhip_gemv_pair_status: stream synchronize failed"); return -1; }
  if (hipMemcpy(&flag, (const char*)ws + 3 * (size_t)PAIR_GRAN * 8, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) {
    ssrhip_set_error("ssrhip_gemv_pair_status: copy failed");
    return -1;
  }
  if (flag) {
    // reported once; tags and flag back to "nothing published" so that a chain restarted from its first launch (after a new prefill) is
    // valid again (the stream is idle: it was synchronised above)
    if (hipMemset(const_cast<void*>(ws), 0, SSRHIP_PAIR_WS_BYTES) != hipSuccess) { ssrhip_set_error("ssrhip_gemv_pair_status: re-zeroing the workspace failed"); return -1; }
  }
  return flag ? 1 : 0;
}

extern "C" int ssrhip_gemv(const ssrhip_gemv_args* a, ssrhip_stream_t stream) {
  SSR_REQUIRE(a && a->W && a->y, "ssrhip_gemv: null argument");
  SSR_REQUIRE(a->N > 0 && a->groups >= 1 && a->K > 0, "ssrhip_gemv: bad N/K/groups");
  if (a->B > 4) return ssrhip_gemv_mfma_launch(a, (hipStream_t)stream);
  SSR_REQUIRE(a->B == 1 || a->B == 2 || a->B == 4, "ssrhip_gemv: B=%d not in {1,2,4} or 5..16", a->B);
  SSR_REQUIRE(!a->x_tiled && !a->y_tiled && !a->w_tiled, "ssrhip_gemv: the tiled activation / weight layouts are for 5..16 rows only");
  SSR_REQUIRE(a->pro != SSRHIP_PRO_ATTN_COMBINE || (a->kv.head_dim > 0 && a->K <= 2048 && a->B * (a->K / a->kv.head_dim) <= 256), "ssrhip_gemv: combine prologue needs K <= 2048 and B*H <= 256");
  SSR_REQUIRE(a->K > 0 && a->K % 4 == 0 && a->K <= 8192, "ssrhip_gemv: K=%d must be a multiple of 4, <= 8192", a->K);
  SSR_REQUIRE(a->N > 0 && a->groups >= 1, "ssrhip_gemv: bad N/groups");
  ensure_num_cu();
  if (a->pro != SSRHIP_PRO_NONE) {
    SSR_REQUIRE(a->groups == 1 || a->pro == SSRHIP_PRO_LAYERNORM, "ssrhip_gemv: combine prologue needs groups==1");
    if (a->pro == SSRHIP_PRO_LAYERNORM) SSR_REQUIRE(a->x && ((a->ln_w && a->ln_b) || (!a->ln_w && !a->ln_b)), "ssrhip_gemv: LayerNorm prologue needs x and either both or none of ln_w/ln_b");
    if (a->pro == SSRHIP_PRO_ATTN_COMBINE) {
      SSR_REQUIRE(a->part_o && a->part_ml && a->row_len && a->kv.head_dim > 0 && a->K % a->kv.head_dim == 0 && a->kv.head_dim % 4 == 0,
                  "ssrhip_gemv: combine prologue needs part_o, part_ml, row_len, kv.head_dim");
    }
  } else {
    SSR_REQUIRE(a->x, "ssrhip_gemv: x is null");
  }
  if (a->epi == SSRHIP_EPI_QKV_APPEND) {
    SSR_REQUIRE(a->N == 3 * a->K && a->groups == 1 && a->kv.pool && a->kv.table && a->kv_pos && a->kv.head_dim > 0,
                "ssrhip_gemv: QKV epilogue needs N==3K and a kv cache");
  }
  if (g_seg_mode < 0) { const char* e = getenv("SSRHIP_GEMV_SEG"); g_seg_mode = (e && e[0] == '0') ? 0 : 1; }
  if (g_seg_mode) {
    bool done = false;
    switch (a->B) {
      case 1: done = try_seg<1>(a, g_num_cu, (hipStream_t)stream); break;
      case 2: done = try_seg<2>(a, g_num_cu, (hipStream_t)stream); break;
      default: done = try_seg<4>(a, g_num_cu, (hipStream_t)stream); break;
    }
    if (done) { SSR_LAUNCH_CHECK(); return 0; }
  }
  GemvK p;
  p.a = *a;
  p.seg_shift = 0;
  p.rows_max = 0;
  p.rows_per = p.rows_rem = 0;
  p.prof = nullptr;
  p.nslice = a->K <= 2048 ? 1 : (a->K <= 4096 ? 2 : 4);
  p.slice_len = ((a->K + p.nslice - 1) / p.nslice + 3) / 4 * 4;
  p.nch = (p.slice_len + 255) / 256;
  SSR_REQUIRE(p.nch <= MAXCH, "ssrhip_gemv: slice too long");
  SSR_REQUIRE(!(a->pro == SSRHIP_PRO_LAYERNORM && a->ln_w == nullptr && p.nslice != 1),
              "ssrhip_gemv: the row-per-wave kernels take a folded LayerNorm (ln_w == NULL) only for K <= 2048 (the segment kernel covers K = 4096 / 8192)");
  const int n_rg = 4 / p.nslice;
  // resident grid: <= 3 workgroups per CU in total (over all groups); rows dealt round-robin to wave-groups
  int max_blocks_x = (g_blocks_per_cu * g_num_cu) / a->groups;
  if (max_blocks_x < 1) max_blocks_x = 1;
  // exactly `max_blocks_x` workgroups (every CU gets the same share) unless there are fewer rows than wave-groups
  int blocks_x = (a->N + n_rg - 1) / n_rg;
  if (blocks_x > max_blocks_x) blocks_x = max_blocks_x;
  if (p.nslice > 1 && (a->N + blocks_x * n_rg - 1) / (blocks_x * n_rg) > MAX_IT)   // LDS partial capacity: more blocks instead
    blocks_x = ((a->N + MAX_IT - 1) / MAX_IT + n_rg - 1) / n_rg;
  p.groups_x = blocks_x * n_rg;
  p.hd = (a->kv.head_dim > 0) ? a->kv.head_dim : 1;
  size_t smem = (4 * MAX_IT * a->B + 16) * sizeof(float);
  if (a->pro != SSRHIP_PRO_NONE) {
    SSR_REQUIRE(a->groups == 1 || a->pro == SSRHIP_PRO_LAYERNORM, "ssrhip_gemv: combine prologue needs groups==1");
    SSR_REQUIRE((size_t)a->B * a->K * 4 <= 60 * 1024 && a->K <= 4096, "ssrhip_gemv: staged prologue needs B*K*4 <= 60 KiB and K <= 4096");
    smem += (size_t)a->B * a->K * sizeof(float);
    if (a->pro == SSRHIP_PRO_ATTN_COMBINE) smem += (size_t)a->B * (a->K / a->kv.head_dim) * a->max_splits * sizeof(float) + 64;
    if (a->pro == SSRHIP_PRO_LAYERNORM) SSR_REQUIRE(a->x && ((a->ln_w && a->ln_b) || (!a->ln_w && !a->ln_b)), "ssrhip_gemv: LayerNorm prologue needs x and either both or none of ln_w/ln_b");
    if (a->pro == SSRHIP_PRO_ATTN_COMBINE) {
      SSR_REQUIRE(a->part_o && a->part_ml && a->row_len && a->kv.head_dim > 0 && a->K % a->kv.head_dim == 0 && a->kv.head_dim % 4 == 0,
                  "ssrhip_gemv: combine prologue needs part_o, part_ml, row_len, kv.head_dim");
    }
  } else {
    SSR_REQUIRE(a->x, "ssrhip_gemv: x is null");
  }
  if (a->epi == SSRHIP_EPI_QKV_APPEND) {
    SSR_REQUIRE(a->N == 3 * a->K && a->groups == 1 && a->kv.pool && a->kv.table && a->kv_pos && a->kv.head_dim > 0,
                "ssrhip_gemv: QKV epilogue needs N==3K and a kv cache");
  }
  dim3 grid(blocks_x, a->groups);
  hipStream_t s = (hipStream_t)stream;
  switch (a->B) {
    case 1: launch_b<1>(p, grid, smem, s); break;
    case 2: launch_b<2>(p, grid, smem, s); break;
    default: launch_b<4>(p, grid, smem, s); break;
  }
  SSR_LAUNCH_CHECK();
  return 0;
}
