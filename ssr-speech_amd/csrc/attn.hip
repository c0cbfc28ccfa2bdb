// attn.hip — single-query attention over the paged KV cache, split over pages (flash-decoding), gfx950.
//
// One workgroup (4 waves) per (page, head, row); wave w scores 32 keys of the page.
// Layout: a key/value row (head_dim fp32) is covered by LPK = head_dim/4 lanes with one float4 each, so one
// wave-instruction loads 64/LPK whole rows, fully coalesced (512 B contiguous per row at head_dim 128).
// All K and V loads of the wave's 32 keys are issued up front (the kernel is latency-, not
// bandwidth-bound: every cache element is used exactly once per step, so there is no reuse for LDS to
// exploit; LDS only carries the 4-wave merge). q.k partial products are reduced across the LPK lanes
// with xor-shuffles on NI independent values at once; softmax statistics (m, l) and the un-normalised
// output are written per page and merged by the consumer (ssrhip_gemv PRO_ATTN_COMBINE or
// ssrhip_attn_combine) — deterministic, no atomics.
// Replaces F.scaled_dot_product_attention (models/modules/activation.py:634); the additive mask the
// reference builds (models/ssr.py:227-255) is exactly "row r sees positions < row_len[r]".
#include <stdlib.h>
#include <algorithm>
using std::min;
#include "common.h"

namespace {

// K/V of a decode step are read exactly once per step: non-temporal loads (measured: 2 rows 8.9 -> 8.0 us per launch, 0.905 -> 0.898
// ms/step; 16 rows 35.1 -> 31.8 us at context 720, 1.475 -> 1.434 ms/step). -DSSR_ATTN_NT=0 builds the plain-load variant.
#ifndef SSR_ATTN_NT
#define SSR_ATTN_NT 1
#endif
__device__ __forceinline__ float4 ld_kv(const float* p) { return SSR_ATTN_NT ? ld_nt(p) : ld4(p); }

template <int HD, bool SEQ, int VAT = -1>   // SEQ: rows carry an explicit sequence id (a.row_seq != NULL: the per-row prefill path); the decode step has none
__global__ __launch_bounds__(256) void attn_decode_kernel(const ssrhip_attn_args a, const int head_fastest) {   // VAT: see the V requests below
  constexpr int LPK = HD / 4;         // lanes per key row
  constexpr int KPI = 64 / LPK;       // key rows per wave-instruction
  constexpr int NI = 32 / KPI;        // load instructions for the wave's 32 keys
  __shared__ __attribute__((aligned(16))) float sm[4][HD + 4];   // row stride keeps float4 stores 16-B aligned
  // Workgroup b runs on XCD b % 8 (observed, for speed only). With the page index as the fastest grid dimension and 8 pages of capacity
  // every workgroup of page p sat on XCD p: a 600-position context used 5 XCDs' L2s and fabric links and left 3 idle. `head_fastest`
  // makes the head the fastest dimension: the live (page, head, row) items spread over all XCDs whatever the context.
  const int split = head_fastest ? blockIdx.y : blockIdx.x, h = head_fastest ? blockIdx.x : blockIdx.y, r = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPK;         // which key row inside one wave-instruction
  const int c4 = (lane % LPK) * 4;    // this lane's 4 columns
  const int H = a.kv.n_head;
  // the row's length, its page id and q are requested TOGETHER, before the early exit (split < max_pages: the table entry exists and
  // holds a valid page — the engine's spare page — also beyond the row's length): one scalar-memory round trip instead of two in a row.
  // Round 4: the ISA showed FOUR serial scalar round trips in front of the first K/V request (kernel arguments fetched piecemeal, then
  // row_seq, then the table entry): the arguments are pinned into ONE batch, and the decode step's common case (row_seq == NULL) reads
  // table[r][split] without waiting for anything but the arguments. Constant address space: scalar loads (row_len / table / row_seq do
  // not change during the launch).
  typedef const int32_t __attribute__((address_space(4))) cint;
  cint* c_len = (cint*)(uintptr_t)a.row_len;
  cint* c_tab = (cint*)(uintptr_t)a.kv.table;
  cint* c_seq = (cint*)(uintptr_t)a.row_seq;
  asm volatile("; kernel arguments in one batch" :: "s"(a.q), "s"(a.kv.pool), "s"(a.kv.max_pages), "s"(a.kv.n_layer), "s"(a.layer), "s"(a.scale),
               "s"(a.part_o), "s"(a.part_ml), "s"(a.max_splits), "s"(a.q_stride), "s"(H), "s"(c_len), "s"(c_tab), "s"(c_seq), "s"(a.prefetch),
               "s"(a.prefetch_floats));
  const int len = c_len[r];
  const int page = SEQ ? c_tab[(size_t)c_seq[r] * a.kv.max_pages + split]      // rows mapped to another sequence: one more round trip
                       : c_tab[(size_t)r * a.kv.max_pages + split];            // the row's own sequence (decode step): together with its length
  const float4 q = ld4(a.q + (size_t)r * (a.q_stride ? a.q_stride : H * HD) + h * HD + c4);
  if (!SEQ && a.prefetch) {
    // Round 6: this launch is latency-bound (a few dependent round trips for 10 MB of K / V) and most of its workgroups at short contexts
    // have nothing to do at all, while the launch behind it starts by streaming 64 KB of W_o per CU from HBM. Workgroup i touches the
    // slice workgroup i of that launch will read (same XCD: i % 8) — plain loads behind the scalar requests and q, results never used: extra
    // loads the compiler does not know of only make its later waits (`vmcnt(n)` = all but the n youngest) stricter, never laxer, and the lines sit in this XCD's L2 when they are wanted.
    const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (wg < 256u) {
      const float* pf = a.prefetch + (size_t)wg * a.prefetch_floats + threadIdx.x * 4;
      for (int i = 0; i < a.prefetch_floats; i += 1024) {
        float4 junk;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(junk) : "v"(pf + i) : "memory");
      }
    }
  }
  asm volatile("; row length and page id arrive together" :: "s"(len), "s"(page));
  const int base = split * SSRHIP_PAGE;
  if (base >= len) return;            // uniform per block
  const float* kp = a.kv.pool + ((((size_t)page * a.kv.n_layer + a.layer) * 2 + 0) * H + h) * SSRHIP_PAGE * HD;
  const float* vp = kp + (size_t)H * SSRHIP_PAGE * HD;

  float4 kk[NI], vv[NI];
  float s[NI];
  // unconditional loads: keys beyond the row's length are read from the last valid key of the page instead (their scores are
  // masked to -inf below, so p == 0 and the duplicate V rows add nothing). Predicated loads would make hipcc drain the
  // memory queue (s_waitcnt vmcnt(0)) between groups of loads (measured: 9.6 -> 7.3 us per launch at 2 rows).
  const int jmax = min(len - base, SSRHIP_PAGE) - 1;     // >= 0: base < len
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int j = min(wave * 32 + i * KPI + sub, jmax);
    kk[i] = ld_kv(kp + (size_t)j * HD + c4);
  }
  // VAT >= 0: the V rows are requested when VAT of the wave's NI K rows have been consumed — pinned with scheduling fences; -1 leaves the
  // order to hipcc (which keeps ~10 K requests in flight and asks for the V rows behind the LAST K row, see below). Measured at head_dim
  // 128 on the 830M step, same box, alternating engines (profiles/r05_microbench/decode_ab_attn_vat.log): hipcc 6.84 us per launch,
  // VAT 4 / 8 / 12: 6.69 / 6.45 / 6.56 — all 16 K rows in flight from the start and the V rows' flight under the second half of the
  // score arithmetic. The decode step takes VAT = 8 (0.8095 -> 0.8033 ms/step); same arithmetic in the same order: identical tokens.
  if constexpr (VAT >= 0) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < VAT; ++i) s[i] = dot4(q, kk[i], 0.f);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int j = min(wave * 32 + i * KPI + sub, jmax);
    vv[i] = ld_kv(vp + (size_t)j * HD + c4);
  }
  if constexpr (VAT >= 0) __builtin_amdgcn_sched_barrier(0);
  // What hipcc makes of the two loops above (read off the ISA, round 5; rounds 1-4 believed all 2 NI requests fly together): its
  // occupancy-driven scheduler consumes the K rows as they arrive, re-uses their registers and requests the V rows only once the last K
  // row has landed (77 VGPRs). Pinning all 2 NI requests in front of the first use (`sched_barrier(0)` here, 146 VGPRs) was measured on
  // the 830M step, same box, alternating engines: 7.60 -> 7.89 us per launch, 0.8224 -> 0.8243 ms/step (profiles/r05_microbench/
  // decode_ab.log, variants r4sched / pin) — SLOWER: with K first the score / softmax arithmetic runs under the V rows' flight, while 64 KB
  // + 64 KB requested at once arrive together behind one CU's 64 B/clk port and leave nothing to overlap. The compiler's order stays.
#pragma unroll
  for (int i = (VAT >= 0 ? VAT : 0); i < NI; ++i) s[i] = dot4(q, kk[i], 0.f);
  // reduce each s[i] over the LPK lanes of its key row (DPP row rotations + permlane16_swap: no LDS)
#pragma unroll
  for (int i = 0; i < NI; ++i) s[i] = (LPK == 32) ? half32_sum(s[i]) : row16_sum(s[i]);
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int j = wave * 32 + i * KPI + sub;
    s[i] = ((base + j) < len) ? s[i] * a.scale : -INFINITY;
    m = fmaxf(m, s[i]);
  }
  if (LPK == 16) m = fmaxf(m, xor16_f(m));
  m = fmaxf(m, xor32_f(m));
  float l = 0.f;
  float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (m > -INFINITY) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float p = expf(s[i] - m);   // exp(-inf) == 0 for masked keys
      l += p;
      o4.x = fmaf(p, vv[i].x, o4.x);
      o4.y = fmaf(p, vv[i].y, o4.y);
      o4.z = fmaf(p, vv[i].z, o4.z);
      o4.w = fmaf(p, vv[i].w, o4.w);
    }
  }
  // merge the KPI key-row groups of the wave (lanes with equal c4)
  if (LPK == 16) {
    l += xor16_f(l);
    o4.x += xor16_f(o4.x); o4.y += xor16_f(o4.y); o4.z += xor16_f(o4.z); o4.w += xor16_f(o4.w);
  }
  l += xor32_f(l);
  o4.x += xor32_f(o4.x); o4.y += xor32_f(o4.y); o4.z += xor32_f(o4.z); o4.w += xor32_f(o4.w);
  if (lane < LPK) {
    *reinterpret_cast<float4*>(&sm[wave][c4]) = o4;
  }
  if (lane == 0) { sm[wave][HD] = m; sm[wave][HD + 1] = l; }
  __syncthreads();
  // 4-wave merge by the first LPK lanes of wave 0 (fixed order)
  if (threadIdx.x < LPK) {
    float M = fmaxf(fmaxf(sm[0][HD], sm[1][HD]), fmaxf(sm[2][HD], sm[3][HD]));
    float L = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float mw = sm[w][HD];
      const float f = (mw > -INFINITY) ? expf(mw - M) : 0.f;
      L = fmaf(f, sm[w][HD + 1], L);
      acc.x = fmaf(f, sm[w][c4 + 0], acc.x);
      acc.y = fmaf(f, sm[w][c4 + 1], acc.y);
      acc.z = fmaf(f, sm[w][c4 + 2], acc.z);
      acc.w = fmaf(f, sm[w][c4 + 3], acc.w);
    }
    const size_t pi = ((size_t)r * H + h) * a.max_splits + split;
    *reinterpret_cast<float4*>(a.part_o + pi * HD + c4) = acc;
    if (threadIdx.x == 0) { a.part_ml[pi * 2] = M; a.part_ml[pi * 2 + 1] = L; }
  }
}

// Merge of the per-page partials (prefill, and the 5..16-row decode step where the out-proj GEMV does not fuse it).
// One wave per workgroup, HD/4 lanes per head (2 or 4 heads per wave), one float4 of the output per lane. The (m, l) pairs
// and the partial outputs of up to 8 pages are requested together with clamped page indices (no predicated loads), so a
// workgroup needs one memory round trip per 8 pages: 5.8 -> ~3 us per launch at 16 rows (was: one workgroup per ROW looping
// over all heads and pages with dependent loads).
template <int HD>
__global__ __launch_bounds__(64) void attn_combine_kernel(const ssrhip_attn_args a, float* out) {
  constexpr int LPH = HD / 4, HPW = 64 / LPH, CH = 8;
  const int H = a.kv.n_head, D = H * HD;
  const int r = blockIdx.y;
  const int h = min((int)blockIdx.x * HPW + (int)threadIdx.x / LPH, H - 1);
  const bool live = (int)blockIdx.x * HPW + (int)threadIdx.x / LPH < H;
  const int d = (threadIdx.x % LPH) * 4;
  const int ns = (a.row_len[r] + SSRHIP_PAGE - 1) / SSRHIP_PAGE;
  const float* ml = a.part_ml + ((size_t)r * H + h) * a.max_splits * 2;
  const float* po = a.part_o + (((size_t)r * H + h) * a.max_splits) * HD + d;
  float M = -INFINITY;
  for (int s0 = 0; s0 < ns; s0 += CH) {
    float m[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) m[i] = ml[2 * min(s0 + i, ns - 1)];
#pragma unroll
    for (int i = 0; i < CH; ++i) M = fmaxf(M, m[i]);           // duplicates of page ns-1 do not change the max
  }
  float den = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s0 = 0; s0 < ns; s0 += CH) {
    float m[CH], l[CH];
    float4 o[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int s2 = min(s0 + i, ns - 1);
      m[i] = ml[2 * s2];
      l[i] = ml[2 * s2 + 1];
      o[i] = ld4(po + (size_t)s2 * HD);
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const float w = (s0 + i < ns) ? expf(m[i] - M) : 0.f;     // same order of operations as before: fixed page order
      den = fmaf(w, l[i], den);
      acc.x = fmaf(w, o[i].x, acc.x);
      acc.y = fmaf(w, o[i].y, acc.y);
      acc.z = fmaf(w, o[i].z, acc.z);
      acc.w = fmaf(w, o[i].w, acc.w);
    }
  }
  if (!live) return;
  const float inv = 1.0f / den;
  const int e = h * HD + d;
  float* dst = a.out_tiled ? out + SSRHIP_TILED(r, e) : out + (size_t)r * D + e;
  *reinterpret_cast<float4*>(dst) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused single-query attention for MANY rows (the 5..16-row decode step: 16 rows x 16 heads = 256 (row, head) pairs = one
// workgroup per CU). One 8-wave workgroup owns a whole (row, head): it walks the row's pages itself, so there are no
// per-page partials, no merge launch and every CU streams the same number of bytes (rows of a lock-step batch have similar
// lengths). rocprofv3 at 16 rows, context ~520: the split kernel + combine took 40 + 6.7 us per layer for 136 MB of K/V
// (3.4 TB/s); here the workgroup keeps two pages in flight (wave w owns keys [16w, 16w+16) of every page: 16 KB of K/V per
// page per wave, double-buffered = 256 KB per CU) and folds them into a running (m, l, o) — the online softmax — so K/V stream
// at the HBM rate. The 8 waves' states are merged once through LDS and the normalised output row is written directly in
// the layout the out-projection GEMV wants (row-major or SSRHIP_TILED).
// Loads are never predicated (see above): keys past the row's length re-read key 0 of the last page and are masked to -inf.
constexpr int ATTN_ROWS_MAX_PAGES = 256;       // 32,768 positions per row

template <int HD>
__global__ __launch_bounds__(512) void attn_rows_kernel(const ssrhip_attn_args a, float* out) {
  constexpr int LPK = HD / 4, KPI = 64 / LPK, NW = 8, KPW = SSRHIP_PAGE / NW, NI = KPW / KPI;
  __shared__ __attribute__((aligned(16))) float sm[NW][HD + 4];
  const int h = blockIdx.x, r = blockIdx.y;
  const int len = __builtin_amdgcn_readfirstlane(a.row_len[r]);
  const int seq = __builtin_amdgcn_readfirstlane(a.row_seq ? a.row_seq[r] : r);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane / LPK, c4 = (lane % LPK) * 4;
  const int H = a.kv.n_head;
  const int npages = (len + SSRHIP_PAGE - 1) / SSRHIP_PAGE;           // >= 1: a decode row always sees its own key
  // the row's page ids live in 4 VGPRs (lane i of register b holds page 64b + i) and are picked with v_readlane: a table
  // lookup inside the loop would be a VECTOR load (the compiler cannot prove the table is not written by this kernel), and
  // waiting for it — or for an LDS copy of it — drains every K/V load in flight (seen in the ISA: s_waitcnt vmcnt(0))
  int pid[ATTN_ROWS_MAX_PAGES / 64];
#pragma unroll
  for (int b = 0; b < ATTN_ROWS_MAX_PAGES / 64; ++b)
    pid[b] = (b * 64 < npages) ? a.kv.table[(size_t)seq * a.kv.max_pages + min(b * 64 + lane, npages - 1)] : 0;
  const size_t head_off = (size_t)h * SSRHIP_PAGE * HD, v_off = (size_t)H * SSRHIP_PAGE * HD;
  const size_t page_stride = (size_t)a.kv.n_layer * 2 * H * SSRHIP_PAGE * HD;
  const float* pool = a.kv.pool + (size_t)a.layer * 2 * H * SSRHIP_PAGE * HD + head_off;
  const float4 q = ld4(a.q + (size_t)r * (a.q_stride ? a.q_stride : H * HD) + h * HD + c4);

  float4 kk[2][NI], vv[2][NI];
  float m = -INFINITY, l = 0.f;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);

#define ATTN_ISSUE(BUF, PG)                                                                           \
  {                                                                                                   \
    const int pg_ = min((PG), npages - 1);                                                            \
    const int pb_ = pg_ >> 6;                                                                         \
    const int pv_ = pb_ == 0 ? pid[0] : (pb_ == 1 ? pid[1] : (pb_ == 2 ? pid[2] : pid[3]));           \
    const float* kp_ = pool + (size_t)__builtin_amdgcn_readlane(pv_, pg_ & 63) * page_stride;         \
    const int jmax_ = ((PG) < npages) ? min(len - pg_ * SSRHIP_PAGE, SSRHIP_PAGE) - 1 : 0;            \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                  \
      const int j_ = min(wave * KPW + i * KPI + sub, jmax_);                                          \
      kk[BUF][i] = ld_kv(kp_ + (size_t)j_ * HD + c4);                                                  \
    }                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                  \
      const int j_ = min(wave * KPW + i * KPI + sub, jmax_);                                          \
      vv[BUF][i] = ld_kv(kp_ + v_off + (size_t)j_ * HD + c4);                                          \
    }                                                                                                 \
  }
#define ATTN_FOLD(BUF, PG)                                                                            \
  {                                                                                                   \
    float s_[NI];                                                                                     \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) s_[i] = dot4(q, kk[BUF][i], 0.f);                  \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) s_[i] = (LPK == 32) ? half32_sum(s_[i]) : row16_sum(s_[i]); \
    float mloc_ = -INFINITY;                                                                          \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                  \
      const int pos_ = (PG) * SSRHIP_PAGE + wave * KPW + i * KPI + sub;                               \
      s_[i] = (pos_ < len) ? s_[i] * a.scale : -INFINITY;                                             \
      mloc_ = fmaxf(mloc_, s_[i]);                                                                    \
    }                                                                                                 \
    if (LPK == 16) mloc_ = fmaxf(mloc_, xor16_f(mloc_));                                              \
    mloc_ = fmaxf(mloc_, xor32_f(mloc_));                                                             \
    const float mnew_ = fmaxf(m, mloc_);                                                              \
    if (mnew_ > -INFINITY) {                                                                          \
      const float al_ = (m > -INFINITY) ? expf(m - mnew_) : 0.f;                                      \
      l *= al_; o.x *= al_; o.y *= al_; o.z *= al_; o.w *= al_;                                       \
      _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                \
        const float p_ = expf(s_[i] - mnew_);                                                         \
        l += p_;                                                                                      \
        o.x = fmaf(p_, vv[BUF][i].x, o.x); o.y = fmaf(p_, vv[BUF][i].y, o.y);                         \
        o.z = fmaf(p_, vv[BUF][i].z, o.z); o.w = fmaf(p_, vv[BUF][i].w, o.w);                         \
      }                                                                                               \
      m = mnew_;                                                                                      \
    }                                                                                                 \
  }

  ATTN_ISSUE(0, 0)
  for (int pg = 0; pg < npages; pg += 2) {
    ATTN_ISSUE(1, pg + 1)
    ATTN_FOLD(0, pg)
    ATTN_ISSUE(0, pg + 2)
    ATTN_FOLD(1, pg + 1)
  }
#undef ATTN_ISSUE
#undef ATTN_FOLD
  // the KPI key-row groups of the wave share m: their (l, o) simply add
  if (LPK == 16) {
    l += xor16_f(l);
    o.x += xor16_f(o.x); o.y += xor16_f(o.y); o.z += xor16_f(o.z); o.w += xor16_f(o.w);
  }
  l += xor32_f(l);
  o.x += xor32_f(o.x); o.y += xor32_f(o.y); o.z += xor32_f(o.z); o.w += xor32_f(o.w);
  if (lane < LPK) *reinterpret_cast<float4*>(&sm[wave][c4]) = o;
  if (lane == 0) { sm[wave][HD] = m; sm[wave][HD + 1] = l; }
  __syncthreads();
  if (threadIdx.x < LPK) {
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) M = fmaxf(M, sm[w][HD]);
    float L = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < NW; ++w) {                                   // fixed wave order: deterministic
      const float mw = sm[w][HD];
      const float f = (mw > -INFINITY) ? expf(mw - M) : 0.f;
      L = fmaf(f, sm[w][HD + 1], L);
      acc.x = fmaf(f, sm[w][c4 + 0], acc.x);
      acc.y = fmaf(f, sm[w][c4 + 1], acc.y);
      acc.z = fmaf(f, sm[w][c4 + 2], acc.z);
      acc.w = fmaf(f, sm[w][c4 + 3], acc.w);
    }
    const float inv = 1.0f / L;
    const int e = h * HD + c4;
    float* dst = a.out_tiled ? out + SSRHIP_TILED(r, e) : out + (size_t)r * H * HD + e;
    *reinterpret_cast<float4*>(dst) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Causal PREFILL attention with K/V tile reuse (flash-style, fp32 on the matrix core). The decode kernel run once per query
// row re-reads a row's whole K/V prefix per row (28,704 workgroups, 97 us per layer for the bench prompt); here a workgroup owns
// 128 consecutive queries of one (sequence, head) — 4 waves x 32 queries — and walks the key tiles (32 keys) up to its diagonal:
//   * the K and V tiles are fetched ONCE per workgroup, coalesced, from the paged cache (the prefill scattered them there just
//     before) into LDS, one tile ahead in registers, and shared by the 4 waves;
//   * S^T = K . Q^T on v_mfma_f32_32x32x2_f32 (keys as M, queries as N, head_dim as K; the wave's Q slice lives in registers),
//     so that a lane holds 16 scores of ITS query column: the running max / sum of the online softmax are per-lane scalars
//     (one xor-32 exchange between the two half-waves), and the probabilities sit exactly where the B operand of the second
//     product wants them:  O^T = V^T . P^T  (head_dim as M, queries as N, keys as K) — no shuffle, no LDS round trip for P
//     (the k order of the second product follows the accumulator's row order; V is read from LDS in that order);
//   * the output row is normalised and stored straight into the [R][D] buffer the out-projection GEMM reads: no partials.
// Exact fp32 (v_mfma_f32 is an fmaf chain). Replaces F.scaled_dot_product_attention with the causal mask of ssr.py:227-255 for
// the prompt rows (activation.py:634).
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int HD>
__global__ __launch_bounds__(256) void attn_prefill_kernel(const ssrhip_attn_args a, const int32_t* __restrict__ seq_start, float* __restrict__ out) {
  constexpr int KT = 32, LDK = HD + 4, F4 = HD / 4, NLD = KT * F4 / 256;     // float4 loads per thread per tile (HD 128: 4, 64: 2)
  constexpr int NJ = HD / 8, NMB = HD / 32;
  __shared__ __attribute__((aligned(16))) float Ks[KT * LDK];
  __shared__ __attribute__((aligned(16))) float Vs[KT * LDK];
  const int qb = blockIdx.x, h = blockIdx.y, seq = blockIdx.z;
  const int r0 = seq_start[seq], S = seq_start[seq + 1] - r0;
  if (qb * 128 >= S) return;                                               // uniform
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int H = a.kv.n_head, D = H * HD;
  const int qstride = a.q_stride ? a.q_stride : D;
  const int q0 = qb * 128 + wave * 32;                                     // this wave's first query position
  const int qi = q0 + li;                                                  // this lane's query (column of both products)
  const int last_q_wg = min(qb * 128 + 127, S - 1);
  const int ntile = last_q_wg / KT + 1;                                    // key tiles the workgroup walks
  const int my_last = (q0 < S) ? min(q0 + 31, S - 1) / KT : -1;            // last tile this wave needs (-1: no live query)

  // Q slice: lane (query li, half lh) holds Q[qi][8j + 4lh .. +3], j = 0..NJ-1 — the k order both operands of S^T use
  float4 qr[NJ];
  {
    const float* qp = a.q + (size_t)(r0 + min(qi, S - 1)) * qstride + h * HD + 4 * lh;
#pragma unroll
    for (int j = 0; j < NJ; ++j) qr[j] = ld4(qp + 8 * j);
  }
  // segment `seq` of the flattened rows belongs to cache sequence row_seq[first row] (a prefill of SOME of the engine's rows while the
  // others keep decoding); without row_seq segment i is sequence i
  const int32_t* tab = a.kv.table + (size_t)(a.row_seq ? a.row_seq[r0] : seq) * a.kv.max_pages;
  const size_t page_stride = (size_t)a.kv.n_layer * 2 * H * SSRHIP_PAGE * HD;
  const float* pool = a.kv.pool + ((size_t)a.layer * 2 * H + h) * SSRHIP_PAGE * HD;
  const size_t v_off = (size_t)H * SSRHIP_PAGE * HD;

  float4 kreg[NLD], vreg[NLD];
  auto gload = [&](int kt) {                                               // tile kt -> registers (rows past S: clamped, masked later)
    const int key0 = kt * KT;
    const float* base = pool + (size_t)tab[key0 / SSRHIP_PAGE] * page_stride + (size_t)(key0 % SSRHIP_PAGE) * HD;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = i * 256 + t, row = idx / F4, c4 = idx % F4;
      const int rr = min(row, S - 1 - key0);                               // key0 + row < S (>= 0: key0 <= last query < S)
      kreg[i] = ld4(base + (size_t)rr * HD + c4 * 4);
      vreg[i] = ld4(base + v_off + (size_t)rr * HD + c4 * 4);
    }
  };
  f32x16 accO[NMB];
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) accO[mb][r] = 0.f;
  float m = -INFINITY, l = 0.f;

  gload(0);
  for (int kt = 0; kt < ntile; ++kt) {
    __syncthreads();                                                       // the previous tile is fully consumed
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = i * 256 + t, row = idx / F4, c4 = idx % F4;
      *reinterpret_cast<float4*>(Ks + row * LDK + c4 * 4) = kreg[i];
      *reinterpret_cast<float4*>(Vs + row * LDK + c4 * 4) = vreg[i];
    }
    __syncthreads();
    if (kt + 1 < ntile) gload(kt + 1);                                     // next tile under the MFMAs
    if (kt > my_last) continue;                                            // beyond this wave's diagonal (uniform per wave)
    // ---- S^T[key][query] = sum_k K[key][k] Q[query][k]
    f32x16 accS;
#pragma unroll
    for (int r = 0; r < 16; ++r) accS[r] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float4 kf = *reinterpret_cast<const float4*>(Ks + li * LDK + 8 * j + 4 * lh);
      accS = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qr[j].x, accS, 0, 0, 0);
      accS = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qr[j].y, accS, 0, 0, 0);
      accS = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qr[j].z, accS, 0, 0, 0);
      accS = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qr[j].w, accS, 0, 0, 0);
    }
    // ---- online softmax down the lane's query column: register r holds key kt*32 + (r&3) + 8(r>>2) + 4lh
    float mloc = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * KT + (r & 3) + 8 * (r >> 2) + 4 * lh;
      accS[r] = (key <= qi && key < S) ? accS[r] * a.scale : -INFINITY;     // causal: a query sees keys at positions <= its own
      mloc = fmaxf(mloc, accS[r]);
    }
    mloc = fmaxf(mloc, xor32_f(mloc));
    const float mnew = fmaxf(m, mloc);                                     // finite for every live query: key 0 is always visible
    const float alpha = (m > -INFINITY) ? expf(m - mnew) : 0.f;
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      accS[r] = (mnew > -INFINITY) ? expf(accS[r] - mnew) : 0.f;
      lsum += accS[r];
    }
    lsum += xor32_f(lsum);
    l = l * alpha + lsum;
    m = mnew;
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) accO[mb][r] *= alpha;
    // ---- O^T[d][query] += sum_key V[key][d] P[key][query]: step s consumes the keys of accumulator register s (both halves)
#pragma unroll
    for (int sidx = 0; sidx < 16; ++sidx) {
      const float* vrow = Vs + ((sidx & 3) + 8 * (sidx >> 2) + 4 * lh) * LDK + li;
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb) accO[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[mb * 32], accS[sidx], accO[mb], 0, 0, 0);
    }
  }
  if (qi >= S) return;
  const float inv = 1.0f / l;
  float* orow = out + (size_t)(r0 + qi) * D + h * HD;
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
    for (int g = 0; g < 4; ++g)                                            // registers 4g..4g+3 = head_dim mb*32 + 8g + 4lh + 0..3
      *reinterpret_cast<float4*>(orow + mb * 32 + 8 * g + 4 * lh) =
          make_float4(accO[mb][4 * g] * inv, accO[mb][4 * g + 1] * inv, accO[mb][4 * g + 2] * inv, accO[mb][4 * g + 3] * inv);
}

int check(const ssrhip_attn_args* a, const char* who) {
  SSR_REQUIRE(a && a->q && a->kv.pool && a->kv.table && a->row_len, "%s: null argument", who);
  SSR_REQUIRE(a->kv.head_dim == 64 || a->kv.head_dim == 128, "%s: head_dim %d not in {64,128}", who, a->kv.head_dim);
  SSR_REQUIRE(a->R > 0 && a->max_splits > 0 && a->max_splits <= a->kv.max_pages, "%s: bad R/max_splits", who);
  return 0;
}

}  // namespace

// gridDim.z / gridDim.y carry the row index and are limited to 65535: longer row lists (a 16-row prefill has ~4k rows per
// sequence) are issued in slices of at most 65535 rows, each slice seeing its own sub-arrays.
static ssrhip_attn_args row_slice(const ssrhip_attn_args& a, int r0, int n) {
  ssrhip_attn_args s = a;
  const size_t H = a.kv.n_head, HD = a.kv.head_dim;
  s.q = a.q + (size_t)r0 * (a.q_stride ? a.q_stride : H * HD);
  s.row_seq = a.row_seq + r0;
  s.row_len = a.row_len + r0;
  s.part_o = a.part_o + (size_t)r0 * H * a.max_splits * HD;
  s.part_ml = a.part_ml + (size_t)r0 * H * a.max_splits * 2;
  s.R = n;
  return s;
}
enum { MAX_GRID_ROWS = 65535 };

extern "C" int ssrhip_attn_prefill(const ssrhip_attn_args* a, const int32_t* seq_start, int32_t n_seq, int32_t max_len, float* out,
                                   ssrhip_stream_t stream) {
  SSR_REQUIRE(a && a->q && a->kv.pool && a->kv.table && seq_start && out, "ssrhip_attn_prefill: null argument");
  SSR_REQUIRE(a->kv.head_dim == 64 || a->kv.head_dim == 128, "ssrhip_attn_prefill: head_dim %d not in {64,128}", a->kv.head_dim);
  SSR_REQUIRE(n_seq > 0 && n_seq <= 65535 && max_len > 0 && a->kv.n_head <= 65535, "ssrhip_attn_prefill: bad n_seq / max_len");
  SSR_REQUIRE((a->q_stride ? a->q_stride : a->kv.n_head * a->kv.head_dim) % 4 == 0, "ssrhip_attn_prefill: q_stride must be a multiple of 4");
  dim3 grid((max_len + 127) / 128, a->kv.n_head, n_seq);
  if (a->kv.head_dim == 128) hipLaunchKernelGGL(attn_prefill_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream, *a, seq_start, out);
  else hipLaunchKernelGGL(attn_prefill_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, *a, seq_start, out);
  SSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int ssrhip_attn_rows(const ssrhip_attn_args* a, float* out, ssrhip_stream_t stream) {
  if (int e = check(a, "ssrhip_attn_rows")) return e;
  SSR_REQUIRE(out && out != a->q, "ssrhip_attn_rows: out is null or aliases q");
  SSR_REQUIRE(!a->out_tiled || a->R <= 16, "ssrhip_attn_rows: tiled output needs R <= 16");
  SSR_REQUIRE(a->R <= MAX_GRID_ROWS, "ssrhip_attn_rows: R too large");
  SSR_REQUIRE(a->kv.max_pages <= ATTN_ROWS_MAX_PAGES, "ssrhip_attn_rows: more than %d pages per row", ATTN_ROWS_MAX_PAGES);
  dim3 grid(a->kv.n_head, a->R);
  if (a->kv.head_dim == 128) hipLaunchKernelGGL(attn_rows_kernel<128>, grid, dim3(512), 0, (hipStream_t)stream, *a, out);
  else hipLaunchKernelGGL(attn_rows_kernel<64>, grid, dim3(512), 0, (hipStream_t)stream, *a, out);
  SSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int ssrhip_attn_decode(const ssrhip_attn_args* a, ssrhip_stream_t stream) {
  if (int e = check(a, "ssrhip_attn_decode")) return e;
  SSR_REQUIRE(a->part_o && a->part_ml, "ssrhip_attn_decode: null partial buffers");
  SSR_REQUIRE(a->R <= MAX_GRID_ROWS || a->row_seq, "ssrhip_attn_decode: more than %d rows need an explicit row_seq", MAX_GRID_ROWS);
  for (int r0 = 0; r0 < a->R; r0 += MAX_GRID_ROWS) {
    const int n = min(a->R - r0, (int)MAX_GRID_ROWS);
    const ssrhip_attn_args s = a->R <= MAX_GRID_ROWS ? *a : row_slice(*a, r0, n);
    int hf = 1;                                                      // SSRHIP_ATTN_HEAD_FASTEST=0: round 4's grid order (A/B knob, read per call)
    if (const char* e = getenv("SSRHIP_ATTN_HEAD_FASTEST")) hf = e[0] != '0';
    dim3 grid(hf ? s.kv.n_head : s.max_splits, hf ? s.max_splits : s.kv.n_head, n);
    if (s.kv.head_dim == 128) {
      if (s.row_seq) hipLaunchKernelGGL((attn_decode_kernel<128, true>), grid, dim3(256), 0, (hipStream_t)stream, s, hf);
      else {
        int vat = 8;                                                 // SSRHIP_ATTN_VAT = -1 | 4 | 12: A/B knob (profiles/r05_microbench/decode_ab_attn_vat.log), read per call
        if (const char* e = getenv("SSRHIP_ATTN_VAT")) vat = atoi(e);
        if (vat == 4) hipLaunchKernelGGL((attn_decode_kernel<128, false, 4>), grid, dim3(256), 0, (hipStream_t)stream, s, hf);
        else if (vat == 12) hipLaunchKernelGGL((attn_decode_kernel<128, false, 12>), grid, dim3(256), 0, (hipStream_t)stream, s, hf);
        else if (vat < 0) hipLaunchKernelGGL((attn_decode_kernel<128, false>), grid, dim3(256), 0, (hipStream_t)stream, s, hf);
        else hipLaunchKernelGGL((attn_decode_kernel<128, false, 8>), grid, dim3(256), 0, (hipStream_t)stream, s, hf);
      }
    } else {
      if (s.row_seq) hipLaunchKernelGGL((attn_decode_kernel<64, true>), grid, dim3(256), 0, (hipStream_t)stream, s, hf);
      else hipLaunchKernelGGL((attn_decode_kernel<64, false>), grid, dim3(256), 0, (hipStream_t)stream, s, hf);
    }
    SSR_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int ssrhip_attn_combine(const ssrhip_attn_args* a, float* out, ssrhip_stream_t stream) {
  if (int e = check(a, "ssrhip_attn_combine")) return e;
  SSR_REQUIRE(out && a->part_o && a->part_ml, "ssrhip_attn_combine: out or the partial buffers are null");
  SSR_REQUIRE(!a->out_tiled || a->R <= 16, "ssrhip_attn_combine: tiled output needs R <= 16");
  const size_t D = (size_t)a->kv.n_head * a->kv.head_dim;
  for (int r0 = 0; r0 < a->R; r0 += MAX_GRID_ROWS) {
    const int n = min(a->R - r0, (int)MAX_GRID_ROWS);
    ssrhip_attn_args s = *a;
    float* o = out;
    if (a->R > MAX_GRID_ROWS) {            // the combine never reads q / row_seq: only the per-row arrays move
      s.row_len = a->row_len + r0;
      s.part_o = a->part_o + (size_t)r0 * a->kv.n_head * a->max_splits * a->kv.head_dim;
      s.part_ml = a->part_ml + (size_t)r0 * a->kv.n_head * a->max_splits * 2;
      s.R = n;
      o = out + (size_t)r0 * D;
    }
    if (s.kv.head_dim == 128) hipLaunchKernelGGL(attn_combine_kernel<128>, dim3((s.kv.n_head + 1) / 2, n), dim3(64), 0, (hipStream_t)stream, s, o);
    else hipLaunchKernelGGL(attn_combine_kernel<64>, dim3((s.kv.n_head + 3) / 4, n), dim3(64), 0, (hipStream_t)stream, s, o);
    SSR_LAUNCH_CHECK();
  }
  return 0;
}
