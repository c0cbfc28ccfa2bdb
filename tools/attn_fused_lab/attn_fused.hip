// attn_fused.hip — single-query paged attention AND the out-projection GEMV of a decode layer in ONE launch (<= 4 rows), gfx950.
//
// Why: at 2 rows the two launches `attn_decode_kernel` -> `gemv_seg_kernel<B, PRO_ATTN_COMBINE>` take 6.7 + 7.1 us per layer for
// 10 MB of K/V + 16.8 MB of W_o, where ONE streaming launch of 27 MB costs ~6.3 us on this GPU (launch-chain floor 2.6 us + bytes /
// 7.3 TB/s). The boundary between them cannot overlap anything: W_o does not depend on the attention, yet its first
// byte is requested only after the attention kernel has drained. Here
//   phase 0  every workgroup (one per CU) requests its WHOLE slice of W_o (8 rows x 2048 floats = 64 KB per CU, 32 VGPRs per lane),
//   phase 1  workgroups 0 .. n_items-1 each compute one (page, head, row) attention partial — the arithmetic of attn_decode_kernel,
//            operation for operation — and publish it: write-through stores, release fence, one relaxed atomic on one of 8 sharded
//            arrival counters,
//   phase 2  one wave per workgroup polls the 8 counters (relaxed agent-scope loads + s_sleep, bounded) until all items have arrived,
//            acquire fence; the last workgroup through re-zeroes the counters (the block is reusable by the next launch on the stream),
//   phase 3  split-KV merge -> x in LDS -> the two (row, segment) units of every wave against the W_o registers that landed long ago
//            -> bias + residual: the arithmetic of gemv_seg_kernel<B, PRO_ATTN_COMBINE>, operation for operation.
// Results are bit-identical to the two-launch path (tests/test_gpu_kernels.py::test_attn_outproj_fused_equals_two_launches), so the
// parity evidence of the decode step carries over unchanged. The in-launch hand-off is the guide's "fanin" + "allgather" pattern
// (MI355X_MICROARCH.md price list): it costs more than a kernel boundary by itself; it pays here because 27 MB of loads hide under it.
// All 256 workgroups are co-resident (one 512-thread workgroup per CU), items are taken by the FIRST n_items workgroups, and nothing
// in phase 1 waits: no deadlock however the dispatcher orders them. The spin is bounded; a give-up is recorded in the last sync word.
// Replaces activation.py:634 (F.scaled_dot_product_attention, tgt_len == 1) + :637 (out_proj) + transformer.py:328 (residual).
#include <stdlib.h>
#include <algorithm>
using std::min;
using std::max;
#include "common.h"

namespace {

struct FusedK {
  ssrhip_attn_args at;
  const float* W;        // [N][K] out-projection weight, N == K == n_head * head_dim == 2048
  const float* bias;     // [N] or NULL
  float* y;              // [B][y_stride] residual stream: y += out_proj(attn)
  int32_t y_stride;
  int32_t N, K;
  int32_t rows_per, rows_rem;   // N / grid and N % grid
  int32_t flags;         // A/B knobs (SSRHIP_FUSED_FLAGS): 1 = acquire fence after the wait, 2 = release fence (L2 write-back) before the arrival
  long long* prof;       // debug (ssrhip_debug_fused_prof): 8 time stamps per workgroup, or NULL
  int32_t* sync;         // [SYNC_WORDS]: arrival counter replicas + pass counter (zero on entry AND on exit) + sticky give-up flag (last word)
};

constexpr int FSEG = 1024;       // floats per (row, segment) unit
constexpr int F_TH = 512, F_NW = 8;
template <int B> struct FusedCS { static constexpr int v = (B <= 2) ? 6 : 2; };   // pages whose partial outputs are requested together (= gemv_seg_kernel's SegCS)
constexpr int SPIN_LIMIT = 1 << 22;
// Arrival counters: NREP replicas, one per 256-byte block of `sync` (different memory channels). Every arriving item adds 1 to ALL of
// them (one 16-lane atomic instruction, no return); a waiting workgroup polls ONLY replica blockIdx % NREP. With a single counter line
// all 256 pollers and all arrivals queue on one channel: the last arrival became visible 4-7 us late (tools/fused_prof.py).
constexpr int NREP = 16, REP_STRIDE = 64;          // ints
constexpr int SYNC_PASS = NREP * REP_STRIDE, SYNC_GIVEUP = SYNC_PASS + 63, SYNC_WORDS = NREP * REP_STRIDE + 64;

__device__ __forceinline__ int ld_agent(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent_f(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int B, int HD>
__global__ __launch_bounds__(F_TH, 2) void attn_outproj_kernel(const FusedK p) {
  constexpr int LPK = HD / 4;         // lanes per key row
  constexpr int KPI = 64 / LPK;       // key rows per wave-instruction
  constexpr int NI = 32 / KPI;        // load instructions for a wave's 32 keys
  constexpr int S = 2, SH = 1;        // K == 2048: two segments per row
  constexpr int F_CS = FusedCS<B>::v;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const ssrhip_attn_args& a = p.at;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int H = a.kv.n_head, K = p.K, G = gridDim.x, MS = a.max_splits;
  const int r0 = (int)blockIdx.x * p.rows_per + min((int)blockIdx.x, p.rows_rem);     // no division in the prologue (see gemv.hip)
  const int nrows = p.rows_per + ((int)blockIdx.x < p.rows_rem ? 1 : 0), nu = nrows * S;
  const int rows_max = p.rows_per + (p.rows_rem ? 1 : 0);
  const int seg = wave & (S - 1);
  float* part = smem;                                   // [rows_max][S][B] partial sums of the GEMV
  float* xs = smem + rows_max * S * B;                  // [B][K] merged attention output
  float* wtab = xs + B * K;                             // [B*H][MS] merge weights
  float* sm = wtab + B * H * MS;                        // [4][HD + 4] 4-wave merge of one attention item

#define FSTAMP(i) do { if (p.prof && t == 0) p.prof[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
  FSTAMP(0);
  // ---- the row lengths first (one scalar round trip): they decide which workgroups hold an attention item
  // row_len and the page table do not change during the launch: constant address space = scalar loads (a plain load is a VECTOR load
  // here — the compiler cannot rule out an alias with the stores above — and costs a second, serial round trip before the K/V requests)
  typedef const int32_t __attribute__((address_space(4))) cint;
  cint* c_row_len = (cint*)(uintptr_t)a.row_len;
  cint* c_table = (cint*)(uintptr_t)a.kv.table;
  // the page of this workgroup's first item slot: its address needs no row length (clamped into the table) — same round trip as row_len
  const int page_first = c_table[(size_t)(((int)blockIdx.x / H) % B) * a.kv.max_pages + min(((int)blockIdx.x / H) / B, a.kv.max_pages - 1)];
  int ns[B], lens[B];
  int ns_max = 0;
#pragma unroll
  for (int b = 0; b < B; ++b) { lens[b] = c_row_len[b]; ns[b] = (lens[b] + SSRHIP_PAGE - 1) / SSRHIP_PAGE; ns_max = max(ns_max, ns[b]); }
  asm volatile("; page_first and the row lengths arrive together" :: "s"(page_first), "s"(lens[0]));   // keeps hipcc from sinking the table load into its use
  const int n_items = ns_max * B * H;                          // item slots (rows shorter than the longest leave empty ones)

  // attention item -> (head, page, row); K/V/q requests of waves 0..3 (the arithmetic of attn_decode_kernel<HD>, 4 waves x 32 keys)
  const int sub = lane / LPK, c4 = (lane % LPK) * 4;
  float4 kk[NI], vv[NI], q4;
  // item -> (head, row, page) by index arithmetic alone (head fastest, then row, then page): the page-table entry's address does not
  // depend on the row lengths, so row_len, table entry and q travel in ONE round trip. Slots beyond a row's last page arrive empty.
  auto item_coords = [&](int item, int& h, int& split, int& r, int& len) {
    h = item % H;
    const int ps = item / H;
    r = ps % B;
    split = ps / B;
    len = lens[0];
#pragma unroll
    for (int b = 1; b < B; ++b) len = (r == b) ? lens[b] : len;
  };
  auto attn_issue = [&](int item, int page) {                                 // page = table[r][split]
    int h, split, r, len;
    item_coords(item, h, split, r, len);
    const int base = split * SSRHIP_PAGE;
    q4 = ld4(a.q + (size_t)r * (a.q_stride ? a.q_stride : H * HD) + h * HD + c4);
    const float* kp = a.kv.pool + ((((size_t)page * a.kv.n_layer + a.layer) * 2 + 0) * H + h) * SSRHIP_PAGE * HD;
    const float* vp = kp + (size_t)H * SSRHIP_PAGE * HD;
    const int jmax = max(min(len - base, SSRHIP_PAGE) - 1, 0);               // an empty slot (base >= len) loads key 0 and drops it
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int j = min(wave * 32 + i * KPI + sub, jmax);
      kk[i] = ld_nt(kp + (size_t)j * HD + c4);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int j = min(wave * 32 + i * KPI + sub, jmax);
      vv[i] = ld_nt(vp + (size_t)j * HD + c4);
    }
  };
  auto attn_finish = [&](int item) {
    int h, split, r, len;
    item_coords(item, h, split, r, len);
    const int base = split * SSRHIP_PAGE;
    if (base >= len) {                                          // uniform: nothing to compute, the slot just arrives
      if (t < NREP) __hip_atomic_fetch_add(p.sync + t * REP_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (wave < 4) {
      float s[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) s[i] = dot4(q4, kk[i], 0.f);
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
          const float pe = expf(s[i] - m);
          l += pe;
          o4.x = fmaf(pe, vv[i].x, o4.x);
          o4.y = fmaf(pe, vv[i].y, o4.y);
          o4.z = fmaf(pe, vv[i].z, o4.z);
          o4.w = fmaf(pe, vv[i].w, o4.w);
        }
      }
      if (LPK == 16) {
        l += xor16_f(l);
        o4.x += xor16_f(o4.x); o4.y += xor16_f(o4.y); o4.z += xor16_f(o4.z); o4.w += xor16_f(o4.w);
      }
      l += xor32_f(l);
      o4.x += xor32_f(o4.x); o4.y += xor32_f(o4.y); o4.z += xor32_f(o4.z); o4.w += xor32_f(o4.w);
      if (lane < LPK) *reinterpret_cast<float4*>(&sm[wave * (HD + 4) + c4]) = o4;
      if (lane == 0) { sm[wave * (HD + 4) + HD] = m; sm[wave * (HD + 4) + HD + 1] = l; }
    }
    __syncthreads();
    if (wave == 0) {
      if (lane < LPK) {
        const float m0 = sm[0 * (HD + 4) + HD], m1 = sm[1 * (HD + 4) + HD], m2 = sm[2 * (HD + 4) + HD], m3 = sm[3 * (HD + 4) + HD];
        const float M = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        float L = 0.f;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float mw = sm[w * (HD + 4) + HD];
          const float f = (mw > -INFINITY) ? expf(mw - M) : 0.f;
          L = fmaf(f, sm[w * (HD + 4) + HD + 1], L);
          acc.x = fmaf(f, sm[w * (HD + 4) + c4 + 0], acc.x);
          acc.y = fmaf(f, sm[w * (HD + 4) + c4 + 1], acc.y);
          acc.z = fmaf(f, sm[w * (HD + 4) + c4 + 2], acc.z);
          acc.w = fmaf(f, sm[w * (HD + 4) + c4 + 3], acc.w);
        }
        const size_t pi = ((size_t)r * H + h) * MS + split;
        float* po = a.part_o + pi * HD + c4;
        st_agent(po + 0, acc.x); st_agent(po + 1, acc.y); st_agent(po + 2, acc.z); st_agent(po + 3, acc.w);   // write-through (sc1)
        if (lane == 0) { st_agent(a.part_ml + pi * 2, M); st_agent(a.part_ml + pi * 2 + 1, L); }
      }
      // publish: the partial went out as write-through (sc1) stores; once they are acknowledged (vmcnt drained) the arrival may become
      // visible (guide "handoff-flag": sc1 payload -> asm vmcnt(0) -> flag). No L2 write-back (buffer_wbl2): nothing else is dirty here.
      if (p.flags & 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane < NREP) __hip_atomic_fetch_add(p.sync + lane * REP_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                                          // sm is reused by the workgroup's next item
  };

  // ---- phase 0/1: a workgroup that holds an item requests its K/V FIRST (loads return in order), then — every workgroup — the
  // WHOLE W_o slice, unconditional (clamped to its last unit), non-temporal: it lands while the attention phase and the hand-off run
  const bool first = (int)blockIdx.x < n_items;                // uniform
  if (first && wave < 4) attn_issue(blockIdx.x, page_first);
  const float* Wg = p.W + (size_t)r0 * K + seg * FSEG + lane * 4;
  float4 wa[4], wb[4];
  const int ua = wave, ub = wave + F_NW;
  {
    const int ca = min(ua, nu - 1) >> SH, cb = min(ub, nu - 1) >> SH;
#pragma unroll
    for (int i = 0; i < 4; ++i) wa[i] = ld_nt(Wg + (size_t)ca * K + i * 256);
#pragma unroll
    for (int i = 0; i < 4; ++i) wb[i] = ld_nt(Wg + (size_t)cb * K + i * 256);
  }
  const int bfin = t % B, rfin = min(t / B, nrows - 1), nfin = r0 + rfin;
  const float ebias = p.bias ? p.bias[nfin] : 0.f;
  const float eres = p.y[(size_t)bfin * p.y_stride + nfin];
  FSTAMP(1);
  if (first) attn_finish(blockIdx.x);
  for (int item = blockIdx.x + G; item < n_items; item += G) {   // contexts with more items than workgroups (> 8 pages per row at 2 rows)
    if (wave < 4) attn_issue(item, c_table[(size_t)((item / H) % B) * a.kv.max_pages + (item / H) / B]);   // split < ns_max <= max_pages
    attn_finish(item);
  }

  // ---- phase 2: wait until every item of this launch has arrived
  int passed = -1;
  FSTAMP(2);
  if (wave == 0) {
    const int32_t* mine = p.sync + ((int)blockIdx.x % NREP) * REP_STRIDE;
    int spins = 0;
    while (true) {
      const int v = __builtin_amdgcn_readfirstlane(ld_agent(mine));
      if (v >= n_items) break;
      if (++spins > SPIN_LIMIT) {
        if (lane == 0) __hip_atomic_store(p.sync + SYNC_GIVEUP, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    // the LAST workgroup to get here puts the counters back to zero for the next launch that uses this block (nobody polls any
    // more: every other workgroup has already passed). The returning atomic's latency hides under phase 3; its result is used at the end.
    if (lane == 0) passed = __hip_atomic_fetch_add(p.sync + SYNC_PASS, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  FSTAMP(3);
  // No cache invalidate here: the launch's own kernel-start acquire emptied L1 / this XCD's L2, and nobody touches a part_o line between
  // that and this point (the producers' stores are write-through), so the first read of a line after the wait comes from memory; the
  // (m, l) pairs — 8-byte pieces of lines that several producers share — are read with agent-scope (sc1) loads. 2,048 waves each
  // invalidating the XCD's L2 here cost 15 us per launch (tools/fused_prof.py); flag bit 0 restores it for A/B.
  if (p.flags & 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");

  // ---- phase 3: split-KV merge (gemv_seg_kernel<B, PRO_ATTN_COMBINE>, K == 2048: thread t owns float4 column 4t of every row)
  {
    float2 cml[F_CS];
    float4 co[B][F_CS];
    const int tt = t % (B * H);
    const float* ml = a.part_ml + (size_t)tt * MS * 2;
#pragma unroll
    for (int i = 0; i < F_CS; ++i) {
      const unsigned long long raw = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(ml + 2 * min(i, MS - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      cml[i] = make_float2(__uint_as_float((unsigned)raw), __uint_as_float((unsigned)(raw >> 32)));
    }
    const int e = t * 4, h = e / HD, d = e % HD;
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const float* po = a.part_o + (((size_t)b * H + h) * MS) * HD + d;
#pragma unroll
      for (int s2 = 0; s2 < F_CS; ++s2) co[b][s2] = ld4(po + (size_t)min(s2, MS - 1) * HD);
    }
    if (t < B * H) {
      int n = ns[0];                                            // ns[t / H] as a select chain: a dynamically indexed array lives in scratch
#pragma unroll
      for (int b = 1; b < B; ++b) n = (t / H == b) ? ns[b] : n;
      float M = -INFINITY;
#pragma unroll
      for (int i = 0; i < F_CS; ++i)
        if (i < n) M = fmaxf(M, cml[i].x);
      for (int s2 = F_CS; s2 < n; ++s2) M = fmaxf(M, ld_agent_f(ml + 2 * s2));
      float den = 0.f;
#pragma unroll
      for (int i = 0; i < F_CS; ++i)
        if (i < n) den = fmaf(expf(cml[i].x - M), cml[i].y, den);
      for (int s2 = F_CS; s2 < n; ++s2) den = fmaf(expf(ld_agent_f(ml + 2 * s2) - M), ld_agent_f(ml + 2 * s2 + 1), den);
      const float inv = 1.0f / den;
#pragma unroll
      for (int i = 0; i < F_CS; ++i)
        if (i < n) wtab[t * MS + i] = expf(cml[i].x - M) * inv;
      for (int s2 = F_CS; s2 < n; ++s2) wtab[t * MS + s2] = expf(ld_agent_f(ml + 2 * s2) - M) * inv;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const float* w = wtab + (b * H + h) * MS;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int s2 = 0; s2 < F_CS; ++s2) {
        const bool in = s2 < ns[b];
        const float ws = in ? w[s2] : 0.f;
        acc.x = fmaf(ws, in ? co[b][s2].x : 0.f, acc.x);
        acc.y = fmaf(ws, in ? co[b][s2].y : 0.f, acc.y);
        acc.z = fmaf(ws, in ? co[b][s2].z : 0.f, acc.z);
        acc.w = fmaf(ws, in ? co[b][s2].w : 0.f, acc.w);
      }
      const float* po = a.part_o + (((size_t)b * H + h) * MS) * HD + d;
      for (int s2 = F_CS; s2 < ns[b]; ++s2) {
        const float ws = w[s2];
        const float4 o = ld4(po + (size_t)s2 * HD);
        acc.x = fmaf(ws, o.x, acc.x);
        acc.y = fmaf(ws, o.y, acc.y);
        acc.z = fmaf(ws, o.z, acc.z);
        acc.w = fmaf(ws, o.w, acc.w);
      }
      *reinterpret_cast<float4*>(xs + b * K + e) = acc;
    }
    __syncthreads();
  }
  FSTAMP(4);
  // ---- the GEMV: this wave's two units against the registers requested in phase 0
  float4 xr[B][4];
#pragma unroll
  for (int b = 0; b < B; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) xr[b][i] = *reinterpret_cast<const float4*>(xs + b * K + seg * FSEG + (i * 64 + lane) * 4);
  auto unit = [&](const float4 (&w)[4], int u) {
    float acc[B][2];
#pragma unroll
    for (int b = 0; b < B; ++b) acc[b][0] = acc[b][1] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int b = 0; b < B; ++b) acc[b][i & 1] = dot4(w[i], xr[b][i], acc[b][i & 1]);
    float mine = 0.f;
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const float sum = wave_sum(acc[b][0] + acc[b][1]);
      if (lane == b) mine = sum;
    }
    if (lane < B) part[u * B + lane] = mine;
  };
  if (ua < nu) unit(wa, ua);
  if (ub < nu) unit(wb, ub);
  __syncthreads();
  if (t < nrows * B) {
    float v = 0.f;
    for (int s2 = 0; s2 < S; ++s2) v += part[(rfin * S + s2) * B + bfin];
    v += ebias;
    p.y[(size_t)bfin * p.y_stride + nfin] = eres + v;
  }
  FSTAMP(5);
  if (passed == G - 1) {                                       // wave 0, lane 0 of the last workgroup
#pragma unroll
    for (int i = 0; i < NREP; ++i) __hip_atomic_store(p.sync + i * REP_STRIDE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p.sync + SYNC_PASS, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int g_fused_cu = 0;
long long* g_fused_prof = nullptr;

template <int B>
int launch(const FusedK& p, int G, size_t smem, hipStream_t s) {
  static ssr_once_per_device once64, once128;
  if (p.at.kv.head_dim == 128) {
    if (smem > 48 * 1024 && once128.need()) SSR_HIP(hipFuncSetAttribute((const void*)attn_outproj_kernel<B, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipLaunchKernelGGL((attn_outproj_kernel<B, 128>), dim3(G), dim3(F_TH), smem, s, p);
  } else {
    if (smem > 48 * 1024 && once64.need()) SSR_HIP(hipFuncSetAttribute((const void*)attn_outproj_kernel<B, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipLaunchKernelGGL((attn_outproj_kernel<B, 64>), dim3(G), dim3(F_TH), smem, s, p);
  }
  SSR_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// debug hook (not part of the ABI in include/ssrhip.h; tools/fused_prof.py): 8 wall_clock64 stamps per workgroup of every later launch
extern "C" void ssrhip_debug_fused_prof(void* dev_ptr) { g_fused_prof = (long long*)dev_ptr; }

extern "C" int ssrhip_attn_outproj_sync_words(void) { return SYNC_WORDS; }

extern "C" int ssrhip_attn_outproj_supported(const ssrhip_attn_args* a, const ssrhip_gemv_args* g) {
  if (!a || !g) return 0;
  const int H = a->kv.n_head, hd = a->kv.head_dim;
  if (g->B != 1 && g->B != 2 && g->B != 4) return 0;
  if (a->R != g->B || a->row_seq) return 0;
  if (hd != 64 && hd != 128) return 0;
  if (g->K != 2048 || g->N != g->K || H * hd != g->K || g->groups != 1) return 0;
  if (g->B * H > F_TH) return 0;
  if (g->x_tiled || g->y_tiled || g->w_tiled) return 0;
  if (g->act != SSRHIP_ACT_NONE || g->epi != SSRHIP_EPI_RESIDUAL) return 0;
  return 1;
}

extern "C" int ssrhip_attn_outproj(const ssrhip_attn_args* a, const ssrhip_gemv_args* g, int32_t* sync, ssrhip_stream_t stream) {
  SSR_REQUIRE(a && g && sync, "ssrhip_attn_outproj: null argument");
  SSR_REQUIRE(ssrhip_attn_outproj_supported(a, g), "ssrhip_attn_outproj: shape not supported (needs B in {1,2,4}, N == K == 2048 == n_head*head_dim, "
              "head_dim in {64,128}, residual epilogue, row-major operands)");
  SSR_REQUIRE(a->q && a->kv.pool && a->kv.table && a->row_len && a->part_o && a->part_ml && g->W && g->y, "ssrhip_attn_outproj: null buffer");
  SSR_REQUIRE(a->max_splits > 0 && a->max_splits <= a->kv.max_pages, "ssrhip_attn_outproj: bad max_splits");
  if (g_fused_cu == 0) {
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cu > 0) g_fused_cu = cu;
    else g_fused_cu = 256;
  }
  int G = g_fused_cu;                       // one workgroup per CU: all co-resident (the in-launch hand-off relies on it for progress only, not safety)
  if (G > g->N) G = g->N;
  const int rows_max = (g->N + G - 1) / G;
  SSR_REQUIRE(rows_max * g->B <= F_TH && rows_max * 2 <= 2 * F_NW, "ssrhip_attn_outproj: %d rows per workgroup do not fit two units per wave", rows_max);
  FusedK p;
  p.at = *a;
  p.W = g->W; p.bias = g->bias; p.y = g->y; p.y_stride = g->y_stride; p.N = g->N; p.K = g->K;
  p.rows_per = g->N / G; p.rows_rem = g->N % G;
  p.sync = sync; p.prof = g_fused_prof;
  { const char* e = getenv("SSRHIP_FUSED_FLAGS"); p.flags = e ? atoi(e) : 0; }
  const int H = a->kv.n_head, hd = a->kv.head_dim;
  size_t smem = ((size_t)rows_max * 2 * g->B + (size_t)g->B * g->K + (size_t)g->B * H * a->max_splits + 4 * (hd + 4)) * sizeof(float);
  smem = (smem + 15) / 16 * 16;
  SSR_REQUIRE(smem <= 64 * 1024, "ssrhip_attn_outproj: %zu bytes of LDS", smem);
  hipStream_t s = (hipStream_t)stream;
  switch (g->B) {
    case 1: return launch<1>(p, G, smem, s);
    case 2: return launch<2>(p, G, smem, s);
    default: return launch<4>(p, G, smem, s);
  }
}
