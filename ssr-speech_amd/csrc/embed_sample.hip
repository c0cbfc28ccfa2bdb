// embed_sample.hip — token embedding and the device-side sampler / logit state machine, gfx950.
//
// ssrhip_embed : sum of K codebook embeddings (or one text embedding) + alpha * sinusoidal pe row.
// ssrhip_sample: everything the reference does on the host between `predict_layer` and the next
//   `embed` (models/ssr.py:689-761): CFG combine, special-token edits, eog cascade, silence penalty,
//   temperature / top-k / top-p filtering, one multinomial draw per codebook, stop rules, span hand-over,
//   and (fused) the embedding of the tokens it just chose, i.e. the next step's input row.
//   One workgroup per utterance, wave k owns codebook k; the ~2k logits of a codebook live in VGPRs
//   (<= 34 per lane), top-k / top-p thresholds are found by bisection on the order-preserving integer
//   image of the logits (ballot+popcount for counts, fixed-order DPP wave sums for mass) — no sort, no
//   atomics, bit-reproducible.  No host round trip per step.
#include <stdlib.h>
#include "common.h"

namespace {

__device__ __forceinline__ void embed_row(const ssrhip_embed_args& a, int r, int kind, int pos, const int* tok, int t0, int nt) {
  const int D = a.D;
  const float alpha = kind ? a.alpha_audio : a.alpha_text;
  for (int d = t0 * 4; d < D; d += nt * 4) {
    float4 e;
    if (kind == 0) {
      e = ld4(a.text_emb + (size_t)tok[0] * D + d);
    } else {
      // torch.stack(...).sum(dim=0): sequential sum over codebooks (ssr.py:193-196); the K row loads are
      // issued together (independent HBM/L2 round trips), then added in codebook order
      float4 t[SSRHIP_MAX_CODEBOOKS];
#pragma unroll
      for (int k = 0; k < SSRHIP_MAX_CODEBOOKS; ++k)
        t[k] = (k < a.K) ? ld4(a.audio_emb + ((size_t)k * a.card + tok[k]) * D + d) : make_float4(0.f, 0.f, 0.f, 0.f);
      e = t[0];
#pragma unroll
      for (int k = 1; k < SSRHIP_MAX_CODEBOOKS; ++k) {
        if (k < a.K) { e.x += t[k].x; e.y += t[k].y; e.z += t[k].z; e.w += t[k].w; }
      }
    }
    const float4 p = ld4(a.pe + (size_t)pos * D + d);
    // x * 1.0 + alpha * pe (embedding.py:96): product rounded, then sum rounded (no fma contraction)
    e.x = __fadd_rn(e.x, __fmul_rn(alpha, p.x));
    e.y = __fadd_rn(e.y, __fmul_rn(alpha, p.y));
    e.z = __fadd_rn(e.z, __fmul_rn(alpha, p.z));
    e.w = __fadd_rn(e.w, __fmul_rn(alpha, p.w));
    float* dst = a.out_tiled ? a.out + SSRHIP_TILED(r, d) : a.out + (size_t)r * D + d;
    *reinterpret_cast<float4*>(dst) = e;
  }
}

__global__ __launch_bounds__(256) void embed_kernel(const ssrhip_embed_args a) {
  const int r = blockIdx.x;
  embed_row(a, r, a.kind ? a.kind[r] : 1, a.pos[r], a.tok + (size_t)r * SSRHIP_MAX_CODEBOOKS, threadIdx.x, 256);
}

constexpr int MAXE = 9;     // logits per lane: 4 waves per codebook -> card <= 4*64*9 = 2304
constexpr int WPC = 4;      // waves per codebook
constexpr int SAMPLE_THREADS = 64 * WPC * SSRHIP_MAX_CODEBOOKS;   // 1024

#ifdef SSR_SAMPLE_PROFILE   // tools/sampler_bench.hip: phase time stamps (shader clock) of wave 0
__device__ unsigned long long g_sample_prof[16];
#define STAMP(i) do { if (threadIdx.x == 0) g_sample_prof[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ uint32_t okey(float f) {   // order-preserving float -> uint32
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <class T>
__device__ __forceinline__ T suffix_excl(T v, int lane) {   // sum of v over lanes > lane
  T inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const T t = __shfl_down(inc, o, 64);
    if (lane + o < 64) inc += t;
  }
  return inc - v;
}

// Exact selection without a sort: 4-pass radix select (8 bits per pass) on the order-preserving integer image of the
// logits. The WPC waves of a codebook each fill their own 256-bin histogram in LDS with LDS atomics (integer counts
// for top-k, 2^-40 fixed-point probability mass for top-p: exact integer sums, order-independent => bit-reproducible);
// after a workgroup barrier the codebook's first wave adds the WPC copies, scans the 256 bins from the top and
// publishes (bin, remainder) for the next pass. All 16 waves take part in every barrier.
struct SelShared {
  unsigned long long hist[SSRHIP_MAX_CODEBOOKS][WPC][256];   // counts alias this as unsigned [..][256]
  unsigned long long carry[SSRHIP_MAX_CODEBOOKS];            // count still needed / mass above the current bucket
  unsigned bin[SSRHIP_MAX_CODEBOOKS];
  int found[SSRHIP_MAX_CODEBOOKS];
  // bin_select: the elements of the selected value bin (candidates), their masses, and the published result
  static constexpr int MAXC = 64;
  unsigned ckey[SSRHIP_MAX_CODEBOOKS][MAXC];
  unsigned long long cmass[SSRHIP_MAX_CODEBOOKS][MAXC];
  int ncand[SSRHIP_MAX_CODEBOOKS];
  unsigned result[SSRHIP_MAX_CODEBOOKS];
  int overflow;
};

// MODE 0: key of the kk-th largest valid key (valid keys != 0).  MODE 1: smallest key t with mass(keys > t) <= lim (0: keep all).
template <int MODE>
__device__ __forceinline__ uint32_t radix_select(const uint32_t (&key)[MAXE], const float (&p)[MAXE], unsigned kk, float lim, bool active,
                                                 SelShared& sh, int k, int sub, int lane) {
  const float SC = 1099511627776.0f;   // 2^40
  const unsigned long long limfx = (unsigned long long)((double)lim * (double)SC);
  uint32_t prefix = 0, mask = 0;
  unsigned long long carry = (MODE == 0) ? (unsigned long long)kk : 0ull;
  bool keep_all = false;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    unsigned long long* mine64 = sh.hist[k][sub];
    unsigned* mine32 = reinterpret_cast<unsigned*>(mine64);
    if (MODE == 0) { for (int j = lane; j < 256; j += 64) mine32[j] = 0u; }
    else { for (int j = lane; j < 256; j += 64) mine64[j] = 0ull; }
    __syncthreads();
    if (active && !keep_all) {
#pragma unroll
      for (int e = 0; e < MAXE; ++e) {
        const bool in = (MODE == 0) ? (key[e] != 0u) : (p[e] > 0.f);
        if (in && (key[e] & mask) == prefix) {
          if (MODE == 0) atomicAdd(&mine32[(key[e] >> shift) & 255u], 1u);
          else atomicAdd(&mine64[(key[e] >> shift) & 255u], (unsigned long long)(p[e] * SC));
        }
      }
    }
    __syncthreads();
    if (sub == 0 && active && !keep_all) {          // this codebook's scanning wave: bins 4*lane .. 4*lane+3
      unsigned long long c[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
      for (int w = 0; w < WPC; ++w)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          c[j] += (MODE == 0) ? (unsigned long long)reinterpret_cast<unsigned*>(sh.hist[k][w])[lane * 4 + j] : sh.hist[k][w][lane * 4 + j];
      const unsigned long long base = (MODE == 0) ? 0ull : carry;
      const unsigned long long a3 = base + suffix_excl<unsigned long long>(c[0] + c[1] + c[2] + c[3], lane);
      const unsigned long long a2 = a3 + c[3], a1 = a2 + c[2], a0 = a1 + c[1];
      int f = -1;
      unsigned long long above = 0;
      if (MODE == 0) {       // need-th largest falls into the bucket with above < need <= above + c
        if (carry > a3 && carry <= a3 + c[3]) { f = 3; above = a3; }
        else if (carry > a2 && carry <= a2 + c[2]) { f = 2; above = a2; }
        else if (carry > a1 && carry <= a1 + c[1]) { f = 1; above = a1; }
        else if (carry > a0 && carry <= a0 + c[0]) { f = 0; above = a0; }
      } else {               // running mass from the top first exceeds lim in the bucket with above <= lim < above + c
        if (a3 <= limfx && limfx < a3 + c[3]) { f = 3; above = a3; }
        else if (a2 <= limfx && limfx < a2 + c[2]) { f = 2; above = a2; }
        else if (a1 <= limfx && limfx < a1 + c[1]) { f = 1; above = a1; }
        else if (a0 <= limfx && limfx < a0 + c[0]) { f = 0; above = a0; }
      }
      const unsigned long long bal = __ballot(f >= 0);
      if (lane == 0) sh.found[k] = bal ? 1 : 0;
      if (f >= 0) {          // exactly one lane
        sh.bin[k] = (unsigned)(lane * 4 + f);
        sh.carry[k] = (MODE == 0) ? (carry - above) : above;
      }
    }
    __syncthreads();
    if (active && !keep_all) {
      if (!sh.found[k]) keep_all = true;           // MODE 1 only: the whole mass is <= lim
      else {
        prefix |= sh.bin[k] << shift;
        mask |= 255u << shift;
        carry = sh.carry[k];
      }
    }
  }
  return keep_all ? 0u : prefix;
}

// Same two selections in ~1/3 of the time for ordinary logits. The first radix pass above is slow because the top byte of a
// float key is sign + 7 exponent bits: most logits share a dozen bins and the LDS atomics serialise (measured 7.8k clocks
// for pass 0 alone, and 3.6k fixed per further pass). Here the ONE histogram pass bins by VALUE: 254 equal-width bins over
// the range [lo, hi] of the ordinary logits (the +-1e4 entries the reference's edits write get the two outer bins) —
// monotone in the key order, and a Gaussian-ish logit vector puts a handful of elements in each bin. The
// scan picks the bin that holds the answer exactly as before; its elements (usually < 10) go to an LDS list and every
// candidate counts the mass / number of candidates above it — the one that straddles the limit publishes its key. Same
// integer arithmetic and the same definition of the threshold as radix_select => identical results. More than MAXC
// candidates in the bin (heavy ties, flat logits): `fallback` is raised for the WHOLE workgroup (barrier-uniform) and the
// caller runs radix_select.
template <int MODE>
__device__ __forceinline__ uint32_t bin_select(const uint32_t (&key)[MAXE], const float (&p)[MAXE], const float (&l)[MAXE], float lo, float scale,
                                               unsigned kk, float lim, bool active, SelShared& sh, int k, int sub, int lane, int force_radix, bool& fallback) {
  const float SC = 1099511627776.0f;   // 2^40
  const unsigned long long limfx = (unsigned long long)((double)lim * (double)SC);
  unsigned long long* mine64 = sh.hist[k][sub];
  unsigned* mine32 = reinterpret_cast<unsigned*>(mine64);
  if (MODE == 0) { for (int j = lane; j < 256; j += 64) mine32[j] = 0u; }
  else { for (int j = lane; j < 256; j += 64) mine64[j] = 0ull; }
  if (threadIdx.x == 0) sh.overflow = force_radix;
  if (sub == 0 && lane == 0) { sh.ncand[k] = 0; sh.result[k] = 0u; }
  if (MODE == 0) STAMP(9);
  int vb[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const bool in = (MODE == 0) ? (key[e] != 0u) : (p[e] > 0.f);
    // ordinary logits -> bins 1..254 over [lo, hi]; the reference's forced (+1e4) / banned (-1e4) entries -> bins 255 / 0
    const int ob = min(254, max(1, 1 + (int)((l[e] - lo) * scale)));
    vb[e] = !in ? -1 : (l[e] >= 9000.f ? 255 : (l[e] <= -9000.f ? 0 : ob));
  }
  __syncthreads();
  if (MODE == 0) STAMP(10);
  if (active) {
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      if (vb[e] >= 0) {
        if (MODE == 0) atomicAdd(&mine32[vb[e]], 1u);
        else atomicAdd(&mine64[vb[e]], (unsigned long long)(p[e] * SC));
      }
    }
  }
  __syncthreads();
  if (MODE == 0) STAMP(11);
  if (sub == 0 && active) {          // scanning wave: bins 4*lane .. 4*lane+3, from the top
    unsigned long long c[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
    for (int w = 0; w < WPC; ++w)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        c[j] += (MODE == 0) ? (unsigned long long)reinterpret_cast<unsigned*>(sh.hist[k][w])[lane * 4 + j] : sh.hist[k][w][lane * 4 + j];
    const unsigned long long a3 = suffix_excl<unsigned long long>(c[0] + c[1] + c[2] + c[3], lane);
    const unsigned long long a2 = a3 + c[3], a1 = a2 + c[2], a0 = a1 + c[1];
    const unsigned long long need = (unsigned long long)kk;
    int f = -1;
    unsigned long long above = 0;
    if (MODE == 0) {
      if (need > a3 && need <= a3 + c[3]) { f = 3; above = a3; }
      else if (need > a2 && need <= a2 + c[2]) { f = 2; above = a2; }
      else if (need > a1 && need <= a1 + c[1]) { f = 1; above = a1; }
      else if (need > a0 && need <= a0 + c[0]) { f = 0; above = a0; }
    } else {
      if (a3 <= limfx && limfx < a3 + c[3]) { f = 3; above = a3; }
      else if (a2 <= limfx && limfx < a2 + c[2]) { f = 2; above = a2; }
      else if (a1 <= limfx && limfx < a1 + c[1]) { f = 1; above = a1; }
      else if (a0 <= limfx && limfx < a0 + c[0]) { f = 0; above = a0; }
    }
    const unsigned long long bal = __ballot(f >= 0);
    if (lane == 0) sh.found[k] = bal ? 1 : 0;
    if (f >= 0) {
      sh.bin[k] = (unsigned)(lane * 4 + f);
      sh.carry[k] = above;             // count / mass in the bins above the selected one
    }
  }
  __syncthreads();
  if (MODE == 0) STAMP(12);
  const bool found = active && sh.found[k];          // MODE 1, not found: the whole mass is <= lim -> keep all
  const int bsel = found ? (int)sh.bin[k] : -2;
  const unsigned long long above_bins = found ? sh.carry[k] : 0ull;
  if (found) {
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      if (vb[e] == bsel) {
        const int idx = atomicAdd(&sh.ncand[k], 1);
        if (idx < SelShared::MAXC) {
          sh.ckey[k][idx] = key[e];
          sh.cmass[k][idx] = (MODE == 0) ? 1ull : (unsigned long long)(p[e] * SC);
        }
      }
    }
  }
  __syncthreads();
  if (MODE == 0) STAMP(13);
  const int n = found ? sh.ncand[k] : 0;
  if (n > SelShared::MAXC) sh.overflow = 1;
  __syncthreads();
  if (MODE == 0) STAMP(14);
  fallback = sh.overflow != 0;                        // identical for every thread of the workgroup
  if (!fallback) {
    // a lane rarely holds more than one candidate: walk "my next candidate" until no lane of the wave has one left, so
    // the n-iteration list scan runs once or twice per wave instead of once per element slot
    unsigned todo = 0u;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) todo |= (found && vb[e] == bsel) ? (1u << e) : 0u;
    while (__any(todo != 0u)) {
      uint32_t ke = 0u;
      bool mine = false;
      if (todo) {
        const int e0 = __builtin_ctz(todo);
        todo &= todo - 1u;
        mine = true;
#pragma unroll
        for (int e = 0; e < MAXE; ++e) ke = (e == e0) ? key[e] : ke;
      }
      unsigned long long A = above_bins, M = 0ull;
      for (int i = 0; i < n; ++i) {
        const unsigned ck = sh.ckey[k][i];
        const unsigned long long m = sh.cmass[k][i];
        A += (ck > ke) ? m : 0ull;
        M += (ck == ke) ? m : 0ull;
      }
      const bool hit = (MODE == 0) ? ((unsigned long long)kk > A && (unsigned long long)kk <= A + M) : (A <= limfx && limfx < A + M);
      if (mine && hit) sh.result[k] = ke;             // all writers (equal keys) write the same value
    }
  }
  __syncthreads();
  if (MODE == 0) STAMP(15);
  return (found && !fallback) ? sh.result[k] : 0u;
}

__global__ __launch_bounds__(SAMPLE_THREADS) void sample_kernel(const ssrhip_sample_args a, const int force_radix) {
  __shared__ SelShared sel;
  __shared__ float sh_f[SSRHIP_MAX_CODEBOOKS][WPC];
  __shared__ float sh_g[SSRHIP_MAX_CODEBOOKS][WPC];
  __shared__ int sh_i[SSRHIP_MAX_CODEBOOKS][WPC];
  __shared__ int sh_sample[SSRHIP_MAX_CODEBOOKS];
  __shared__ int sh_argmax0;
  __shared__ int sh_next[SSRHIP_MAX_CODEBOOKS + 2];   // next tokens, next audio pos, live flag
  const int u = blockIdx.x;
  const ssrhip_sampler_cfg& c = a.cfg[u];
  ssrhip_sampler_state& st = a.state[u];
  if (st.done) return;                                 // uniform for the whole workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = wave / WPC, sub = wave % WPC;
  const int K = a.K, card = a.card;
  const bool active = k < K;
  const int rows = c.use_cfg ? 2 : 1;     // (read again below as c_usecfg: same value)
  const int row0 = u * rows;
  // snapshot of the state (every wave reads it before thread 0 mutates it, after the barriers below)
  const int num_gen = st.num_gen, num_eog = st.num_eog, cfg_tag = st.num_cfg_tag;
  const int prev_token = st.prev_token, consec = st.consec_silence;
  const int step = st.n_steps;
  // cfg fields into registers once
  const int c_eos = c.eos, c_sos = c.sos, c_mts = c.mts, c_mts_end = c.mts + c.max_n_spans;
  const int c_empty = c.empty_token, c_eog = c.eog, c_topk = c.top_k, c_stoprep = c.stop_repetition;
  const float c_topp = c.top_p, c_temp = c.temperature, c_coef = c.cfg_coef, c_om = c.cfg_one_minus;
  const int c_maxsteps = c.max_steps, c_nsil = c.n_silence;
  const uint32_t c_seedlo = c.seed_lo, c_seedhi = c.seed_hi;
  // everything the single-thread state machine at the end reads, fetched NOW with the other loads (round 4: it re-read these from
  // memory one dependent load after the other — the compiler cannot keep them across the stores to `st` / `generated` — ~8 serial
  // L1 / L2 round trips on the tail of the step's last kernel)
  const int c_usecfg = c.use_cfg, c_stride = c.cfg_stride, c_textlen = c.text_len, c_nspans = c.n_spans;
  const int st_audio_pos = st.audio_pos, st_span = st.span;
  int c_sil[SSRHIP_MAX_SILENCE];
#pragma unroll
  for (int j = 0; j < SSRHIP_MAX_SILENCE; ++j) c_sil[j] = c.silence[j];
  STAMP(0);

  // ---- CFG combine (:690-696) + edits (:699-730); element e of this lane is index e*256 + sub*64 + lane
  const int kc = active ? k : 0;
  const float* lc = a.logits + ((size_t)row0 * K + kc) * card;
  const float* lu = lc + (size_t)K * card;
  const bool guided = c_usecfg && (cfg_tag == c_stride);
  bool penal = false;     // silence-repetition penalty applies to logits[0][prev_token] (:726-730)
  if (k == 0 && num_eog == 0 && c_stoprep > 0 && consec > c_stoprep) {
#pragma unroll
    for (int s = 0; s < SSRHIP_MAX_SILENCE; ++s) penal |= (s < c_nsil) && (c_sil[s] == prev_token);
  }
  const float npen = (float)(consec - (c_stoprep - 1));
  const bool force_empty = (num_gen < K - 1) && (k > num_gen);     // :705-707
  const bool cut_eog_empty = (num_eog > 0) && (k > num_eog);       // :710-712
  const bool cut_eog = (num_eog == 0) && (k >= 1);                 // :722-723
  float l[MAXE], lun[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int i = e * 256 + sub * 64 + lane;
    const int ic = (i < card) ? i : 0;
    l[e] = lc[ic];
    lun[e] = guided ? lu[ic] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const int i = e * 256 + sub * 64 + lane;
    float v = guided ? __fadd_rn(__fmul_rn(c_coef, l[e]), __fmul_rn(c_om, lun[e])) : l[e];
    const bool special = (i == c_eos) | (i == c_sos) | ((i >= c_mts) & (i < c_mts_end));
    v = special ? -10000.f : v;
    v = (force_empty & (i == c_empty)) ? 10000.f : v;
    v = (cut_eog_empty & ((i == c_eog) | (i == c_empty))) ? -10000.f : v;
    v = (cut_eog & (i == c_eog)) ? -10000.f : v;
    if (penal && i == prev_token) v = (v < 0.f) ? __fmul_rn(v, npen) : __fdiv_rn(v, npen);
    l[e] = (active && i < card) ? v : -INFINITY;
  }
  if (a.dbg_logits && active) {
    float* dbg = a.dbg_logits + ((size_t)u * K + k) * card;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int i = e * 256 + sub * 64 + lane;
      if (i < card) dbg[i] = l[e];
    }
  }
  STAMP(1);
  // ---- max / argmax of the edited logits (first index on ties), needed for the stop rule :739
  float mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < MAXE; ++e) mx = fmaxf(mx, l[e]);
  mx = wave_max(mx);
  int am = 0x7fffffff;
#pragma unroll
  for (int e = 0; e < MAXE; ++e) if (l[e] == mx) am = min(am, e * 256 + sub * 64 + lane);
  am = wave_min_i(am);
  if (lane == 0 && active) { sh_f[k][sub] = mx; sh_i[k][sub] = am; }
  __syncthreads();
  if (active) {
    float m2 = sh_f[k][0];
    int a2 = sh_i[k][0];
#pragma unroll
    for (int w = 1; w < WPC; ++w)
      if (sh_f[k][w] > m2 || (sh_f[k][w] == m2 && sh_i[k][w] < a2)) { m2 = sh_f[k][w]; a2 = sh_i[k][w]; }
    mx = m2;
    am = a2;
  }
  __syncthreads();
  // ---- temperature (:80-81): true division, like the reference
  if (c_temp != 1.0f) {
#pragma unroll
    for (int e = 0; e < MAXE; ++e) l[e] = __fdiv_rn(l[e], c_temp);
    float m3 = -INFINITY;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) m3 = fmaxf(m3, l[e]);
    m3 = wave_max(m3);
    if (lane == 0 && active) sh_f[k][sub] = m3;
    __syncthreads();
    if (active) mx = fmaxf(fmaxf(sh_f[k][0], sh_f[k][1]), fmaxf(sh_f[k][2], sh_f[k][3]));
    __syncthreads();
  }
  uint32_t key[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) key[e] = (active && (e * 256 + sub * 64 + lane) < card) ? okey(l[e]) : 0u;   // 0 = padding
  const uint32_t kmax = okey(mx);
  // range of the ordinary logits of this codebook (for bin_select's value bins)
  float vlo = INFINITY, vhi = -INFINITY;
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    const bool ord = (key[e] != 0u) && (l[e] < 9000.f) && (l[e] > -9000.f);
    vlo = ord ? fminf(vlo, l[e]) : vlo;
    vhi = ord ? fmaxf(vhi, l[e]) : vhi;
  }
  vlo = -wave_max(-vlo);
  vhi = wave_max(vhi);
  if (lane == 0 && active) { sh_f[k][sub] = vlo; sh_g[k][sub] = vhi; }
  __syncthreads();
  if (active) {
    vlo = fminf(fminf(sh_f[k][0], sh_f[k][1]), fminf(sh_f[k][2], sh_f[k][3]));
    vhi = fmaxf(fmaxf(sh_g[k][0], sh_g[k][1]), fmaxf(sh_g[k][2], sh_g[k][3]));
  }
  __syncthreads();
  const float vscale = (vhi > vlo) ? 253.0f / (vhi - vlo) : 0.f;
  float p[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) p[e] = 0.f;
  STAMP(2);
  // ---- top-k (:38-44): keep logits >= k-th largest value (ties kept)
  uint32_t thr = 0;   // keep keys >= thr
  if (c_topk > 0) {   // uniform
    const int kk = min(max(c_topk, 1), card);
    if (kk == 1) thr = kmax;
    else if (kk < card) {
      bool fb;
      thr = bin_select<0>(key, p, l, vlo, vscale, (unsigned)kk, 0.f, active, sel, kc, sub, lane, force_radix, fb);
      if (fb) thr = radix_select<0>(key, p, (unsigned)kk, 0.f, active, sel, kc, sub, lane);          // workgroup-uniform
    }
  }
  STAMP(3);
  // ---- softmax numerators over the kept set, then top-p (:46-67)
  float Z = 0.f;
#pragma unroll
  for (int e = 0; e < MAXE; ++e) {
    p[e] = (key[e] != 0u && key[e] >= thr) ? __expf(l[e] - mx) : 0.f;
    Z += p[e];
  }
  Z = wave_sum(Z);
  if (lane == 0 && active) sh_f[k][sub] = Z;
  __syncthreads();
  if (active) Z = (sh_f[k][0] + sh_f[k][1]) + (sh_f[k][2] + sh_f[k][3]);
  __syncthreads();
  if (c_topp < 1.0f) {   // uniform
    bool fb;
    uint32_t tp = bin_select<1>(key, p, l, vlo, vscale, 0u, c_topp * Z, active, sel, kc, sub, lane, force_radix, fb);
    if (fb) tp = radix_select<1>(key, p, 0u, c_topp * Z, active, sel, kc, sub, lane);                // workgroup-uniform
    if (tp > thr) {
      thr = tp;
#pragma unroll
      for (int e = 0; e < MAXE; ++e) p[e] = (key[e] >= thr) ? p[e] : 0.f;
    }
  }
  STAMP(4);
  // ---- multinomial(softmax, 1) == argmax(prob / q), q ~ Exp(1)   (torch CPU fast path; :85).
  // The positive normaliser does not change the argmax, so it is dropped.
  {
    const float* nz = (a.noise && c.use_noise) ? a.noise + (((size_t)u * c_maxsteps + step) * K + kc) * card : nullptr;
    const uint32_t sd = hash32(c_seedlo + (uint32_t)step * 0x9E3779B1u) ^ hash32(c_seedhi + (uint32_t)kc * 0x85EBCA6Bu + 0x632BE5ABu);
    float best = -1.f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int i = e * 256 + sub * 64 + lane;
      float q;
      if (nz) q = nz[(key[e] != 0u) ? i : 0];                       // wave-uniform branch
      else {
        const uint32_t hsh = hash32(sd + (uint32_t)i * 0x9E3779B1u);
        q = fmaxf(-__logf(((float)(hsh >> 8) + 1.0f) * (1.0f / 16777216.0f)), 1e-30f);
      }
      const float sc = (p[e] > 0.f) ? __fdividef(p[e], q) : -1.f;
      const bool better = sc > best;                                 // e ascending => lowest index wins ties within the lane
      best = better ? sc : best;
      bi = better ? i : bi;
    }
    const float wb = wave_max(best);
    bi = (best == wb) ? bi : 0x7fffffff;
    bi = wave_min_i(bi);
    if (lane == 0 && active) { sh_f[k][sub] = wb; sh_i[k][sub] = bi; }
    __syncthreads();
    if (active && sub == 0 && lane == 0) {
      float b2 = sh_f[k][0];
      int i2 = sh_i[k][0];
#pragma unroll
      for (int w = 1; w < WPC; ++w)
        if (sh_f[k][w] > b2 || (sh_f[k][w] == b2 && sh_i[k][w] < i2)) { b2 = sh_f[k][w]; i2 = sh_i[k][w]; }
      sh_sample[k] = i2;
      if (k == 0) sh_argmax0 = am;
    }
  }
  STAMP(5);
  __syncthreads();
  STAMP(6);
  if (threadIdx.x == 0) {
    // ---- state machine (:709-761), single thread; reads only registers and LDS
    int s[SSRHIP_MAX_CODEBOOKS];
    for (int j = 0; j < K; ++j) s[j] = sh_sample[j];
    int ne_og = num_eog, cs = consec, pt = prev_token;
    if (num_eog > 0) {
      for (int j = 0; j < num_eog; ++j) s[j] = c_empty;
      s[num_eog] = c_eog;
      ne_og = num_eog + 1;
    } else {
      if (s[0] == c_eog || sh_argmax0 == c_eog || (st_audio_pos + 1) > c_textlen * 10) {
        s[0] = c_eog;
        ne_og = 1;
      }
      bool sil = false;
#pragma unroll
      for (int j = 0; j < SSRHIP_MAX_SILENCE; ++j) sil |= (j < c_nsil) && (c_sil[j] == s[0]);
      cs = (sil && s[0] == pt) ? cs + 1 : 0;
      pt = s[0];
    }
    if (c_usecfg) st.num_cfg_tag = (cfg_tag == c_stride) ? 1 : cfg_tag + 1;
    int* gen = a.generated + ((size_t)u * c_maxsteps + step) * K;
    for (int j = 0; j < K; ++j) gen[j] = s[j];
    st.n_steps = step + 1;
    st.num_gen = num_gen + 1;
    st.num_eog = ne_og;
    st.consec_silence = cs;
    st.prev_token = pt;
    bool done = false;
    int span = st_span;
    if (ne_og == K) {                 // span finished (:753): the all-eog sample is NOT fed back
      st.span_end[span] = step + 1;
      span += 1;
      st.span = span;
      if (span >= c_nspans) { st.done = 1; done = true; }
      else {
        st.num_gen = 0; st.num_eog = 0; st.num_cfg_tag = 1; st.prev_token = -1; st.consec_silence = 0;
        for (int j = 0; j < K; ++j) s[j] = c_mts + span;   // next span starts from its mask token (:655)
      }
    }
    if (!done && step + 1 >= c_maxsteps) { st.done = 2; done = true; }
    sh_next[SSRHIP_MAX_CODEBOOKS + 1] = done ? 0 : 1;
    if (!done) {
      const int apos = st_audio_pos + 1;
      st.audio_pos = apos;
      sh_next[SSRHIP_MAX_CODEBOOKS] = apos;
      for (int j = 0; j < SSRHIP_MAX_CODEBOOKS; ++j) sh_next[j] = (j < K) ? s[j] : 0;
      for (int rr = 0; rr < rows; ++rr) {
        const int b = row0 + rr;
        for (int j = 0; j < SSRHIP_MAX_CODEBOOKS; ++j) a.next_tok[b * SSRHIP_MAX_CODEBOOKS + j] = (j < K) ? s[j] : 0;
        a.next_pos[b] = apos;
        const int kp = a.kv_pos[b] + 1;
        a.kv_pos[b] = kp;
        a.row_len[b] = kp + 1;
      }
    }
  }
  STAMP(7);
  // ---- fused embedding of the next input row(s) (ssr.py:757-763): saves a launch per step
  if (a.embed.out) {
    __syncthreads();
    if (sh_next[SSRHIP_MAX_CODEBOOKS + 1]) {
      for (int rr = 0; rr < rows; ++rr) embed_row(a.embed, row0 + rr, 1, sh_next[SSRHIP_MAX_CODEBOOKS], sh_next, threadIdx.x, SAMPLE_THREADS);
    }
  }
  STAMP(8);
}

}  // namespace

extern "C" int ssrhip_embed(const ssrhip_embed_args* a, ssrhip_stream_t stream) {
  SSR_REQUIRE(a && a->audio_emb && a->pe && a->tok && a->pos && a->out, "ssrhip_embed: null argument");
  SSR_REQUIRE(a->R > 0 && a->D % 4 == 0 && a->K >= 1 && a->K <= SSRHIP_MAX_CODEBOOKS, "ssrhip_embed: bad R/D/K");
  SSR_REQUIRE(!a->kind || a->text_emb, "ssrhip_embed: text rows need text_emb");
  hipLaunchKernelGGL(embed_kernel, dim3(a->R), dim3(256), 0, (hipStream_t)stream, *a);
  SSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int ssrhip_sample(const ssrhip_sample_args* a, ssrhip_stream_t stream) {
  SSR_REQUIRE(a && a->logits && a->cfg && a->state && a->generated && a->next_tok && a->next_pos && a->kv_pos && a->row_len,
              "ssrhip_sample: null argument");
  SSR_REQUIRE(a->K >= 1 && a->K <= SSRHIP_MAX_CODEBOOKS, "ssrhip_sample: K=%d", a->K);
  SSR_REQUIRE(a->card > 0 && a->card <= 64 * WPC * MAXE, "ssrhip_sample: card=%d exceeds %d", a->card, 64 * WPC * MAXE);
  if (a->embed.out) SSR_REQUIRE(a->embed.audio_emb && a->embed.pe && a->embed.D % 4 == 0 && a->embed.K == a->K && a->embed.card == a->card,
                                "ssrhip_sample: fused embed needs audio_emb, pe, matching K/card");
  const char* e = getenv("SSRHIP_SAMPLE_RADIX");          // test knob: always take the radix-select path (value baked in at graph capture)
  hipLaunchKernelGGL(sample_kernel, dim3(a->n_utt), dim3(SAMPLE_THREADS), 0, (hipStream_t)stream, *a, (e && atoi(e)) ? 1 : 0);
  SSR_LAUNCH_CHECK();
  return 0;
}
