// embed_sample.hip — token embedding and the device-side sampler / logit state machine, gfx950.
//
// ssrhip_embed : sum of K codebook embeddings (or one text embedding) + alpha * sinusoidal pe row.
// ssrhip_sample: everything the reference does on the host between `predict_layer` and the next
//   `embed` (models/ssr.py:689-761): CFG combine, special-token edits, eog cascade, silence penalty,
//   temperature / top-k / top-p filtering, one multinomial draw per codebook, stop rules, span hand-over.
//   One workgroup per utterance, wave k owns codebook k; the ~2k logits of a codebook live in VGPRs
//   (<= 34 per lane), top-k / top-p thresholds are found by a 32-step bisection on the order-preserving
//   integer image of the logits (ballot+popcount for counts, fixed-order wave sums for mass) — no sort,
//   no atomics, bit-reproducible.  No host round trip per step.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void embed_kernel(const ssrhip_embed_args a) {
  const int r = blockIdx.x;
  const int D = a.D;
  const int kind = a.kind ? a.kind[r] : 1;
  const int pos = a.pos[r];
  const int* tok = a.tok + (size_t)r * SSRHIP_MAX_CODEBOOKS;
  const float alpha = kind ? a.alpha_audio : a.alpha_text;
  for (int d = threadIdx.x * 4; d < D; d += 1024) {
    float4 e;
    if (kind == 0) {
      e = ld4(a.text_emb + (size_t)tok[0] * D + d);
    } else {
      // torch.stack(...).sum(dim=0): sequential sum over codebooks (ssr.py:193-196)
      e = ld4(a.audio_emb + (size_t)tok[0] * D + d);
      for (int k = 1; k < a.K; ++k) {
        const float4 t = ld4(a.audio_emb + ((size_t)k * a.card + tok[k]) * D + d);
        e.x += t.x; e.y += t.y; e.z += t.z; e.w += t.w;
      }
    }
    const float4 p = ld4(a.pe + (size_t)pos * D + d);
    // x * 1.0 + alpha * pe (embedding.py:96): product rounded, then sum rounded (no fma contraction)
    e.x = __fadd_rn(e.x, __fmul_rn(alpha, p.x));
    e.y = __fadd_rn(e.y, __fmul_rn(alpha, p.y));
    e.z = __fadd_rn(e.z, __fmul_rn(alpha, p.z));
    e.w = __fadd_rn(e.w, __fmul_rn(alpha, p.w));
    *reinterpret_cast<float4*>(a.out + (size_t)r * D + d) = e;
  }
}

constexpr int MAXE = 34;   // logits per lane: card <= 64*34 = 2176

__device__ __forceinline__ uint32_t okey(float f) {   // order-preserving float -> uint32
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void sample_kernel(const ssrhip_sample_args a) {
  __shared__ int sh_sample[4];
  __shared__ int sh_argmax0;
  const int u = blockIdx.x;
  const ssrhip_sampler_cfg& c = a.cfg[u];
  ssrhip_sampler_state& st = a.state[u];
  if (st.done) return;
  const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
  const int K = a.K, card = a.card;
  const int ne = (card + 63) / 64;
  const int rows = c.use_cfg ? 2 : 1;
  const int row0 = u * rows;
  // snapshot of the state (all waves read before thread 0 mutates it after the barrier)
  const int num_gen = st.num_gen, num_eog = st.num_eog, cfg_tag = st.num_cfg_tag;
  const int prev_token = st.prev_token, consec = st.consec_silence;
  const int step = st.n_steps;

  if (k < K) {
    const float* lc = a.logits + ((size_t)row0 * K + k) * card;
    const float* lu = lc + (size_t)K * card;
    const bool guided = c.use_cfg && (cfg_tag == c.cfg_stride);
    float l[MAXE];
    // ---- CFG combine (:690-696) + edits (:699-730)
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int i = e * 64 + lane;
      float v = -INFINITY;
      if (e < ne && i < card) {
        v = lc[i];
        if (guided) v = __fadd_rn(__fmul_rn(c.cfg_coef, v), __fmul_rn(c.cfg_one_minus, lu[i]));
        if (i == c.eos || i == c.sos || (i >= c.mts && i < c.mts + c.max_n_spans)) v = -10000.f;
        if (num_gen < K - 1 && k > num_gen && i == c.empty_token) v = 10000.f;
        if (num_eog > 0) {
          if (k > num_eog && (i == c.eog || i == c.empty_token)) v = -10000.f;
        } else {
          if (k >= 1 && i == c.eog) v = -10000.f;
          if (k == 0 && i == prev_token && c.stop_repetition > 0 && consec > c.stop_repetition) {
            bool sil = false;
            for (int s = 0; s < c.n_silence; ++s) sil |= (c.silence[s] == prev_token);
            if (sil) {
              const float n = (float)(consec - (c.stop_repetition - 1));
              v = (v < 0.f) ? __fmul_rn(v, n) : __fdiv_rn(v, n);
            }
          }
        }
        if (a.dbg_logits) a.dbg_logits[((size_t)u * K + k) * card + i] = v;
      }
      l[e] = v;
    }
    // ---- argmax of the edited logits (first index on ties), needed for the stop rule :739
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) mx = fmaxf(mx, l[e]);
    mx = wave_max(mx);
    int am = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) if (l[e] == mx) am = min(am, e * 64 + lane);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = min(am, __shfl_xor(am, o, 64));
    // ---- temperature (:80-81)
    if (c.temperature != 1.0f) {
#pragma unroll
      for (int e = 0; e < MAXE; ++e) l[e] = __fdiv_rn(l[e], c.temperature);
      mx = __fdiv_rn(mx, c.temperature);
      if (c.temperature < 0.f) {  // not meaningful, keep max consistent
        mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < MAXE; ++e) mx = fmaxf(mx, l[e]);
        mx = wave_max(mx);
      }
    }
    // ---- top-k (:38-44): keep logits >= k-th largest value (ties kept)
    uint32_t thr = 0;   // keep keys >= thr
    if (c.top_k > 0) {
      const int kk = min(max(c.top_k, 1), card);
      if (kk == 1) {
        thr = okey(mx);
      } else if (kk < card) {
        uint32_t t = 0;   // largest key with count(keys > t) > kk-1
        for (int bit = 31; bit >= 0; --bit) {
          const uint32_t cand = t | (1u << bit);
          int cnt = 0;
#pragma unroll
          for (int e = 0; e < MAXE; ++e) cnt += __popcll(__ballot(okey(l[e]) > cand && (e * 64 + lane) < card));
          if (cnt > kk - 1) t = cand;
        }
        thr = t + 1;
      }
    }
    // ---- softmax over the kept set, then top-p (:46-67)
    float p[MAXE];
    float Z = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const bool keep = (e * 64 + lane) < card && okey(l[e]) >= thr;
      p[e] = keep ? expf(l[e] - mx) : 0.f;
      Z += p[e];
    }
    Z = wave_sum(Z);
    if (c.top_p < 1.0f) {
#pragma unroll
      for (int e = 0; e < MAXE; ++e) p[e] = __fdiv_rn(p[e], Z);
      // smallest key t* such that mass(keys > t*) <= top_p ; keep keys >= t*
      uint32_t t = 0;
      for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = t | (1u << bit);
        float m = 0.f;
#pragma unroll
        for (int e = 0; e < MAXE; ++e) m += (okey(l[e]) > cand) ? p[e] : 0.f;
        m = wave_sum(m);
        if (m > c.top_p) t = cand;
      }
      // is mass(keys > 0) <= top_p already? then everything is kept (t stays 0)
      const uint32_t thr_p = t + 1;
      thr = max(thr, thr_p);
      Z = 0.f;
#pragma unroll
      for (int e = 0; e < MAXE; ++e) {
        const bool keep = (e * 64 + lane) < card && okey(l[e]) >= thr;
        p[e] = keep ? expf(l[e] - mx) : 0.f;
        Z += p[e];
      }
      Z = wave_sum(Z);
    }
    // ---- multinomial(softmax, 1) == argmax(prob / q), q ~ Exp(1)   (torch CPU fast path; :85)
    const float* nz = a.noise ? a.noise + (((size_t)u * c.max_steps + step) * K + k) * card : nullptr;
    float best = -1.f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int i = e * 64 + lane;
      if (e < ne && i < card && p[e] > 0.f) {
        float q;
        if (nz) q = nz[i];
        else {
          const uint32_t hsh = hash32(hash32(c.seed_lo + (uint32_t)step * 0x9E3779B1u) ^ hash32(c.seed_hi + (uint32_t)(k * card + i) * 0x85EBCA6Bu));
          q = -logf(((float)(hsh >> 8) + 1.0f) * (1.0f / 16777216.0f));
          q = fmaxf(q, 1e-30f);
        }
        const float sc = __fdiv_rn(__fdiv_rn(p[e], Z), q);
        if (sc > best || (sc == best && i < bi)) { best = sc; bi = i; }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) {
      sh_sample[k] = bi;
      if (k == 0) sh_argmax0 = am;
    }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;

  // ---- state machine (:709-761), single thread
  int s[SSRHIP_MAX_CODEBOOKS];
  for (int j = 0; j < K; ++j) s[j] = sh_sample[j];
  int ne_og = num_eog, cs = consec, pt = prev_token;
  if (num_eog > 0) {
    for (int j = 0; j < num_eog; ++j) s[j] = c.empty_token;
    s[num_eog] = c.eog;
    ne_og = num_eog + 1;
  } else {
    if (s[0] == c.eog || sh_argmax0 == c.eog || (st.audio_pos + 1) > c.text_len * 10) {
      s[0] = c.eog;
      ne_og = 1;
    }
    bool sil = false;
    for (int j = 0; j < c.n_silence; ++j) sil |= (c.silence[j] == s[0]);
    cs = (sil && s[0] == pt) ? cs + 1 : 0;
    pt = s[0];
  }
  if (c.use_cfg) st.num_cfg_tag = (cfg_tag == c.cfg_stride) ? 1 : cfg_tag + 1;
  int* gen = a.generated + ((size_t)u * c.max_steps + step) * K;
  for (int j = 0; j < K; ++j) gen[j] = s[j];
  st.n_steps = step + 1;
  st.num_gen = num_gen + 1;
  st.num_eog = ne_og;
  st.consec_silence = cs;
  st.prev_token = pt;
  bool done = false;
  if (ne_og == K) {                 // span finished (:753): the all-eog sample is NOT fed back
    st.span_end[st.span] = step + 1;
    st.span += 1;
    if (st.span >= c.n_spans) { st.done = 1; done = true; }
    else {
      st.num_gen = 0; st.num_eog = 0; st.num_cfg_tag = 1; st.prev_token = -1; st.consec_silence = 0;
      for (int j = 0; j < K; ++j) s[j] = c.mts + st.span;   // next span starts from its mask token (:655)
    }
  }
  if (!done && step + 1 >= c.max_steps) { st.done = 2; done = true; }
  if (!done) {
    st.audio_pos += 1;
    for (int rr = 0; rr < rows; ++rr) {
      const int b = row0 + rr;
      for (int j = 0; j < SSRHIP_MAX_CODEBOOKS; ++j) a.next_tok[b * SSRHIP_MAX_CODEBOOKS + j] = (j < K) ? s[j] : 0;
      a.next_pos[b] = st.audio_pos;
      a.kv_pos[b] += 1;
      a.row_len[b] = a.kv_pos[b] + 1;
    }
  }
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
  SSR_REQUIRE(a->card > 0 && a->card <= 64 * MAXE, "ssrhip_sample: card=%d exceeds %d", a->card, 64 * MAXE);
  hipLaunchKernelGGL(sample_kernel, dim3(a->n_utt), dim3(256), 0, (hipStream_t)stream, *a);
  SSR_LAUNCH_CHECK();
  return 0;
}
