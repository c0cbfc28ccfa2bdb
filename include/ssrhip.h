/* ssrhip.h — C-ABI of libssrhip.so: the MI355X (gfx950) kernels and decode engine behind the
 * `ssr_speech_amd` Python classes that mirror SSR-Speech's inference surface.
 *
 * The reference (WangHelin1997/SSR-Speech) has NO native/FFI layer on this path — it is plain
 * PyTorch modules — so every entry point below names the reference *Python*
 * function(s) whose arithmetic it replaces (paths relative to the reference root).  A maintainer's
 * binding is a ctypes stub.
 *
 * Conventions
 *   - plain C types only; all `*_dev` / `float*` / `int*` arguments are DEVICE pointers into buffers
 *     the caller owns (torch-allocated); the library never allocates or frees device memory.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*); no internal device sync.
 *   - return 0 on success, negative on error; `ssrhip_last_error()` returns a thread-local message.
 *   - fp32 everywhere (the reference's CPU path is fp32); integer tokens are int32 on the device.
 *   - "rows": B = utterances x (2 if classifier-free guidance else 1); row 2u is the conditional row
 *     of utterance u and row 2u+1 its unconditional row (models/ssr.py:571-577, 690-696).
 */
#ifndef SSRHIP_H
#define SSRHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSRHIP_VERSION 107
#define SSRHIP_PAGE 128          /* KV-cache page = 128 positions */
#define SSRHIP_MAX_CODEBOOKS 4
#define SSRHIP_MAX_SILENCE 8

typedef void* ssrhip_stream_t;

int ssrhip_version(void);
/* sizeof() of the ABI structs, for binding self-checks: 0 kv, 1 gemv_args, 2 attn_args, 3 embed_args,
 * 4 sampler_cfg, 5 sampler_state, 6 sample_args, 7 gemm_args, 8 lm_weights, 9 lm_dims, 10 lm_buffers, 11 prefill_args,
 * 12 lstm_args */
int ssrhip_sizeof(int which);
const char* ssrhip_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Paged KV cache (replaces the dense, re-concatenated `past` tensor: models/ssr.py:685-686,
 * models/modules/activation.py:626-631).
 *   pool  [n_pages][n_layer][2][n_head][SSRHIP_PAGE][head_dim] fp32
 *   table [n_seq][max_pages] int32 : logical page -> physical page of that sequence (row)
 * ---------------------------------------------------------------------------------------------- */
typedef struct ssrhip_kv {
  float* pool;
  const int32_t* table;
  int32_t max_pages;   /* table row stride */
  int32_t n_layer, n_head, head_dim;
} ssrhip_kv;

/* ------------------------------------------------------------------------------------------------
 * Fused weight-streaming GEMV for B<=4 rows:  y[b][n] = epi( sum_k pro(x)[b][k] * W[n][k] + bias[n] )
 * replaces F.linear at: activation.py:86 (packed in-proj), :637 (out-proj), transformer.py:386-388
 * (linear1/ReLU/linear2), models/ssr.py:175-179 (prediction heads), fused with
 * transformer.py:58-75 (LayerNorm, eps 1e-5) as prologue and the residual add (transformer.py:328-329).
 * ---------------------------------------------------------------------------------------------- */
enum { SSRHIP_PRO_NONE = 0, SSRHIP_PRO_LAYERNORM = 1, SSRHIP_PRO_ATTN_COMBINE = 2 };
enum { SSRHIP_ACT_NONE = 0, SSRHIP_ACT_RELU = 1, SSRHIP_ACT_GELU_ERF = 2, SSRHIP_ACT_ELU = 3 };
enum { SSRHIP_EPI_STORE = 0, SSRHIP_EPI_RESIDUAL = 1, SSRHIP_EPI_QKV_APPEND = 2 };

typedef struct ssrhip_gemv_args {
  const float* W;        /* [groups][N][K] row-major (PyTorch Linear.weight layout) */
  const float* bias;     /* [groups][N] or NULL */
  const float* x;        /* PRO_NONE/LAYERNORM: [B][x_stride] (+ g*K);  ATTN_COMBINE: unused */
  float* y;              /* STORE: [B][y_stride] (+ g*N); RESIDUAL: y += ...; QKV_APPEND: q out [B][K] */
  int32_t B, N, K, groups;
  int32_t x_stride, y_stride;
  int32_t pro, act, epi;
  const float* ln_w; const float* ln_b; float ln_eps;           /* PRO_LAYERNORM */
  /* PRO_ATTN_COMBINE: split-KV partials written by ssrhip_attn_decode */
  const float* part_o; const float* part_ml; int32_t max_splits; /* [R][H][max_splits][hd], [R][H][max_splits][2] */
  const int32_t* row_len;                                        /* [B] keys visible to each row */
  /* EPI_QKV_APPEND: N == 3K; q -> y, k/v -> cache at position kv_pos[b] of sequence b */
  ssrhip_kv kv; int32_t layer; const int32_t* kv_pos;
  /* 5..16 rows only (matrix-core path): activations in the 16-column tiled layout SSRHIP_TILED(b,k) below instead of
   * row-major [B][stride]; x_stride / y_stride are ignored for a tiled operand. QKV_APPEND's q output is always row-major. */
  int32_t x_tiled, y_tiled;
  /* 5..16 rows only: W is stored in the matrix core's streaming order instead of [N][K] (see SSRHIP_WTILED_INDEX): one
   * wave-level load then reads 8 full 128-byte lines instead of 64 sixteen-byte pieces of 16 different rows. */
  int32_t w_tiled;
} ssrhip_gemv_args;

/* Streaming-order weight layout for the 5..16-row GEMV (`w_tiled`): rows are grouped in 8-row units (the last unit zero-padded),
 * K in 16-float steps; the 512-byte block of (unit u, k-step t) holds, at float4 index ks*8 + c, the four weights
 * W[8u + c][16t + 4ks .. 16t + 4ks + 3]   (c = 0..7 row inside the unit, ks = 0..3 k-slot of the 16x16x4 MFMA).
 * Blocks of one unit are contiguous along t, units follow each other: float index of W[n][k] is
 *   ((n/8) * (K/16) + k/16) * 128 + (((k%16)/4) * 8 + n%8) * 4 + k%4 ;  a group's matrix takes ceil(N/8)*8*K floats. */
#define SSRHIP_WTILED_INDEX(n, k, K) ((((size_t)(n) / 8) * ((size_t)(K) / 16) + (size_t)(k) / 16) * 128 + ((((k) % 16) / 4) * 8 + (n) % 8) * 4 + (k) % 4)

/* 16-column tiled activation layout used between the kernels of the 5..16-row decode step: element (row b, feature k) of a
 * [<=16][K] activation lives at float index ((k/4)*16 + b)*4 + k%4, i.e. 4 consecutive features of the 16 rows are 256
 * contiguous bytes — exactly what one 64-lane MFMA B-operand load wants (a whole KiB per wave instruction). Buffer size is
 * always 16*K floats. */
#define SSRHIP_TILED(b, k) ((((size_t)(k) >> 2) * 16 + (size_t)(b)) * 4 + ((k) & 3))

int ssrhip_gemv(const ssrhip_gemv_args* a, ssrhip_stream_t stream);

/* Two consecutive launches of the 2-row decode step as ONE: `a` = a GEMV with the residual epilogue and `b` = the LayerNorm + Linear that
 * reads a's output, with the all-to-all edge between them inside the launch (csrc/gemv.hip gemv_pair_kernel / gemv_pair_merge_kernel:
 * tagged 8-byte granules, write-through stores, one gather round trip). Two forms:
 *   a = FFN2 (models/modules/transformer.py:386-388 linear2 + the residual add at :328-329), b = the next layer's packed QKV projection
 *       (activation.py:86, with the KV append) or the first Linear of the prediction heads (models/ssr.py:175-179);
 *   a = split-KV merge + out-projection (activation.py:637) + residual, b = LayerNorm + linear1 + ReLU (transformer.py:386-388).
 * Results are bit-identical to ssrhip_gemv(a) followed by ssrhip_gemv(b).
 *   returns 0 = launched, 1 = this (a, b) does not qualify (nothing launched: call ssrhip_gemv twice), < 0 = error.
 *   Qualifies: B == 2 rows; a: ACT_NONE / EPI_RESIDUAL, N == 2048, and either PRO_NONE with K == 8192 or PRO_ATTN_COMBINE with K == 2048;
 *   b: PRO_LAYERNORM with folded gamma / beta, K == 2048, x == a->y, EPI_STORE or EPI_QKV_APPEND, N in {4096, 6144, 8192} (8192 only
 *   behind the merge form); >= 256 CUs; SSRHIP_GEMV_PAIR (0 = never, 1 = only the FFN2 form).
 *   ws: SSRHIP_PAIR_WS_BYTES of device memory, zeroed once by the caller and then owned by the chain of pair launches: three granule
 *   buffers + the give-up flag. `buf` is the buffer this launch uses, `buf_next` (!= buf) the one the NEXT pair launch on this workspace
 *   will use — this launch resets it. Consecutive pair launches must therefore follow each other's buf_next, cyclically.
 *   The launch needs its 256 workgroups resident together (ssrhip_gemv_pair_applicable also asks the occupancy calculator that each pair
 *   kernel fits a CU and refuses when a CU mask is set in the environment); a workgroup that waits longer than ~1 s for the others
 *   gives up, sets the flag and the outputs are garbage: ssrhip_gemv_pair_status (synchronises `stream`) returns 1 ONCE and re-zeroes the
 *   workspace (tags and flag), so the chain can be started again from its first launch. Who may pair at all on a device is decided one
 *   level up, per decode engine: ssrhip_lm_create below. */
#define SSRHIP_PAIR_WS_BYTES (3 * 4096 * 8 + 64)
int ssrhip_pair_buffer(int32_t i, int32_t n);   /* granule buffer (0..2) of the i-th of n cyclically consecutive pair launches; -1: bad i / n < 2 */
int ssrhip_gemv_pair_applicable(const ssrhip_gemv_args* a, const ssrhip_gemv_args* b);
int ssrhip_gemv_pair(const ssrhip_gemv_args* a, const ssrhip_gemv_args* b, void* ws, int32_t buf, int32_t buf_next, ssrhip_stream_t stream);
int ssrhip_gemv_pair_status(const void* ws, ssrhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Single-query attention over the paged cache, split over pages (one workgroup per page):
 * replaces F.scaled_dot_product_attention at activation.py:634 for tgt_len==1 (and, row by row,
 * the causal prefill).  Row r attends to positions [0, row_len[r]) of sequence row_seq[r]
 * (row_seq==NULL -> r).  Writes per-page partials (o, m, l); combine via ssrhip_gemv PRO_ATTN_COMBINE
 * or ssrhip_attn_combine.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ssrhip_attn_args {
  const float* q;          /* [R][q_stride] (first n_head*head_dim floats of each row) */
  int32_t q_stride;        /* 0 -> n_head*head_dim */
  ssrhip_kv kv; int32_t layer;
  const int32_t* row_seq;  /* [R] or NULL */
  const int32_t* row_len;  /* [R] */
  int32_t R, max_splits;   /* max_splits >= ceil(max(row_len)/SSRHIP_PAGE) */
  float scale;             /* 1/sqrt(head_dim) */
  float* part_o; float* part_ml;
  int32_t out_tiled;       /* ssrhip_attn_combine only: write `out` in the SSRHIP_TILED layout (R <= 16) */
  /* ssrhip_attn_decode only, optional (NULL = off; an experiment that did not pay, DESIGN.md Part I.5): while this latency-bound launch runs, workgroup i (linear launch index, the first 256)
   * touches floats [i * prefetch_floats, (i + 1) * prefetch_floats) of `prefetch` with plain loads whose results nobody reads — the weights
   * the NEXT launch's workgroup i will stream (workgroup i of both launches runs on XCD i % 8, so they land in the right L2). */
  const float* prefetch; int32_t prefetch_floats;
} ssrhip_attn_args;

int ssrhip_attn_decode(const ssrhip_attn_args* a, ssrhip_stream_t stream);
int ssrhip_attn_combine(const ssrhip_attn_args* a, float* out /* [R][n_head*head_dim] */, ssrhip_stream_t stream);
/* The same attention (activation.py:634, tgt_len == 1) WITHOUT splitting over pages: one workgroup per (row, head) walks the
 * row's pages with an online softmax and writes the normalised output row directly (part_o / part_ml unused, may be NULL;
 * `out` must not alias q; out_tiled as for ssrhip_attn_combine). Meant for many rows (rows x heads >= the number of CUs): the
 * 5..16-row decode step. */
int ssrhip_attn_rows(const ssrhip_attn_args* a, float* out /* [R][n_head*head_dim] */, ssrhip_stream_t stream);
/* Causal attention of whole PROMPTS (activation.py:634 with the mask of ssr.py:227-255, tgt_len = prompt length): the rows of
 * sequence s are rows seq_start[s] .. seq_start[s+1]-1 of q / out (positions 0 .. len-1, in order); K/V are read from the paged
 * cache (already scattered there), a query at position i sees keys 0..i. K/V tiles are staged in LDS once per 128 queries and
 * both products run on the matrix core (fp32). Uses a->q, q_stride, kv, layer, scale; `seq_start` is a DEVICE array of n_seq+1
 * ints, `max_len` an upper bound of the sequence lengths (grid size). out is row-major [R][n_head*head_dim]. Segment s reads the cache
 * of sequence a->row_seq[seq_start[s]] (row_seq == NULL: sequence s), so a SUBSET of an engine's rows can be prefilled. */
int ssrhip_attn_prefill(const ssrhip_attn_args* a, const int32_t* seq_start, int32_t n_seq, int32_t max_len, float* out,
                        ssrhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Token embedding + sinusoidal position: replaces embed_y (models/ssr.py:191-198, :655-660, :757-761),
 * TokenEmbedding / SinePositionalEmbedding.forward (models/modules/embedding.py:44-48, 94-97).
 *   kind[r]==0: text row   x = text_emb[tok[r][0]] + alpha_text * pe[pos[r]]
 *   kind[r]==1: audio row  x = sum_k audio_emb[k][tok[r][k]] + alpha_audio * pe[pos[r]]
 * ---------------------------------------------------------------------------------------------- */
typedef struct ssrhip_embed_args {
  const float* text_emb;   /* [n_text][D] */
  const float* audio_emb;  /* [K][card][D] */
  const float* pe;         /* [max_pos][D] */
  float alpha_text, alpha_audio;
  const int32_t* tok;      /* [R][SSRHIP_MAX_CODEBOOKS] */
  const int32_t* pos;      /* [R] */
  const int32_t* kind;     /* [R] or NULL (all audio) */
  int32_t R, D, K, card;
  float* out;              /* [R][D] */
  int32_t out_tiled;       /* write `out` in the SSRHIP_TILED layout (R <= 16) */
} ssrhip_embed_args;

int ssrhip_embed(const ssrhip_embed_args* a, ssrhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Per-step sampler + logit state machine, one workgroup per utterance, no host sync:
 * replaces models/ssr.py:689-754 (CFG combine, special-token edits, eog cascade, silence penalty,
 * topk_sampling/top_k_top_p_filtering :26-86, stop rules, span hand-over :646-660).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ssrhip_sampler_cfg {     /* per utterance, device memory, read-only during decode */
  int32_t top_k; float top_p; float temperature; int32_t stop_repetition;
  float cfg_coef; float cfg_one_minus; /* fp32(1 - cfg_coef) computed in double like the reference :692 */
  int32_t cfg_stride; int32_t use_cfg;
  int32_t n_silence; int32_t silence[SSRHIP_MAX_SILENCE];
  int32_t text_len;        /* L of the (doubled) text batch, for the 10*L cap :739 */
  int32_t n_spans;         /* num_task */
  int32_t empty_token, eog, eos, sos, mts, max_n_spans;
  int32_t max_steps;       /* capacity of `generated` per utterance */
  uint32_t seed_lo, seed_hi; /* on-device RNG stream when noise==NULL or use_noise==0 */
  int32_t use_noise;       /* 1: take the multinomial's Exp(1) draws from `noise` (host-drawn, reproduces torch's CPU stream);
                              0: on-device hash RNG. Lets one engine keep ONE persistent noise buffer (stable pointer, no graph
                              re-capture) whether or not a given generation uses it. */
} ssrhip_sampler_cfg;

typedef struct ssrhip_sampler_state {   /* per utterance, device memory, mutated every step */
  int32_t span;            /* current span index */
  int32_t num_gen, num_eog, num_cfg_tag, prev_token, consec_silence;
  int32_t audio_pos;       /* position of the token being fed this step (y_input.shape[1]-1) */
  int32_t n_steps;         /* samples written so far (all spans) */
  int32_t done;            /* 1 when all spans finished (or max_steps hit: 2) */
  int32_t span_end[3];     /* n_steps at the end of each finished span */
  int32_t pad[3];
} ssrhip_sampler_state;

typedef struct ssrhip_sample_args {
  const float* logits;     /* [B][K][card]; rows 2u,2u+1 when use_cfg else row u */
  int32_t n_utt, K, card;
  const ssrhip_sampler_cfg* cfg;
  ssrhip_sampler_state* state;
  const float* noise;      /* [n_utt][max_steps][K][card] Exp(1) draws, or NULL */
  int32_t* generated;      /* [n_utt][max_steps][K] */
  int32_t* next_tok;       /* [B][SSRHIP_MAX_CODEBOOKS] token ids fed to ssrhip_embed next step */
  int32_t* next_pos;       /* [B] audio position of the next input */
  int32_t* kv_pos;         /* [B] incremented for live utterances */
  int32_t* row_len;        /* [B] = kv_pos+1 */
  float* dbg_logits;       /* optional [n_utt][K][card] post-edit logits (tests) or NULL */
  ssrhip_embed_args embed; /* embed.out != NULL: also write the next input rows x[b] = embed(next_tok) (fused, saves a launch) */
} ssrhip_sample_args;

int ssrhip_sample(const ssrhip_sample_args* a, ssrhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Dense fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 FMA chain) for the
 * prefill rows and the codec:  C[M][N] = epi( A[M][K] . W[N][K]^T + bias[N] ), act/residual as GEMV.
 * replaces the same F.linear call sites as ssrhip_gemv when tgt_len > 4.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ssrhip_gemm_args {
  const float* A; const float* W; const float* bias; float* C;
  int32_t M, N, K, lda, ldc;
  int32_t act, residual;   /* residual: C += ... (in place) */
  /* --- extensions used by the codec (all zero = plain GEMM) --- */
  int32_t act_in;          /* SSRHIP_ACT_ELU: apply ELU(alpha=1) to A while staging it (conv input activation) */
  const float* R; int32_t ldr;      /* R != NULL: C = epi(...) + R[m*ldr+n]  (SEANet residual block skip) */
  int32_t batch;           /* grid.z; 0/1 = single problem */
  int64_t strideA, strideC, strideR; /* per-batch element strides */
  int32_t tm_c, tm_lo, tm_hi;       /* tm_c > 0: element (m,n) belongs to time row u = (m*N+n)/tm_c and is stored only
                                       if tm_lo <= u < tm_hi (transposed-conv trimming, modules/conv.py:236-243) */
  /* per-row CLASS bias (all NULL/0 = off): C[m][n] += rbias[rclass[z * rclass_stride + m / rrep] * N + n]  (z = batch index).
   * The watermark decoder's label conditioning (modules/seanet.py:577-591): the 1x1 convolution over cat(skip, embed(label))
   * splits into W_a . ELU(skip) + (W_b . ELU(embed(label))), and the second term takes only as many values as there are labels —
   * so the concatenated tensor is never built and the GEMM's K shrinks by the embedding width. */
  const float* rbias; const int32_t* rclass; int32_t rrep, rclass_stride;
  /* optional (NULL = always the fp32 FMA chain): W as three bf16 planes [3][N][K] made by ssrhip_split_weights. When given — and the
   * problem is large enough, N > 64, K % 8 == 0 — the GEMM runs on the bf16 matrix cores with every fp32 operand split EXACTLY into
   * three bf16 pieces and the six largest cross products accumulated in fp32 (csrc/gemm_split.hip): fp32 accuracy (error against an
   * fp64 reference no larger than the fp32 chain's), ~1.3-1.5x the speed, NOT bit-identical to the k-ordered fp32 chain. The codec
   * passes it (parity bar: waveform tolerance); the LM prefill does not (greedy tokens are compared bit for bit). */
  const uint16_t* W_split;
  /* SSRHIP_ACT_ELU: apply ELU(alpha = 1) LAST — after act, residual, R and the class bias — i.e. store what the consumer would compute
   * on load. For tensors that are only ever read through ELU (SEANet: a residual block's output feeds `ELU -> conv`, seanet.py:137-141,
   * 236-247) the activation then runs once per element in the producer instead of once per (tap, column block) in every consumer. */
  int32_t act_out;
} ssrhip_gemm_args;
int ssrhip_gemm(const ssrhip_gemm_args* a, ssrhip_stream_t stream);
/* W fp32 [n_elems] -> out bf16 [3][n_elems]: piece p of element i at out[p * n_elems + i], w = w0 + w1 + w2 exactly
 * (w0 = bf16_rne(w), w1 = bf16_rne(w - w0), w2 = bf16_rne(w - w0 - w1)). One-time preparation of a weight matrix for W_split. */
int ssrhip_split_weights(const float* W, uint16_t* out, int64_t n_elems, ssrhip_stream_t stream);


/* ------------------------------------------------------------------------------------------------
 * Codec (watermarked Encodec) kernels. Activations are TIME-MAJOR fp32: one item = [rows][C] with C contiguous,
 * so a Conv1d (kernel k, stride s) over a zero/reflect-padded buffer is a plain GEMM on a strided view:
 * A row t = k consecutive time rows = k*C contiguous floats starting at padded row t*s (lda = s*C), W repacked to
 * [Cout][k][Cin]; a ConvTranspose1d (k = 2s) is ONE GEMM with N = s*Cout (the s output phases side by side) and
 * K = 2*Cin, trimmed by the time mask. replaces StreamableConv1d / StreamableConvTranspose1d
 * (audiocraft/modules/conv.py:185-243) and SEANetResnetBlock (modules/seanet.py:16-60) via ssrhip_gemm.
 * ---------------------------------------------------------------------------------------------- */
/* first conv of SEANet (Cin == 1): out[b][t][co] = bias[co] + sum_kk w[co][kk] * x[b][t*stride + kk]  (x pre-padded) */
int ssrhip_conv_cin1(const float* x, const float* w, const float* bias, float* out, int32_t B, int32_t T_out, int32_t k,
                     int32_t stride, int32_t Cout, int64_t x_bstride, int64_t out_bstride, ssrhip_stream_t stream);
/* convolution with at most 4 OUTPUT channels on a time-major, pre-padded input (SEANet's last layer, 64 -> 1, k = 7; conv.py:185-201):
 * out[b][t][co] = bias[co] + sum_{kk, ci} w[co][kk][ci] * act_in(x[b][t + kk][ci]);  w is [Cout][k][Cin]; act_in NONE or ELU */
int ssrhip_conv_few_out(const float* x, const float* w, const float* bias, float* out, int32_t B, int32_t T_out, int32_t k,
                        int32_t Cin, int32_t Cout, int32_t act_in, int64_t x_bstride, int64_t out_bstride, ssrhip_stream_t stream);
/* reflect padding of a time-major buffer (conv.py:71-88): rows [0,padL) and [padL+T, padL+T+padR) mirror the interior */
int ssrhip_pad_reflect(float* buf, int32_t B, int32_t T, int32_t padL, int32_t padR, int32_t C, int64_t bstride,
                       ssrhip_stream_t stream);
/* halo rows of a RAGGED batch in a time-major buffer [B][padL + T + padR][C] whose producer wrote T rows for every item although
 * item b holds only lens[b] <= T valid ones (device array): the padR rows right behind item b's last valid row (rows padL + lens[b] ..,
 * partly inside the dense interior) and, for reflect, its padL leading rows get the padding an item of that length ALONE would have —
 * zeros (reflect == 0) or the mirror image about its own ends (conv.py:71-88, incl. the zero extension of inputs shorter than the
 * pad). Every layer of the batch then computes, for t < its own length, exactly what a batch-1 call computes (the SEANet
 * convolutions are not causal, so dense zero padding of a short item would change its tail; wmencodec.py:341-375 per item). */
int ssrhip_pad_ragged(float* buf, const int32_t* lens, int32_t B, int32_t T, int32_t padL, int32_t padR, int32_t C, int64_t bstride,
                      int32_t reflect, ssrhip_stream_t stream);
/* one LSTM layer over T steps (torch.nn.LSTM semantics, gates i,f,g,o; modules/lstm.py:10-25):
 * gin[b][t][4C] = x_t W_ih^T + b_ih + b_hh (precomputed by ssrhip_gemm); h,c start at 0.
 * out[b][t][C] = h_t (+ skip[b][t][C] when skip != NULL).  cbuf: [B][C]; hbuf: [2][ceil(B/16)*16][C] floats (row-major
 * [B][C] on the small-batch path B <= 4 && C in {256,512,1024,2048}; otherwise kept in the SSRHIP_TILED layout per 16-item
 * tile, which needs C <= 1024); gates: unused (may be NULL). C % 16 == 0. */
typedef struct ssrhip_lstm_args {
  const float* gin; const float* w_hh; float* out; const float* skip;
  float* hbuf; float* cbuf; float* gates;
  int32_t B, T, C;
  int64_t gin_bstride, out_bstride, skip_bstride;   /* element strides between items */
  /* time window of this call: steps [t_begin, t_end) of the T-step sequence (t_end == 0 means T). t_begin == 0 starts from
   * h = c = 0; a later window continues from the state the previous call left in hbuf / cbuf. Lets a caller run two stacked
   * layers as a software pipeline over chunks of steps on two streams (layer 2 chunk i beside layer 1 chunk i+1). */
  int32_t t_begin, t_end;
  /* matrix-core path only (B > 4, or C not in {256,512,1024,2048}): w_hh is given as 16 x 16 blocks in lane order,
   * packed[C/4 tiles][C/16 k-steps][4 k-slots][16 rows][4 floats] with row r of tile j = W_hh[(r % 4) * C + 4j + r / 4] (gate r % 4 of
   * hidden unit 4j + r / 4), so that one wave-level load is one contiguous KiB instead of 64 pieces of 16 rows */
  int32_t w_packed;
  int32_t out_act;   /* SSRHIP_ACT_ELU: out = ELU(h_t (+ skip)) (the layer's only consumer is `ELU -> conv`: seanet.py:147-150, 226-236) */
  /* optional (both or none; used when C % 128 == 0 and B >= 32): the recurrent product on the bf16 matrix cores with exactly split fp32
   * operands (csrc/lstm_split.hip), both operands in MFMA fragment order — KS = C/64 k-steps of 16 per wave, a block = 64 lanes x 8 bf16:
   *   w_split [C/16][4][KS][2][3][64][8]: lane l (li = l % 32, lh = l / 32), element e of block (ub, w, s, mb, q) = piece q
   *            (ssrhip_split_weights) of w_hh[g C + 16 ub + u][w C/4 + 16 s + 8 lh + e] with 32 mb + li = 16 g + u;
   *   hsplit  workspace, 2 x ceil(B/64) x 64 x C x 3 bf16 (two buffers by step parity): h_t in the same order, written and zeroed by
   *            the library. `hbuf` is not used on this path (it still has to be a valid pointer). */
  const uint16_t* w_split; uint16_t* hsplit;
} ssrhip_lstm_args;
int ssrhip_lstm_layer(const ssrhip_lstm_args* a, ssrhip_stream_t stream);
/* residual vector quantisation (quantization/core_vq.py:164-179, 382-400): emb [B][T][D] time-major;
 * codebooks [n_q][bins][D]; e2 [n_q][bins] = |e|^2; codes int32 [B][n_q][T] */
int ssrhip_rvq_encode(const float* emb, const float* codebooks, const float* e2, int32_t* codes, int32_t B, int32_t T,
                      int32_t D, int32_t n_q, int32_t bins, int64_t emb_bstride, ssrhip_stream_t stream);
int ssrhip_rvq_decode(const int32_t* codes, const float* codebooks, float* out, int32_t B, int32_t T, int32_t D,
                      int32_t n_q, int32_t bins, int64_t out_bstride, ssrhip_stream_t stream);
/* Fused SEANetResnetBlock (modules/seanet.py:16-60; true_skip, dilation 1, kernel sizes 3 and 1) for C in {64, 128, 256, 512}
 * channels (64: one persistent kernel with W3 resident in LDS; wider: two chained GEMMs with the C/2 intermediate kept in LDS):
 *   y[b][t][:] = x[b][t][:] + b1 + W1 . ELU(b3 + W3 . ELU(x[b][t-1 : t+2][:]))
 * x points at the row BEFORE t = 0 of item 0 of a time-major buffer [B][1 + T + 1][C] (halo rows hold the zero / reflect
 * padding); w3 is [C/2][3][C] (output channel, tap, input channel), w1 is [C][C/2]; y points at row t = 0. */
typedef struct ssrhip_resblock_args {
  const float* x; float* y;
  const float* w3; const float* b3; const float* w1; const float* b1;
  int32_t B, T, C;
  int64_t x_bstride, y_bstride;   /* elements between items */
  int32_t out_act;                /* SSRHIP_ACT_ELU: y = ELU(block(x)): the block's only consumer is `ELU -> conv` */
  /* optional (both or none; C in {64, 128}): the two weight matrices as three bf16 planes each (ssrhip_split_weights) — the block then runs
   * on the bf16 matrix cores with exactly split fp32 operands, weights streamed global -> LDS by DMA (csrc/resblock_split.hip).
   *   w3_split: planes of w3 as it is, [3][C/2][3*C];
   *   w1_split: planes of w1 with its COLUMNS permuted into the order the kernel's stage-1 accumulator hands the hidden channels on:
   *             column k' = 16 j + 8 g + i (j = k'/16, g = (k'/8) % 2, i = k' % 8) holds hidden channel h = 16 j + (i % 4) + 8 (i / 4) + 4 g. */
  const uint16_t* w3_split; const uint16_t* w1_split;
} ssrhip_resblock_args;
int ssrhip_resblock(const ssrhip_resblock_args* a, ssrhip_stream_t stream);

/* LayerNorm over rows (transformer.py:58-75) */
int ssrhip_layernorm(const float* x, const float* w, const float* b, float eps, float* y, int32_t R, int32_t D,
                     ssrhip_stream_t stream);

/* Scatter the k/v thirds of a packed qkv buffer [R][3D] into the paged cache; row r -> sequence
 * row_seq[r], position row_pos[r].  (activation.py:626-631 for the prefill rows) */
int ssrhip_kv_scatter(const float* qkv, const ssrhip_kv* kv, int32_t layer, const int32_t* row_seq,
                      const int32_t* row_pos, int32_t R, ssrhip_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Decode engine: one LM, B rows; prefill + N decode steps replayed from a captured hipGraph.
 * replaces the body of SSR_Speech.inference (models/ssr.py:597-754) on the device.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ssrhip_lm_weights {      /* device pointers; per-layer arrays have n_layer entries (host arrays) */
  const float* text_emb; const float* audio_emb; const float* pe; float alpha_text, alpha_audio;
  const float* const* ln1_w; const float* const* ln1_b;
  const float* const* in_proj_w; const float* const* in_proj_b;
  const float* const* out_proj_w; const float* const* out_proj_b;
  const float* const* ln2_w; const float* const* ln2_b;
  const float* const* ffn1_w; const float* const* ffn1_b;
  const float* const* ffn2_w; const float* const* ffn2_b;
  const float* lnf_w; const float* lnf_b;
  const float* head1_w; const float* head1_b;   /* [K*Hh][D], [K*Hh]   (predict_layer.k.0 stacked) */
  const float* head2_w; const float* head2_b;   /* [K][card][Hh], [K][card] (predict_layer.k.2 stacked) */
  /* optional second copy of the six matrices in the streaming order of the 5..16-row GEMV (SSRHIP_WTILED_INDEX; all NULL = absent).
   * Used by the decode steps of engines with more than 4 rows; the prefill GEMMs and the <= 4-row GEMV read the [N][K] copies. */
  const float* const* in_proj_wt; const float* const* out_proj_wt; const float* const* ffn1_wt; const float* const* ffn2_wt;
  const float* head1_wt; const float* head2_wt;
  /* optional (all four or none; NULL = the fp32 FMA chain): the four matrices of every layer as three bf16 planes each (ssrhip_split_weights)
   * for the PREFILL GEMMs (ssrhip_lm_prefill: tgt_len = prompt length): fp32 operands split exactly, six cross products on the bf16 matrix
   * cores (ssrhip_gemm_args.W_split: error against fp64 no larger than the chain's, not bit-identical to it). The decode step never reads them. */
  const uint16_t* const* in_proj_ws; const uint16_t* const* out_proj_ws; const uint16_t* const* ffn1_ws; const uint16_t* const* ffn2_ws;
} ssrhip_lm_weights;

typedef struct ssrhip_lm_dims {
  int32_t d_model, n_head, n_layer, d_ffn, n_codebooks, card, head_hidden, n_text, max_pos;
  int32_t ln_folded;   /* 1: the LayerNorm affine (gamma, beta) of norm1/norm2/decoder.norm is already folded into in_proj/ffn1/head1
                          (W' = W diag(gamma), b' = b + W beta); ln*_w / ln*_b then hold ones / zeros (used by the prefill LayerNorm) */
} ssrhip_lm_dims;

typedef struct ssrhip_lm_buffers {      /* caller-allocated device workspaces */
  int32_t B, n_utt, max_splits;
  int32_t pair_mode;   /* 2-row engines: 0 = pair launches if this engine may hold the device's pairing slot (ssrhip_lm_create), 1 = never,
                          2 = always (tests of the give-up path: no slot taken, no guard) */
  float* x;        /* [B][D] residual stream */
  float* q;        /* [B][D] */
  float* h;        /* [B][max(d_ffn, K*Hh)] */
  float* logits;   /* [B][K][card] */
  float* part_o;   /* [B][H][max_splits][hd] */
  float* part_ml;  /* [B][H][max_splits][2] */
  int32_t* next_tok; int32_t* next_pos; int32_t* kv_pos; int32_t* row_len;
  ssrhip_kv kv;
  ssrhip_sampler_cfg* cfg; ssrhip_sampler_state* state;
  const float* noise; int32_t* generated; float* dbg_logits;
} ssrhip_lm_buffers;

typedef struct ssrhip_lm ssrhip_lm;     /* opaque host-side object (graph + launch descriptors) */

/* A 2-row engine (one utterance x CFG) runs its step with pair launches (ssrhip_gemv_pair) only while it holds the PAIRING SLOT of its
 * device: two pair chains on one GPU can starve each other of CUs (each launch spins until all 256 of its workgroups are resident).
 * The slot is taken here and given back by ssrhip_lm_destroy: at most one live engine per process and device (a table in the library)
 * and at most one process per device (an exclusive flock on /dev/shm/ssrhip_pair_<pci bus id>.lock, released by the kernel when the
 * process dies). An engine that does not get the slot — or finds a CU mask in the environment, fewer than 256 CUs, a pair kernel that
 * does not fit a CU, SSRHIP_GEMV_PAIR=0, pair_mode 1 — steps with the ordinary launches (same tokens, bit for bit) and says why once
 * on stderr; ssrhip_lm_pairing reports it. */
int ssrhip_lm_create(const ssrhip_lm_dims* d, const ssrhip_lm_weights* w, const ssrhip_lm_buffers* b, ssrhip_lm** out);
/* 1 = this engine's decode step uses pair launches, 0 = it does not; `why` (may be NULL) receives the reason as text */
int ssrhip_lm_pairing(const ssrhip_lm* lm, char* why, int32_t why_len);
void ssrhip_lm_destroy(ssrhip_lm* lm);
/* enqueue `n_steps` decode steps (graph replays when use_graph!=0) */
int ssrhip_lm_decode(ssrhip_lm* lm, int32_t n_steps, int32_t use_graph, ssrhip_stream_t stream);
/* prefill R rows ([text || audio] of every sequence, flattened): fills the cache for all layers.
 * ws: workspace floats: x[R][D], xn[R][D], qkv[R][3D], o[R][D], h[R][d_ffn], part_o, part_ml sized for R. */
typedef struct ssrhip_prefill_args {
  const int32_t* tok; const int32_t* pos; const int32_t* kind;   /* embed inputs, [R].. */
  const int32_t* row_seq; const int32_t* row_pos; const int32_t* row_len; /* cache coordinates, [R] */
  int32_t R, max_splits;
  float* x; float* xn; float* qkv; float* o; float* h; float* part_o; float* part_ml;
  /* optional (NULL/0 = per-row attention through part_o / part_ml): the rows of sequence s are seq_start[s] .. seq_start[s+1]-1,
   * in position order -> the tiled prefill attention (ssrhip_attn_prefill); part_o / part_ml may then be NULL */
  const int32_t* seq_start; int32_t n_seq, max_len;
  /* two-phase admission (continuous batching: a new utterance is prefilled on a SIDE stream while the graph keeps stepping the live rows):
   * `table` != NULL replaces the engine's page table for THIS prefill (same layout) — the rows being filled are still parked on the
   * scratch page in the table the decode step reads; `no_embed` != 0 skips the closing embedding of the rows' pending tokens into the
   * residual stream (it is live decode state): the caller issues ssrhip_lm_embed_pending on the decode stream when it activates the rows. */
  const int32_t* table; int32_t no_embed, reserved_;
} ssrhip_prefill_args;
int ssrhip_lm_prefill(ssrhip_lm* lm, const ssrhip_prefill_args* p, ssrhip_stream_t stream);
/* x[b] = embedding of row b's pending input token (next_tok / next_pos) for EVERY row of the engine — the closing step of ssrhip_lm_prefill
 * on its own (rows in mid-decode get exactly what the sampler's fused embedding left there: same function, same inputs). */
int ssrhip_lm_embed_pending(ssrhip_lm* lm, ssrhip_stream_t stream);
/* 0 = fine, 1 = a paired GEMV launch of this engine's decode step gave up waiting (ssrhip_gemv_pair): every token since the last call
 * that returned 0 is invalid and so are the KV cache and the residual stream of the rows in flight; reported ONCE (the workspace is
 * re-zeroed: after a new prefill the engine decodes correctly again). Synchronises `stream`. Engines that do not pair always return 0
 * without touching the stream. */
int ssrhip_lm_pair_status(ssrhip_lm* lm, ssrhip_stream_t stream);

/* Test hook (tests/test_gpu_lm.py: the pair launches' give-up path): `n_wg` workgroups that each hold `lds_bytes` of LDS (<= 160 KB) and
 * spin for `ms` milliseconds of the constant 100 MHz clock — a foreign kernel that keeps CUs away from everybody else. `started` (may be
 * NULL): an int32 the device can reach (pinned host memory), incremented once by every workgroup when it has become resident, so that the
 * caller can wait for all of them before it starts what the squatters are meant to starve. */
int ssrhip_debug_occupy(int32_t n_wg, int32_t lds_bytes, float ms, int32_t* started, ssrhip_stream_t stream);

/* run `n_steps` eager decode steps with a hipEvent pair around EVERY kernel launch (bench.py roofline).
 * out_us[i] = average microseconds of launch slot i of a step, out_kind[i] = 0 gemv | 1 attention | 2 sampler;
 * returns the number of slots (<= n_out) or a negative error. */
int ssrhip_lm_time_steps(ssrhip_lm* lm, int32_t n_steps, ssrhip_stream_t stream, float* out_us, int32_t* out_kind, int32_t n_out);

/* Launch duration of ONE kernel category inside a decode step, measured the way the product runs it: the launches of
 * `category` (0 gemv | 1 attention | 2 sampler) of one step are captured into a hipGraph (the others are left out), the
 * graph is replayed `n_replays` times between one hipEvent pair on `stream`, and the elapsed time is divided by the number
 * of launches. No per-launch event overhead, so the figure agrees with rocprofv3's kernel durations (+ the ~0.1 us
 * in-graph gap). The decode state is NOT advanced and the hidden-state buffers are left with garbage: call
 * ssrhip_lm_prefill (DecodeEngine.start) again before decoding. */
int ssrhip_lm_time_category(ssrhip_lm* lm, int32_t category, int32_t n_replays, ssrhip_stream_t stream, float* out_us_per_launch,
                            int32_t* out_launches_per_step);

#ifdef __cplusplus
}
#endif
#endif /* SSRHIP_H */
