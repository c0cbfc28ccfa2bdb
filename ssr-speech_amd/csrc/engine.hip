// engine.hip — host side of the decode engine: launch descriptors for one decode step, hipGraph
// capture/replay, prefill orchestration, per-kernel event timing. No device allocation here: all
// buffers belong to the caller (torch tensors), the engine owns only the graph objects and a private
// capture stream.
//
// One decode step = embed -> n_layer x [LN1+QKV GEMV(+KV append) -> paged attention (split-KV partials)
// -> combine+out-proj GEMV(+residual) -> LN2+FFN1 GEMV(+ReLU) -> FFN2 GEMV(+residual)]
// -> final-LN + 4 head MLP-1 GEMV(+GELU) -> grouped head MLP-2 GEMV -> sampler/state machine.
// The step touches no host state, so the captured graph replays unchanged for every step
// (positions, tokens, stop flags all live in device memory).
// Replaces the per-iteration body of SSR_Speech.inference (models/ssr.py:671-770).
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <errno.h>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>
#include <mutex>
#include <vector>
#include "common.h"

static thread_local char g_err[512] = "";
void ssrhip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* ssrhip_last_error(void) { return g_err; }
extern "C" int ssrhip_version(void) { return SSRHIP_VERSION; }
extern "C" int ssrhip_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(ssrhip_kv);
    case 1: return sizeof(ssrhip_gemv_args);
    case 2: return sizeof(ssrhip_attn_args);
    case 3: return sizeof(ssrhip_embed_args);
    case 4: return sizeof(ssrhip_sampler_cfg);
    case 5: return sizeof(ssrhip_sampler_state);
    case 6: return sizeof(ssrhip_sample_args);
    case 7: return sizeof(ssrhip_gemm_args);
    case 8: return sizeof(ssrhip_lm_weights);
    case 9: return sizeof(ssrhip_lm_dims);
    case 10: return sizeof(ssrhip_lm_buffers);
    case 11: return sizeof(ssrhip_prefill_args);
    case 12: return sizeof(ssrhip_lstm_args);
    case 13: return sizeof(ssrhip_resblock_args);
    default: return -1;
  }
}

struct ssrhip_lm {
  ssrhip_lm_dims d;
  ssrhip_lm_weights w;
  ssrhip_lm_buffers b;
  std::vector<const float*> ptrs[16];
  std::vector<const uint16_t*> sptrs[4];
  bool prefill_split = false;   // bf16 planes present and SSRHIP_PREFILL_SPLIT != 0
  hipStream_t cap_stream = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  void* pair_ws = nullptr;      // granules of the paired GEMV launches (2-row step; SSRHIP_PAIR_WS_BYTES, owned)
  int pair_dev = -1;            // >= 0: this engine holds the pairing slot of that device (released in ssrhip_lm_destroy)
  char pair_why[200] = "";      // why the step pairs / does not pair (ssrhip_lm_pairing)
};

namespace {

enum { CAT_GEMV = 0, CAT_ATTN = 1, CAT_SAMPLE = 2 };

// ---- the pairing slot: ONE chain of pair launches per device (include/ssrhip.h ssrhip_lm_create). A pair launch spins until its 256
// workgroups are resident together; two such chains at once (two engines on two streams, two processes) can each hold half the CUs and
// wait for the other half — bounded (~1 s), flagged, but garbage. So the right to pair is a resource: a per-process table (one live
// engine per device) plus an exclusive flock on a file named after the GPU's PCI address (one process per device; the kernel drops the
// lock when the process dies, however it dies).
constexpr int PAIR_MAX_DEV = 64;
std::mutex g_pair_mu;
struct PairSlot { int live = 0; int fd = -1; };
PairSlot g_pair_slot[PAIR_MAX_DEV];

// true: slot taken (give it back with pair_slot_release); false: `why` says who has it / what went wrong
bool pair_slot_acquire(int dev, char* why, size_t why_len) {
  std::lock_guard<std::mutex> lk(g_pair_mu);
  if (dev < 0 || dev >= PAIR_MAX_DEV) { snprintf(why, why_len, "device index %d outside the pairing table", dev); return false; }
  PairSlot& sl = g_pair_slot[dev];
  if (sl.live > 0) { snprintf(why, why_len, "another decode engine of this process already runs pair launches on device %d", dev); return false; }
  char bus[64] = "";
  if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), dev) != hipSuccess || !bus[0]) snprintf(bus, sizeof(bus), "dev%d", dev);
  for (char* c = bus; *c; ++c) if (*c == ':' || *c == '/' || *c == '.') *c = '_';
  const char* dirs[2] = {"/dev/shm", "/tmp"};
  int fd = -1;
  char path[160] = "";
  for (int i = 0; i < 2 && fd < 0; ++i) {
    snprintf(path, sizeof(path), "%s/ssrhip_pair_%s.lock", dirs[i], bus);
    fd = open(path, O_RDWR | O_CREAT | O_CLOEXEC, 0666);
    if (fd >= 0) fchmod(fd, 0666);                                  // the next user's process must be able to open it too (umask)
    else fd = open(path, O_RDONLY | O_CLOEXEC);                     // someone else's file: a shared descriptor is enough for flock
  }
  if (fd < 0) { snprintf(why, why_len, "cannot create the pair-launch lock file (%s: %s)", path, strerror(errno)); return false; }
  if (flock(fd, LOCK_EX | LOCK_NB) != 0) {
    snprintf(why, why_len, "another process holds the pair-launch lock of this GPU (%s)", path);
    close(fd);
    return false;
  }
  sl.fd = fd;
  sl.live = 1;
  return true;
}

void pair_slot_release(int dev) {
  std::lock_guard<std::mutex> lk(g_pair_mu);
  if (dev < 0 || dev >= PAIR_MAX_DEV) return;
  PairSlot& sl = g_pair_slot[dev];
  if (sl.live > 0) sl.live = 0;
  if (sl.fd >= 0) { flock(sl.fd, LOCK_UN); close(sl.fd); sl.fd = -1; }
}

__global__ void occupy_kernel(long long ticks, int lds_floats, int* started) {
  extern __shared__ float occ[];
  for (int i = threadIdx.x; i < lds_floats; i += blockDim.x) occ[i] = (float)i;
  __syncthreads();
  if (started && threadIdx.x == 0) __hip_atomic_fetch_add(started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // "this workgroup is resident"
  const long long t0 = wall_clock64();
  float acc = 0.f;
  while (wall_clock64() - t0 < ticks) { acc += occ[(threadIdx.x * 7) % (lds_floats > 0 ? lds_floats : 1)]; __builtin_amdgcn_s_sleep(32); }
  if (acc == -1.f) occ[0] = acc;
}

bool getenv_flag(const char* name) {      // tuning / A-B knobs, read once per process
  const char* e = getenv(name);
  return e && e[0] && e[0] != '0';
}

struct Timer {   // optional per-launch event timing: one accumulator per launch slot of a step
  bool on = false;
  hipStream_t s = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int slot = 0;
  int only = -1;          // >= 0: enqueue only launches of this category (graph-chained category timing)
  int n_enq = 0;
  std::vector<float> ms;
  std::vector<int> kind;
};

#define STEP_CALL(cat, call)                                    \
  do {                                                          \
    if (tm && tm->only >= 0 && tm->only != (cat)) break;        \
    if (tm) tm->n_enq += 1;                                     \
    if (tm && tm->on) hipEventRecord(tm->e0, tm->s);            \
    int _rc = (call);                                           \
    if (_rc) return _rc;                                        \
    if (tm && tm->on) {                                         \
      hipEventRecord(tm->e1, tm->s);                            \
      hipEventSynchronize(tm->e1);                              \
      float _ms = 0.f;                                          \
      hipEventElapsedTime(&_ms, tm->e0, tm->e1);                \
      if ((int)tm->ms.size() <= tm->slot) { tm->ms.push_back(0.f); tm->kind.push_back(cat); } \
      tm->ms[tm->slot] += _ms;                                  \
      tm->slot += 1;                                            \
    }                                                           \
  } while (0)

// Launch descriptors of the five GEMVs of a decode step (shared by enqueue_step and by ssrhip_lm_create's "would this engine pair?")
struct StepShapes {
  const ssrhip_lm_dims& d;
  const ssrhip_lm_weights& w;
  const ssrhip_lm_buffers& b;
  const int D, B, K, Hh;
  const int tiled;     // 5..16 rows: x, the combined attention output and h live in the 16-column tiled layout (SSRHIP_TILED)
  const bool wt;       // streaming-order weight copies for the matrix-core GEMV (include/ssrhip.h w_tiled)
  explicit StepShapes(const ssrhip_lm* lm)
      : d(lm->d), w(lm->w), b(lm->b), D(lm->d.d_model), B(lm->b.B), K(lm->d.n_codebooks), Hh(lm->d.head_hidden), tiled(lm->b.B > 4 ? 1 : 0),
        wt(lm->b.B > 4 && lm->w.in_proj_wt) {}
  ssrhip_gemv_args qkv_args(int l) const {
    ssrhip_gemv_args g;
    // LN1 + packed QKV projection, q -> b.q, k/v appended in place into the paged cache
    memset(&g, 0, sizeof(g));
    g.W = w.in_proj_w[l]; g.bias = w.in_proj_b[l]; g.x = b.x; g.y = b.q;
    g.B = B; g.N = 3 * D; g.K = D; g.groups = 1; g.x_stride = D; g.y_stride = D;
    g.pro = SSRHIP_PRO_LAYERNORM; g.act = SSRHIP_ACT_NONE; g.epi = SSRHIP_EPI_QKV_APPEND;
    if (!d.ln_folded) { g.ln_w = w.ln1_w[l]; g.ln_b = w.ln1_b[l]; }
    g.ln_eps = 1e-5f;
    g.kv = b.kv; g.layer = l; g.kv_pos = b.kv_pos;
    g.x_tiled = tiled;                         // q stays row-major for the attention kernel
    if (wt) { g.W = w.in_proj_wt[l]; g.w_tiled = 1; }
    return g;
  }
  ssrhip_gemv_args head1_args() const {
    ssrhip_gemv_args g;
    // final LayerNorm + first Linear of the K prediction heads (stacked) + GELU
    memset(&g, 0, sizeof(g));
    g.W = w.head1_w; g.bias = w.head1_b; g.x = b.x; g.y = b.h;
    g.B = B; g.N = K * Hh; g.K = D; g.groups = 1; g.x_stride = D; g.y_stride = K * Hh;
    g.pro = SSRHIP_PRO_LAYERNORM; g.act = SSRHIP_ACT_GELU_ERF; g.epi = SSRHIP_EPI_STORE;
    if (!d.ln_folded) { g.ln_w = w.lnf_w; g.ln_b = w.lnf_b; }
    g.ln_eps = 1e-5f;
    g.x_tiled = tiled; g.y_tiled = tiled;
    if (wt) { g.W = w.head1_wt; g.w_tiled = 1; }
    return g;
  }
  ssrhip_gemv_args outproj_args(int l) const {               // the <= 4-row form: split-KV merge prologue + out-proj + residual
    ssrhip_gemv_args g;
    memset(&g, 0, sizeof(g));
    g.W = w.out_proj_w[l]; g.bias = w.out_proj_b[l]; g.x = nullptr; g.y = b.x;
    g.B = B; g.N = D; g.K = D; g.groups = 1; g.x_stride = D; g.y_stride = D;
    g.pro = SSRHIP_PRO_ATTN_COMBINE; g.act = SSRHIP_ACT_NONE; g.epi = SSRHIP_EPI_RESIDUAL;
    g.part_o = b.part_o; g.part_ml = b.part_ml; g.max_splits = b.max_splits; g.row_len = b.row_len;
    g.kv = b.kv;
    return g;
  }
  ssrhip_gemv_args ffn1_args(int l) const {
    ssrhip_gemv_args g;
    // LN2 + FFN1 + ReLU
    memset(&g, 0, sizeof(g));
    g.W = w.ffn1_w[l]; g.bias = w.ffn1_b[l]; g.x = b.x; g.y = b.h;
    g.B = B; g.N = d.d_ffn; g.K = D; g.groups = 1; g.x_stride = D; g.y_stride = d.d_ffn;
    g.pro = SSRHIP_PRO_LAYERNORM; g.act = SSRHIP_ACT_RELU; g.epi = SSRHIP_EPI_STORE;
    if (!d.ln_folded) { g.ln_w = w.ln2_w[l]; g.ln_b = w.ln2_b[l]; }
    g.ln_eps = 1e-5f;
    g.x_tiled = tiled; g.y_tiled = tiled;
    if (wt) { g.W = w.ffn1_wt[l]; g.w_tiled = 1; }
    return g;
  }
  ssrhip_gemv_args ffn2_args(int l) const {
    ssrhip_gemv_args g;
    memset(&g, 0, sizeof(g));
    g.W = w.ffn2_w[l]; g.bias = w.ffn2_b[l]; g.x = b.h; g.y = b.x;
    g.B = B; g.N = D; g.K = d.d_ffn; g.groups = 1; g.x_stride = d.d_ffn; g.y_stride = D;
    g.pro = SSRHIP_PRO_NONE; g.act = SSRHIP_ACT_NONE; g.epi = SSRHIP_EPI_RESIDUAL;
    g.x_tiled = tiled; g.y_tiled = tiled;
    if (wt) { g.W = w.ffn2_wt[l]; g.w_tiled = 1; }
    return g;
  }
  // which launches of the 2-row step qualify for the pair forms, and how many pair launches a step then has (0: fewer than 2 -> none)
  int pair_plan(bool* qkv, bool* head, bool* ffn1) const {
    *qkv = *head = *ffn1 = false;
    if (B != 2) return 0;
    const ssrhip_gemv_args fa = ffn2_args(0);
    if (d.n_layer > 1) { const ssrhip_gemv_args qa = qkv_args(1); *qkv = ssrhip_gemv_pair_applicable(&fa, &qa) != 0; }
    { const ssrhip_gemv_args ha = head1_args(); *head = ssrhip_gemv_pair_applicable(&fa, &ha) != 0; }
    { const ssrhip_gemv_args oa = outproj_args(0), f1 = ffn1_args(0); *ffn1 = ssrhip_gemv_pair_applicable(&oa, &f1) != 0; }
    const int n = (*qkv ? d.n_layer - 1 : 0) + (*head ? 1 : 0) + (*ffn1 ? d.n_layer : 0);
    if (n < 2) { *qkv = *head = *ffn1 = false; return 0; }
    return n;
  }
};

int enqueue_step(ssrhip_lm* lm, hipStream_t s, Timer* tm) {
  const ssrhip_lm_dims& d = lm->d;
  const ssrhip_lm_weights& w = lm->w;
  const ssrhip_lm_buffers& b = lm->b;
  const int D = d.d_model, B = b.B, K = d.n_codebooks, Hh = d.head_hidden;
  // 5..16 rows: the residual stream x, the combined attention output and the hidden h live in the 16-column tiled layout
  // (include/ssrhip.h SSRHIP_TILED) so that the matrix-core GEMV's operand loads are contiguous KiBs
  const int tiled = B > 4 ? 1 : 0;
  const bool wt = tiled && w.in_proj_wt;      // streaming-order weight copies for the matrix-core GEMV (include/ssrhip.h w_tiled)

  const StepShapes sh(lm);
  auto qkv_args = [&](int l) { return sh.qkv_args(l); };
  auto head1_args = [&]() { return sh.head1_args(); };
  auto ffn1_args = [&](int l) { return sh.ffn1_args(l); };
  auto ffn2_args = [&](int l) { return sh.ffn2_args(l); };
  // 2-row step: FFN2 of layer l and the launch that consumes its output (QKV of layer l + 1; the head MLP after the last layer) run as ONE
  // launch with the all-to-all edge inside it (csrc/gemv.hip gemv_pair_kernel), and so do the out-projection (with its split-KV merge
  // prologue) and FFN1 (gemv_pair_merge_kernel): attention, pair, pair per layer. The pairs of a step use the three granule buffers of
  // lm->pair_ws cyclically: pair i uses buffer i % 3 and resets the buffer of pair (i + 1) % n — closed over the step, so that graph
  // replays (and eager steps) always find their buffer reset by the launch before them; with n % 3 == 1 the last pair takes buffer 1.
  bool pair_qkv = false, pair_head = false, pair_ffn1 = false;
  const int n_pairs = lm->pair_ws ? sh.pair_plan(&pair_qkv, &pair_head, &pair_ffn1) : 0;
  auto pair_buf = [&](int i) { return ssrhip_pair_buffer(i, n_pairs); };
  int pair_i = 0;
  bool qkv_done = false;                        // this layer's QKV already ran inside the previous layer's pair launch

  if (tm) tm->slot = 0;

  for (int l = 0; l < d.n_layer; ++l) {
    ssrhip_gemv_args g;
    if (!qkv_done) {
      g = qkv_args(l);
      STEP_CALL(CAT_GEMV, ssrhip_gemv(&g, s));
    }
    qkv_done = false;

    ssrhip_attn_args at;
    memset(&at, 0, sizeof(at));
    at.q = b.q; at.kv = b.kv; at.layer = l; at.row_seq = nullptr; at.row_len = b.row_len;
    at.R = B; at.max_splits = b.max_splits; at.scale = 1.0f / sqrtf((float)(D / d.n_head));
    at.part_o = b.part_o; at.part_ml = b.part_ml;
    // 2-row paired step, OPT-IN experiment (SSRHIP_ATTN_PREFETCH=1, read when the step is enqueued): the attention launch pre-touches the
    // out-projection slice the merge pair launch behind it starts with (include/ssrhip.h ssrhip_attn_args.prefetch). Measured: the attention
    // launch pays for the 16.8 MB (6.55 -> 8.80 us) and the pair launch gains nothing (18.33 vs 18.35 us): 0.7477 -> 0.7707 ms/step
    // (profiles/r06_microbench/decode_ab_attn_prefetch.log). Like every cross-launch prefetch tried since round 1, it loses.
    if (pair_ffn1 && D % 256 == 0 && getenv("SSRHIP_ATTN_PREFETCH") && getenv("SSRHIP_ATTN_PREFETCH")[0] == '1') {
      at.prefetch = w.out_proj_w[l];
      at.prefetch_floats = (D / 256) * D;                    // workgroup i of the pair launch owns rows [8 i, 8 i + 8) of W_o [D][D]
    }
    // 5..16 rows with enough (row, head) pairs to give every CU one: the fused walk over the pages (no partials, no merge
    // launch); its output goes to b.h (free until FFN1 of this layer) because q is still being read by other workgroups
    const bool fused_attn = B > 4 && B * d.n_head >= 192 && b.kv.max_pages <= 256 && !getenv_flag("SSRHIP_ATTN_SPLIT");   // 256 pages: the kernel's page-id registers
    // split-KV combine + out-proj + residual
    memset(&g, 0, sizeof(g));
    g.W = w.out_proj_w[l]; g.bias = w.out_proj_b[l]; g.x = nullptr; g.y = b.x;
    g.B = B; g.N = D; g.K = D; g.groups = 1; g.x_stride = D; g.y_stride = D;
    g.pro = SSRHIP_PRO_ATTN_COMBINE; g.act = SSRHIP_ACT_NONE; g.epi = SSRHIP_EPI_RESIDUAL;
    g.part_o = b.part_o; g.part_ml = b.part_ml; g.max_splits = b.max_splits; g.row_len = b.row_len;
    g.kv = b.kv;
    if (fused_attn) {
      at.out_tiled = 1;
      STEP_CALL(CAT_ATTN, ssrhip_attn_rows(&at, b.h, s));
    } else {
      STEP_CALL(CAT_ATTN, ssrhip_attn_decode(&at, s));
    }
    if (B > 4) {
      if (fused_attn) {
        g.x = b.h;
      } else {
        // the combine is its own small launch; q is dead after the attention, reuse it
        at.out_tiled = 1;
        STEP_CALL(CAT_ATTN, ssrhip_attn_combine(&at, b.q, s));
        g.x = b.q;
      }
      g.pro = SSRHIP_PRO_NONE; g.x_tiled = 1; g.y_tiled = 1;
      if (wt) { g.W = w.out_proj_wt[l]; g.w_tiled = 1; }
    }
    if (pair_ffn1) {                              // out-projection + FFN1 as one launch (2 rows)
      const ssrhip_gemv_args f1 = ffn1_args(l);
      const int bufi = pair_buf(pair_i), bufn = pair_buf((pair_i + 1) % n_pairs);
      STEP_CALL(CAT_GEMV, ssrhip_gemv_pair(&g, &f1, lm->pair_ws, bufi, bufn, (ssrhip_stream_t)s));
      pair_i += 1;
    } else {
      STEP_CALL(CAT_GEMV, ssrhip_gemv(&g, s));
      g = ffn1_args(l);
      STEP_CALL(CAT_GEMV, ssrhip_gemv(&g, s));
    }

    // FFN2 + residual — paired with the next launch where that applies
    g = ffn2_args(l);
    const bool last = l == d.n_layer - 1;
    if (last ? pair_head : pair_qkv) {
      const ssrhip_gemv_args nx = last ? head1_args() : qkv_args(l + 1);
      const int bufi = pair_buf(pair_i), bufn = pair_buf((pair_i + 1) % n_pairs);
      STEP_CALL(CAT_GEMV, ssrhip_gemv_pair(&g, &nx, lm->pair_ws, bufi, bufn, (ssrhip_stream_t)s));
      pair_i += 1;
      qkv_done = true;                          // (after the last layer: the head MLP's first launch)
    } else {
      STEP_CALL(CAT_GEMV, ssrhip_gemv(&g, s));
    }
  }
  {
    ssrhip_gemv_args g;
    if (!qkv_done) {
      g = head1_args();
      STEP_CALL(CAT_GEMV, ssrhip_gemv(&g, s));
    }
    // second Linear of each head: K groups
    memset(&g, 0, sizeof(g));
    g.W = w.head2_w; g.bias = w.head2_b; g.x = b.h; g.y = b.logits;
    g.B = B; g.N = d.card; g.K = Hh; g.groups = K; g.x_stride = K * Hh; g.y_stride = K * d.card;
    g.pro = SSRHIP_PRO_NONE; g.act = SSRHIP_ACT_NONE; g.epi = SSRHIP_EPI_STORE;
    g.x_tiled = tiled;                         // logits stay row-major for the sampler
    if (wt) { g.W = w.head2_wt; g.w_tiled = 1; }
    STEP_CALL(CAT_GEMV, ssrhip_gemv(&g, s));
  }
  ssrhip_sample_args sa;
  memset(&sa, 0, sizeof(sa));
  sa.logits = b.logits; sa.n_utt = b.n_utt; sa.K = K; sa.card = d.card;
  sa.cfg = b.cfg; sa.state = b.state; sa.noise = b.noise; sa.generated = b.generated;
  sa.next_tok = b.next_tok; sa.next_pos = b.next_pos; sa.kv_pos = b.kv_pos; sa.row_len = b.row_len;
  sa.dbg_logits = b.dbg_logits;
  // the sampler also embeds the tokens it chose: x of the next step (no separate embed launch)
  sa.embed.text_emb = w.text_emb; sa.embed.audio_emb = w.audio_emb; sa.embed.pe = w.pe;
  sa.embed.alpha_text = w.alpha_text; sa.embed.alpha_audio = w.alpha_audio;
  sa.embed.R = B; sa.embed.D = D; sa.embed.K = K; sa.embed.card = d.card; sa.embed.out = b.x; sa.embed.out_tiled = tiled;
  STEP_CALL(CAT_SAMPLE, ssrhip_sample(&sa, s));
  return 0;
}

}  // namespace

extern "C" int ssrhip_lm_create(const ssrhip_lm_dims* d, const ssrhip_lm_weights* w, const ssrhip_lm_buffers* b, ssrhip_lm** out) {
  SSR_REQUIRE(d && w && b && out, "ssrhip_lm_create: null argument");
  SSR_REQUIRE(d->d_model % d->n_head == 0, "ssrhip_lm_create: d_model %% n_head != 0");
  const int hd = d->d_model / d->n_head;
  SSR_REQUIRE(hd == 64 || hd == 128, "ssrhip_lm_create: head_dim %d not in {64,128}", hd);
  SSR_REQUIRE(b->B == 1 || b->B == 2 || b->B == 4 || (b->B >= 5 && b->B <= 16), "ssrhip_lm_create: B=%d rows not in {1,2,4,5..16}", b->B);
  SSR_REQUIRE(b->B <= 4 || d->ln_folded, "ssrhip_lm_create: B > 4 rows needs LayerNorm gamma/beta folded into the weights (ln_folded)");
  SSR_REQUIRE(d->n_codebooks <= SSRHIP_MAX_CODEBOOKS, "ssrhip_lm_create: too many codebooks");
  // a row's text and audio positions are both below its sequence capacity: the sinusoidal table must cover it (embed reads pe[pos])
  SSR_REQUIRE((int64_t)b->kv.max_pages * SSRHIP_PAGE <= d->max_pos, "ssrhip_lm_create: sequence capacity %lld exceeds the position table (%d rows)",
              (long long)b->kv.max_pages * SSRHIP_PAGE, d->max_pos);
  ssrhip_lm* lm = new ssrhip_lm();
  lm->d = *d; lm->w = *w; lm->b = *b;
  // deep-copy the per-layer pointer arrays (the caller's ctypes arrays may be temporaries)
  const float* const** fields[12] = {&lm->w.ln1_w, &lm->w.ln1_b, &lm->w.in_proj_w, &lm->w.in_proj_b, &lm->w.out_proj_w, &lm->w.out_proj_b,
                                     &lm->w.ln2_w, &lm->w.ln2_b, &lm->w.ffn1_w, &lm->w.ffn1_b, &lm->w.ffn2_w, &lm->w.ffn2_b};
  for (int f = 0; f < 12; ++f) {
    const float* const* src = *fields[f];
    if (!src) { delete lm; ssrhip_set_error("ssrhip_lm_create: null per-layer pointer array %d", f); return -1; }
    lm->ptrs[f].assign(src, src + d->n_layer);
    *fields[f] = lm->ptrs[f].data();
  }
  // optional streaming-order copies (all or none)
  const float* const** opt[4] = {&lm->w.in_proj_wt, &lm->w.out_proj_wt, &lm->w.ffn1_wt, &lm->w.ffn2_wt};
  const bool has_wt = lm->w.in_proj_wt && lm->w.out_proj_wt && lm->w.ffn1_wt && lm->w.ffn2_wt && lm->w.head1_wt && lm->w.head2_wt;
  for (int f = 0; f < 4; ++f) {
    if (has_wt) {
      const float* const* src = *opt[f];
      lm->ptrs[12 + f].assign(src, src + d->n_layer);
      *opt[f] = lm->ptrs[12 + f].data();
    } else {
      *opt[f] = nullptr;
    }
  }
  if (!has_wt) lm->w.head1_wt = lm->w.head2_wt = nullptr;
  // optional bf16 planes for the prefill GEMMs (all four or none)
  const uint16_t* const** sp[4] = {&lm->w.in_proj_ws, &lm->w.out_proj_ws, &lm->w.ffn1_ws, &lm->w.ffn2_ws};
  const bool has_ws = lm->w.in_proj_ws && lm->w.out_proj_ws && lm->w.ffn1_ws && lm->w.ffn2_ws;
  for (int f = 0; f < 4; ++f) {
    if (has_ws) {
      const uint16_t* const* src = *sp[f];
      lm->sptrs[f].assign(src, src + d->n_layer);
      *sp[f] = lm->sptrs[f].data();
    } else {
      *sp[f] = nullptr;
    }
  }
  { const char* e = getenv("SSRHIP_PREFILL_SPLIT"); lm->prefill_split = has_ws && !(e && e[0] == '0'); }
  if (b->B != 2) {
    snprintf(lm->pair_why, sizeof(lm->pair_why), "pair launches exist for the 2-row step only (this engine has %d rows)", b->B);
  } else {
    // does the 2-row step pair at all (shapes, CU count, occupancy, CU mask, SSRHIP_GEMV_PAIR), and may THIS engine do it?
    bool want = b->pair_mode != 1;
    bool loud = false;                                              // a refusal the user did not ask for: say it once on stderr
    if (!want) snprintf(lm->pair_why, sizeof(lm->pair_why), "switched off by the caller (pair_mode 1)");
    if (want) {
      bool pq, ph, pf;
      if (StepShapes(lm).pair_plan(&pq, &ph, &pf) == 0) {
        want = false;
        snprintf(lm->pair_why, sizeof(lm->pair_why), "%s", ssrhip_gemv_pair_why());
      }
    }
    if (want && b->pair_mode != 2) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess) dev = -1;
      if (pair_slot_acquire(dev, lm->pair_why, sizeof(lm->pair_why))) lm->pair_dev = dev;
      else { want = false; loud = true; }
    }
    if (want) {                                   // granules + give-up flag of the paired GEMV launches (zeroed: every tag invalid)
      if (hipMalloc(&lm->pair_ws, SSRHIP_PAIR_WS_BYTES) != hipSuccess || hipMemset(lm->pair_ws, 0, SSRHIP_PAIR_WS_BYTES) != hipSuccess) {
        if (lm->pair_ws) hipFree(lm->pair_ws);
        if (lm->pair_dev >= 0) pair_slot_release(lm->pair_dev);
        delete lm;
        ssrhip_set_error("ssrhip_lm_create: allocating the pair workspace failed");
        return -1;
      }
      snprintf(lm->pair_why, sizeof(lm->pair_why), b->pair_mode == 2 ? "forced on by the caller (pair_mode 2: no slot, no guard)"
                                                                     : "this engine holds the pairing slot of its device");
    } else if (loud) {
      fprintf(stderr, "ssrhip: this 2-row decode engine steps WITHOUT pair launches (same tokens, ~5 %% slower): %s\n", lm->pair_why);
    }
  }
  *out = lm;
  return 0;
}

extern "C" int ssrhip_lm_pairing(const ssrhip_lm* lm, char* why, int32_t why_len) {
  if (!lm) return 0;
  if (why && why_len > 0) snprintf(why, (size_t)why_len, "%s", lm->pair_why);
  return lm->pair_ws ? 1 : 0;
}

extern "C" int ssrhip_lm_pair_status(ssrhip_lm* lm, ssrhip_stream_t stream) {
  SSR_REQUIRE(lm, "ssrhip_lm_pair_status: null engine");
  if (!lm->pair_ws) return 0;
  return ssrhip_gemv_pair_status(lm->pair_ws, stream);
}

extern "C" int ssrhip_debug_occupy(int32_t n_wg, int32_t lds_bytes, float ms, int32_t* started, ssrhip_stream_t stream) {
  SSR_REQUIRE(n_wg > 0 && n_wg <= 65535 && lds_bytes >= 0 && lds_bytes <= 160 * 1024 && ms >= 0.f && ms <= 20000.f, "ssrhip_debug_occupy: bad argument");
  if (lds_bytes > 64 * 1024)
    SSR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  hipLaunchKernelGGL(occupy_kernel, dim3(n_wg), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (long long)(ms * 100000.0f), lds_bytes / 4, started);
  SSR_LAUNCH_CHECK();
  return 0;
}

extern "C" void ssrhip_lm_destroy(ssrhip_lm* lm) {
  if (!lm) return;
  if (lm->exec) hipGraphExecDestroy(lm->exec);
  if (lm->graph) hipGraphDestroy(lm->graph);
  if (lm->cap_stream) hipStreamDestroy(lm->cap_stream);
  if (lm->pair_ws) hipFree(lm->pair_ws);
  if (lm->pair_dev >= 0) pair_slot_release(lm->pair_dev);
  delete lm;
}

extern "C" int ssrhip_lm_decode(ssrhip_lm* lm, int32_t n_steps, int32_t use_graph, ssrhip_stream_t stream) {
  SSR_REQUIRE(lm && n_steps >= 0, "ssrhip_lm_decode: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if (!use_graph) {
    for (int i = 0; i < n_steps; ++i)
      if (int rc = enqueue_step(lm, s, nullptr)) return rc;
    return 0;
  }
  if (!lm->exec) {
    if (!lm->cap_stream) SSR_HIP(hipStreamCreateWithFlags(&lm->cap_stream, hipStreamNonBlocking));
    SSR_HIP(hipStreamBeginCapture(lm->cap_stream, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_step(lm, lm->cap_stream, nullptr);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(lm->cap_stream, &g);
    if (rc) { if (g) hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) { ssrhip_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return -2; }
    lm->graph = g;
    SSR_HIP(hipGraphInstantiate(&lm->exec, lm->graph, nullptr, nullptr, 0));
  }
  for (int i = 0; i < n_steps; ++i) SSR_HIP(hipGraphLaunch(lm->exec, s));
  return 0;
}

extern "C" int ssrhip_lm_time_steps(ssrhip_lm* lm, int32_t n_steps, ssrhip_stream_t stream, float* out_us, int32_t* out_kind, int32_t n_out) {
  SSR_REQUIRE(lm && n_steps > 0 && out_us && out_kind, "ssrhip_lm_time_steps: bad argument");
  Timer tm;
  tm.on = true;
  tm.s = (hipStream_t)stream;
  SSR_HIP(hipEventCreate(&tm.e0));
  SSR_HIP(hipEventCreate(&tm.e1));
  int rc = 0;
  for (int i = 0; i < n_steps && !rc; ++i) rc = enqueue_step(lm, tm.s, &tm);
  hipEventDestroy(tm.e0);
  hipEventDestroy(tm.e1);
  if (rc) return rc;
  const int n = (int)tm.ms.size();
  SSR_REQUIRE(n <= n_out, "ssrhip_lm_time_steps: %d slots > n_out=%d", n, n_out);
  for (int i = 0; i < n; ++i) { out_us[i] = 1000.f * tm.ms[i] / n_steps; out_kind[i] = tm.kind[i]; }
  return n;
}

extern "C" int ssrhip_lm_time_category(ssrhip_lm* lm, int32_t category, int32_t n_replays, ssrhip_stream_t stream, float* out_us_per_launch,
                                       int32_t* out_launches_per_step) {
  SSR_REQUIRE(lm && category >= 0 && category <= CAT_SAMPLE && n_replays > 0 && out_us_per_launch && out_launches_per_step,
              "ssrhip_lm_time_category: bad argument");
  hipStream_t s = (hipStream_t)stream;
  Timer tm;
  tm.only = category;
  hipStream_t cap = nullptr;
  SSR_HIP(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  SSR_HIP(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  int rc = enqueue_step(lm, cap, &tm);
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(cap, &g);
  hipGraphExec_t ex = nullptr;
  if (!rc && e == hipSuccess) e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  float ms = 0.f;
  if (!rc && e == hipSuccess) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipGraphLaunch(ex, s);
    hipEventRecord(e0, s);
    for (int i = 0; i < n_replays; ++i) hipGraphLaunch(ex, s);
    hipEventRecord(e1, s);
    e = hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  if (ex) hipGraphExecDestroy(ex);
  if (g) hipGraphDestroy(g);
  hipStreamDestroy(cap);
  if (rc) return rc;
  if (e != hipSuccess) { ssrhip_set_error("ssrhip_lm_time_category: %s", hipGetErrorString(e)); return -2; }
  SSR_REQUIRE(tm.n_enq > 0, "ssrhip_lm_time_category: no launches in category %d", category);
  *out_launches_per_step = tm.n_enq;
  *out_us_per_launch = 1000.f * ms / ((float)n_replays * (float)tm.n_enq);
  return 0;
}

extern "C" int ssrhip_lm_prefill(ssrhip_lm* lm, const ssrhip_prefill_args* p, ssrhip_stream_t stream) {
  SSR_REQUIRE(lm && p && p->tok && p->pos && p->kind && p->row_seq && p->row_pos && p->row_len, "ssrhip_lm_prefill: null argument");
  const bool tiled_attn = p->seq_start && p->n_seq > 0 && p->max_len > 0 && !getenv_flag("SSRHIP_PREFILL_ATTN_ROWWISE");
  SSR_REQUIRE(p->x && p->xn && p->qkv && p->o && p->h && (tiled_attn || (p->part_o && p->part_ml)), "ssrhip_lm_prefill: null workspace");
  const ssrhip_lm_dims& d = lm->d;
  const ssrhip_lm_weights& w = lm->w;
  const int D = d.d_model, R = p->R;
  hipStream_t s = (hipStream_t)stream;
  ssrhip_kv kv = lm->b.kv;
  if (p->table) kv.table = p->table;              // two-phase admission: the rows being filled are not in the decode step's table yet

  ssrhip_embed_args ea;
  memset(&ea, 0, sizeof(ea));
  ea.text_emb = w.text_emb; ea.audio_emb = w.audio_emb; ea.pe = w.pe;
  ea.alpha_text = w.alpha_text; ea.alpha_audio = w.alpha_audio;
  ea.tok = p->tok; ea.pos = p->pos; ea.kind = p->kind;
  ea.R = R; ea.D = D; ea.K = d.n_codebooks; ea.card = d.card; ea.out = p->x;
  if (int rc = ssrhip_embed(&ea, s)) return rc;

  for (int l = 0; l < d.n_layer; ++l) {
    if (int rc = ssrhip_layernorm(p->x, w.ln1_w[l], w.ln1_b[l], 1e-5f, p->xn, R, D, s)) return rc;
    ssrhip_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = p->xn; g.W = w.in_proj_w[l]; g.bias = w.in_proj_b[l]; g.C = p->qkv;
    g.M = R; g.N = 3 * D; g.K = D; g.lda = D; g.ldc = 3 * D;
    if (lm->prefill_split) g.W_split = w.in_proj_ws[l];
    if (int rc = ssrhip_gemm(&g, s)) return rc;
    if (int rc = ssrhip_kv_scatter(p->qkv, &kv, l, p->row_seq, p->row_pos, R, s)) return rc;

    ssrhip_attn_args at;
    memset(&at, 0, sizeof(at));
    at.kv = kv; at.layer = l; at.row_seq = p->row_seq; at.row_len = p->row_len;
    at.R = R; at.max_splits = p->max_splits; at.scale = 1.0f / sqrtf((float)(D / d.n_head));
    at.part_o = p->part_o; at.part_ml = p->part_ml;
    at.q = p->qkv; at.q_stride = 3 * D;   // q is the first third of each packed qkv row
    if (tiled_attn) {
      // whole prompts: K/V tiles staged in LDS once per 128 queries, both products on the matrix core, no partials
      if (int rc = ssrhip_attn_prefill(&at, p->seq_start, p->n_seq, p->max_len, p->o, s)) return rc;
    } else {
      if (int rc = ssrhip_attn_decode(&at, s)) return rc;
      if (int rc = ssrhip_attn_combine(&at, p->o, s)) return rc;
    }

    memset(&g, 0, sizeof(g));
    g.A = p->o; g.W = w.out_proj_w[l]; g.bias = w.out_proj_b[l]; g.C = p->x;
    g.M = R; g.N = D; g.K = D; g.lda = D; g.ldc = D; g.residual = 1;
    if (lm->prefill_split) g.W_split = w.out_proj_ws[l];
    if (int rc = ssrhip_gemm(&g, s)) return rc;

    if (int rc = ssrhip_layernorm(p->x, w.ln2_w[l], w.ln2_b[l], 1e-5f, p->xn, R, D, s)) return rc;
    memset(&g, 0, sizeof(g));
    g.A = p->xn; g.W = w.ffn1_w[l]; g.bias = w.ffn1_b[l]; g.C = p->h;
    g.M = R; g.N = d.d_ffn; g.K = D; g.lda = D; g.ldc = d.d_ffn; g.act = SSRHIP_ACT_RELU;
    if (lm->prefill_split) g.W_split = w.ffn1_ws[l];
    if (int rc = ssrhip_gemm(&g, s)) return rc;
    memset(&g, 0, sizeof(g));
    g.A = p->h; g.W = w.ffn2_w[l]; g.bias = w.ffn2_b[l]; g.C = p->x;
    g.M = R; g.N = D; g.K = d.d_ffn; g.lda = d.d_ffn; g.ldc = D; g.residual = 1;
    if (lm->prefill_split) g.W_split = w.ffn2_ws[l];
    if (int rc = ssrhip_gemm(&g, s)) return rc;
  }
  if (p->no_embed) return 0;
  return ssrhip_lm_embed_pending(lm, stream);
}

extern "C" int ssrhip_lm_embed_pending(ssrhip_lm* lm, ssrhip_stream_t stream) {
  SSR_REQUIRE(lm, "ssrhip_lm_embed_pending: null argument");
  const ssrhip_lm_dims& d = lm->d;
  const ssrhip_lm_weights& w = lm->w;
  const int D = d.d_model;
  hipStream_t s = (hipStream_t)stream;
  // x of the first decode step: the rows' pending input tokens (the span-0 mask token, ssr.py:655-662)
  const ssrhip_lm_buffers& b = lm->b;
  ssrhip_embed_args ea;
  memset(&ea, 0, sizeof(ea));
  ea.text_emb = w.text_emb; ea.audio_emb = w.audio_emb; ea.pe = w.pe;
  ea.alpha_text = w.alpha_text; ea.alpha_audio = w.alpha_audio;
  ea.tok = b.next_tok; ea.pos = b.next_pos; ea.kind = nullptr;
  ea.R = b.B; ea.D = D; ea.K = d.n_codebooks; ea.card = d.card; ea.out = b.x; ea.out_tiled = b.B > 4 ? 1 : 0;
  return ssrhip_embed(&ea, s);
}
