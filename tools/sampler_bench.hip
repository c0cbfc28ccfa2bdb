// tools/sampler_bench.hip — stand-alone timing of ssrhip_sample (one utterance, 830M-shape logits) with
// per-phase shader-clock stamps.  Build: hipcc -O3 --offload-arch=gfx950 -DSSR_SAMPLE_PROFILE -Iinclude
//   -Issr-speech_amd/csrc -ffp-contract=off tools/sampler_bench.hip -o tools/bin/sampler_bench
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../ssr-speech_amd/csrc/embed_sample.hip"
#include <stdarg.h>
void ssrhip_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
  const int K = 4, card = 2056, D = 2048, B = 2, max_steps = 512;
  int top_k = argc > 1 ? atoi(argv[1]) : 40;
  float top_p = argc > 2 ? atof(argv[2]) : 0.8f;
  std::vector<float> h(B * K * card);
  srand(1);
  for (auto& v : h) v = 3.0f * ((rand() / (float)RAND_MAX) - 0.5f) * 4.f;
  float *logits, *aemb, *pe, *x; int *gen, *ntok, *npos, *kvp, *rl; ssrhip_sampler_cfg* cfg; ssrhip_sampler_state* st;
  CK(hipMalloc(&logits, h.size() * 4)); CK(hipMemcpy(logits, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&aemb, (size_t)K * card * D * 4)); CK(hipMemset(aemb, 0, (size_t)K * card * D * 4));
  CK(hipMalloc(&pe, (size_t)4096 * D * 4)); CK(hipMemset(pe, 0, (size_t)4096 * D * 4));
  CK(hipMalloc(&x, B * D * 4)); CK(hipMalloc(&gen, max_steps * K * 4)); CK(hipMalloc(&ntok, B * 4 * 4)); CK(hipMalloc(&npos, B * 4));
  CK(hipMalloc(&kvp, B * 4)); CK(hipMalloc(&rl, B * 4)); CK(hipMemset(kvp, 0, B * 4));
  CK(hipMalloc(&cfg, sizeof(*cfg))); CK(hipMalloc(&st, sizeof(*st)));
  ssrhip_sampler_cfg c; memset(&c, 0, sizeof(c));
  c.top_k = top_k; c.top_p = top_p; c.temperature = 1.f; c.stop_repetition = 2; c.cfg_coef = 1.5f; c.cfg_one_minus = -0.5f; c.cfg_stride = 5; c.use_cfg = 1;
  c.n_silence = 3; c.silence[0] = 1388; c.silence[1] = 1898; c.silence[2] = 131; c.text_len = 100000; c.n_spans = 1;
  c.empty_token = 2048; c.eog = 2049; c.eos = 2051; c.sos = 2052; c.mts = 2053; c.max_n_spans = 3; c.max_steps = max_steps; c.seed_lo = 7;
  CK(hipMemcpy(cfg, &c, sizeof(c), hipMemcpyHostToDevice));
  ssrhip_sample_args a; memset(&a, 0, sizeof(a));
  a.logits = logits; a.n_utt = 1; a.K = K; a.card = card; a.cfg = cfg; a.state = st; a.generated = gen; a.next_tok = ntok; a.next_pos = npos; a.kv_pos = kvp; a.row_len = rl;
  a.embed.audio_emb = aemb; a.embed.pe = pe; a.embed.alpha_audio = 1.f; a.embed.R = B; a.embed.D = D; a.embed.K = K; a.embed.card = card; a.embed.out = x;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 200;
  ssrhip_sampler_state s0; memset(&s0, 0, sizeof(s0)); s0.num_cfg_tag = 1; s0.prev_token = -1; s0.num_gen = 5;
  float tot = 0; unsigned long long acc[16] = {0}, fine[6] = {0};
  for (int r = 0; r < reps; ++r) {
    s0.n_steps = r; CK(hipMemcpy(st, &s0, sizeof(s0), hipMemcpyHostToDevice));
    CK(hipEventRecord(e0, 0));
    if (ssrhip_sample(&a, 0)) return 1;
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 10) tot += ms;
#ifdef SSR_SAMPLE_PROFILE
    unsigned long long p[16]; CK(hipMemcpyFromSymbol(p, HIP_SYMBOL(g_sample_prof), sizeof(p)));
    if (r >= 10) for (int i = 1; i < 9; ++i) acc[i] += p[i] - p[i - 1];
    if (r >= 10) { fine[0] += p[10] - p[9]; fine[1] += p[11] - p[10]; fine[2] += p[12] - p[11]; fine[3] += p[13] - p[12]; fine[4] += p[14] - p[13]; fine[5] += p[15] - p[14]; }
#endif
  }
  printf("top_k=%d top_p=%.2f: %.2f us per launch (event)\n", top_k, top_p, 1000 * tot / (reps - 10));
  const char* nm[9] = {"", "load+edit", "argmax/temp/keys", "(kmin..)", "top-k", "softmax+top-p", "sample", "barrier", "state"};
  for (int i = 1; i < 9; ++i) printf("  phase %d %-18s %8.0f clk\n", i, i < 8 ? nm[i] : "embed", (double)acc[i] / (reps - 10));
  const char* fn[6] = {"bin_select<0>: zero+bins+barrier", "atomics+barrier", "scan+barrier", "collect+barrier", "overflow flag+barrier", "rank+barrier"};
  for (int i = 0; i < 6; ++i) printf("    %-30s %8.0f clk\n", fn[i], (double)fine[i] / (reps - 10));
  return 0;
}
