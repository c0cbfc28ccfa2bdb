// tools/gemvm_lab.hip — where a 16-row decode GEMV launch spends its time ("attribute the 5 us above the chain floor,
// do not guess"). Includes the SHIPPED csrc/gemv_mfma.hip built with -DSSR_GEMVM_PROFILE: wave 0 of every workgroup leaves wall_clock64
// stamps (100 MHz) at entry / requests issued / LayerNorm done / last MFMA / behind the partial-tile barrier / stores issued. Per shape of the
// 830M step (16 rows, tiled activations, streaming-order weights as the engine passes them): the launch's graph-chained time, then the
// stamps relative to the earliest workgroup entry (min / median / max over workgroups).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I ssr-speech_amd/csrc -I include tools/gemvm_lab.hip -o tools/bin/gemvm_lab
#define SSR_GEMVM_PROFILE 1
#include "../ssr-speech_amd/csrc/gemv_mfma.hip"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <vector>
void ssrhip_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((float)(h & 0xFFFF) / 32768.0f - 1.0f) * scale;
  }
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 16;
  struct Shape { const char* name; int N, K, pro, act, epi, groups; } shapes[] = {
      {"ln1+qkv", 6144, 2048, 1, 0, 2, 1}, {"out_proj", 2048, 2048, 0, 0, 1, 1}, {"ln2+ffn1", 8192, 2048, 1, 1, 0, 1},
      {"ffn2", 2048, 8192, 0, 0, 1, 1}, {"lnf+head1", 4096, 2048, 1, 2, 0, 1}, {"head2", 2056, 1024, 0, 0, 0, 4}};
  const int NBUF = 16;
  const size_t wsz = (size_t)8224 * 2048;
  float* W; CK(hipMalloc(&W, NBUF * wsz * 4));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, W, NBUF * wsz, 17u, 0.02f);
  float *x, *y, *bias; CK(hipMalloc(&x, 16 * 8192 * 4)); CK(hipMalloc(&y, 16 * 8224 * 4)); CK(hipMalloc(&bias, 8224 * 4));
  hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, x, (size_t)16 * 8192, 3u, 1.0f);
  hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, y, (size_t)16 * 8224, 5u, 1.0f);
  hipLaunchKernelGGL(fill_kernel, dim3(8), dim3(256), 0, 0, bias, (size_t)8224, 7u, 0.1f);
  // a KV cache for the QKV launch's append (2 layers, 16 heads x 128, 8 pages per row)
  const int H = 16, hd = 128, n_layer = 2, max_pages = 8;
  float* pool; CK(hipMalloc(&pool, (size_t)(16 * max_pages + 1) * n_layer * 2 * H * 128 * hd * 4));
  std::vector<int> tab(16 * max_pages), pos(16);
  for (int r = 0; r < 16; ++r) { pos[r] = 300 + 17 * r; for (int pg = 0; pg < max_pages; ++pg) tab[r * max_pages + pg] = (r * 5 + pg * 16 + 3) % (16 * max_pages); }
  int *dtab, *dpos; CK(hipMalloc(&dtab, tab.size() * 4)); CK(hipMalloc(&dpos, 64));
  CK(hipMemcpy(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dpos, pos.data(), 64, hipMemcpyHostToDevice));
  long long* prof; CK(hipMalloc(&prof, 4096 * 8 * 8));
  hipStream_t s; CK(hipStreamCreate(&s));
  CK(hipDeviceSynchronize());
  for (auto& sh : shapes) {
    auto args = [&](int i) {
      ssrhip_gemv_args a; memset(&a, 0, sizeof(a));
      a.W = W + (size_t)(i % NBUF) * wsz; a.x = x; a.y = y; a.bias = bias; a.B = B; a.N = sh.N; a.K = sh.K; a.groups = sh.groups;
      a.x_stride = sh.K * sh.groups; a.y_stride = (sh.epi == 2 ? sh.K : sh.N) * sh.groups; a.x_tiled = 1; a.y_tiled = sh.epi == 2 ? 0 : 1; a.w_tiled = 1;
      a.pro = sh.pro; a.act = sh.act; a.epi = sh.epi; a.ln_eps = 1e-5f;
      if (sh.epi == 2) { a.kv.pool = pool; a.kv.table = dtab; a.kv.max_pages = max_pages; a.kv.n_layer = n_layer; a.kv.n_head = H; a.kv.head_dim = hd; a.layer = 1; a.kv_pos = dpos; }
      return a;
    };
    long long* null_prof = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemvm_prof), &null_prof, sizeof(null_prof)));
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 64; ++i) { ssrhip_gemv_args a = args(i); if (ssrhip_gemv_mfma_launch(&a, s)) return 1; }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ex, s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ex, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / (5 * 64), mb = (double)sh.groups * sh.N * sh.K * 4 / 1e6;
    printf("== %-10s N=%d K=%d g=%d, %d rows: %6.2f us per launch without stamps (%.2f TB/s)\n", sh.name, sh.N, sh.K, sh.groups, B, us, mb / us);
    // stamped: a chain of 8 launches, the LAST one's stamps are read (its predecessors keep the chain's steady state)
    CK(hipMemsetAsync(prof, 0, 4096 * 8 * 8, s));
    CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_gemvm_prof), &prof, sizeof(prof), 0, hipMemcpyHostToDevice, s));
    for (int i = 0; i < 8; ++i) { ssrhip_gemv_args a = args(i); if (ssrhip_gemv_mfma_launch(&a, s)) return 1; }
    CK(hipStreamSynchronize(s));
    std::vector<long long> h(4096 * 8);
    CK(hipMemcpy(h.data(), prof, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<int> wgs;
    for (int w = 0; w < 4096; ++w) if (h[w * 8] != 0) wgs.push_back(w);
    if (wgs.empty()) { printf("   no stamps\n"); continue; }
    long long t0 = h[wgs[0] * 8];
    for (int w : wgs) t0 = std::min(t0, h[w * 8]);
    const char* names[8] = {"entry", "x in registers", "LayerNorm done", "last MFMA issued", "behind the tile barrier", "stores issued", "x requests issued", "first W requests issued"};
    const int order[8] = {0, 6, 7, 1, 2, 3, 4, 5};
    printf("   %zu workgroups; us after the first workgroup's entry (min / median / max over workgroups):\n", wgs.size());
    for (int kk = 0; kk < 8; ++kk) {
      const int k = order[kk];
      std::vector<double> v;
      for (int w : wgs) if (h[w * 8 + k]) v.push_back((h[w * 8 + k] - t0) * 0.01);
      if (v.empty()) continue;
      std::sort(v.begin(), v.end());
      printf("     %-24s %6.2f / %6.2f / %6.2f\n", names[k], v.front(), v[v.size() / 2], v.back());
    }
  }
  return 0;
}
