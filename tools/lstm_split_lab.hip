// lstm_split_lab.hip — the LSTM recurrence through the C-ABI (ssrhip_lstm_layer) with and without the split-operand planes
// (csrc/lstm_split.hip vs lstm_step_wide_kernel): microseconds per step alone and with two layers' chains on two streams (the codec's
// pipeline), and the largest difference between the two paths' outputs. No torch: a GPU call of a few seconds.
// Written at the end of round 4 without GPU minutes left; `tools/r05_labs.sh lstm` runs the pytest form first.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include tools/lstm_split_lab.hip -Lssr-speech_amd/csrc -lssrhip \
//        -Wl,-rpath,'$ORIGIN/../../ssr-speech_amd/csrc' -o tools/bin/lstm_split_lab
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "ssrhip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define SK(x) do { if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, ssrhip_last_error()); exit(1); } } while (0)

static unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
static float rnd(unsigned& s) { return ((int)(lcg(s) >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }      // (-1, 1)

struct Layer {
  float *gin, *whh_packed, *out, *hbuf, *cbuf;
  uint16_t *wsplit, *hsplit;
};

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256, C = argc > 2 ? atoi(argv[2]) : 1024, T = argc > 3 ? atoi(argv[3]) : 200;
  if (C % 128 || B < 32) { fprintf(stderr, "need C %% 128 == 0 and B >= 32\n"); return 1; }
  const int KS = C / 64, rows = (B + 15) / 16 * 16, nbg = (B + 63) / 64;
  unsigned seed = 99;
  std::vector<float> whh((size_t)4 * C * C), gin((size_t)B * T * 4 * C);
  for (auto& v : whh) v = rnd(seed) / sqrtf((float)C) * 1.7f;
  for (auto& v : gin) v = rnd(seed) * 1.5f;
  // the fp32 kernels' packed order (include/ssrhip.h w_packed): [C/4 tiles][C/16 k-steps][4 k-slots][16 rows][4 floats], row r of tile j =
  // W_hh[(r % 4) C + 4 j + r / 4]  ==  whh.view(4, C/4, 4, C/16, 4, 4).permute(1, 3, 4, 2, 0, 5) of wmencodec._Lstm
  std::vector<float> packed((size_t)4 * C * C);
  {
    size_t o = 0;
    for (int j = 0; j < C / 4; ++j)
      for (int s = 0; s < C / 16; ++s)
        for (int ks = 0; ks < 4; ++ks)
          for (int u = 0; u < 4; ++u)
            for (int g = 0; g < 4; ++g)
              for (int e = 0; e < 4; ++e) packed[o++] = whh[((size_t)g * C + 4 * j + u) * C + 16 * s + 4 * ks + e];
  }
  float *d_whh;
  uint16_t* d_planes;
  CK(hipMalloc(&d_whh, whh.size() * 4)); CK(hipMalloc(&d_planes, whh.size() * 6));
  CK(hipMemcpy(d_whh, whh.data(), whh.size() * 4, hipMemcpyHostToDevice));
  SK(ssrhip_split_weights(d_whh, d_planes, (int64_t)whh.size(), nullptr));
  std::vector<uint16_t> planes(whh.size() * 3), wsplit(whh.size() * 3);
  CK(hipMemcpy(planes.data(), d_planes, planes.size() * 2, hipMemcpyDeviceToHost));
  // wmencodec.pack_lstm_whh_planes: out[ub][w][s][mb][q][lh][li][e] = planes[q][g C + 16 ub + u][w C/4 + 16 s + 8 lh + e], 32 mb + li = 16 g + u
  {
    size_t o = 0;
    for (int ub = 0; ub < C / 16; ++ub)
      for (int w = 0; w < 4; ++w)
        for (int s = 0; s < KS; ++s)
          for (int mb = 0; mb < 2; ++mb)
            for (int q = 0; q < 3; ++q)
              for (int lh = 0; lh < 2; ++lh)
                for (int li = 0; li < 32; ++li) {
                  const int m = 32 * mb + li, g = m / 16, u = m % 16;
                  for (int e = 0; e < 8; ++e)
                    wsplit[o++] = planes[((size_t)q * 4 * C + (size_t)g * C + 16 * ub + u) * C + w * (C / 4) + 16 * s + 8 * lh + e];
                }
  }
  Layer L[2];
  for (int l = 0; l < 2; ++l) {
    CK(hipMalloc(&L[l].gin, gin.size() * 4)); CK(hipMalloc(&L[l].whh_packed, packed.size() * 4)); CK(hipMalloc(&L[l].out, (size_t)B * T * C * 4));
    CK(hipMalloc(&L[l].hbuf, (size_t)2 * rows * C * 4)); CK(hipMalloc(&L[l].cbuf, (size_t)B * C * 4));
    CK(hipMalloc(&L[l].wsplit, wsplit.size() * 2)); CK(hipMalloc(&L[l].hsplit, (size_t)2 * nbg * 64 * C * 3 * 2));
    CK(hipMemcpy(L[l].gin, gin.data(), gin.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(L[l].whh_packed, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(L[l].wsplit, wsplit.data(), wsplit.size() * 2, hipMemcpyHostToDevice));
  }
  auto args = [&](int l, bool split) {
    ssrhip_lstm_args a = {};
    a.gin = L[l].gin; a.w_hh = L[l].whh_packed; a.out = L[l].out; a.hbuf = L[l].hbuf; a.cbuf = L[l].cbuf;
    a.B = B; a.T = T; a.C = C; a.gin_bstride = (int64_t)T * 4 * C; a.out_bstride = (int64_t)T * C; a.w_packed = 1;
    if (split) { a.w_split = L[l].wsplit; a.hsplit = L[l].hsplit; }
    return a;
  };
  hipStream_t s0, s1;
  CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ref((size_t)B * T * C), got(ref.size());
  printf("B = %d, C = %d, T = %d steps (%d workgroups per step)\n", B, C, T, (C / 16) * nbg);
  for (int split = 0; split < 2; ++split) {
    ssrhip_lstm_args a0 = args(0, split), a1 = args(1, split);
    SK(ssrhip_lstm_layer(&a0, s0));                                     // warm (and the result that is compared)
    CK(hipStreamSynchronize(s0));
    CK(hipMemcpy(split ? got.data() : ref.data(), L[0].out, ref.size() * 4, hipMemcpyDeviceToHost));
    float alone = 0, both = 0;
    CK(hipEventRecord(e0, s0));
    SK(ssrhip_lstm_layer(&a0, s0));
    CK(hipEventRecord(e1, s0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&alone, e0, e1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, s0));
    CK(hipStreamWaitEvent(s1, e0, 0));
    SK(ssrhip_lstm_layer(&a0, s0));
    SK(ssrhip_lstm_layer(&a1, s1));                                     // a second chain beside it, as the codec's two-layer pipeline has
    CK(hipEventRecord(e1, s1));
    CK(hipStreamWaitEvent(s0, e1, 0));
    CK(hipEventRecord(e1, s0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&both, e0, e1));
    printf("  %-46s %7.2f us per step alone, %7.2f with a second chain on another stream\n", split ? "split operands, bf16 matrix cores (lstm_split.hip)" : "fp32 matrix pipe (lstm_step_wide_kernel)",
           alone * 1e3 / T, both * 1e3 / T);
  }
  double worst = 0, mean = 0;
  for (size_t i = 0; i < ref.size(); ++i) { const double d = fabs((double)ref[i] - got[i]); worst = d > worst ? d : worst; mean += d; }
  printf("  outputs of the two paths over %d steps: max |diff| %.3g, mean %.3g (both carry fp32-level rounding; a layout bug shows as O(0.1))\n", T, worst, mean / ref.size());
  return 0;
}
