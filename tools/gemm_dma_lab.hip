// gemm_dma_lab.hip — where the time of gemm_split_dma_kernel goes, taken on the SHIPPED source (this file includes
// ssr-speech_amd/csrc/gemm_split.hip; the kernel's KO template parameter is 0 in the library), for the codec's launch shapes at config 5
// (profiles/r04_codec_b256_kernel_trace_summary.md), 32 items per batch instead of 256:
//   knock-outs: the same launch with ONE component removed (results wrong, timing meaningful): the MFMAs (+ their LDS reads), the A loads,
//               the A split + LDS store, the W DMA, the C stores; and the MFMA loop alone;
//   timestamps: wave 0's first 40 k-steps (100 MHz clock) for a sample of the workgroups: loop top -> tile ready (barrier, A store, DMA
//               drain, barrier) -> MFMA block issued; entry and end of the epilogue.
// Written at the end of round 4 (no GPU minutes left to run it): the first thing to run in round 5.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I ssr-speech_amd/csrc -I include tools/gemm_dma_lab.hip -o tools/bin/gemm_dma_lab
#include "../ssr-speech_amd/csrc/gemm_split.hip"
#include <stdarg.h>
#include <stdio.h>
#include <vector>

void ssrhip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

template <int BM, bool ELU, int KO>
static float run(const ssrhip_gemm_args& a, int reps, int wide = 1) {
  const int lds = dma_lds(BM) + ((KO & GD_PROF) ? GD_NSTAMP * 4 : 0);
  auto kern = gemm_split_dma_kernel<BM, ELU, KO>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  dim3 grid((a.N + 127) / 128, (a.M + BM - 1) / BM, a.batch > 1 ? a.batch : 1);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, 0, a, wide);
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, grid, dim3(512), lds, 0, a, wide);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

struct Shape { const char* what; int M, N, K, lda; bool elu, withR; };

template <int BM, bool ELU>
static void lab(const Shape& sh, int batch, int reps) {
  const int M = sh.M, N = sh.N, K = sh.K, lda = sh.lda;
  const size_t rowsA = (size_t)M + (K + lda - 1) / lda + 8;             // an overlapping strided view (convolution windows) reads past row M - 1
  const size_t nA = (size_t)batch * rowsA * lda, nC = (size_t)batch * M * N, nW = (size_t)N * K;
  float *A, *C, *R = nullptr, *bias;
  uint16_t* Ws;
  CK(hipMalloc(&A, nA * 4)); CK(hipMalloc(&C, nC * 4)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&Ws, 3 * nW * 2));
  if (sh.withR) { CK(hipMalloc(&R, nC * 4)); CK(hipMemset(R, 0, nC * 4)); }
  {
    unsigned s = 4321;
    std::vector<float> h(rowsA * lda);
    for (auto& v : h) v = ((int)(lcg(s) >> 8) - (1 << 23)) * (2.0f / (1 << 23));
    for (int b = 0; b < batch; ++b) CK(hipMemcpy(A + (size_t)b * rowsA * lda, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint16_t> w(3 * nW);
    for (auto& v : w) v = (uint16_t)(0x3C00u + (lcg(s) >> 22) % 0x100u + ((lcg(s) >> 31) << 15));
    CK(hipMemcpy(Ws, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, N * 4));
  }
  ssrhip_gemm_args a = {};
  a.A = A; a.W = nullptr; a.bias = bias; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = N;
  a.act_in = ELU ? SSRHIP_ACT_ELU : 0; a.R = R; a.ldr = N; a.batch = batch;
  a.strideA = (int64_t)rowsA * lda; a.strideC = (int64_t)M * N; a.strideR = (int64_t)M * N; a.W_split = Ws;
  const double tflop = 2.0 * M * N * K * batch / 1e12, gb = ((double)batch * M * (lda < K ? lda : K) + nC * (sh.withR ? 2 : 1)) * 4 / 1e9;
  printf("== %s: %d x %d x %d x %d items, lda %d, BM %d%s%s: %.2f TFLOP fp32-equivalent, %.2f GB of A (once) + C%s\n", sh.what, M, N, K, batch, lda, BM,
         ELU ? ", ELU on load" : "", sh.withR ? ", + R" : "", tflop, gb, sh.withR ? " + R" : "");
  run<BM, ELU, 0>(a, reps);                                             // settle the clocks
  float base = 0.f, dw = 0.f;
  for (int k = 0; k < 2; ++k) { base += run<BM, ELU, 0>(a, reps, 1) / 2; dw += run<BM, ELU, 0>(a, reps, 0) / 2; }
  printf("  %-46s %8.3f ms   %6.1f TFLOP/s, %5.2f TB/s\n", "shipped kernel (16-byte epilogue)", base, tflop / base * 1e3, gb / base);
  printf("  %-46s %8.3f ms   %+7.3f\n", "dword epilogue (SSRHIP_EPILOGUE_WIDE=0)", dw, dw - base);
#define KOLINE(ko, what) { const float t_ = run<BM, ELU, ko>(a, reps); printf("  %-46s %8.3f ms   %+7.3f\n", what, t_, t_ - base); }
  KOLINE(GD_KO_MFMA, "without the MFMAs and their LDS reads");
  KOLINE(GD_KO_ALOAD, "without the A loads");
  KOLINE(GD_KO_ASTORE, "without the A split + LDS store");
  KOLINE(GD_KO_DMA, "without the W DMA");
  KOLINE(GD_KO_STORE, "without the C stores");
  KOLINE(GD_KO_ALOAD | GD_KO_ASTORE | GD_KO_DMA | GD_KO_STORE, "MFMAs + LDS reads + barriers only");
  KOLINE(GD_KO_MFMA | GD_KO_ASTORE, "memory only (no MFMA, no A split)");
  // timestamps
  const int gy = (M + BM - 1) / BM, nsy = (gy + 31) / 37, nslot = ((batch + 7) / 8) * nsy;
  unsigned* prof;
  CK(hipMalloc(&prof, (size_t)nslot * GD_NSTAMP * 4));
  CK(hipMemset(prof, 0, (size_t)nslot * GD_NSTAMP * 4));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gd_prof), &prof, sizeof(prof)));
  const float tp = run<BM, ELU, GD_PROF>(a, 1);
  std::vector<unsigned> h((size_t)nslot * GD_NSTAMP);
  CK(hipMemcpy(h.data(), prof, h.size() * 4, hipMemcpyDeviceToHost));
  const int nstep = (K + 31) / 32 < GD_STEPS ? (K + 31) / 32 : GD_STEPS;
  std::vector<double> sum(GD_NSTAMP, 0.0);
  int n = 0;
  for (int s = 0; s < nslot; ++s) {
    const unsigned* p = &h[(size_t)s * GD_NSTAMP];
    if (p[124] == 0 || p[125] == 0) continue;
    ++n;
    for (int i = 0; i < GD_NSTAMP; ++i) sum[i] += (double)(unsigned)(p[i] - p[124]) * 0.01;      // 100 MHz -> us
  }
  printf("  timestamps (%.3f ms with them; %d workgroups sampled), us after entry, wave 0; %d k-steps of 32 (24 MFMAs per wave = 0.32 us alone):\n", tp, n,
         (K + 31) / 32);
  if (n) {
    auto at = [&](int i) { return sum[i] / n; };
    double wait = 0, mm = 0;
    for (int s = 0; s < nstep; ++s) {
      if (s < 6 || s == nstep - 1)
        printf("    step %2d: top %7.2f  -> tile ready +%5.2f (barrier, A store, DMA drain, barrier)  -> MFMA block issued +%5.2f\n", s, at(3 * s), at(3 * s + 1) - at(3 * s),
               at(3 * s + 2) - at(3 * s + 1));
      wait += at(3 * s + 1) - at(3 * s);
      mm += at(3 * s + 2) - at(3 * s + 1);
    }
    printf("    first %d steps: %.2f us to the first loop top, %.2f us waiting for tiles, %.2f us in MFMA blocks; loop end %.2f, epilogue done %.2f (+%.2f)\n", nstep, at(0),
           wait, mm, at(123), at(125), at(125) - at(123));
  }
  CK(hipFree(prof)); CK(hipFree(A)); CK(hipFree(C)); CK(hipFree(bias)); CK(hipFree(Ws));
  if (R) CK(hipFree(R));
}

int main(int argc, char** argv) {
  const int batch = argc > 1 ? atoi(argv[1]) : 32, reps = argc > 2 ? atoi(argv[2]) : 3;
  // the launches of one config-5 pass that weigh most (shape, lda of the strided view the codec passes)
  const Shape big[] = {
      {"downsampling conv 256 -> 512 (k 8, stride 4... as a view)", 60001, 512, 512, 256, false, false},
      {"conv 512 -> 1280-wide view", 12001, 1280, 1024, 512, false, false},
      {"transposed conv as GEMM 1024 -> 256 x 4", 60000, 256, 1024, 1024, false, false},
      {"residual block 256: k 3 conv, ELU on load", 60000, 128, 768, 256, true, false},
      {"residual block 256: 1x1 conv + R", 60000, 256, 128, 128, false, true},
      {"first strided conv 64 -> 128 (k 4, stride 2)", 240000, 128, 256, 128, false, false},
      {"LSTM layer-1 input projection", 1500, 4096, 1024, 1024, false, false},
  };
  for (const Shape& sh : big) {
    if (sh.elu) lab<128, true>(sh, batch, reps);
    else lab<128, false>(sh, batch, reps);
  }
  const Shape chunk = {"LSTM layer-2 chunk projection (64 steps)", 64, 4096, 1024, 1024, false, false};
  lab<64, false>(chunk, batch * 8, reps);
  return 0;
}
