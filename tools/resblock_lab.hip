// resblock_lab.hip — where the time of resblock_split_dma_kernel goes, taken on the SHIPPED source (this file includes
// ssr-speech_amd/csrc/resblock_split.hip; the kernel's KO template parameter is 0 in the library).
//   knock-outs: the same launch with ONE component removed (results wrong, timing meaningful): the DMA waits, the ELU(x) tile build, the
//               residual re-read, the MFMAs (+ their LDS reads), the weight DMA, the x tile loads, the stores;
//   timestamps: wave 0's phases (100 MHz clock) of a sample of workgroups: entry, tile-ready / MFMA-block-done per weight tile, end.
// Shapes: the two residual blocks of config 5 (C = 128 at T = 240000, C = 64 at T = 480000), B clips (default 32 = 1/8 of the bench).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I ssr-speech_amd/csrc -I include tools/resblock_lab.hip -o tools/bin/resblock_lab
#include "../ssr-speech_amd/csrc/resblock_split.hip"
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

template <int CC, int RING, int KO>
static float run(const ssrhip_resblock_args& a, int reps, int wide = 1) {
  const int lds = rb_lds(CC, RING) + ((KO & RB_PROF) ? RB_NSTAMP * 4 : 0);
  auto kern = resblock_split_dma_kernel<CC, RING, KO>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  dim3 grid((a.T + RB_BM - 1) / RB_BM, a.B);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, grid, dim3(RB_TH), lds, 0, a, wide);      // warm
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, grid, dim3(RB_TH), lds, 0, a, wide);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

__global__ void count_diff(const unsigned* a, const unsigned* b, size_t n, unsigned long long* out) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}

template <int CC>
static void lab(int B, int T, int reps) {
  constexpr int HH = CC / 2;
  const size_t nx = (size_t)B * (T + 2) * CC, ny = (size_t)B * T * CC;
  float *x, *y, *b3, *b1;
  uint16_t *w3s, *w1s;
  CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&y, ny * 4)); CK(hipMalloc(&b3, HH * 4)); CK(hipMalloc(&b1, CC * 4));
  const size_t n3 = (size_t)3 * HH * 3 * CC, n1 = (size_t)3 * CC * HH;
  CK(hipMalloc(&w3s, n3 * 2)); CK(hipMalloc(&w1s, n1 * 2));
  {
    unsigned s = 12345;
    std::vector<float> h((size_t)(T + 2) * CC);
    for (auto& v : h) v = ((int)(lcg(s) >> 8) - (1 << 23)) * (2.0f / (1 << 23));      // (-2, 2)
    for (int b = 0; b < B; ++b) CK(hipMemcpy(x + (size_t)b * (T + 2) * CC, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint16_t> w(n3 > n1 ? n3 : n1);
    for (auto& v : w) v = (uint16_t)(0x3C00u + (lcg(s) >> 22) % 0x100u + ((lcg(s) >> 31) << 15));   // bf16 around +-0.008
    CK(hipMemcpy(w3s, w.data(), n3 * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(w1s, w.data(), n1 * 2, hipMemcpyHostToDevice));
    CK(hipMemset(b3, 0, HH * 4)); CK(hipMemset(b1, 0, CC * 4));
  }
  ssrhip_resblock_args a = {};
  a.x = x; a.y = y; a.b3 = b3; a.b1 = b1; a.B = B; a.T = T; a.C = CC;
  a.x_bstride = (int64_t)(T + 2) * CC; a.y_bstride = (int64_t)T * CC; a.out_act = SSRHIP_ACT_ELU;
  a.w3_split = w3s; a.w1_split = w1s;
  const double gb = (nx + ny) * 4 / 1e9;
  printf("== C = %d, B = %d, T = %d: %.2f GB in + out once, %d x %d workgroups\n", CC, B, T, gb, (T + RB_BM - 1) / RB_BM, B);
  run<CC, 2, 0>(a, reps);                                   // settle clocks and the allocator's pages before anything is compared
  float base = 0.f, dw = 0.f;
  for (int k = 0; k < 3; ++k) {                              // alternate the two epilogue forms: order effects cancel
    const float tw = run<CC, 2, 0>(a, reps, 1), td = run<CC, 2, 0>(a, reps, 0);
    printf("  round %d: 16-byte epilogue %8.3f ms, dword epilogue %8.3f ms\n", k, tw, td);
    base += tw / 3; dw += td / 3;
  }
  printf("  %-44s %8.3f ms   (%.2f TB/s of in + out once)\n", "shipped kernel (ring 2, 16-byte epilogue)", base, gb / base);
  printf("  %-44s %8.3f ms   %+7.3f\n", "dword epilogue (SSRHIP_EPILOGUE_WIDE=0)", dw, dw - base);
#define KOLINE(ko, what) { const float t_ = run<CC, 2, ko>(a, reps); printf("  %-44s %8.3f ms   %+7.3f\n", what, t_, t_ - base); }
  KOLINE(RB_KO_WAIT, "without the DMA waits");
  KOLINE(RB_KO_DMA, "without the weight DMA (and its waits)");
  KOLINE(RB_KO_ESTORE, "without the ELU(x) tile build");
  KOLINE(RB_KO_ELOAD, "without the x tile loads");
  KOLINE(RB_KO_RESID, "without the residual re-read");
  KOLINE(RB_KO_MFMA, "without the MFMAs and their LDS reads");
  KOLINE(RB_KO_STORE, "without the stores");
  KOLINE(RB_KO_RESID | RB_KO_STORE | RB_KO_ELOAD, "without any HBM traffic");
  KOLINE(RB_KO_MFMA | RB_KO_ESTORE, "memory only (no MFMA, no tile build)");
  KOLINE(RB_KO_RESID | RB_KO_STORE | RB_KO_ELOAD | RB_KO_DMA | RB_KO_ESTORE, "MFMAs + LDS reads + barriers only");
  {   // the two epilogue forms do the same arithmetic per element: their outputs have to be bit-identical
    float* y2;
    unsigned long long *d, h = 0;
    CK(hipMalloc(&y2, ny * 4)); CK(hipMalloc(&d, 8)); CK(hipMemset(d, 0, 8));
    run<CC, 2, 0>(a, 1, 1);
    ssrhip_resblock_args a2 = a;
    a2.y = y2;
    run<CC, 2, 0>(a2, 1, 0);
    hipLaunchKernelGGL(count_diff, dim3(4096), dim3(256), 0, 0, (const unsigned*)y, (const unsigned*)y2, ny, d);
    CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    printf("  16-byte vs dword epilogue: %llu of %zu outputs differ\n", h, ny);
    CK(hipFree(y2)); CK(hipFree(d));
  }
  // timestamps
  const int gx = (T + RB_BM - 1) / RB_BM, nsx = (gx + 53) / 61, nslot = ((B + 7) / 8) * nsx;
  unsigned* prof;
  CK(hipMalloc(&prof, (size_t)nslot * RB_NSTAMP * 4));
  CK(hipMemset(prof, 0, (size_t)nslot * RB_NSTAMP * 4));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rb_prof), &prof, sizeof(prof)));
  const float tp = run<CC, 2, RB_PROF>(a, 1);
  std::vector<unsigned> h((size_t)nslot * RB_NSTAMP);
  CK(hipMemcpy(h.data(), prof, h.size() * 4, hipMemcpyDeviceToHost));
  constexpr int NU = NS1X(CC) + HH / 16;
  std::vector<double> sum(RB_NSTAMP, 0.0);
  int n = 0;
  for (int s = 0; s < nslot; ++s) {
    const unsigned* p = &h[(size_t)s * RB_NSTAMP];
    if (p[0] == 0 || p[2 + 2 * NU] == 0) continue;
    ++n;
    for (int i = 0; i < RB_NSTAMP; ++i) sum[i] += (double)(unsigned)(p[i] - p[0]) * 0.01;      // 100 MHz -> us
  }
  printf("  timestamps (%.3f ms with them; %d workgroups sampled), us after entry, wave 0:\n", tp, n);
  if (n) {
    auto at = [&](int i) { return sum[i] / n; };
    double prev = 0;
    for (int u = 0; u < NU; ++u) {
      const int ct = u / 3;
      if (u < NS1X(CC) && u % 3 == 0) { printf("    ELU(x) tile %d built            at %7.2f (+%5.2f)\n", ct, at(40 + ct), at(40 + ct) - prev); prev = at(40 + ct); }
      printf("    tile %2d (%s) ready %7.2f (+%5.2f wait + barrier)   MFMA block issued %7.2f (+%5.2f)\n", u, u < NS1X(CC) ? "W3" : "W1", at(2 + 2 * u),
             at(2 + 2 * u) - prev, at(3 + 2 * u), at(3 + 2 * u) - at(2 + 2 * u));
      prev = at(3 + 2 * u);
    }
    printf("    epilogue done (stores drained)  %7.2f (+%5.2f)\n", at(2 + 2 * NU), at(2 + 2 * NU) - prev);
  }
  CK(hipFree(prof)); CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(b3)); CK(hipFree(b1)); CK(hipFree(w3s)); CK(hipFree(w1s));
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, reps = argc > 2 ? atoi(argv[2]) : 5;
  lab<128>(B, 240000, reps);
  lab<64>(B, 480000, reps);
  return 0;
}
