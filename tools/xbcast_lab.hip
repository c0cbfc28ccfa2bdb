// tools/xbcast_lab.hip — what it costs EVERY CU to read the SAME activation block right after a kernel boundary (the x operand of the
// 16-row decode GEMVs: 128 KB at K = 2048, 512 KB at K = 8192; tools/gemvm_lab.hip's stamps show it landing 5.5-6.9 us after entry while
// 128 KB through a CU's 64 B/clk L2 port are 0.9 us). A writer kernel produces the block (256 workgroups, a slice each), then the reader
// — 256 workgroups x 8 waves, wave w asks for 1-KB pieces w * NL .. w * NL + NL - 1 — stamps entry and "everything has arrived"; the pair is
// chained 64 times in a hipGraph. Variants: size; piece order rotated per workgroup (so that the CUs of an XCD do not all miss on the same
// line at the same moment); a 1/32 slice per CU first ("warm"), then everything; sc1 / nt cache policies; block NOT rewritten between reads.
// Build: hipcc -O3 --offload-arch=gfx950 tools/xbcast_lab.hip -o tools/bin/xbcast_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void writer(float* x, int n4, float v) {        // n4 float4 in all; workgroup b writes a contiguous slice
  const int per = (n4 + gridDim.x - 1) / gridDim.x;
  for (int i = threadIdx.x; i < per; i += blockDim.x) {
    const int j = blockIdx.x * per + i;
    if (j < n4) reinterpret_cast<v4f*>(x)[j] = (v4f){v, v + 1.f, v + 2.f, (float)j};
  }
}

// MODE 0 plain order, 1 rotated per workgroup, 2 warm slice first then plain, 3 plain with sc1 loads, 4 plain with nt loads
template <int NL, int MODE>
__global__ __launch_bounds__(512) void reader(const float* __restrict__ x, float* __restrict__ out, long long* __restrict__ prof) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long t0 = wall_clock64();
  constexpr int NP = NL * 8;                               // 1-KB pieces in the block
  const int rot = (MODE == 1) ? (int)((blockIdx.x >> 3) * (NP / 32)) % NP : 0;    // blockIdx >> 3 = index inside the XCD (b % 8 = XCD)
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 2) {                                         // this CU's own 1/32 of the block first: one miss per line chip-wide
    const int own = (int)(blockIdx.x >> 3) * (NP / 32);
    if (wave < NP / 32) { const v4f w = reinterpret_cast<const v4f*>(x)[(own + wave) * 64 + lane]; acc += w; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  v4f r[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const v4f* p = reinterpret_cast<const v4f*>(x) + (size_t)((wave * NL + i + rot) % NP) * 64 + lane;
    if (MODE == 3) r[i] = __builtin_nontemporal_load(p);   // placeholder, replaced below
    else if (MODE == 4) r[i] = __builtin_nontemporal_load(p);
    else r[i] = *p;
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) acc += r[i];
  const float s = acc[0] + acc[1] + acc[2] + acc[3];
  const long long t1 = wall_clock64();                     // behind the last use: everything has arrived
  if (s == 123.456f) out[threadIdx.x] = s;
  __syncthreads();
  const long long t2 = wall_clock64();
  if (threadIdx.x == 0) { prof[blockIdx.x * 4] = t0; prof[blockIdx.x * 4 + 1] = t1; prof[blockIdx.x * 4 + 2] = t2; }
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = s;      // something the next writer's launch depends on
}

template <int NL, int MODE>
void run(const char* name, float* x, float* out, long long* prof, bool rewrite) {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int n4 = NL * 8 * 64;
  hipGraph_t g; hipGraphExec_t ex;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 32; ++i) {
    if (rewrite) hipLaunchKernelGGL(writer, dim3(256), dim3(256), 0, s, x, n4, (float)i);
    else hipLaunchKernelGGL(writer, dim3(256), dim3(256), 0, s, out + 65536, 64, (float)i);     // a boundary, but x untouched
    hipLaunchKernelGGL((reader<NL, MODE>), dim3(256), dim3(512), 0, s, x, out, prof);
  }
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ex, s));
  CK(hipStreamSynchronize(s));
  std::vector<long long> h(256 * 4);
  CK(hipMemcpy(h.data(), prof, h.size() * 8, hipMemcpyDeviceToHost));
  long long t0 = h[0];
  for (int b = 0; b < 256; ++b) t0 = std::min(t0, h[b * 4]);
  std::vector<double> arr, all;
  for (int b = 0; b < 256; ++b) { arr.push_back((h[b * 4 + 1] - t0) * 0.01); all.push_back((h[b * 4 + 2] - t0) * 0.01); }
  std::sort(arr.begin(), arr.end()); std::sort(all.begin(), all.end());
  printf("  %-46s %4d KB: wave 0 has its pieces at %5.2f / %5.2f / %5.2f us, whole workgroup at %5.2f / %5.2f / %5.2f (min / median / max over CUs)\n", name,
         NL * 8, arr[0], arr[128], arr[255], all[0], all[128], all[255]);
  CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
}

int main() {
  float *x, *out; long long* prof;
  CK(hipMalloc(&x, 1 << 20)); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&prof, 256 * 4 * 8));
  CK(hipMemset(x, 0, 1 << 20)); CK(hipMemset(out, 0, 1 << 20));
  printf("every CU reads the same block behind a kernel boundary (256 workgroups x 8 waves):\n");
  run<2, 0>("16 KB (the 2-row x), plain order", x, out, prof, true);
  run<8, 0>("64 KB, plain order", x, out, prof, true);
  run<16, 0>("128 KB (16 rows x 2048), plain order", x, out, prof, true);
  run<16, 0>("128 KB, block NOT rewritten between the reads", x, out, prof, false);
  run<16, 1>("128 KB, piece order rotated per CU of an XCD", x, out, prof, true);
  run<16, 2>("128 KB, own 1/32 first, barrier, then all", x, out, prof, true);
  run<16, 4>("128 KB, nt loads", x, out, prof, true);
  run<32, 0>("256 KB, plain order", x, out, prof, true);
  run<32, 1>("256 KB, rotated", x, out, prof, true);
  return 0;
}
