// gemm_trace.cpp -- timeline of the trailing-update kernel's chunk loop (diagnostic; built with -DBIEM_GEMM_TRACE together
// with the library sources):  hipcc -O3 -std=c++17 --offload-arch=gfx950 -DBIEM_GEMM_TRACE tools/gemm_trace.cpp \
//     biem_helmholtz_sphere_amd/csrc/{abi.cpp,plan.cpp,kernels_fill.hip,kernels_uscat.hip,kernels_lu.hip} -o tools/gemm_trace
#include <cstdio>
#include <cstdlib>
#include <vector>
extern "C" int biem_debug_gemm(int nb, int n, int kd, int reps, unsigned long long* trace_out, float* ms_out);
int main(int argc, char** argv) {
  int nb = argc > 1 ? atoi(argv[1]) : 8, n = argc > 2 ? atoi(argv[2]) : 6272, kd = argc > 3 ? atoi(argv[3]) : 128;
  std::vector<unsigned long long> tr(16 * 64 * 8);
  float ms = 0;
  if (biem_debug_gemm(nb, n, kd, 3, tr.data(), &ms)) { printf("failed\n"); return 1; }
  double flops = 8.0 * nb * (double)n * n * kd;
  printf("nb=%d n=%d kd=%d  %.3f ms  %.2f TFLOP/s (algorithmic)\n", nb, n, kd, ms, flops / (ms * 1e-3) / 1e12);
  const char* seg[7] = {"vmcnt", "barrier", "dma-issue", "lds-read", "sums", "mfma", "c-add"};
  for (int b = 0; b < 16; ++b) {
    unsigned long long* t = &tr[(size_t)b * 64 * 8];
    unsigned hw = (unsigned)t[7]; t[7] = t[6];
    printf("wg %d wave %d hw_id=0x%04x wave=%u simd=%u cu=%u sh=%u se=%u\n", b >> 2, b & 3, hw, hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7);
    double sum[8] = {0}; int cnt = 0;
    for (int c = 20; c < 60; ++c) {                  // steady state
      unsigned long long* s = t + c * 8; unsigned long long* nx = t + (c + 1) * 8;
      if (!s[0] || !nx[0]) continue;
      for (int i = 0; i < 7; ++i) sum[i] += (double)(s[i + 1] - s[i]);
      sum[7] += (double)(nx[0] - s[0]); ++cnt;
    }
    if (!cnt) continue;
    printf("   avg cycles/chunk:");
    for (int i = 0; i < 7; ++i) printf(" %s %.0f", seg[i], sum[i] / cnt);
    printf(" | period %.0f\n", sum[7] / cnt);
    if (b < 8) {
      printf("   chunk start stamps (rel.):");
      for (int c = 20; c < 26; ++c) printf(" [%llu M@%llu..%llu]", t[c * 8] - tr[(size_t)(b & ~3) * 64 * 8 + 20 * 8], t[c * 8 + 5] - tr[(size_t)(b & ~3) * 64 * 8 + 20 * 8], t[c * 8 + 6] - tr[(size_t)(b & ~3) * 64 * 8 + 20 * 8]);
      printf("\n");
    }
  }
  return 0;
}
