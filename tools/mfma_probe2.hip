// mfma_probe2.hip -- does the issue rate of v_mfma_f64_4x4x4_4b_f64 depend on operand variety / VGPR vs AGPR accumulators?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int MINW, bool VARY>
__global__ void __launch_bounds__(256, MINW) probe(int iters, const double* in, double* sink, unsigned long long* cyc) {
  double acc[4][4][3];
  double a[4], as[4], b[4], bs[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x + 256 * i]; as[i] = in[threadIdx.x + 256 * (i + 4)]; b[i] = in[threadIdx.x + 256 * (i + 8)]; bs[i] = in[threadIdx.x + 256 * (i + 12)]; }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int p = 0; p < 3; ++p) acc[i][j][p] = 0.0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (VARY) {
          acc[i][j][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[i][j][0], 0, 0, 1);
          acc[i][j][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(as[i], bs[j], acc[i][j][1], 0, 0, 0);
          acc[i][j][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[j], bs[i], acc[i][j][2], 0, 0, 1);
        } else {
          acc[i][j][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], b[0], acc[i][j][0], 0, 0, 1);
          acc[i][j][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], b[0], acc[i][j][1], 0, 0, 0);
          acc[i][j][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], b[0], acc[i][j][2], 0, 0, 1);
        }
      }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int p = 0; p < 3; ++p) s += acc[i][j][p];
  if (s == 123.456) sink[0] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MINW, bool VARY>
void run(int bpc, int iters, const char* name) {
  int blocks = 256 * bpc;
  double *in, *sink; unsigned long long* cyc;
  hipMalloc(&in, 256 * 16 * 8); hipMemset(in, 0, 256 * 16 * 8); hipMalloc(&sink, 8); hipMalloc(&cyc, blocks * 4 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<MINW, VARY>), dim3(blocks), dim3(256), 0, 0, iters, in, sink, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MINW, VARY>), dim3(blocks), dim3(256), 0, 0, iters, in, sink, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> hc(blocks * 4);
  hipMemcpy(hc.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
  std::sort(hc.begin(), hc.end());
  double flops = (double)blocks * 4 * iters * 48 * 512.0;
  printf("%-34s waves/SIMD=%d  %.1f TFLOP/s  cycles/MFMA/wave median %.1f\n", name, bpc, flops / (ms * 1e-3) / 1e12,
         (double)hc[hc.size() / 2] / (iters * 48.0));
}

int main() {
  int iters = 2000;
  run<1, false>(1, iters, "same operands, launch_bounds(256,1)");
  run<1, true>(1, iters, "varied operands, launch_bounds(256,1)");
  run<2, false>(1, iters, "same operands, launch_bounds(256,2)");
  run<2, true>(1, iters, "varied operands, launch_bounds(256,2)");
  run<2, true>(2, iters, "varied operands, launch_bounds(256,2)");
  run<3, true>(3, iters, "varied operands, launch_bounds(256,3)");
  run<4, true>(4, iters, "varied operands, launch_bounds(256,4)");
  return 0;
}
