// mfma_probe.hip -- issue rate and sustained clock of v_mfma_f64_16x16x4_f64 on gfx950 (evidence for DESIGN.md).
// usage: mfma_probe [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) probe4(int iters, double* sink, unsigned long long* cyc, unsigned long long* rt) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (s == 123.456) sink[0] = s;
  if ((threadIdx.x & 63) == 0) {
    int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    cyc[w] = t1 - t0; rt[w] = r1 - r0;
  }
}

template <int NACC>
__global__ void __launch_bounds__(256) probe(int iters, double* sink, unsigned long long* cyc, unsigned long long* rt) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (s == 123.456) sink[0] = s;
  if ((threadIdx.x & 63) == 0) {
    int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    cyc[w] = t1 - t0; rt[w] = r1 - r0;
  }
}

template <int NACC, bool SMALL = false>
void run(int blocks_per_cu, int iters) {
  int blocks = 256 * blocks_per_cu;
  double* sink; unsigned long long *cyc, *rt;
  hipMalloc(&sink, 8); hipMalloc(&cyc, blocks * 4 * 8); hipMalloc(&rt, blocks * 4 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // warm: ~1 s of back-to-back launches so the chip settles at its sustained clock
  auto kern = SMALL ? probe4<NACC> : probe<NACC>;
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, sink, cyc, rt);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, iters, sink, cyc, rt);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> hc(blocks * 4), hr(blocks * 4);
  hipMemcpy(hc.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hr.data(), rt, blocks * 4 * 8, hipMemcpyDeviceToHost);
  std::vector<double> cpm, clk;
  for (int i = 0; i < blocks * 4; ++i) { cpm.push_back((double)hc[i] / ((double)iters * NACC)); clk.push_back((double)hc[i] / (double)hr[i] * 0.1); }
  std::sort(cpm.begin(), cpm.end()); std::sort(clk.begin(), clk.end());
  double flops = (double)blocks * 4 * iters * NACC * (SMALL ? 512.0 : 2048.0);
  printf("%s acc=%2d waves/SIMD=%d  %.1f TFLOP/s  cycles/MFMA/wave median %.1f  in-kernel clock median %.3f GHz  (kernel %.2f ms)\n",
         SMALL ? "4x4x4_4b " : "16x16x4  ", NACC, blocks_per_cu, flops / (ms * 1e-3) / 1e12, cpm[cpm.size() / 2], clk[clk.size() / 2], ms);
  hipFree(sink); hipFree(cyc); hipFree(rt);
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 4000;
  run<1>(1, iters); run<2>(1, iters); run<4>(1, iters); run<8>(1, iters);
  run<4>(2, iters); run<8>(2, iters);
  run<2>(3, iters); run<4>(3, iters);
  run<2>(4, iters); run<4>(4, iters); run<8>(4, iters);
  run<2>(6, iters); run<2>(8, iters); run<4>(8, iters);
  run<4, true>(1, iters); run<8, true>(1, iters); run<4, true>(2, iters); run<4, true>(4, iters); run<4, true>(8, iters);
  return 0;
}
