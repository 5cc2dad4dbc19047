// mfma_valu_mix.hip -- what does one VALU / SALU / LDS instruction of wave B cost while wave A on the same SIMD streams
// independent v_mfma_f64_4x4x4_4b_f64?  One workgroup of 8 waves per CU: waves 0-3 stream MFMAs, waves 4-7 run the probe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP32(x) REP16(x) REP16(x)
template <int KIND, bool MFMA_ON, int PRIO>
__global__ void __launch_bounds__(512, 1) k_mix(int iters, unsigned long long* cyc, double* sink, const double* src) {
  __shared__ double lds[4096];
  const int wave = threadIdx.x >> 6;
  lds[threadIdx.x] = 1.0; lds[threadIdx.x + 512] = 2.0;
  __syncthreads();
  if (wave < 4) {
    if (!MFMA_ON) return;
    double acc[48]; for (int i = 0; i < 48; ++i) acc[i] = 0.0;
    double a = threadIdx.x, b = 1.0;
    for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
      for (int i = 0; i < 48; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0; for (int i = 0; i < 48; ++i) s += acc[i];
    if (s == 123.456) sink[0] = s;
    return;
  }
  if (PRIO == 3) __builtin_amdgcn_s_setprio(3);
  __builtin_amdgcn_s_sleep(20);     // let the MFMA stream start
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) asm volatile(REP32("v_mov_b32 v10, v11\n\t") ::: "v10");
    if (KIND == 1) asm volatile(REP32("v_add_f64 v[10:11], v[12:13], v[14:15]\n\t") ::: "v10", "v11");
    if (KIND == 2) asm volatile(REP32("s_add_u32 s20, s21, s22\n\t") ::: "s20", "scc");
    if (KIND == 3) asm volatile(REP32("v_add_u32 v10, v11, v12\n\t") ::: "v10");
    if (KIND == 4) asm volatile(REP32("ds_read_b128 v[10:13], %0\n\t") "s_waitcnt lgkmcnt(0)" ::"v"((threadIdx.x & 63) * 16) : "v10", "v11", "v12", "v13");
    if (KIND == 5) asm volatile(REP32("v_fma_f64 v[10:11], v[12:13], v[14:15], v[16:17]\n\t") ::: "v10", "v11");
    if (KIND == 6) {   // latency of one L2-resident 1 KiB wave load to VGPRs (32 dependent round trips)
      const double* p = src + (size_t)(blockIdx.x * 512 + threadIdx.x) * 2;
      asm volatile(REP32("global_load_dwordx4 v[10:13], %0, off\n\ts_waitcnt vmcnt(0)\n\t") ::"v"(p) : "v10", "v11", "v12", "v13", "memory");
    }
    if (KIND == 7) {   // the same through LDS-DMA
      const double* p = src + (size_t)(blockIdx.x * 512 + threadIdx.x) * 2;
      unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) double*)(lds + 1024 + (wave - 4) * 128));
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t" REP32("global_load_lds_dwordx4 %0, off\n\ts_waitcnt vmcnt(0)\n\t") ::"v"(p), "s"(dst) : "memory");
    }
    if (KIND == 8) {   // throughput: 32 LDS-DMA in flight, one wait
      const double* p = src + (size_t)(blockIdx.x * 512 + threadIdx.x) * 2;
      unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) double*)(lds + 1024 + (wave - 4) * 128));
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t" REP32("global_load_lds_dwordx4 %0, off\n\t") "s_waitcnt vmcnt(0)" ::"v"(p), "s"(dst) : "memory");
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + wave - 4] = t1 - t0;
}
template <int KIND, bool ON, int PRIO> void run(const char* name) {
  int blocks = 256, iters = 200; unsigned long long* cyc; double* sink; double* src; hipMalloc(&cyc, blocks * 4 * 8); hipMalloc(&sink, 8);
  hipMalloc(&src, (size_t)blocks * 512 * 16); hipMemset(src, 0, (size_t)blocks * 512 * 16);
  hipLaunchKernelGGL((k_mix<KIND, ON, PRIO>), dim3(blocks), dim3(512), 0, 0, iters, cyc, sink, src); hipDeviceSynchronize();
  hipLaunchKernelGGL((k_mix<KIND, ON, PRIO>), dim3(blocks), dim3(512), 0, 0, iters, cyc, sink, src); hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 4); hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost); std::sort(h.begin(), h.end());
  printf("%-14s mfma partner %-3s prio %d: %.2f cycles per instruction (median wave)\n", name, ON ? "on" : "off", PRIO, (double)h[h.size() / 2] / (iters * 32.0));
  hipFree(cyc); hipFree(sink); hipFree(src);
}
int main() {
  run<0, false, 0>("v_mov_b32"); run<0, true, 0>("v_mov_b32"); run<0, true, 3>("v_mov_b32");
  run<1, false, 0>("v_add_f64"); run<1, true, 0>("v_add_f64"); run<1, true, 3>("v_add_f64");
  run<5, false, 0>("v_fma_f64"); run<5, true, 0>("v_fma_f64"); run<5, true, 3>("v_fma_f64");
  run<3, false, 0>("v_add_u32"); run<3, true, 0>("v_add_u32"); run<3, true, 3>("v_add_u32");
  run<2, false, 0>("s_add_u32"); run<2, true, 0>("s_add_u32"); run<2, true, 3>("s_add_u32");
  run<4, false, 0>("ds_read_b128"); run<4, true, 0>("ds_read_b128"); run<4, true, 3>("ds_read_b128");
  run<6, false, 0>("gload x4 lat"); run<6, true, 0>("gload x4 lat"); run<6, true, 3>("gload x4 lat");
  run<7, false, 0>("glds x4 lat"); run<7, true, 0>("glds x4 lat"); run<7, true, 3>("glds x4 lat");
  run<8, false, 0>("glds x4 thru"); run<8, true, 0>("glds x4 thru"); run<8, true, 3>("glds x4 thru");
  return 0;
}
