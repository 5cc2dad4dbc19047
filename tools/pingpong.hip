// pingpong.hip -- one-hop latency of a 16-byte {payload, tag} mailbox granule between two workgroups (CUs), for the cache
// policy flavours of store and load; block 0 and block 1 (+8: same XCD class) bounce a tag ITER times.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long v2ull __attribute__((ext_vector_type(2)));
template <int F> __device__ inline void st(v2ull* p, v2ull v) {
  if (F == 0) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  if (F == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (F == 2) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (F == 3) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}
template <int F> __device__ inline v2ull ld(const v2ull* p) {
  v2ull r;
  if (F == 0) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  if (F == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  if (F == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  if (F == 3) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  return r;
}
template <int FS, int FL>
__global__ void k_pp(v2ull* box, int iters, int partner_block, unsigned long long* out, int* fail) {
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == partner_block ? 1 : -1);
  if (me < 0 || threadIdx.x != 0) return;
  v2ull* mine = box + me * 64;          // separate 1 KiB regions
  v2ull* theirs = box + (1 - me) * 64;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 1; it <= iters; ++it) {
    if (me == 0) {
      v2ull v; v.x = it * 3; v.y = it; st<FS>(mine, v);
      v2ull g; int spins = 0;
      do { g = ld<FL>(theirs); if (++spins > (1 << 18)) { *fail = 1; return; } } while (g.y != (unsigned long long)it);
      if (g.x != (unsigned long long)it * 5) *fail = 2;
    } else {
      v2ull g; int spins = 0;
      do { g = ld<FL>(theirs); if (++spins > (1 << 18)) { *fail = 1; return; } } while (g.y != (unsigned long long)it);
      if (g.x != (unsigned long long)it * 3) *fail = 2;
      v2ull v; v.x = it * 5; v.y = it; st<FS>(mine, v);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (me == 0) out[0] = t1 - t0;
}
template <int FS, int FL> void run(const char* name, int partner) {
  v2ull* box; unsigned long long* out; int* fail;
  hipMalloc(&box, 4096); hipMemset(box, 0, 4096); hipMalloc(&out, 8); hipMalloc(&fail, 4); hipMemset(fail, 0, 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_pp<FS, FL>), dim3(64), dim3(64), 0, 0, box, iters, partner, out, fail);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  int f; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
  printf("%-34s partner block %2d: %.2f us per hop  %s\n", name, partner, ms * 1e3 / (2.0 * iters), f == 0 ? "ok" : (f == 1 ? "TIMEOUT (stale)" : "WRONG PAYLOAD"));
  hipFree(box); hipFree(out); hipFree(fail);
}
int main() {
  for (int partner : {8, 1}) {     // 8: same XCD class as block 0; 1: a different XCD
    if (partner == 8) {
      run<0, 0>("store sc0 sc1 / load sc0 sc1", 8); run<1, 1>("store sc1 / load sc1", 8); run<2, 1>("store plain / load sc1", 8);
      run<2, 2>("store plain / load sc0", 8); run<3, 3>("store nt / load nt", 8); run<1, 0>("store sc1 / load sc0 sc1", 8);
    } else {
      run<0, 0>("store sc0 sc1 / load sc0 sc1", 1); run<1, 1>("store sc1 / load sc1", 1); run<2, 1>("store plain / load sc1", 1);
      run<2, 2>("store plain / load sc0", 1); run<3, 3>("store nt / load nt", 1); run<1, 0>("store sc1 / load sc0 sc1", 1);
    }
  }
  return 0;
}
