// mfma_layout.hip -- empirical check of v_mfma_f64_4x4x4_4b_f64 lane layout, CBSZ/ABID broadcast and the f64 NEG bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

template <int CBSZ, int ABID, int BLGP>
__global__ void k(const double* a, const double* b, const double* c, double* d) {
  int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], CBSZ, ABID, BLGP);
}

double ha[64], hb[64], hc[64], hd[64];
double *da, *db, *dc, *dd;

template <int CBSZ, int ABID, int BLGP>
void run(const char* name) {
  hipLaunchKernelGGL((k<CBSZ, ABID, BLGP>), dim3(1), dim3(64), 0, 0, da, db, dc, dd);
  hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
  // hypothesis: A lane l: i = l&3, blk = (l>>2)&3, k = l>>4;  B lane l: j = l&3, blk = (l>>2)&3, k = l>>4
  //             D lane l: j = l&3, blk = (l>>2)&3, i = l>>4;  cbsz=2: A block ABID is used for every block slot
  double sa = (BLGP & 1) ? -1 : 1, sb = (BLGP & 2) ? -1 : 1, sc = (BLGP & 4) ? -1 : 1;
  double worst = 0;
  for (int l = 0; l < 64; ++l) {
    int j = l & 3, blk = (l >> 2) & 3, i = l >> 4;
    int ablk = CBSZ == 2 ? ABID : (CBSZ == 1 ? ((blk & ~1) | (ABID & 1)) : blk);
    double e = sc * hc[l];
    for (int kk = 0; kk < 4; ++kk) e += sa * ha[i + 4 * ablk + 16 * kk] * sb * hb[j + 4 * blk + 16 * kk];
    worst = fmax(worst, fabs(e - hd[l]));
  }
  printf("%-28s max |expected - got| = %.3e  %s\n", name, worst, worst < 1e-12 ? "OK" : "MISMATCH");
}

int main() {
  srand(1);
  for (int i = 0; i < 64; ++i) { ha[i] = rand() / (double)RAND_MAX - 0.5; hb[i] = rand() / (double)RAND_MAX - 0.5; hc[i] = rand() / (double)RAND_MAX - 0.5; }
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dc, 512); hipMalloc(&dd, 512);
  hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice); hipMemcpy(dc, hc, 512, hipMemcpyHostToDevice);
  run<0, 0, 0>("cbsz=0");
  run<2, 0, 0>("cbsz=2 abid=0");
  run<2, 1, 0>("cbsz=2 abid=1");
  run<2, 2, 0>("cbsz=2 abid=2");
  run<2, 3, 0>("cbsz=2 abid=3");
  run<1, 0, 0>("cbsz=1 abid=0");
  run<1, 1, 0>("cbsz=1 abid=1");
  run<0, 0, 1>("blgp=1 (neg A?)");
  run<0, 0, 2>("blgp=2 (neg B?)");
  run<0, 0, 4>("blgp=4 (neg C?)");
  run<2, 3, 1>("cbsz=2 abid=3 blgp=1");
  return 0;
}
