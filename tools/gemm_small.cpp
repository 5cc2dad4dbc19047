// gemm_small.cpp -- time of the small update launches a single system issues (one 64-row strip x n columns, K = 64 .. 256), warm and cold.
// build: tools/build_small.sh (the library sources with -DBIEM_GEMM_TRACE)
#include <cstdio>
#include <cstdlib>
extern "C" int biem_debug_gemm_strip(int n, int kd, int rows, int reps, int cold, float* us_out, int extra_cols, int col0);
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 4096;
  for (int extra : {0, 1})
    for (int col0 : {0, 2048})
      for (int kd : {64, 128, 192, 256}) {
        float us = 0;
        if (biem_debug_gemm_strip(n, kd, 64, 50, 1, &us, extra, col0)) { printf("failed\n"); return 1; }
        printf("n=%d rows=64 cols %d..%d K=%3d: %.1f us per launch\n", n, col0, n + extra, kd, us);
      }
  return 0;
}
