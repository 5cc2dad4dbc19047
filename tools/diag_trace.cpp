// diag_trace.cpp -- timeline of the diagonal-block kernel (k_diag_utu_reg) of the symmetric factorisation; build: tools/build_diag.sh
#include <cstdio>
#include <vector>
#ifndef BIEM_DIAG_THREADS
#define BIEM_DIAG_THREADS 1024
#endif
constexpr int NWV = BIEM_DIAG_THREADS / 64;
extern "C" int biem_debug_diag(int reps, unsigned long long* trace_out, float* us_out);
int main() {
  std::vector<unsigned long long> t(16 * 66 * 4);
  float us = 0;
  if (biem_debug_diag(12, t.data(), &us)) { printf("failed\n"); return 1; }
  auto T = [&](int w, int s, int i) { return t[((size_t)w * 66 + s) * 4 + i]; };
  printf("%.1f us per launch (events, back to back)\n", us);
  const unsigned long long t0 = T(0, 64, 0);
  printf("wave 0 (s_memtime ticks, 100 MHz): start 0, loads done + row 0 published %llu, loop end %llu, kernel end %llu\n", T(0, 64, 1) - t0, T(0, 64, 2) - t0, T(0, 64, 3) - t0);
  // per step: time from the publisher's stamp of step c-1 (row c published) to the publisher's stamp of step c
  double sum_step = 0, sum_bar = 0, sum_lds = 0, sum_upd = 0; int n = 0;
  for (int c = 1; c < 63; ++c) {
    const int wp = (c + 1) & (NWV - 1), wprev = c & (NWV - 1);
    const unsigned long long pub_prev = T(wprev, c - 1, 3), pub = T(wp, c, 3);
    const unsigned long long bar = T(wp, c, 1), lds = T(wp, c, 2);
    sum_step += (double)(pub - pub_prev); sum_bar += (double)(bar - pub_prev); sum_lds += (double)(lds - bar); sum_upd += (double)(pub - lds); ++n;
    if (c < 6 || c > 57) printf("  step %2d: published->barrier passed %llu, LDS reads %llu, update+publish %llu  (ticks)\n", c, bar - pub_prev, lds - bar, pub - lds);
  }
  printf("mean per step (ticks of 10 ns): total %.1f = wait for barrier %.1f + LDS reads %.1f + update and publish %.1f\n", sum_step / n, sum_bar / n, sum_lds / n, sum_upd / n);
  // how long after the publisher do the slowest waves reach the next barrier?
  double lag = 0;
  for (int c = 1; c < 63; ++c) { unsigned long long mx = 0; for (int w = 0; w < NWV; ++w) if (T(w, c + 1, 0) > mx) mx = T(w, c + 1, 0); lag += (double)mx - (double)T((c + 1) & (NWV - 1), c, 3); }
  printf("mean (last wave arrives at the next barrier) - (publisher done): %.1f ticks\n", lag / 62);
  return 0;
}
