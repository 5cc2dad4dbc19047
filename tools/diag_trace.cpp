// diag_trace.cpp -- timeline of the diagonal-block kernel (k_diag_utu_reg) of the symmetric factorisation; build: tools/build_diag.sh
#include <cstdio>
#include <cstdlib>
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
  printf("wave 0 (s_memtime: shader cycles): start 0, loads done + row 0 published %llu, loop end %llu, kernel end %llu\n", T(0, 64, 1) - t0, T(0, 64, 2) - t0, T(0, 64, 3) - t0);
  if (!getenv("BIEM_DIAG_FORM")) {
    // four pivots per barrier (k_diag_utu_blk): the chain is the owner of block b + 1 in block step b
    double s_bar = 0, s_app = 0, s_bc = 0, s_sc = 0, s_vec = 0, s_tot = 0; int n = 0;
    for (int b = 1; b < 15; ++b) {
      const int wc = b + 1;
      const unsigned long long pub_prev = T(b, 16 + b, 2), bar = T(wc, b, 1), app = T(wc, b, 2), bc = T(wc, 16 + wc, 0), sc = T(wc, 16 + wc, 1), vec = T(wc, 16 + wc, 2);
      s_bar += (double)(bar - pub_prev); s_app += (double)(app - bar); s_bc += (double)(bc - app); s_sc += (double)(sc - bc); s_vec += (double)(vec - sc); s_tot += (double)(vec - pub_prev); ++n;
      if (b < 4 || b > 11) printf("  block step %2d: published->barrier passed %llu, rank-4 update %llu, 10 broadcasts %llu, 4 x 4 elimination %llu, row updates + publication %llu (cycles)\n", b, bar - pub_prev, app - bar, bc - app, sc - bc, vec - sc);
    }
    printf("mean per block step (cycles): total %.0f = barrier %.0f + rank-4 update %.0f + broadcasts %.0f + 4 x 4 elimination %.0f + row updates and publication %.0f\n", s_tot / n, s_bar / n, s_app / n, s_bc / n, s_sc / n, s_vec / n);
    return 0;
  }
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
