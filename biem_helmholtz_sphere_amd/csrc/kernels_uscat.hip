// kernels_uscat.hip -- K6: scattered-field evaluation (reference biem_u, _biem.py:822-977).
//   near : u(x) = sum_b sum_h density[b][h] blc_n(rho_b) h_n(k |x - c_b|) Y_h(dir(x - c_b))     (_biem.py:896-966)
//   far  : no radial factor, times (-i)^n e^{-i k x.c_b} / (i k)^{(d-1)/2}; Y is still taken at dir(x - c_b),
//          exactly as the reference does (_biem.py:885,930-959)
//   NaN fill where the point is inside a ball (outer) / outside (inner)                          (_biem.py:971-976)
//   kind = "inner": the points left valid lie INSIDE the spheres (r <= rho), where the layer potentials of a density Y_h on
//   |y| = rho expand in the regular functions: slc_in = i k^{d-2} rho^{d-1} h_n(k rho) j_n(k r), dlc_in = i k^{d-1} rho^{d-1}
//   h_n'(k rho) j_n(k r) (j and h exchange roles across the sphere; potential_coef(x_abs = r, y_abs = rho) of the un-vendored
//   ultrasphere presumably selects it - parity unpinned, no reference fixture has kind = "inner").  The exterior form is
//   singular at r -> 0 and is not the potential there.  tests: jump relation u(rho+) - u(rho-) = density . Y, regularity at
//   r = 0, Helmholtz residual inside the ball.
#include "common.hpp"

namespace biem {

constexpr int kMaxRadU = 320;

// c[s][b][h] = density * blc_{n(h)}(rho_b) (inner: the interior coefficient with h_n, h_n' at k rho): one wave per (system, ball)
__global__ void __launch_bounds__(64) k_uscat_coef(int d, int H, int n_end, const int* __restrict__ deg, int B, int inner,
                                                    const cplx* __restrict__ k, const double* __restrict__ eta,
                                                    const double* __restrict__ radii, int geom_batched,
                                                    const cplx* __restrict__ dens, cplx* __restrict__ c, cplx* __restrict__ scratch) {
  __shared__ cplx sJl[kMaxRadU + 3], sHl[kMaxRadU + 3];
  __shared__ cplx sBl[kMaxRadU];
  int b = blockIdx.x, s = blockIdx.y;
  // orders beyond kMaxRadU (2-D only): 3 (n_end + 3) complex of global scratch per block (written by thread 0, read after the barrier)
  cplx* sJ = scratch ? scratch + ((size_t)s * gridDim.x + b) * 3 * (n_end + 3) : sJl;
  cplx* sH = scratch ? sJ + (n_end + 3) : sHl;
  cplx* sB = scratch ? sH + (n_end + 3) : sBl;
  const cplx kk = k[s];
  const double et = eta[s];
  double rho = radii[(geom_batched ? (size_t)s * B : 0) + b];
  if (threadIdx.x == 0) {
    const cplx x = cscale(kk, rho), ix = crecip(x);
    radial_jh(d, n_end, x, sJ, sH);
    double rp = 1.0; for (int q = 0; q < d - 1; ++q) rp *= rho;
    cplx kd2 = make_double2(1.0, 0.0); for (int q = 0; q < d - 2; ++q) kd2 = cmul(kd2, kk);
    for (int n = 0; n < n_end; ++n) {
      const cplx j = inner ? sH[n] : sJ[n];
      const cplx kjp = cmul(kk, csub(cscale(cmul(ix, j), (double)n), inner ? sH[n + 1] : sJ[n + 1]));
      sB[n] = cscale(cmul(kd2, make_double2(et * j.x - kjp.y, et * j.y + kjp.x)), rp);   // blc = k^{d-2} rho^{d-1} (eta j + i k j')
    }
  }
  __syncthreads();
  size_t base = ((size_t)s * B + b) * H;
  for (int h = threadIdx.x; h < H; h += 64) c[base + h] = cmul(dens[base + h], sB[deg[h]]);
}

__device__ inline double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__global__ void __launch_bounds__(256) k_uscat(int tree, int d, int H, int n_end, const int* __restrict__ labels,
                                                const int* __restrict__ deg, int nb, int B, int P, const cplx* __restrict__ k,
                                                const double* __restrict__ centers, const double* __restrict__ radii,
                                                int geom_batched, const cplx* __restrict__ c, const double* __restrict__ pts,
                                                int flags, cplx* __restrict__ out) {
  __shared__ cplx sJ[4][kMaxRadU + 3], sH[4][kMaxRadU + 3];
  __shared__ int sBad;
  extern __shared__ cplx sBall[];   // [B]
  const int p = blockIdx.x, s = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool far = (flags & BIEM_USCAT_FAR_FIELD) != 0, per_ball = (flags & BIEM_USCAT_PER_BALL) != 0;
  const bool inner = (flags & BIEM_USCAT_KIND_INNER) != 0, pb = (flags & BIEM_USCAT_POINTS_BATCHED) != 0;
  if (threadIdx.x == 0) sBad = 0;
  double x[4];
  for (int i = 0; i < d; ++i) x[i] = pb ? pts[((size_t)i * P + p) * nb + s] : pts[(size_t)i * P + p];
  const cplx kk = k[s];
  __syncthreads();
  for (int bb = 0; bb < B; bb += 4) {
    const int b = bb + wave;
    const bool act = b < B;
    double rel[4] = {0, 0, 0, 0}, r = 1.0, xc = 0.0;
    if (act) {
      const double* cb = centers + ((geom_batched ? (size_t)s * B : 0) + b) * d;
      double rho = radii[(geom_batched ? (size_t)s * B : 0) + b];
      double r2 = 0.0;
      for (int i = 0; i < d; ++i) { rel[i] = x[i] - cb[i]; r2 += rel[i] * rel[i]; xc += x[i] * cb[i]; }
      r = sqrt(r2);
      if (lane == 0 && !far) {
        if ((!inner && r < rho) || (inner && r > rho)) atomicOr(&sBad, 1);
        if (r > 0.0) radial_jh(d, n_end - 1, cscale(kk, r), sJ[wave], sH[wave]);
        else {   // centre of a ball (inner kind): z_n(0) = delta_{n0} sqrt(pi/2) 2^{1-d/2} / Gamma(d/2); the exterior form has no value there
          const double z0 = d == 2 ? kSqrtHalfPi : d == 3 ? 1.0 : 0.5 * kSqrtHalfPi;
          const double qn = __longlong_as_double(0x7ff8000000000000LL);
          for (int n = 0; n < n_end; ++n) { sJ[wave][n] = make_double2(n == 0 ? z0 : 0.0, 0.0); sH[wave][n] = make_double2(qn, qn); }
        }
      }
    }
    __syncthreads();
    if (act) {
      Dir dir = make_dir(tree, rel);
      const cplx* cs = c + ((size_t)s * B + b) * H;
      double ar = 0.0, ai = 0.0;
      for (int h = lane; h < H; h += 64) {
        double yr, yi;
        harmonic_single(tree, labels[3 * h], labels[3 * h + 1], labels[3 * h + 2], dir, &yr, &yi);
        int n = deg[h];
        cplx rad;
        if (far) {
          // (-i)^n
          int q = n & 3;
          rad = q == 0 ? make_double2(1, 0) : q == 1 ? make_double2(0, -1) : q == 2 ? make_double2(-1, 0) : make_double2(0, 1);
        } else {
          rad = inner ? sJ[wave][n] : sH[wave][n];
        }
        cplx v = cmul(cmul(cs[h], rad), make_double2(yr, yi));
        ar += v.x; ai += v.y;
      }
      ar = wave_sum(ar); ai = wave_sum(ai);
      if (lane == 0) {
        cplx v = make_double2(ar, ai);
        if (far) {
          // e^{-i k x.c_b} / (i k)^{(d-1)/2} = exp(-i k x.c_b - p log(i k)),  p = (d-1)/2, principal branch (k > 0 real:
          // |k|^{-p} e^{-i (k x.c_b + pi p / 2)})
          const double pw = 0.5 * (d - 1);
          const cplx lik = zlog(make_double2(-kk.y, kk.x));                    // log(i k)
          v = cmul(v, zexp(make_double2(kk.y * xc - pw * lik.x, -kk.x * xc - pw * lik.y)));
        }
        sBall[b] = v;
      }
    }
    __syncthreads();
  }
  const double qnan = __longlong_as_double(0x7ff8000000000000LL);
  const bool bad = !far && sBad;
  if (per_ball) {
    for (int b = threadIdx.x; b < B; b += 256)
      out[((size_t)p * nb + s) * B + b] = bad ? make_double2(qnan, 0.0) : sBall[b];
  } else if (threadIdx.x == 0) {
    double ar = 0.0, ai = 0.0;
    for (int b = 0; b < B; ++b) { ar += sBall[b].x; ai += sBall[b].y; }
    out[(size_t)p * nb + s] = bad ? make_double2(qnan, 0.0) : make_double2(ar, ai);
  }
}

// ---------------------------------------------------------------------------------------------
// Near field (both kinds) and the far field, trees a (2-D), ba (3-D; bpa is ba in permuted axes), bba (4-D; bpbpa likewise) and
// caa: ONE POINT PER LANE.
// The generic kernel above spends one workgroup per (point, system), evaluates every harmonic from scratch (an O(n) Legendre
// recurrence and a sin / cos per harmonic) and leaves 63 lanes idle while lane 0 runs the radial recurrences: 2.0e6
// point-systems/s at cfg 3 (16 balls, H = 400) - the 100 x 100 plot grid of the reference's second hot loop took as long as the
// solve.  Here a lane owns a point and walks the harmonics by recurrence, all in registers:
//   h_n(k r): upward three-term recurrence from h_0, h_1 (the outgoing function is the dominant solution: stable), restarted
//             for every order m (n_end^2 / 2 extra steps - cheaper than an n_end-entry array per lane);
//   Pbar_n^m: the normalised recurrence of pbar_single with its square-root coefficients tabulated once per workgroup in LDS;
//   e^{i m phi}: rotation by (cos phi, sin phi) = (u1, u2) / |(u1, u2)| - no trigonometric call at all;
//   the +m and -m harmonics of a degree share Pbar and h;  the ball's coefficients c[h] = density * blc sit in LDS in (n, m)
//   order (every lane reads the same entry: a broadcast).
// ---------------------------------------------------------------------------------------------
constexpr int kFastNendMax3 = 48;          // LDS: 2 n_end^2 doubles of recurrence coefficients + n_end^2 complex of c
constexpr int kFastNendMax2 = kMaxRadU;    // 2-D: 2 n_end - 1 complex of c
constexpr int kFastNendMax4 = 14;          // 4-D (bba): c in a dense [n][l][m] store, n_end^2 (2 n_end - 1) complex = 85 KB at 14
constexpr int kFastNendMaxCaa = 12;        // 4-D (caa): c in a dense [n][m1][m2] store, n_end (2 n_end - 1)^2 complex = 101 KB at 12
// INNER (kind = "inner", near field): the radial factor is the regular function j_n(k r), whose upward recurrence is unstable
// for n > |k r|: each lane computes j_0 .. j_{n_end-1} once per ball by the backward recurrence of radial_jh into its own LDS row
// (64-thread workgroups; odd row stride: conflict-free 16-byte reads) and the harmonic loops read it by degree.
template <int TREE, bool FAR, bool INNER>
__global__ void __launch_bounds__(256) k_uscat_fast(int d, int H, int n_end, const int* __restrict__ labels, int nb, int B, int P,
                                                     const cplx* __restrict__ k, const double* __restrict__ centers,
                                                     const double* __restrict__ radii, int geom_batched, const cplx* __restrict__ c,
                                                     const double* __restrict__ pts, int flags, cplx* __restrict__ out) {
  extern __shared__ double sfast[];
  // 3-D: ra[q * n_end + m], rb[q * n_end + m] (q > m), cm[m]; then the ball's coefficients sC[pos]
  constexpr bool LEG = TREE == TREE_BA || TREE == TREE_BBA;       // a Legendre factor Pbar_l^m
  double* ra = sfast;
  double* rb = ra + (LEG ? n_end * n_end : 0);
  double* cmm = rb + (LEG ? n_end * n_end : 0);
  double* ga = cmm + (LEG ? ((n_end + 1) & ~1) : 0);              // bba: Gegenbauer a_q of order lam = l + 1 at [l * n_end + q]
  double* gia = ga + (TREE == TREE_BBA ? n_end * n_end : 0);       //      1 / a_q
  double* g0 = gia + (TREE == TREE_BBA ? n_end * n_end : 0);       //      p_0 = 1 / sqrt(h_0(l))
  // caa: the Jacobi recurrence p_{m+1} = (jA x + jB) p_m - jC p_{m-1} of cbar_single and its norm, at [(a * n_end + b) * K2 + m]
  const int K2 = (n_end + 1) / 2, ncaa = TREE == TREE_CAA ? n_end * n_end * K2 : 0;
  double* jA = g0 + (TREE == TREE_BBA ? ((n_end + 1) & ~1) : 0);
  double* jB = jA + ncaa;
  double* jC = jB + ncaa;
  double* jN = jC + ncaa;
  cplx* sC = (cplx*)(jN + ncaa);
  const int ms = 2 * n_end - 1;
  const int nC = TREE == TREE_BA ? n_end * n_end : TREE == TREE_BBA ? n_end * n_end * ms : TREE == TREE_CAA ? n_end * ms * ms : ms;
  const int js = (n_end + 2) | 1;           // INNER: row stride of the per-lane j_n store (radial_jh wants n_end + 1 slots at d = 4)
  const int s = blockIdx.y, tid = threadIdx.x, T = blockDim.x;
  cplx* sJl = sC + nC + (size_t)tid * js;
  const int p = blockIdx.x * T + tid, pc = p < P ? p : P - 1;
  const bool per_ball = (flags & BIEM_USCAT_PER_BALL) != 0, pb = (flags & BIEM_USCAT_POINTS_BATCHED) != 0;
  if (LEG) {
    for (int e = tid; e < n_end * n_end; e += T) {
      const int q = e / n_end, m = e - q * n_end;
      double a = 0.0, b = 0.0;
      if (q > m) {
        a = sqrt((double)(4 * q * q - 1) / (double)(q * q - m * m));
        b = sqrt((double)((q - 1) * (q - 1) - m * m) / (double)(4 * (q - 1) * (q - 1) - 1));
      }
      ra[e] = a; rb[e] = b;
    }
    for (int m = tid; m < n_end; m += T) cmm[m] = m == 0 ? 0.0 : sqrt((double)(2 * m + 1) / (double)(2 * m));
  }
  if (TREE == TREE_CAA) {
    for (int e = tid; e < ncaa; e += T) {
      const int a = e / (n_end * K2), b = (e / K2) % n_end, m = e % K2;
      const double al = (double)b, be = (double)a;            // Jacobi P^{(alpha = b, beta = a)}
      double A, Bc, C;
      if (m == 0) { A = 0.5 * (al + be + 2.0); Bc = (al + 1.0) - A; C = 0.0; }
      else {
        const double t = 2.0 * m + al + be, den = 2.0 * (m + 1.0) * (m + al + be + 1.0) * t;
        A = (t + 1.0) * (t + 2.0) * t / den; Bc = (t + 1.0) * (al * al - be * be) / den; C = 2.0 * (m + al) * (m + be) * (t + 2.0) / den;
      }
      double nr = 2.0 * (2.0 * m + a + b + 1.0);
      for (int i = 1; i <= a; ++i) nr *= (double)(m + b + i) / (double)(m + i);
      jA[e] = A; jB[e] = Bc; jC[e] = C; jN[e] = sqrt(nr);
    }
  }
  if (TREE == TREE_BBA) {                   // the coefficients of gbar_single
    for (int e = tid; e < n_end * n_end; e += T) {
      const int l = e / n_end, q = e - l * n_end;
      const double lam = (double)(l + 1);
      const double aq = q == 0 ? 1.0 : 0.5 * sqrt((double)q * ((double)q + 2.0 * lam - 1.0) / (((double)q + lam - 1.0) * ((double)q + lam)));
      ga[e] = q == 0 ? 0.0 : aq; gia[e] = 1.0 / aq;
    }
    for (int l = tid; l < n_end; l += T) {
      double h0 = 0.5 * kPi;
      for (int i = 1; i <= l; ++i) h0 *= ((double)i + 0.5) / ((double)i + 1.0);
      g0[l] = 1.0 / sqrt(h0);
    }
  }
  double x[4];
  for (int i = 0; i < d; ++i) x[i] = pb ? pts[((size_t)i * P + pc) * nb + s] : pts[(size_t)i * P + pc];
  const cplx kk = k[s];
  bool bad = false;
  double tr = 0.0, ti = 0.0;
  const double dd2 = (double)(d - 2);
  for (int b = 0; b < B; ++b) {
    __syncthreads();                       // the previous ball's coefficients are no longer read (and the tables are written)
    const cplx* cs = c + ((size_t)s * B + b) * H;
    for (int h = tid; h < H; h += T) {
      int pos;
      if (TREE == TREE_BA) { const int n = labels[3 * h], m = labels[3 * h + 1]; pos = n * n + n + m; }
      else if (TREE == TREE_BBA) pos = (labels[3 * h] * n_end + labels[3 * h + 1]) * (2 * n_end - 1) + labels[3 * h + 2] + n_end - 1;
      else if (TREE == TREE_CAA) pos = (labels[3 * h] * ms + labels[3 * h + 1] + n_end - 1) * ms + labels[3 * h + 2] + n_end - 1;
      else pos = labels[3 * h] + n_end - 1;
      sC[pos] = cs[h];
    }
    __syncthreads();
    const double* cb = centers + ((geom_batched ? (size_t)s * B : 0) + b) * d;
    const double rho = radii[(geom_batched ? (size_t)s * B : 0) + b];
    double u[4] = {0.0, 0.0, 0.0, 0.0}, r2 = 0.0;
    for (int i = 0; i < d; ++i) { u[i] = x[i] - cb[i]; r2 += u[i] * u[i]; }
    const double r = sqrt(r2);
    if (!FAR && (INNER ? r > rho : r < rho)) bad = true;
    // h_0, h_1 at k r (r = 0 only inside a ball: the value is discarded).  Far field: the radial factor is (-i)^n, i.e. the
    // "recurrence" h_{n+1} = -i h_n from h_0 = 1 (advance() below)
    cplx h0 = make_double2(1.0, 0.0), h1 = make_double2(0.0, -1.0), ix = make_double2(0.0, 0.0);
    if (INNER) {
      if (r > 0.0) radial_jh(d, n_end - 1, cscale(kk, r), (zc*)sJl, nullptr);
      else {                                // centre of the ball: z_n(0) = delta_{n0} sqrt(pi/2) 2^{1-d/2} / Gamma(d/2)
        const double z0 = d == 2 ? kSqrtHalfPi : d == 3 ? 1.0 : 0.5 * kSqrtHalfPi;
        for (int n = 0; n < n_end; ++n) sJl[n] = make_double2(n == 0 ? z0 : 0.0, 0.0);
      }
    } else if (!FAR) {
      zc J2[4], H2[4];
      radial_jh(d, 1, cscale(kk, r > 0.0 ? r : rho), J2, H2);
      h0 = H2[0]; h1 = H2[1];
      ix = crecip(cscale(kk, r > 0.0 ? r : rho));
    }
    // next of (h_{q-1}, h_q): h_{q+1} = ((2 q + d - 2) / x) h_q - h_{q-1}
    auto advance = [&](const cplx& hprev, const cplx& hcur, double two_q) -> cplx {
      if (FAR) return make_double2(hcur.y, -hcur.x);
      return csub(cmul(cscale(ix, two_q + dd2), hcur), hprev);
    };
    auto radial = [&](int n, const cplx& hup) -> cplx { if (INNER) return sJl[n]; return hup; };   // the radial factor of degree n
    double ar = 0.0, ai = 0.0;
    if (TREE == TREE_A) {
      // Y_m = e^{i m theta} / sqrt(2 pi); degree n = |m|
      const double e1x = r > 0.0 ? u[0] / r : 1.0, e1y = r > 0.0 ? u[1] / r : 0.0;
      double ex = 1.0, ey = 0.0;
      cplx hp = h0, hc = h1;               // h_n, h_{n+1}
      for (int n = 0; n < n_end; ++n) {
        const cplx cp = sC[n_end - 1 + n];
        cplx t = make_double2(cp.x * ex - cp.y * ey, cp.x * ey + cp.y * ex);
        if (n > 0) { const cplx cn = sC[n_end - 1 - n]; t.x += cn.x * ex + cn.y * ey; t.y += cn.y * ex - cn.x * ey; }
        const cplx hv = radial(n, hp);
        ar += hv.x * t.x - hv.y * t.y; ai += hv.x * t.y + hv.y * t.x;
        const cplx hn = advance(hp, hc, 2.0 * n + 2.0);
        hp = hc; hc = hn;
        const double nx = ex * e1x - ey * e1y; ey = ex * e1y + ey * e1x; ex = nx;
      }
    } else if (TREE == TREE_BBA) {
      // Y_{n l m} = s0^l g_{n-l}^{(l+1)}(c0) Pbar_l^{|m|}(c1) e^{i m phi} / sqrt(2 pi): three nested recurrences; (h_m, h_{m+1})
      // runs along m, (h_l, h_{l+1}) along l from it, (h_n, h_{n+1}) along n from that - no restart from h_0
      const double rho2 = sqrt(u[2] * u[2] + u[3] * u[3]), rho1 = sqrt(u[1] * u[1] + rho2 * rho2);
      const double c0 = r > 0.0 ? u[0] / r : 1.0, s0 = r > 0.0 ? rho1 / r : 0.0;
      const double c1 = rho1 > 0.0 ? u[1] / rho1 : 1.0, s1 = rho1 > 0.0 ? rho2 / rho1 : 0.0;
      const double e1x = rho2 > 0.0 ? u[2] / rho2 : 1.0, e1y = rho2 > 0.0 ? u[3] / rho2 : 0.0;
      const int mstride = 2 * n_end - 1;
      double ex = 1.0, ey = 0.0, pmm = 0.70710678118654752440, s0m = 1.0;
      cplx hm = h0, hm1 = h1;
      for (int m = 0; m < n_end; ++m) {
        if (m > 0) {
          pmm *= cmm[m] * s1; s0m *= s0;
          const cplx hn = advance(hm, hm1, 2.0 * m);
          hm = hm1; hm1 = hn;
          const double nx = ex * e1x - ey * e1y; ey = ex * e1y + ey * e1x; ex = nx;
        }
        double p0 = 0.0, p1 = pmm, sl = s0m;
        cplx hl = hm, hl1 = hm1;
        double sr = 0.0, si = 0.0, qr = 0.0, qi = 0.0;
        for (int l = m; l < n_end; ++l) {
          double gp0 = 0.0, gp1 = g0[l];
          cplx hp = hl, hc = hl1;
          const double alm = sl * p1;
          for (int n = l; n < n_end; ++n) {
            const double amp = alm * gp1;
            const cplx hv = radial(n, hp);
            const double wr = hv.x * amp, wi = hv.y * amp;
            const cplx* cc = sC + (n * n_end + l) * mstride + n_end - 1;
            const cplx cp = cc[m];
            sr += wr * cp.x - wi * cp.y; si += wr * cp.y + wi * cp.x;
            if (m > 0) { const cplx cn = cc[-m]; qr += wr * cn.x - wi * cn.y; qi += wr * cn.y + wi * cn.x; }
            const int q = n - l + 1;
            if (n + 1 < n_end) {
              const double gp2 = (c0 * gp1 - ga[l * n_end + q - 1] * gp0) * gia[l * n_end + q];
              gp0 = gp1; gp1 = gp2;
              const cplx hn = advance(hp, hc, 2.0 * (n + 1));
              hp = hc; hc = hn;
            }
          }
          const int ql = l + 1;
          if (ql < n_end) {
            const double p2 = ra[ql * n_end + m] * (c1 * p1 - rb[ql * n_end + m] * p0);
            p0 = p1; p1 = p2;
            sl *= s0;
            const cplx hn = advance(hl, hl1, 2.0 * ql);
            hl = hl1; hl1 = hn;
          }
        }
        ar += sr * ex - si * ey + qr * ex + qi * ey;
        ai += sr * ey + si * ex + qi * ex - qr * ey;
      }
    } else if (TREE == TREE_CAA) {
      // Y_{n m1 m2} = cos^a sin^b Pbar_k^{(b,a)}(cos 2 t0) e^{i (m1 t1 + m2 t2)} / (2 pi), a = |m1|, b = |m2|, n = a + b + 2 k: the
      // Jacobi recurrence runs along k inside (a, b); the four sign combinations share it and the radial factor
      const double r01 = sqrt(u[0] * u[0] + u[1] * u[1]), r23 = sqrt(u[2] * u[2] + u[3] * u[3]);
      const double c0 = r > 0.0 ? r01 / r : 1.0, s0 = r > 0.0 ? r23 / r : 0.0, xx = c0 * c0 - s0 * s0;
      const double e1x = r01 > 0.0 ? u[0] / r01 : 1.0, e1y = r01 > 0.0 ? u[1] / r01 : 0.0;
      const double e2x = r23 > 0.0 ? u[2] / r23 : 1.0, e2y = r23 > 0.0 ? u[3] / r23 : 0.0;
      double ca = 1.0, eax = 1.0, eay = 0.0;
      cplx ha = h0, ha1 = h1;              // h_a, h_{a+1}
      for (int a = 0; a < n_end; ++a) {
        if (a > 0) {
          ca *= c0;
          const cplx hn = advance(ha, ha1, 2.0 * a);
          ha = ha1; ha1 = hn;
          const double nx = eax * e1x - eay * e1y; eay = eax * e1y + eay * e1x; eax = nx;
        }
        double sb = 1.0, ebx = 1.0, eby = 0.0;
        cplx hb = ha, hb1 = ha1;           // h_{a+b}, h_{a+b+1}
        for (int b = 0; a + b < n_end; ++b) {
          if (b > 0) {
            sb *= s0;
            const cplx hn = advance(hb, hb1, 2.0 * (a + b));
            hb = hb1; hb1 = hn;
            const double nx = ebx * e2x - eby * e2y; eby = ebx * e2y + eby * e2x; ebx = nx;
          }
          const int tb = (a * n_end + b) * K2;
          const double amp0 = ca * sb;
          double p0 = 0.0, p1 = 1.0;
          cplx hp = hb, hc = hb1;
          double ppr = 0.0, ppi = 0.0, mpr = 0.0, mpi = 0.0, pmr = 0.0, pmi = 0.0, mmr = 0.0, mmi = 0.0;   // sums of the (+-a, +-b) coefficients
          for (int kq = 0, n = a + b; n < n_end; ++kq, n += 2) {
            const cplx hv = radial(n, hp);
            const double amp = amp0 * jN[tb + kq] * p1;
            const double wr = hv.x * amp, wi = hv.y * amp;
            const cplx* cc = sC + (n * ms + n_end - 1) * ms + n_end - 1;
            { const cplx cv = cc[a * ms + b]; ppr += wr * cv.x - wi * cv.y; ppi += wr * cv.y + wi * cv.x; }
            if (a > 0) { const cplx cv = cc[-a * ms + b]; mpr += wr * cv.x - wi * cv.y; mpi += wr * cv.y + wi * cv.x; }
            if (b > 0) { const cplx cv = cc[a * ms - b]; pmr += wr * cv.x - wi * cv.y; pmi += wr * cv.y + wi * cv.x; }
            if (a > 0 && b > 0) { const cplx cv = cc[-a * ms - b]; mmr += wr * cv.x - wi * cv.y; mmi += wr * cv.y + wi * cv.x; }
            if (n + 2 < n_end) {
              const double p2 = (jA[tb + kq] * xx + jB[tb + kq]) * p1 - jC[tb + kq] * p0;
              p0 = p1; p1 = p2;
              cplx hn = advance(hp, hc, 2.0 * (n + 1));
              hp = hc; hc = hn;
              hn = advance(hp, hc, 2.0 * (n + 2));
              hp = hc; hc = hn;
            }
          }
          // e^{i (+-a t1 +- b t2)}
          const double fx = eax * ebx - eay * eby, fy = eax * eby + eay * ebx;     // e^{i (a t1 + b t2)}
          const double gx = eax * ebx + eay * eby, gy = eax * eby - eay * ebx;     // e^{i (-a t1 + b t2)}
          ar += ppr * fx - ppi * fy + mmr * fx + mmi * fy + mpr * gx - mpi * gy + pmr * gx + pmi * gy;
          ai += ppr * fy + ppi * fx + mmi * fx - mmr * fy + mpr * gy + mpi * gx + pmi * gx - pmr * gy;
        }
      }
      ar *= kInvSqrt2Pi; ai *= kInvSqrt2Pi;   // (the second 1 / sqrt(2 pi) below)
    } else {
      const double rxy = sqrt(u[1] * u[1] + u[2] * u[2]);
      const double c0 = r > 0.0 ? u[0] / r : 1.0, s0 = r > 0.0 ? rxy / r : 0.0;
      const double e1x = rxy > 0.0 ? u[1] / rxy : 1.0, e1y = rxy > 0.0 ? u[2] / rxy : 0.0;
      double ex = 1.0, ey = 0.0, pmm = 0.70710678118654752440;
      cplx hm = h0, hm1 = h1;              // h_m, h_{m+1}: advanced by one per order m
      for (int m = 0; m < n_end; ++m) {
        if (m > 0) {
          pmm *= cmm[m] * s0;
          const cplx hn = advance(hm, hm1, 2.0 * m);
          hm = hm1; hm1 = hn;
          const double nx = ex * e1x - ey * e1y; ey = ex * e1y + ey * e1x; ex = nx;
        }
        cplx hp = hm, hc = hm1;            // h_n, h_{n+1} for n = m ..
        double p0 = 0.0, p1 = pmm;
        double sr = 0.0, si = 0.0;         // sum over n of h_n Pbar_n^m c_{n, +-m} (the e^{+- i m phi} factors applied once per m)
        double qr = 0.0, qi = 0.0;
        for (int n = m; n < n_end; ++n) {
          const cplx cp = sC[n * n + n + m];
          const cplx hv = radial(n, hp);
          const double wr = hv.x * p1, wi = hv.y * p1;
          sr += wr * cp.x - wi * cp.y; si += wr * cp.y + wi * cp.x;
          if (m > 0) { const cplx cn = sC[n * n + n - m]; qr += wr * cn.x - wi * cn.y; qi += wr * cn.y + wi * cn.x; }
          const int q = n + 1;
          if (q < n_end) {
            const double p2 = ra[q * n_end + m] * (c0 * p1 - rb[q * n_end + m] * p0);
            p0 = p1; p1 = p2;
            const cplx hn = advance(hp, hc, 2.0 * q);
            hp = hc; hc = hn;
          }
        }
        ar += sr * ex - si * ey + qr * ex + qi * ey;
        ai += sr * ey + si * ex + qi * ex - qr * ey;
      }
    }
    ar *= kInvSqrt2Pi; ai *= kInvSqrt2Pi;
    if (FAR) {
      // e^{-i k x.c_b} / (i k)^{(d-1)/2}, as in the generic kernel
      double xc = 0.0;
      for (int i = 0; i < d; ++i) xc += x[i] * cb[i];
      const double pw = 0.5 * (d - 1);
      const cplx lik = zlog(make_double2(-kk.y, kk.x));
      const cplx v = cmul(make_double2(ar, ai), zexp(make_double2(kk.y * xc - pw * lik.x, -kk.x * xc - pw * lik.y)));
      ar = v.x; ai = v.y;
    }
    if (per_ball) { if (p < P) out[((size_t)p * nb + s) * B + b] = make_double2(ar, ai); }
    else { tr += ar; ti += ai; }
  }
  const double qnan = __longlong_as_double(0x7ff8000000000000LL);
  if (p >= P) return;
  if (per_ball) {
    if (bad) for (int b = 0; b < B; ++b) out[((size_t)p * nb + s) * B + b] = make_double2(qnan, 0.0);
  } else out[(size_t)p * nb + s] = bad ? make_double2(qnan, 0.0) : make_double2(tr, ti);
}

int launch_uscat(const biem_plan* p, int nb, int B, int P, const double* d_k, const double* d_eta, const double* d_centers,
                 const double* d_radii, int geom_batched, const double* d_density, const double* d_points, int flags,
                 double* d_out, void* d_work, size_t work_bytes, hipStream_t st) {
  if (nb <= 0 || B <= 0 || P <= 0) return BIEM_OK;
  // orders beyond the LDS tables: 2-D only, through the per-lane kernel (its only table is the ball's 2 n_end - 1 coefficients)
  const bool big = p->n_end > kMaxRadU;
  if (big && (p->tree != TREE_A || (size_t)(2 * p->n_end - 1) * sizeof(cplx) > 150 * 1024 ||
              ((flags & BIEM_USCAT_KIND_INNER) && !(flags & BIEM_USCAT_FAR_FIELD)) || nb > 65535)) {
    set_error("biem_uscat: n_end=%d too large for this tree / kind", p->n_end); return BIEM_ERR_UNSUPPORTED;
  }
  size_t need = (size_t)nb * B * p->H * sizeof(cplx);
  if (work_bytes < need) { set_error("biem_uscat: workspace too small"); return BIEM_ERR_ARG; }
  cplx* c = (cplx*)d_work;
  cplx* scratch = nullptr;
  if (big) BIEM_HIPCHK(hipMallocAsync((void**)&scratch, (size_t)nb * B * 3 * (p->n_end + 3) * sizeof(cplx), st));
  hipLaunchKernelGGL(k_uscat_coef, dim3(B, nb), dim3(64), 0, st, p->d, p->H, p->n_end, p->d_deg, B,
                     (flags & BIEM_USCAT_KIND_INNER) && !(flags & BIEM_USCAT_FAR_FIELD) ? 1 : 0, (const cplx*)d_k, d_eta, d_radii,
                     geom_batched, (const cplx*)d_density, c, scratch);
  BIEM_LAUNCHCHK();
  if (scratch) BIEM_HIPCHK(hipFreeAsync(scratch, st));
  // the far field does not depend on the kind; the near field of kind inner reads j_n from a per-lane LDS row (INNER)
  const bool far = (flags & BIEM_USCAT_FAR_FIELD) != 0;
  const bool inner = !far && (flags & BIEM_USCAT_KIND_INNER);
  const int ne = p->n_end, ms = 2 * ne - 1;
  const bool fast_tree = (p->tree == TREE_BA && ne <= kFastNendMax3) || (p->tree == TREE_A && (big || ne <= kFastNendMax2)) ||
                         (p->tree == TREE_BBA && ne <= kFastNendMax4) || (p->tree == TREE_CAA && ne <= kFastNendMaxCaa);
  if (fast_tree && !getenv("BIEM_USCAT_GENERIC") && nb <= 65535) {
    const int T = inner ? 64 : 256;
    size_t tab = 0, nC = 0;               // doubles of tables, complex of coefficients (the layout at the head of k_uscat_fast)
    if (p->tree == TREE_BA) { tab = (size_t)2 * ne * ne + ((ne + 1) & ~1); nC = (size_t)ne * ne; }
    else if (p->tree == TREE_BBA) { tab = (size_t)4 * ne * ne + 2 * ((ne + 1) & ~1); nC = (size_t)ne * ne * ms; }
    else if (p->tree == TREE_CAA) { tab = (size_t)4 * ne * ne * ((ne + 1) / 2); nC = (size_t)ne * ms * ms; }
    else nC = (size_t)ms;
    const size_t shm = tab * sizeof(double) + (nC + (inner ? (size_t)T * ((ne + 2) | 1) : 0)) * sizeof(cplx);
    if (shm <= 160 * 1024) {
#define BIEM_USCAT_FAST(TREE, FARF, INNERF)                                                                                      \
  {                                                                                                                              \
    BIEM_HIPCHK(hipFuncSetAttribute((const void*)k_uscat_fast<TREE, FARF, INNERF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
    hipLaunchKernelGGL((k_uscat_fast<TREE, FARF, INNERF>), dim3((P + T - 1) / T, nb), dim3(T), shm, st, p->d, p->H, ne, p->d_labels, nb, \
                       B, P, (const cplx*)d_k, d_centers, d_radii, geom_batched, (const cplx*)c, d_points, flags, (cplx*)d_out); \
  }
#define BIEM_USCAT_FAST_TREE(TREE)                                                                                               \
  { if (far) BIEM_USCAT_FAST(TREE, true, false) else if (inner) BIEM_USCAT_FAST(TREE, false, true) else BIEM_USCAT_FAST(TREE, false, false) }
      if (p->tree == TREE_BA) BIEM_USCAT_FAST_TREE(TREE_BA)
      else if (p->tree == TREE_BBA) BIEM_USCAT_FAST_TREE(TREE_BBA)
      else if (p->tree == TREE_CAA) BIEM_USCAT_FAST_TREE(TREE_CAA)
      else BIEM_USCAT_FAST_TREE(TREE_A)
#undef BIEM_USCAT_FAST_TREE
#undef BIEM_USCAT_FAST
      BIEM_LAUNCHCHK();
      return BIEM_OK;
    }
  }
  hipLaunchKernelGGL(k_uscat, dim3(P, nb), dim3(256), (size_t)B * sizeof(cplx), st, p->tree, p->d, p->H, p->n_end, p->d_labels,
                     p->d_deg, nb, B, P, (const cplx*)d_k, d_centers, d_radii, geom_batched, (const cplx*)c, d_points, flags, (cplx*)d_out);
  BIEM_LAUNCHCHK();
  return BIEM_OK;
}

}  // namespace biem
