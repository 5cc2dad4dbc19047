// special.hpp -- special functions shared by the host table builder and the gfx950 kernels.
//
// d-dimensional spherical Bessel functions z_n(x) = sqrt(pi/2) Z_{n+d/2-1}(x) / x^{d/2-1}
// (what the reference obtains from ultrasphere.shn1 / potential_coef, _biem.py:654-684,723-789)
// and orthonormal hyperspherical harmonics of the trees a / ba / bba (ush.harmonics, _biem.py:922).
// Written from the published mathematics; no third-party code.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define BIEM_HD __host__ __device__ inline

namespace biem {

constexpr double kPi = 3.14159265358979323846;
constexpr double kEulerGamma = 0.57721566490153286061;
constexpr double kSqrtHalfPi = 1.25331413731550025121;   // sqrt(pi/2)
constexpr double kInvSqrt2Pi = 0.39894228040143267794;   // 1/sqrt(2 pi)

enum Tree { TREE_A = 0, TREE_BA = 1, TREE_BBA = 2, TREE_CAA = 3 };

BIEM_HD int tree_dim(int tree) { return tree == TREE_A ? 2 : (tree == TREE_BA ? 3 : 4); }   // bba, caa: 4

// number of harmonics of degree < n
BIEM_HD int harm_count(int tree, int n) {
  if (n <= 0) return 0;
  if (tree == TREE_A) return 2 * n - 1;
  if (tree == TREE_BA) return n * n;
  return n * (n + 1) * (2 * n + 1) / 6;
}

// ---------------------------------------------------------------------------------------------
// Integer-order cylindrical Bessel functions J_0..J_nmax, Y_0..Y_nmax at real x > 0.
// J: Miller backward recurrence normalised by 1 = J_0 + 2 sum J_{2k}; Y_0, Y_1 from the Neumann series
//   Y_0 = (2/pi)(ln(x/2)+gamma) J_0 - (4/pi) sum_{k>=1} (-1)^k J_{2k}/k
//   Y_1 = (2/pi)(ln(x/2)+gamma) J_1 - (2/pi) J_0/x + (2/pi) sum_{k>=1} (-1)^k (J_{2k-1} - J_{2k+1})/k
// accumulated in the same backward pass; Y_n by (stable) forward recurrence.
// J, Y must hold nmax+1 doubles (any address space reachable by a generic pointer).
// ---------------------------------------------------------------------------------------------
BIEM_HD void bessel_jy_int(int nmax, double x, double* J, double* Y) {
  int M = (int)x;
  if (M < nmax + 1) M = nmax + 1;
  M += 32 + (int)sqrt(48.0 * (double)M);
  M += (M & 1);  // even
  const double big = 1e200, small = 1e-200;
  double jp1 = 0.0, jc = 1e-250;  // J_{M+1}, J_M (unnormalised)
  double norm = 0.0, s0 = 0.0, s1 = 0.0;
  const double tx = 2.0 / x;
  for (int k = M; k >= 1; --k) {
    // here jc = J_k, jp1 = J_{k+1}
    double jm1 = (double)k * tx * jc - jp1;  // J_{k-1}
    if ((k & 1) == 0) {
      int kk = k >> 1;
      double sg = (kk & 1) ? -1.0 : 1.0;  // (-1)^kk
      norm += 2.0 * jc;
      s0 += sg * jc / (double)kk;
      s1 += sg * (jm1 - jp1) / (double)kk;
    }
    if (k <= nmax) J[k] = jc;
    jp1 = jc;
    jc = jm1;
    if (fabs(jc) > big) {
      jc *= small; jp1 *= small; norm *= small; s0 *= small; s1 *= small;
      for (int q = (k <= nmax ? k : nmax + 1); q <= nmax; ++q) J[q] *= small;  // entries k..nmax are stored
    }
  }
  J[0] = jc;
  norm += jc;
  double inv = 1.0 / norm;
  for (int q = 0; q <= nmax; ++q) J[q] *= inv;
  s0 *= inv; s1 *= inv;
  if (Y == nullptr) return;                     // regular functions only
  double j0 = jc * inv, j1 = jp1 * inv;
  double lg = log(0.5 * x) + kEulerGamma;
  double y0 = (2.0 / kPi) * (lg * j0 - 2.0 * s0);
  double y1 = (2.0 / kPi) * (lg * j1 - j0 / x + s1);
  Y[0] = y0;
  if (nmax >= 1) Y[1] = y1;
  for (int n = 1; n < nmax; ++n) {
    double y2 = (double)n * tx * y1 - y0;
    Y[n + 1] = y2;
    y0 = y1; y1 = y2;
  }
}

// Spherical Bessel functions (d = 3) j_0..j_nmax, y_0..y_nmax at real x > 0.
BIEM_HD void bessel_jy_sph(int nmax, double x, double* J, double* Y) {
  int M = (int)x;
  if (M < nmax + 1) M = nmax + 1;
  M += 32 + (int)sqrt(48.0 * (double)M);
  const double big = 1e200, small = 1e-200;
  double jp1 = 0.0, jc = 1e-250;
  const double ix = 1.0 / x;
  for (int k = M; k >= 1; --k) {
    double jm1 = (double)(2 * k + 1) * ix * jc - jp1;
    if (k <= nmax) J[k] = jc;
    jp1 = jc;
    jc = jm1;
    if (fabs(jc) > big) {
      jc *= small; jp1 *= small;
      for (int q = (k <= nmax ? k : nmax + 1); q <= nmax; ++q) J[q] *= small;
    }
  }
  J[0] = jc;
  double s = sin(x), c = cos(x);
  double j0 = s * ix, j1 = (s * ix - c) * ix;
  double scale = (fabs(j0) >= fabs(j1)) ? j0 / jc : j1 / jp1;
  for (int q = 0; q <= nmax; ++q) J[q] *= scale;
  if (Y == nullptr) return;                     // regular functions only
  double y0 = -c * ix, y1 = (-c * ix - s) * ix;
  Y[0] = y0;
  if (nmax >= 1) Y[1] = y1;
  for (int n = 1; n < nmax; ++n) {
    double y2 = (double)(2 * n + 1) * ix * y1 - y0;
    Y[n + 1] = y2;
    y0 = y1; y1 = y2;
  }
}

// d-dimensional z_n, n = 0..nmax. For d = 4 the caller must provide nmax+2 slots (order n+1 is needed).
BIEM_HD void radial_d(int d, int nmax, double x, double* J, double* Y) {
  if (d == 3) {
    bessel_jy_sph(nmax, x, J, Y);
  } else if (d == 2) {
    bessel_jy_int(nmax, x, J, Y);
    for (int n = 0; n <= nmax; ++n) { J[n] *= kSqrtHalfPi; if (Y) Y[n] *= kSqrtHalfPi; }
  } else {  // d == 4: sqrt(pi/2) Z_{n+1}(x) / x
    bessel_jy_int(nmax + 1, x, J, Y);
    double f = kSqrtHalfPi / x;
    for (int n = 0; n <= nmax; ++n) { J[n] = J[n + 1] * f; if (Y) Y[n] = Y[n + 1] * f; }
  }
}

// ---------------------------------------------------------------------------------------------
// Complex argument (complex wavenumber, reference gui.py:296-301).  Outputs are the regular functions z_n = J and the
// OUTGOING ones h_n = j_n + i y_n directly: for Im z > 0 both j and y grow like e^{Im z} while h decays, so h is never
// formed as a sum.  (zc = double2: .x real, .y imaginary.)
//   d = 3 : j_n by Miller's backward recurrence normalised with the larger of j_0 = sin z / z, j_1;  h_0 = -i e^{iz}/z,
//           h_1 = -(z + i) e^{iz}/z^2, h_n by forward recurrence (h is the dominant solution in n).
//   d = 2,4: J_n by Miller's recurrence normalised with e^{-+iz} = J_0 + 2 sum_k (-+i)^k J_k (the sign that makes the
//           right-hand side the large exponential); H_0, H_1 from the modified functions K_0, K_1 at w = -iz:
//           H_0 = (2/(pi i)) K_0(w), H_1 = -(2/pi) K_1(w), with K by power series for |w| <= 2 and Steed's continued
//           fraction (CF2) otherwise; Im z < 0 through H1(z) = 2 J(z) - conj(H1(conj z)).  H_n by forward recurrence.
// ---------------------------------------------------------------------------------------------
typedef double2 zc;
BIEM_HD zc zmk(double a, double b) { zc r; r.x = a; r.y = b; return r; }
BIEM_HD zc zadd(zc a, zc b) { return zmk(a.x + b.x, a.y + b.y); }
BIEM_HD zc zsub(zc a, zc b) { return zmk(a.x - b.x, a.y - b.y); }
BIEM_HD zc zmul(zc a, zc b) { return zmk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
BIEM_HD zc zscl(zc a, double s) { return zmk(a.x * s, a.y * s); }
BIEM_HD zc zinv(zc a) {
  if (fabs(a.x) >= fabs(a.y)) { double r = a.y / a.x, den = a.x + a.y * r; return zmk(1.0 / den, -r / den); }
  double r = a.x / a.y, den = a.x * r + a.y;
  return zmk(r / den, -1.0 / den);
}
BIEM_HD zc zdiv(zc a, zc b) { return zmul(a, zinv(b)); }
BIEM_HD double zabs1(zc a) { return fabs(a.x) + fabs(a.y); }
BIEM_HD zc zexp(zc a) { double e = exp(a.x); return zmk(e * cos(a.y), e * sin(a.y)); }
BIEM_HD zc zlog(zc a) { return zmk(log(hypot(a.x, a.y)), atan2(a.y, a.x)); }
BIEM_HD zc zsqrt(zc a) {
  double m = hypot(a.x, a.y);
  if (m == 0.0) return zmk(0.0, 0.0);
  double re = sqrt(0.5 * (m + fabs(a.x)));
  double im = 0.5 * a.y / re;
  return a.x >= 0.0 ? zmk(re, im) : zmk(fabs(im), a.y >= 0.0 ? re : -re);
}
// sin z, cos z without overflow for the sizes met here (|Im z| < 700)
BIEM_HD void zsincos(zc z, zc* s, zc* c) {
  double sh = sinh(z.y), ch = cosh(z.y), sn = sin(z.x), cs = cos(z.x);
  *s = zmk(sn * ch, cs * sh);
  *c = zmk(cs * ch, -sn * sh);
}

BIEM_HD int miller_start(int nmax, double az) {
  int M = (int)az;
  if (M < nmax + 1) M = nmax + 1;
  return M + 32 + (int)sqrt(48.0 * (double)M);
}

// spherical j_0..j_nmax, h_0..h_nmax at complex z != 0
BIEM_HD void bessel_jh_sph_c(int nmax, zc z, zc* J, zc* H) {
  const int M = miller_start(nmax, hypot(z.x, z.y));
  const double big = 1e200, small = 1e-200;
  zc jp1 = zmk(0.0, 0.0), jc = zmk(1e-250, 0.0);
  const zc iz = zinv(z);
  for (int k = M; k >= 1; --k) {
    zc jm1 = zsub(zscl(zmul(iz, jc), (double)(2 * k + 1)), jp1);
    if (k <= nmax) J[k] = jc;
    jp1 = jc;
    jc = jm1;
    if (zabs1(jc) > big) {
      jc = zscl(jc, small); jp1 = zscl(jp1, small);
      for (int q = (k <= nmax ? k : nmax + 1); q <= nmax; ++q) J[q] = zscl(J[q], small);
    }
  }
  J[0] = jc;
  zc sn, cs;
  zsincos(z, &sn, &cs);
  zc j0 = zmul(sn, iz), j1 = zmul(zsub(zmul(sn, iz), cs), iz);
  zc scale = (zabs1(j0) >= zabs1(j1)) ? zdiv(j0, jc) : zdiv(j1, jp1);
  for (int q = 0; q <= nmax; ++q) J[q] = zmul(J[q], scale);
  if (H == nullptr) return;                     // regular functions only
  // h_0 = -i e^{iz} / z,  h_1 = -(z + i) e^{iz} / z^2
  zc e = zexp(zmk(-z.y, z.x));
  zc h0 = zmul(zmk(e.y, -e.x), iz);                         // -i e
  zc h1 = zmul(zmul(zmk(-(z.x), -(z.y + 1.0)), e), zmul(iz, iz));
  H[0] = h0;
  if (nmax >= 1) H[1] = h1;
  for (int n = 1; n < nmax; ++n) {
    zc h2 = zsub(zscl(zmul(iz, h1), (double)(2 * n + 1)), h0);
    H[n + 1] = h2;
    h0 = h1; h1 = h2;
  }
}

// K_0(w), K_1(w) for complex w, Re w >= 0, w != 0
BIEM_HD void bessel_k01_c(zc w, zc* k0, zc* k1) {
  if (hypot(w.x, w.y) <= 2.0) {
    const zc t = zscl(zmul(w, w), 0.25);                    // w^2 / 4
    const zc lg = zadd(zlog(zscl(w, 0.5)), zmk(kEulerGamma, 0.0));
    zc term0 = zmk(1.0, 0.0), term1 = zmk(1.0, 0.0);       // t^k/(k!)^2, t^k/(k!(k+1)!)
    zc i0 = term0, s0 = zmk(0.0, 0.0);                      // I_0, sum t^k/(k!)^2 H_k
    zc i1 = term1, s1 = zscl(term1, 1.0);                   // sum t^k/(k!(k+1)!), sum ... (H_k + H_{k+1}),  k = 0: H_0 + H_1 = 1
    double hk = 0.0;
    for (int k = 1; k < 60; ++k) {
      term0 = zscl(zmul(term0, t), 1.0 / ((double)k * (double)k));
      term1 = zscl(zmul(term1, t), 1.0 / ((double)k * (double)(k + 1)));
      hk += 1.0 / (double)k;
      i0 = zadd(i0, term0);
      s0 = zadd(s0, zscl(term0, hk));
      i1 = zadd(i1, term1);
      s1 = zadd(s1, zscl(term1, 2.0 * hk + 1.0 / (double)(k + 1)));
      if (zabs1(term0) < 1e-18 * zabs1(i0)) break;
    }
    *k0 = zsub(s0, zmul(lg, i0));
    zc halfw = zscl(w, 0.5);
    // K_1 = 1/w + (ln(w/2) + gamma) I_1 - (w/4) s1,  I_1 = (w/2) i1
    *k1 = zsub(zadd(zinv(w), zmul(lg, zmul(halfw, i1))), zmul(zscl(w, 0.25), s1));
    return;
  }
  // Steed's algorithm for the second continued fraction (order 0), complex arithmetic
  zc b = zscl(zadd(w, zmk(1.0, 0.0)), 2.0);
  zc d = zinv(b), h = d, delh = d;
  zc q1 = zmk(0.0, 0.0), q2 = zmk(1.0, 0.0);
  const double a1 = 0.25;
  zc q = zmk(a1, 0.0), c = zmk(a1, 0.0);
  double a = -a1;
  zc s = zadd(zmk(1.0, 0.0), zmul(q, delh));
  for (int i = 2; i < 20000; ++i) {
    a -= 2.0 * (double)(i - 1);
    c = zscl(c, -a / (double)i);
    zc qnew = zscl(zsub(q1, zmul(b, q2)), 1.0 / a);
    q1 = q2; q2 = qnew;
    q = zadd(q, zmul(c, qnew));
    b = zadd(b, zmk(2.0, 0.0));
    d = zinv(zadd(b, zscl(d, a)));
    delh = zmul(zsub(zmul(b, d), zmk(1.0, 0.0)), delh);
    h = zadd(h, delh);
    zc dels = zmul(q, delh);
    s = zadd(s, dels);
    if (zabs1(dels) < 1e-17 * zabs1(s)) break;
  }
  h = zscl(h, a1);
  zc pre = zmul(zsqrt(zscl(zinv(w), 0.5 * kPi)), zexp(zmk(-w.x, -w.y)));
  *k0 = zdiv(pre, s);
  *k1 = zmul(zmul(*k0, zsub(zadd(w, zmk(0.5, 0.0)), h)), zinv(w));
}

// integer order J_0..J_nmax, H^(1)_0..H^(1)_nmax at complex z != 0
BIEM_HD void bessel_jh_int_c(int nmax, zc z, zc* J, zc* H) {
  const int M = miller_start(nmax, hypot(z.x, z.y)) | 1;
  const double big = 1e200, small = 1e-200;
  zc jp1 = zmk(0.0, 0.0), jc = zmk(1e-250, 0.0);
  const zc tz = zscl(zinv(z), 2.0);
  const bool up = z.y >= 0.0;                              // normalise with e^{-iz} (Im z >= 0) or e^{+iz}
  zc sum = zmk(0.0, 0.0);                                  // sum_{k>=1} (-+i)^k J_k
  for (int k = M; k >= 1; --k) {
    zc jm1 = zsub(zscl(zmul(tz, jc), (double)k), jp1);
    // (-i)^k or (+i)^k times jc
    int q4 = k & 3;
    zc ph = q4 == 0 ? jc : q4 == 1 ? (up ? zmk(jc.y, -jc.x) : zmk(-jc.y, jc.x)) : q4 == 2 ? zmk(-jc.x, -jc.y)
                                                                                          : (up ? zmk(-jc.y, jc.x) : zmk(jc.y, -jc.x));
    sum = zadd(sum, ph);
    if (k <= nmax) J[k] = jc;
    jp1 = jc;
    jc = jm1;
    if (zabs1(jc) > big) {
      jc = zscl(jc, small); jp1 = zscl(jp1, small); sum = zscl(sum, small);
      for (int q = (k <= nmax ? k : nmax + 1); q <= nmax; ++q) J[q] = zscl(J[q], small);
    }
  }
  J[0] = jc;
  zc norm = zadd(jc, zscl(sum, 2.0));
  zc target = up ? zexp(zmk(z.y, -z.x)) : zexp(zmk(-z.y, z.x));   // e^{-iz} or e^{+iz}
  zc scale = zdiv(target, norm);
  for (int q = 0; q <= nmax; ++q) J[q] = zmul(J[q], scale);
  if (H == nullptr) return;                     // regular functions only
  // H_0, H_1 at zz = z (Im z >= 0) or conj z
  zc zz = up ? z : zmk(z.x, -z.y);
  zc w = zmk(zz.y, -zz.x);                                 // -i zz
  zc k0, k1;
  bessel_k01_c(w, &k0, &k1);
  zc h0 = zscl(zmk(k0.y, -k0.x), 2.0 / kPi);               // (2/(pi i)) K_0 = -(2 i/pi) K_0
  zc h1 = zscl(k1, -2.0 / kPi);
  const zc tzz = up ? tz : zmk(tz.x, -tz.y);
  H[0] = h0;
  if (nmax >= 1) H[1] = h1;
  for (int n = 1; n < nmax; ++n) {
    zc h2 = zsub(zscl(zmul(tzz, h1), (double)n), h0);
    H[n + 1] = h2;
    h0 = h1; h1 = h2;
  }
  if (!up) {                                               // H1(z) = 2 J(z) - conj(H1(conj z))
    for (int n = 0; n <= nmax; ++n) H[n] = zsub(zscl(J[n], 2.0), zmk(H[n].x, -H[n].y));
  }
}

// d-dimensional regular and outgoing radial functions at a complex argument; real arguments (Im z == 0) take the real
// routines above (bit-identical to the real-k path).  J, H: nmax + 1 entries (d = 4: nmax + 2 slots each).  H == nullptr: the
// regular functions only (the outgoing ones are not computed).
BIEM_HD void radial_jh(int d, int nmax, zc z, zc* J, zc* H) {
  if (z.y == 0.0) {
    double* Jd = (double*)J;
    double* Yd = (double*)H;
    radial_d(d, nmax, z.x, Jd, Yd);        // reals in the first nmax + 1 (+1) doubles of each buffer
    for (int n = nmax; n >= 0; --n) { double j = Jd[n]; if (H) H[n] = zmk(j, Yd[n]); J[n] = zmk(j, 0.0); }
    return;
  }
  if (d == 3) {
    bessel_jh_sph_c(nmax, z, J, H);
  } else if (d == 2) {
    bessel_jh_int_c(nmax, z, J, H);
    for (int n = 0; n <= nmax; ++n) { J[n] = zscl(J[n], kSqrtHalfPi); if (H) H[n] = zscl(H[n], kSqrtHalfPi); }
  } else {
    bessel_jh_int_c(nmax + 1, z, J, H);
    zc f = zscl(zinv(z), kSqrtHalfPi);
    for (int n = 0; n <= nmax; ++n) { J[n] = zmul(J[n + 1], f); if (H) H[n] = zmul(H[n + 1], f); }
  }
}

// ---------------------------------------------------------------------------------------------
// Orthonormal building blocks (no Condon-Shortley phase; positive leading coefficients)
// ---------------------------------------------------------------------------------------------
// Pbar_n^m(x), int_{-1}^{1} Pbar^2 = 1; s = sqrt(1 - x^2) supplied by the caller.
BIEM_HD double pbar_single(int n, int m, double x, double s) {
  double pmm = 0.70710678118654752440;
  for (int i = 1; i <= m; ++i) pmm *= sqrt((double)(2 * i + 1) / (double)(2 * i)) * s;
  if (n == m) return pmm;
  double p1 = sqrt((double)(2 * m + 3)) * x * pmm;
  double p0 = pmm;
  for (int q = m + 2; q <= n; ++q) {
    double a = sqrt((double)(4 * q * q - 1) / (double)(q * q - m * m));
    double b = sqrt((double)((q - 1) * (q - 1) - m * m) / (double)(4 * (q - 1) * (q - 1) - 1));
    double p2 = a * (x * p1 - b * p0);
    p0 = p1; p1 = p2;
  }
  return p1;
}

// orthonormal Gegenbauer p_k^{(lam)}(x), lam = l + 1 (integer l >= 0), weight (1-x^2)^{lam-1/2}
BIEM_HD double gbar_single(int k, int l, double x) {
  double h0 = 0.5 * kPi;  // lam = 1
  for (int i = 1; i <= l; ++i) h0 *= ((double)i + 0.5) / ((double)i + 1.0);
  double lam = (double)(l + 1);
  double p0 = 1.0 / sqrt(h0);
  if (k == 0) return p0;
  // x p_{q-1} = a_q p_q + a_{q-1} p_{q-2},  a_q = 0.5 sqrt(q (q + 2 lam - 1) / ((q + lam - 1)(q + lam)))
  double a1 = 0.5 * sqrt((2.0 * lam) / (lam * (1.0 + lam)));
  double p1 = x * p0 / a1;
  double aprev = a1;
  for (int q = 2; q <= k; ++q) {
    double aq = 0.5 * sqrt((double)q * ((double)q + 2.0 * lam - 1.0) / (((double)q + lam - 1.0) * ((double)q + lam)));
    double p2 = (x * p1 - aprev * p0) / aq;
    p0 = p1; p1 = p2; aprev = aq;
  }
  return p1;
}

// Angles of a direction for the harmonics of one tree (computed once per direction).
struct Dir {
  double c0, s0;     // root polar angle (ba, bba, caa)
  double c1, s1;     // second polar angle (bba)
  double phi;        // azimuth (caa: azimuth of (x0, x1))
  double phi2;       // caa: azimuth of (x2, x3)
};

// caa polar factor  cos^a sin^b Pbar_k^{(b,a)}(cos 2 theta),  k = (n-a-b)/2,  int_0^{pi/2} ()^2 sin cos dtheta = 1
// (Hopf-type coordinates x0 = r cos t0 cos t1, x1 = r cos t0 sin t1, x2 = r sin t0 cos t2, x3 = r sin t0 sin t2; caa.svg)
BIEM_HD double cbar_single(int n, int a, int b, double c, double s) {
  const int k = (n - a - b) / 2;
  const double x = c * c - s * s;
  const double al = (double)b, be = (double)a;       // Jacobi P_k^{(alpha = b, beta = a)}
  double p0 = 1.0, p1 = 1.0;
  if (k >= 1) {
    p1 = (al + 1.0) + 0.5 * (al + be + 2.0) * (x - 1.0);
    for (int m = 1; m < k; ++m) {
      double t = 2.0 * m + al + be;
      double p2 = ((t + 1.0) * ((t + 2.0) * t * x + al * al - be * be) * p1 - 2.0 * (m + al) * (m + be) * (t + 2.0) * p0) /
                  (2.0 * (m + 1.0) * (m + al + be + 1.0) * t);
      p0 = p1; p1 = p2;
    }
  }
  // norm^2 = 2 (2k+a+b+1) k! (k+a+b)! / ((k+a)! (k+b)!) = 2 (2k+a+b+1) prod_{i=1..a} (k+b+i)/(k+i)
  double r = 2.0 * (2.0 * k + a + b + 1.0);
  for (int i = 1; i <= a; ++i) r *= (double)(k + b + i) / (double)(k + i);
  double f = sqrt(r);
  for (int i = 0; i < a; ++i) f *= c;
  for (int i = 0; i < b; ++i) f *= s;
  return f * p1;
}

BIEM_HD Dir make_dir(int tree, const double* u) {
  Dir d;
  d.c0 = 1.0; d.s0 = 0.0; d.c1 = 1.0; d.s1 = 0.0; d.phi = 0.0; d.phi2 = 0.0;
  if (tree == TREE_CAA) {
    double r01 = sqrt(u[0] * u[0] + u[1] * u[1]), r23 = sqrt(u[2] * u[2] + u[3] * u[3]);
    double r = sqrt(r01 * r01 + r23 * r23);
    if (r > 0.0) { d.c0 = r01 / r; d.s0 = r23 / r; }
    d.phi = atan2(u[1], u[0]);
    d.phi2 = atan2(u[3], u[2]);
  } else if (tree == TREE_A) {
    d.phi = atan2(u[1], u[0]);
  } else if (tree == TREE_BA) {
    double rho = sqrt(u[1] * u[1] + u[2] * u[2]);
    double r = sqrt(u[0] * u[0] + rho * rho);
    if (r > 0.0) { d.c0 = u[0] / r; d.s0 = rho / r; }
    d.phi = atan2(u[2], u[1]);
  } else {
    double rho2 = sqrt(u[2] * u[2] + u[3] * u[3]);
    double rho1 = sqrt(u[1] * u[1] + rho2 * rho2);
    double r = sqrt(u[0] * u[0] + rho1 * rho1);
    if (r > 0.0) { d.c0 = u[0] / r; d.s0 = rho1 / r; }
    if (rho1 > 0.0) { d.c1 = u[1] / rho1; d.s1 = rho2 / rho1; }
    d.phi = atan2(u[3], u[2]);
  }
  return d;
}

// One harmonic Y_label(direction); label = (m,-,-) | (n,m,-) | (n,l,m) | caa: (n,m1,m2).
BIEM_HD void harmonic_single(int tree, int a, int b, int c, const Dir& d, double* re, double* im) {
  double amp;
  int m;
  if (tree == TREE_CAA) {
    amp = cbar_single(a, b < 0 ? -b : b, c < 0 ? -c : c, d.c0, d.s0) * (0.5 / kPi);
    double ang = (double)b * d.phi + (double)c * d.phi2;
    *re = amp * cos(ang);
    *im = amp * sin(ang);
    return;
  }
  if (tree == TREE_A) {
    m = a;
    amp = kInvSqrt2Pi;
  } else if (tree == TREE_BA) {
    m = b;
    amp = pbar_single(a, m < 0 ? -m : m, d.c0, d.s0) * kInvSqrt2Pi;
  } else {
    m = c;
    double sl = 1.0;
    for (int i = 0; i < b; ++i) sl *= d.s0;
    amp = sl * gbar_single(a - b, b, d.c0) * pbar_single(b, m < 0 ? -m : m, d.c1, d.s1) * kInvSqrt2Pi;
  }
  double ang = (double)m * d.phi;
  *re = amp * cos(ang);
  *im = amp * sin(ang);
}

}  // namespace biem
