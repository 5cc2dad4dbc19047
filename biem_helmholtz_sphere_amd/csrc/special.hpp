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
    for (int n = 0; n <= nmax; ++n) { J[n] *= kSqrtHalfPi; Y[n] *= kSqrtHalfPi; }
  } else {  // d == 4: sqrt(pi/2) Z_{n+1}(x) / x
    bessel_jy_int(nmax + 1, x, J, Y);
    double f = kSqrtHalfPi / x;
    for (int n = 0; n <= nmax; ++n) { J[n] = J[n + 1] * f; Y[n] = Y[n + 1] * f; }
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
