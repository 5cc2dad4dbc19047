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
                                                    const cplx* __restrict__ dens, cplx* __restrict__ c) {
  __shared__ cplx sJ[kMaxRadU + 3], sH[kMaxRadU + 3];
  __shared__ cplx sB[kMaxRadU];
  int b = blockIdx.x, s = blockIdx.y;
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
      out[((size_t)p * nb + s) * B + b] = bad ? make_double2(qnan, qnan) : sBall[b];
  } else if (threadIdx.x == 0) {
    double ar = 0.0, ai = 0.0;
    for (int b = 0; b < B; ++b) { ar += sBall[b].x; ai += sBall[b].y; }
    out[(size_t)p * nb + s] = bad ? make_double2(qnan, qnan) : make_double2(ar, ai);
  }
}

int launch_uscat(const biem_plan* p, int nb, int B, int P, const double* d_k, const double* d_eta, const double* d_centers,
                 const double* d_radii, int geom_batched, const double* d_density, const double* d_points, int flags,
                 double* d_out, void* d_work, size_t work_bytes, hipStream_t st) {
  if (nb <= 0 || B <= 0 || P <= 0) return BIEM_OK;
  if (p->n_end > kMaxRadU) { set_error("biem_uscat: n_end too large"); return BIEM_ERR_UNSUPPORTED; }
  size_t need = (size_t)nb * B * p->H * sizeof(cplx);
  if (work_bytes < need) { set_error("biem_uscat: workspace too small"); return BIEM_ERR_ARG; }
  cplx* c = (cplx*)d_work;
  hipLaunchKernelGGL(k_uscat_coef, dim3(B, nb), dim3(64), 0, st, p->d, p->H, p->n_end, p->d_deg, B,
                     (flags & BIEM_USCAT_KIND_INNER) && !(flags & BIEM_USCAT_FAR_FIELD) ? 1 : 0, (const cplx*)d_k, d_eta, d_radii,
                     geom_batched, (const cplx*)d_density, c);
  BIEM_LAUNCHCHK();
  hipLaunchKernelGGL(k_uscat, dim3(P, nb), dim3(256), (size_t)B * sizeof(cplx), st, p->tree, p->d, p->H, p->n_end, p->d_labels,
                     p->d_deg, nb, B, P, (const cplx*)d_k, d_centers, d_radii, geom_batched, (const cplx*)c, d_points, flags, (cplx*)d_out);
  BIEM_LAUNCHCHK();
  return BIEM_OK;
}

}  // namespace biem
