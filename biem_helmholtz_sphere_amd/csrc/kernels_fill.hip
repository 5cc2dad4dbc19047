// kernels_fill.hip -- special-function tables, matrix fill, RHS projection, density scaling (gfx950).
//
// Replaces the span _biem.py:627-639 (RHS) and _biem.py:694-792 (matrix) of the reference, which there is
// ~40 array-API ops and 3-5 full-size temporaries; here every matrix element is written exactly once.
#include "common.hpp"
#include <cstdlib>

namespace biem {

constexpr int kMaxRad = 320;   // max table order handled per thread-local/LDS radial array

// ---------------------------------------------------------------------------------------------
// test entry: radial functions at arbitrary arguments
// ---------------------------------------------------------------------------------------------
__global__ void k_radial(int d, int nmax, int count, const double* __restrict__ x, double* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double* J = out + (size_t)i * 2 * (nmax + 1);
  double* Y = J + (nmax + 1);
  if (d == 4) {
    // needs one more order: compute integer orders 0..nmax+1 in place with a shifted write
    double xi = x[i];
    // order nmax+1 does not fit the output slot: run the recurrence with the last slot as scratch
    // (output layout leaves no spare element) -> compute J/Y of integer order via local arrays
    double lj[kMaxRad + 2], ly[kMaxRad + 2];
    bessel_jy_int(nmax + 1, xi, lj, ly);
    double f = kSqrtHalfPi / xi;
    for (int n = 0; n <= nmax; ++n) { J[n] = lj[n + 1] * f; Y[n] = ly[n + 1] * f; }
  } else {
    radial_d(d, nmax, x[i], J, Y);
  }
}

// complex arguments: out[i][0][n] = z_n (regular), out[i][1][n] = h_n (outgoing), complex128
__global__ void k_radial_c(int d, int nmax, int count, const cplx* __restrict__ z, cplx* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  cplx lj[kMaxRad + 3], lh[kMaxRad + 3];
  radial_jh(d, nmax, z[i], lj, lh);
  cplx* J = out + (size_t)i * 2 * (nmax + 1);
  cplx* Hh = J + (nmax + 1);
  for (int n = 0; n <= nmax; ++n) { J[n] = lj[n]; Hh[n] = lh[n]; }
}

int launch_radial_c(int d, int nmax, int count, const double* d_z, double* d_out, hipStream_t st) {
  if (nmax < 0 || nmax > kMaxRad || (d != 2 && d != 3 && d != 4)) { set_error("biem_radial_complex: bad d/nmax"); return BIEM_ERR_ARG; }
  if (count <= 0) return BIEM_OK;
  hipLaunchKernelGGL(k_radial_c, dim3((count + 63) / 64), dim3(64), 0, st, d, nmax, count, (const cplx*)d_z, (cplx*)d_out);
  BIEM_LAUNCHCHK();
  return BIEM_OK;
}

int launch_radial(int d, int nmax, int count, const double* d_x, double* d_out, hipStream_t st) {
  if (nmax < 0 || nmax > kMaxRad || (d != 2 && d != 3 && d != 4)) { set_error("biem_radial: bad d/nmax"); return BIEM_ERR_ARG; }
  if (count <= 0) return BIEM_OK;
  hipLaunchKernelGGL(k_radial, dim3((count + 63) / 64), dim3(64), 0, st, d, nmax, count, d_x, d_out);
  BIEM_LAUNCHCHK();
  return BIEM_OK;
}

// ---------------------------------------------------------------------------------------------
// test entry + building block: all harmonics of degree < n_end at given directions
// ---------------------------------------------------------------------------------------------
__global__ void k_harmonics(int tree, int d, int H, const int* __restrict__ labels, int count,
                            const double* __restrict__ u, cplx* __restrict__ Y) {
  int p = blockIdx.x;
  if (p >= count) return;
  double v[4];
  for (int i = 0; i < d; ++i) v[i] = u[(size_t)p * d + i];
  Dir dir = make_dir(tree, v);
  for (int h = threadIdx.x; h < H; h += blockDim.x) {
    double re, im;
    harmonic_single(tree, labels[3 * h], labels[3 * h + 1], labels[3 * h + 2], dir, &re, &im);
    Y[(size_t)p * H + h] = make_double2(re, im);
  }
}

int launch_harmonics(const biem_plan* p, int count, const double* d_u, double* d_Y, hipStream_t st) {
  if (count <= 0) return BIEM_OK;
  hipLaunchKernelGGL(k_harmonics, dim3(count), dim3(128), 0, st, p->tree, p->d, p->H, p->d_labels, count, d_u, (cplx*)d_Y);
  BIEM_LAUNCHCHK();
  return BIEM_OK;
}

// ---------------------------------------------------------------------------------------------
// K0a: per-ball tables  gj = alpha j + beta k j',  gh = alpha h + beta k h',  blc = dlc - i eta slc
//   (ush.harmonics_regular_singular_component x4 + potential_coef S/D, _biem.py:723-789)
// one thread per (system, ball); outputs tab[s][b][3][n_end] complex.
// ---------------------------------------------------------------------------------------------
__global__ void k_ball_tables(int d, int n_end, int nb, int B, const cplx* __restrict__ k, const double* __restrict__ eta,
                              const double* __restrict__ radii, int geom_batched, const cplx* __restrict__ alpha,
                              const cplx* __restrict__ beta, int ab_batched, cplx* __restrict__ tab, cplx* __restrict__ scratch) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb * B) return;
  int s = i / B, b = i % B;
  const cplx kk = k[s];
  const double et = eta[s];
  double rho = radii[(geom_batched ? (size_t)s * B : 0) + b];
  cplx al = alpha[(ab_batched ? (size_t)s * B : 0) + b];
  cplx be = beta[(ab_batched ? (size_t)s * B : 0) + b];
  // orders up to kMaxRad: thread-local arrays; beyond (2-D only): 2 (n_end + 3) complex of global scratch per (system, ball)
  cplx Jl[kMaxRad + 3], Hl[kMaxRad + 3];
  cplx* J = scratch ? scratch + (size_t)i * 2 * (n_end + 3) : Jl;
  cplx* Hh = scratch ? J + (n_end + 3) : Hl;
  const cplx x = cscale(kk, rho);
  radial_jh(d, n_end, x, J, Hh);   // orders 0..n_end (one extra for the derivative); Im k = 0 takes the real routines
  const cplx ix = crecip(x);
  cplx* out = tab + (size_t)i * 3 * n_end;
  double rp = 1.0;               // rho^{d-1}
  for (int q = 0; q < d - 1; ++q) rp *= rho;
  cplx kd2 = make_double2(1.0, 0.0);   // k^{d-2}
  for (int q = 0; q < d - 2; ++q) kd2 = cmul(kd2, kk);
  for (int n = 0; n < n_end; ++n) {
    const cplx j = J[n], h = Hh[n];
    const cplx jp = csub(cscale(cmul(ix, j), (double)n), J[n + 1]);
    const cplx hp = csub(cscale(cmul(ix, h), (double)n), Hh[n + 1]);
    const cplx kjp = cmul(kk, jp), khp = cmul(kk, hp);
    // gj = alpha j + beta k j',  gh = alpha h + beta k h'
    cplx gj = cadd(cmul(al, j), cmul(be, kjp));
    cplx gh = cadd(cmul(al, h), cmul(be, khp));
    // blc = i k^{d-1} rho^{d-1} j' - i eta * i k^{d-2} rho^{d-1} j = k^{d-2} rho^{d-1} (eta j + i k j')
    cplx blc = cscale(cmul(kd2, make_double2(et * j.x - kjp.y, et * j.y + kjp.x)), rp);
    out[n] = gj;
    out[n_end + n] = gh;
    out[2 * n_end + n] = blc;
  }
}

int launch_ball_tables(const biem_plan* p, int nb, int B, const double* d_k, const double* d_eta, const double* d_radii,
                       int geom_batched, const double* d_alpha, const double* d_beta, int ab_batched, double* d_tab, hipStream_t st) {
  if (p->n_end > kMaxRad && p->tree != TREE_A) { set_error("n_end=%d exceeds the built table size %d", p->n_end, kMaxRad); return BIEM_ERR_UNSUPPORTED; }
  int total = nb * B;
  if (total <= 0) return BIEM_OK;
  ProfScope ps(PK_TABLES, st);
  cplx* scratch = nullptr;
  if (p->n_end > kMaxRad) BIEM_HIPCHK(hipMallocAsync((void**)&scratch, (size_t)total * 2 * (p->n_end + 3) * sizeof(cplx), st));   // (2-D, large orders: stream-ordered)
  hipLaunchKernelGGL(k_ball_tables, dim3((total + 63) / 64), dim3(64), 0, st, p->d, p->n_end, nb, B, (const cplx*)d_k, d_eta, d_radii,
                     geom_batched, (const cplx*)d_alpha, (const cplx*)d_beta, ab_batched, (cplx*)d_tab, scratch);
  BIEM_LAUNCHCHK();
  if (scratch) BIEM_HIPCHK(hipFreeAsync(scratch, st));
  return BIEM_OK;
}

// ---------------------------------------------------------------------------------------------
// K0b: per-pair translation tables  T[s][b][b'][l] = C_d h_{n''}(k |t|) Y_l(t^),  t = c_b - c_b'
//   (argument order of _biem.py:694-699), l over labels of degree < 2 n_end - 1.
// One wave per (system, ordered pair): lane 0 runs the radial recurrences into LDS, then all lanes
// evaluate harmonics.  Diagonal pairs are skipped (the reference evaluates them at t = 0 and masks
// the inf/nan afterwards, _biem.py:745-746).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_pair_tables(int tree, int d, int n2, int H2, double Cd, const int* __restrict__ labels2,
                                                     const int* __restrict__ deg2, int B, const cplx* __restrict__ k,
                                                     const double* __restrict__ centers, int geom_batched, cplx* __restrict__ T,
                                                     int lower, int nbp, const int* __restrict__ lin2, int H2lin,
                                                     const int* __restrict__ red_of = nullptr, const int* __restrict__ red_first = nullptr,
                                                     const int* __restrict__ ph_mu = nullptr, int E = 0, int NP = 0,
                                                     const cplx* __restrict__ tab = nullptr, int n_end = 0, cplx* __restrict__ rad_scratch = nullptr,
                                                     const int* __restrict__ rep_flag = nullptr) {
  __shared__ cplx sJl[kMaxRad * 2 + 6];
  __shared__ cplx sHl[kMaxRad * 2 + 6];
  int pair = blockIdx.x, s = blockIdx.y;
  // radial functions of the pair: LDS up to order 2 kMaxRad; beyond (2-D only) 2 (n2 + 6) complex of global scratch per block (written by
  // thread 0, read by the block after the barrier: same CU, fresh addresses)
  cplx* sJ = rad_scratch ? rad_scratch + ((size_t)s * gridDim.x + pair) * 2 * (n2 + 6) : sJl;
  cplx* sH = rad_scratch ? sJ + (n2 + 6) : sHl;
  int b = pair / B, bp = pair % B;
  // both fills contract the blocks b < bp (the general fill derives block (bp, b) from (b, bp) by the parity sign, the symmetric
  // fill writes the upper triangle only); lower = 1 (tables of the pairs b > bp instead) is kept for tests
  if (lower ? b <= bp : b >= bp) return;
  if (rep_flag != nullptr && !rep_flag[b * B + bp]) return;      // pair classes: only a class's first pair is read by the fill
  const double* cb = centers + ((geom_batched ? (size_t)s * B : 0) + b) * d;
  const double* cp = centers + ((geom_batched ? (size_t)s * B : 0) + bp) * d;
  double t[4];
  double r2 = 0.0;
  for (int i = 0; i < d; ++i) { t[i] = cb[i] - cp[i]; r2 += t[i] * t[i]; }
  double r = sqrt(r2);
  if (threadIdx.x == 0) radial_jh(d, n2 - 1, cscale(k[s], r), sJ, sH);
  __syncthreads();
  Dir dir = make_dir(tree, t);
  if (red_of != nullptr) {
    // reduced table of the entry-per-lane symmetric fill (plan.hpp): T'[e] = C_d h_{n''} x (real angular factor of the unit's first
    // member: the harmonic at zero azimuth), then the NP phases e^{i mu . phi}
    // ... and the per-degree factors q = gj / sqrt(gj gh) of the row ball b and of the column ball bp (the kernel multiplies the block by
    // q_b[n] q_bp[n']): a table row is everything a (pair, system) combination of the fill needs, E + NP + 2 n_end complex numbers
    cplx* o = T + ((size_t)s * B * B + pair) * (size_t)(E + NP + 2 * n_end);
    for (int i = threadIdx.x; i < 2 * n_end; i += 64) {
      const int which = i >= n_end, n = i - which * n_end;
      const cplx* tb = tab + ((size_t)s * B + (which ? bp : b)) * 3 * n_end;
      o[E + NP + i] = cmul(tb[n], crecip(zsqrt(cmul(tb[n], tb[n_end + n]))));
    }
    if (tree == TREE_BA) {
      for (int m = threadIdx.x; m < n2; m += 64) {          // one lane per |mu|: the degree recurrence of Pbar_n^m once
        double pmm = 0.70710678118654752440;
        for (int i = 1; i <= m; ++i) pmm *= sqrt((double)(2 * i + 1) / (double)(2 * i)) * dir.s0;
        double p0 = 0.0, p1 = pmm;
        for (int n = m; n < n2; ++n) {
          if (n > m) {
            double p2;
            if (n == m + 1) p2 = sqrt((double)(2 * m + 3)) * dir.c0 * pmm;
            else {
              const double a = sqrt((double)(4 * n * n - 1) / (double)(n * n - m * m));
              const double bq = sqrt((double)((n - 1) * (n - 1) - m * m) / (double)(4 * (n - 1) * (n - 1) - 1));
              p2 = a * (dir.c0 * p1 - bq * p0);
            }
            p0 = p1; p1 = p2;
          }
          o[red_of[n * n + n - m]] = cscale(sH[n], Cd * p1 * kInvSqrt2Pi);       // label (n, -m) is the first member of its unit
        }
      }
    } else {
      Dir d0 = dir; d0.phi = 0.0; d0.phi2 = 0.0;
      for (int l = threadIdx.x; l < H2; l += 64) {
        if (!red_first[l]) continue;
        double re, im;
        harmonic_single(tree, labels2[3 * l], labels2[3 * l + 1], labels2[3 * l + 2], d0, &re, &im);
        o[red_of[l]] = cscale(sH[deg2[l]], Cd * re);
      }
    }
    for (int i = threadIdx.x; i < NP; i += 64) {
      double sn, cs;
      sincos((double)ph_mu[2 * i] * dir.phi + (double)ph_mu[2 * i + 1] * dir.phi2, &sn, &cs);
      o[E + i] = make_double2(cs, sn);
    }
    return;
  }
  // nbp > 0: systems-in-lanes layout Tt[group of 64 systems][pair index of (b < bp)][l][system in group]: a group's table rows are
  // contiguous 1-KiB lines (with the systems of ALL groups in one row, the rows one group reads lie 4 KiB apart and every
  // workgroup of a fill - they all work on the same group at a time - hits the same quarter of the L2 channels)
  cplx* out = nbp > 0 ? T + (((size_t)(s >> 6) * (B * (B - 1) / 2) + (bp * (bp - 1) / 2 + b)) * H2) * 64 + (s & 63) : T + ((size_t)s * B * B + pair) * H2;
  const size_t ostride = nbp > 0 ? 64 : 1;
  auto emit = [&](int l, cplx val, bool self_conjugate) {
    if (lin2 == nullptr) { out[(size_t)l * ostride] = val; return; }
    // paired layout of the entry-per-lane symmetric fill: conjugate partners adjacent, a self-conjugate label in both places
    cplx* o2 = T + ((size_t)s * B * B + pair) * H2lin;
    const int i = lin2[l];
    o2[i] = val;
    if (self_conjugate) o2[i + 1] = val;
  };
  if (tree == TREE_BA) {
    // d = 3: one lane per order |m| runs the degree recurrence of Pbar_n^m ONCE (n = m .. n2-1) and writes both signs; evaluating
    // every label from scratch (generic branch below) repeats that recurrence, square roots and divisions included, per label:
    // 10 ms per 256 systems of cfg 3 against the fill's 47
    for (int m = threadIdx.x; m < n2; m += 64) {
      double sn, cs;
      sincos((double)m * dir.phi, &sn, &cs);
      double pmm = 0.70710678118654752440;
      for (int i = 1; i <= m; ++i) pmm *= sqrt((double)(2 * i + 1) / (double)(2 * i)) * dir.s0;
      double p0 = 0.0, p1 = pmm;
      for (int n = m; n < n2; ++n) {
        if (n > m) {
          double p2;
          if (n == m + 1) p2 = sqrt((double)(2 * m + 3)) * dir.c0 * pmm;
          else {
            const double a = sqrt((double)(4 * n * n - 1) / (double)(n * n - m * m));
            const double bq = sqrt((double)((n - 1) * (n - 1) - m * m) / (double)(4 * (n - 1) * (n - 1) - 1));
            p2 = a * (dir.c0 * p1 - bq * p0);
          }
          p0 = p1; p1 = p2;
        }
        const double amp = p1 * kInvSqrt2Pi;
        const cplx h = cscale(sH[n], Cd);
        const int l0 = n * n + n;
        emit(l0 + m, cmul(h, make_double2(amp * cs, amp * sn)), m == 0);
        if (m > 0) emit(l0 - m, cmul(h, make_double2(amp * cs, -amp * sn)), false);
      }
    }
    return;
  }
  for (int l = threadIdx.x; l < H2; l += 64) {
    double re, im;
    harmonic_single(tree, labels2[3 * l], labels2[3 * l + 1], labels2[3 * l + 2], dir, &re, &im);
    int n = deg2[l];
    const cplx val = cmul(cscale(sH[n], Cd), make_double2(re, im));
    emit(l, val, labels2[3 * l + (tree == TREE_A ? 0 : 2)] == 0 && (tree != TREE_CAA || labels2[3 * l + 1] == 0));
  }
}

// ---------------------------------------------------------------------------------------------
// K1-K3: generic fill.  For every ordered pair block (b, b') and every entry (h, h'):
//   off-diagonal:  A = R_b[n(h)] * Cc_{b'}[n(h')] * sum_p coef[p] T_{bb'}[tidx[p]]        ((S|R)^T, _biem.py:769)
//   diagonal    :  A = delta_{hh'} * Dg_b[n(h)]
// with (R, Cc, Dg) = (gj, blc, gh*blc) for the reference scaling (_biem.py:745-789) and (gj, 1/gh, 1) for the
// equilibrated system the LU factors.
// One 1024-thread workgroup per (entry chunk, ball b, system).  The chunk's slice of the term list (8-byte coefficient +
// 2-byte table index per term) and its row pointers are loaded into LDS ONCE and reused for all partner balls b'; per
// partner only its pair table T (H2 complex) and column factors are staged.  Every thread keeps two independent entries
// in flight (the per-term chain idx -> T -> fma is LDS-latency bound).  The first version (256 threads, term list from
// L2 for every pair) ran at 487 GB/s (profiles/r01_bench_cfg3_32sys.json).  Consecutive lanes own consecutive columns
// h': 16-byte coalesced stores, every matrix element written exactly once.
// ---------------------------------------------------------------------------------------------
constexpr int FILL_THREADS = 1024;

__global__ void __launch_bounds__(FILL_THREADS) k_fill(int H, int H2, int n_end, int B, const int* __restrict__ deg,
                                                        const int* __restrict__ chunk_ent, int chunk_terms_max, int chunk_ents_max,
                                                        const uint32_t* __restrict__ ptr, const double* __restrict__ coef,
                                                        const uint16_t* __restrict__ tidx, const cplx* __restrict__ T,
                                                        const cplx* __restrict__ tab, int scaling, cplx* __restrict__ A,
                                                        long long lda, long long sys_stride, int table_global) {
  extern __shared__ char smem[];
  // table_global: the pair table is read where it lies (global memory, served by L2) - orders whose table does not fit LDS; the
  // generic pointer sT then addresses global memory and nothing is staged
  cplx* const sTl = (cplx*)smem;                        // [H2] pair table T_{b,bp}
  cplx* sC = table_global ? sTl : sTl + H2;             // [H] column factors of the partner ball bp
  cplx* sC2 = sC + H;                                   // [H] column factors of the owner ball b (mirrored block)
  double* sCoef = (double*)(sC2 + H);                   // [chunk_terms_max]
  uint32_t* sPtr = (uint32_t*)(sCoef + chunk_terms_max);   // [chunk_ents_max + 1], relative to the chunk's first term
  uint16_t* sIdx = (uint16_t*)(sPtr + chunk_ents_max + 1);
  const int chunk = blockIdx.x, s = blockIdx.z, tid = threadIdx.x;
  const int e0 = chunk_ent[chunk], e1 = chunk_ent[chunk + 1], nent = e1 - e0;
  const uint32_t t0 = ptr[e0], t1 = ptr[e1];
  for (uint32_t q = t0 + tid; q < t1; q += FILL_THREADS) { sCoef[q - t0] = coef[q]; sIdx[q - t0] = tidx[q]; }
  for (int e = tid; e <= nent; e += FILL_THREADS) sPtr[e] = ptr[e0 + e] - t0;
  cplx* As = A + (size_t)s * sys_stride;
  auto colfac = [&](const cplx* tball, int hp) {        // column factor of a ball for harmonic hp
    const int n = deg[hp];
    return scaling == BIEM_FILL_REFERENCE ? tball[2 * n_end + n] : crecip(tball[n_end + n]);
  };
  // the raw sum of an entry is shared by the two blocks of an unordered pair: T_{bp,b}[l] = (-1)^{n''} T_{b,bp}[l] (harmonics of
  // degree n'' at -t) and every term of entry (h, h') has n'' = n + n' (mod 2), so S_{bp,b}[h,h'] = (-1)^{n+n'} S_{b,bp}[h,h'].
  // A workgroup owns balls b1 = blockIdx.y and b2 = B-1-b1 and contracts each with its partners bp > b: B-1 contractions per
  // workgroup whatever b1 is, half the contractions of the one-block-per-contraction form.
  const int b1 = blockIdx.y, b2 = B - 1 - b1;
  for (int own = 0; own < 2; ++own) {
    const int b = own == 0 ? b1 : b2;
    if (own == 1 && b2 == b1) break;
    const cplx* tb = tab + ((size_t)s * B + b) * 3 * n_end;
    {  // diagonal block of b
      cplx* Ab = As + ((size_t)b * H) * lda + (size_t)b * H;
      for (int e = tid; e < nent; e += FILL_THREADS) {
        int h = (e0 + e) / H, hp = (e0 + e) - h * H;
        cplx v = make_double2(0.0, 0.0);
        if (hp == h) {
          int n = deg[h];
          v = scaling == BIEM_FILL_REFERENCE ? cmul(tb[n_end + n], tb[2 * n_end + n]) : make_double2(1.0, 0.0);
        }
        Ab[(size_t)h * lda + hp] = v;
      }
    }
    if (b + 1 >= B) continue;
    __syncthreads();                                     // previous owner's sC2 no longer in use (also orders the chunk loads)
    for (int hp = tid; hp < H; hp += FILL_THREADS) sC2[hp] = colfac(tb, hp);
    for (int bp = b + 1; bp < B; ++bp) {
      const cplx* tbp = tab + ((size_t)s * B + bp) * 3 * n_end;
      const cplx* Tp = T + ((size_t)s * B * B + (size_t)b * B + bp) * H2;
      cplx* Ab = As + ((size_t)b * H) * lda + (size_t)bp * H;     // block (b, bp)
      cplx* Am = As + ((size_t)bp * H) * lda + (size_t)b * H;     // block (bp, b)
      __syncthreads();                                   // previous partner's table no longer in use
      const cplx* sT = table_global ? Tp : sTl;
      if (!table_global) for (int l = tid; l < H2; l += FILL_THREADS) sTl[l] = Tp[l];
      for (int hp = tid; hp < H; hp += FILL_THREADS) sC[hp] = colfac(tbp, hp);
      __syncthreads();
      auto put = [&](int e, double sr, double si) {
        const int h = (e0 + e) / H, hp = (e0 + e) - h * H;
        const int nh = deg[h];
        const cplx raw = make_double2(sr, si);
#ifdef BIEM_ABL_FILL_NOSTORE       // timing ablation: results computed, not stored
        const cplx v1 = cmul(cmul(raw, tb[nh]), sC[hp]);
        const cplx m = cmul(cmul(raw, tbp[nh]), sC2[hp]);
        asm volatile("" ::"v"(v1.x), "v"(v1.y), "v"(m.x), "v"(m.y));
        (void)Ab; (void)Am;
#else
        Ab[(size_t)h * lda + hp] = cmul(cmul(raw, tb[nh]), sC[hp]);
        const cplx m = cmul(cmul(raw, tbp[nh]), sC2[hp]);
        Am[(size_t)h * lda + hp] = ((nh + deg[hp]) & 1) ? make_double2(-m.x, -m.y) : m;
#endif
      };
      for (int e = tid; e < nent; e += 2 * FILL_THREADS) {
        const int eb = e + FILL_THREADS;
        const bool two = eb < nent;
        uint32_t p0 = sPtr[e], p1 = sPtr[e + 1];
        uint32_t q0 = two ? sPtr[eb] : 0, q1 = two ? sPtr[eb + 1] : 0;
        double ar = 0.0, ai = 0.0, br = 0.0, bi = 0.0;
        while (p0 < p1 && q0 < q1) {                     // two independent chains
          double c = sCoef[p0], d = sCoef[q0];
          cplx t = sT[sIdx[p0]], u = sT[sIdx[q0]];
          ar = fma(c, t.x, ar); ai = fma(c, t.y, ai);
          br = fma(d, u.x, br); bi = fma(d, u.y, bi);
          ++p0; ++q0;
        }
        for (; p0 < p1; ++p0) { double c = sCoef[p0]; cplx t = sT[sIdx[p0]]; ar = fma(c, t.x, ar); ai = fma(c, t.y, ai); }
        for (; q0 < q1; ++q0) { double d = sCoef[q0]; cplx u = sT[sIdx[q0]]; br = fma(d, u.x, br); bi = fma(d, u.y, bi); }
        put(e, ar, ai);
        if (two) put(eb, br, bi);
      }
    }
  }
}

__global__ void k_fill_pad(int N, int n_pad, cplx* __restrict__ A, long long lda, long long sys_stride) {
  int s = blockIdx.y;
  cplx* As = A + (size_t)s * sys_stride;
  int npadrows = n_pad - N;
  // region 1: rows [N, n_pad) x cols [0, n_pad);  region 2: rows [0, N) x cols [N, n_pad)
  long long total1 = (long long)npadrows * n_pad, total2 = (long long)N * npadrows;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total1 + total2; i += (long long)gridDim.x * blockDim.x) {
    int r, c;
    if (i < total1) { r = N + (int)(i / n_pad); c = (int)(i % n_pad); }
    else { long long q = i - total1; r = (int)(q / npadrows); c = N + (int)(q % npadrows); }
    As[(size_t)r * lda + c] = (r == c) ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0);
  }
}

// does the one-unit-pair-per-lane form of the symmetric fill fit this plan's tables into LDS?
static bool fill_sym_entry_fits(const biem_plan* p, size_t* shm_out) {
  const size_t shm = (size_t)(p->H2lin + 2 * p->n_end) * sizeof(cplx) + (size_t)(p->qchunk_terms_max + 1) * 10 + (size_t)(2 * p->qchunk_pairs_max + 1) * 4 + 16;
  if (shm_out) *shm_out = shm;
  return p->pair_lists_ok && (int)p->qchunk.size() > 1 && shm <= 160 * 1024 && p->H2lin <= 8 * 1024;
}

// the reduced-table form (k_fill_red): reduced pair table + phases, q factors and the chunk's transposed lists in LDS
static bool fill_red_fits(const biem_plan* p, size_t* shm_out) {
  const size_t shm = (size_t)p->red_nc * (p->E + p->NP + 2 * p->n_end) * sizeof(cplx) + (size_t)p->rchunk_rows_max * 64 * 10 + 33 * 4 + 64;
  if (shm_out) *shm_out = shm;
  return p->red_lists_ok && shm <= (size_t)(p->red_waves == 16 ? 158 : 79) * 1024 && p->E + p->NP + 2 * p->n_end <= 8 * 64 * p->red_waves;
}

// pair classes of the symmetric fill (k_pair_dedupe), behind the pair tables: nrep (+3 pad), rep_list[np], dup_ptr[np + 1], dup_bb[np]
// ... then rep_flag[B * B]: 1 where the pair (b, bp) is the first of its class - only those pairs' tables are read by the fill
static size_t fill_dedupe_bytes(int B) { const size_t np = (size_t)B * (B - 1) / 2; return ((3 * np + 5 + (size_t)B * B) * sizeof(int) + 15) / 16 * 16; }
constexpr int kDedupeMaxPairs = 2048;       // (the class search is quadratic in the pairs, in one workgroup)

size_t fill_workspace_bytes(const biem_plan* p, int nb, int B) {
  // pair tables of the general / entry forms: [nb][B][B][H2 or H2lin]; of the systems-in-lanes form (needed where the entry form
  // does not fit, or when BIEM_FILL_FORM=sys forces it): [nb rounded up to 64][pairs][H2] + q factors
  size_t row = (size_t)(p->H2lin > p->H2 ? p->H2lin : p->H2);            // general / gather forms; reduced form: T', phases, q factors
  if ((size_t)(p->E + p->NP + 2 * p->n_end) > row) row = (size_t)(p->E + p->NP + 2 * p->n_end);
  const size_t a = (size_t)nb * B * B * row, nbp = (size_t)(nb + 63) / 64 * 64;
  const size_t b = ((size_t)(B * (B - 1) / 2) * p->H2 + (size_t)B * p->n_end) * nbp;
  const char* form = getenv("BIEM_FILL_FORM");
  const bool need_sys = (form && form[0] == 's') || !(fill_sym_entry_fits(p, nullptr) || fill_red_fits(p, nullptr));
  return (need_sys && b > a ? b : a) * sizeof(cplx) + fill_dedupe_bytes(B);
}

__global__ void k_fill2d(int n_end, int H, int B, int npairs, const cplx* __restrict__ T, const cplx* __restrict__ tab, int scaling,
                         cplx* __restrict__ A, long long lda, long long sys_stride);        // (defined with the 2-D fills below)
// k_pair_tables in reduced mode (table rows T' | phases | q factors); 2-D orders whose radial arrays exceed the kernel's LDS take
// stream-ordered global scratch
static int launch_pair_tables_red(const biem_plan* p, int nb, int B, const double* d_k, const double* d_centers, int geom_batched,
                                  const double* d_tab, cplx* T, hipStream_t st, const int* rep_flag = nullptr) {
  cplx* scratch = nullptr;
  if (p->n2 + 6 > kMaxRad * 2 + 6) {
    if (p->tree != TREE_A) { set_error("n_end=%d exceeds the built table size", p->n_end); return BIEM_ERR_UNSUPPORTED; }
    BIEM_HIPCHK(hipMallocAsync((void**)&scratch, (size_t)nb * B * B * 2 * (p->n2 + 6) * sizeof(cplx), st));
  }
  hipLaunchKernelGGL(k_pair_tables, dim3(B * B, nb), dim3(64), 0, st, p->tree, p->d, p->n2, p->H2, p->Cd, p->d_labels2, p->d_deg2, B,
                     (const cplx*)d_k, d_centers, geom_batched, T, 0, 0, nullptr, 0, p->d_red_of, p->d_red_first, p->d_ph_mu, p->E, p->NP,
                     (const cplx*)d_tab, p->n_end, scratch, rep_flag);
  BIEM_LAUNCHCHK();
  if (scratch) BIEM_HIPCHK(hipFreeAsync(scratch, st));
  return BIEM_OK;
}
// the list-free 2-D kernels index the table row directly: T'[j] at j, phase e^{i j phi} at E + j (phase ids in order of the orders)
static bool plan_2d_direct(const biem_plan* p) {
  if (p->tree != TREE_A || p->E != p->n2 || p->NP != p->n2 || (int)p->ph_mu.size() != 2 * p->n2) return false;
  for (int i = 0; i < p->n2; ++i) if (p->ph_mu[2 * i] != i || p->red_label[i] != i) return false;
  return true;
}

int launch_fill(const biem_plan* p, int nb, int B, const double* d_k, const double* d_centers, int geom_batched,
                const double* d_tab, int scaling, double* d_A, long long lda, long long sys_stride, int n_pad,
                void* d_work, size_t work_bytes, hipStream_t st) {
  const int H = p->H, N = B * H;
  if (nb <= 0 || B <= 0) return BIEM_OK;
  if (lda < (n_pad > N ? n_pad : N) || n_pad < N) { set_error("biem_fill: lda/n_pad too small"); return BIEM_ERR_ARG; }
  if (work_bytes < fill_workspace_bytes(p, nb, B)) { set_error("biem_fill: workspace too small"); return BIEM_ERR_ARG; }
  if (scaling != BIEM_FILL_REFERENCE && scaling != BIEM_FILL_EQUILIBRATED) { set_error("biem_fill: bad scaling"); return BIEM_ERR_ARG; }
  cplx* T = (cplx*)d_work;
  ProfScope ps(PK_FILL, st, 16.0 * (double)nb * N * (double)N);
  if (p->tree == TREE_A && (!p->lists_built || !getenv("BIEM_FILL_FORM"))) {
    // 2-D: no term lists, any order (BIEM_FILL_FORM set: the generic list kernel below, for A / B tests)
    if (!plan_2d_direct(p)) { set_error("biem_fill: internal: 2-D table order"); return BIEM_ERR_ARG; }
    if (nb > 65535) { set_error("biem_fill: at most 65535 systems per call"); return BIEM_ERR_ARG; }
    const int npairs = B * (B - 1) / 2;
    if (npairs + B > 65535) { set_error("biem_fill (2-D): too many balls for one launch (%d)", B); return BIEM_ERR_UNSUPPORTED; }
    if (B > 1) { const int rc = launch_pair_tables_red(p, nb, B, d_k, d_centers, geom_batched, d_tab, T, st); if (rc) return rc; }
    hipLaunchKernelGGL(k_fill2d, dim3((unsigned)(((long long)H * H + 255) / 256), npairs + B, nb), dim3(256), 0, st, p->n_end, H, B, npairs, T,
                       (const cplx*)d_tab, scaling, (cplx*)d_A, lda, sys_stride);
    BIEM_LAUNCHCHK();
    if (n_pad > N) {
      hipLaunchKernelGGL(k_fill_pad, dim3(64, nb), dim3(256), 0, st, N, n_pad, (cplx*)d_A, lda, sys_stride);
      BIEM_LAUNCHCHK();
    }
    return BIEM_OK;
  }
  if (2 * p->n_end > kMaxRad) { set_error("n_end=%d exceeds the built table size", p->n_end); return BIEM_ERR_UNSUPPORTED; }
  if (B > 1) {
    hipLaunchKernelGGL(k_pair_tables, dim3(B * B, nb), dim3(64), 0, st, p->tree, p->d, p->n2, p->H2, p->Cd, p->d_labels2,
                       p->d_deg2, B, (const cplx*)d_k, d_centers, geom_batched, T, 0, 0, nullptr, 0);
    BIEM_LAUNCHCHK();
  }
  size_t shm = (size_t)((p->fill_table_global ? 0 : p->H2) + 2 * H) * sizeof(cplx) + (size_t)p->chunk_terms_max * 10 + (size_t)(p->chunk_ents_max + 1) * 4 + 16;
  if (shm > 160 * 1024 || (p->chunk_terms_max == 0 && p->coef.size() > 0)) {
    set_error("biem_fill: tables do not fit LDS (H=%d, H2=%d, chunk terms=%d)", H, p->H2, p->chunk_terms_max);
    return BIEM_ERR_UNSUPPORTED;
  }
  BIEM_HIPCHK(hipFuncSetAttribute((const void*)k_fill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  const int nchunks = (int)p->chunk_ent.size() - 1;
  hipLaunchKernelGGL(k_fill, dim3(nchunks, (B + 1) / 2, nb), dim3(FILL_THREADS), shm, st, H, p->H2, p->n_end, B, p->d_deg, p->d_chunk_ent,
                     p->chunk_terms_max, p->chunk_ents_max, p->d_ptr, p->d_coef, p->d_tidx16, T, (const cplx*)d_tab, scaling,
                     (cplx*)d_A, lda, sys_stride, p->fill_table_global ? 1 : 0);
  BIEM_LAUNCHCHK();
  if (n_pad > N) {
    hipLaunchKernelGGL(k_fill_pad, dim3(64, nb), dim3(256), 0, st, N, n_pad, (cplx*)d_A, lda, sys_stride);
    BIEM_LAUNCHCHK();
  }
  return BIEM_OK;
}

// ---------------------------------------------------------------------------------------------
// Symmetric fill (default path): the complex-symmetric form A~ = R W^H M W R^-1 of the equilibrated system, written ONCE and
// only where the symmetric factorisation reads it (UPPER triangle + the diagonal 64 x 64 tiles) - the general fill followed by
// the in-place symmetrising transform moved 4x the bytes.
//   W: unitary map to real harmonics (units (h, p), conj Y_h = Y_p: "cosine" (e_h + e_p)/sqrt2, "sine" i (e_p - e_h)/sqrt2),
//   R = diag(1/sqrt(gj gh)) per (ball, degree).  With q_b[n] = gj / sqrt(gj gh) the block (b, b'), b != b', is
//        A~[(b,x), (b',x')] = q_b[n] q_b'[n'] (W^H S_{bb'} W)[x, x'],      S = (S|R)^T raw sums,  diagonal blocks = identity.
// Internal order of a ball's unknowns: slot u = cosine combination of unit u, slot U + spos[u] = its sine combination
// (plan.hpp); consecutive lanes own consecutive u', so every store instruction covers contiguous runs of a matrix row.
// One 512-thread workgroup owns a chunk of unit pairs (its four term lists per pair stay in LDS) and loops over
// (upper ball pair b < b', system) combinations: per combination only the pair table T (H2 complex) is staged - fetched into
// registers during the previous combination's contraction - so the term lists are read from L2 once per workgroup.
// Each thread runs the four independent chains of its 2 x 2 block (the per-term chain idx -> T -> fma is LDS-latency bound).
// ---------------------------------------------------------------------------------------------
// Pair classes (FillDedupe, common.hpp).  Upper pairs are numbered pr = bp (bp - 1) / 2 + b, b < bp.  out: [0] = number of classes,
// [4 ..] rep_list[class] = its first pair, then dup_ptr[class .. class + 1] into dup_bb[] = (b << 16 | bp) of the class's pairs
// (the representative first).  enable = 0: every pair its own class.  One workgroup; quadratic search, npairs <= kDedupeMaxPairs.
__global__ void __launch_bounds__(256) k_pair_dedupe(int B, int d, int npairs, const double* __restrict__ centers, const double* __restrict__ radii,
                                                      const double* __restrict__ alpha, const double* __restrict__ beta, int enable,
                                                      int* __restrict__ out) {
  extern __shared__ int sd_int[];
  int* cls = sd_int;                 // [B] first ball with the same (radius, alpha, beta)
  int* rep = cls + B;                // [npairs] first pair of the same class
  int* cnt = rep + npairs;           // [npairs] members per representative, then write cursors
  int* rep_list = out + 4;
  int* dup_ptr = rep_list + npairs;
  int* dup_bb = dup_ptr + npairs + 1;
  int* rep_flag = dup_bb + npairs;            // [B * B]
  const int tid = threadIdx.x;
  for (int e = tid; e < B * B; e += 256) rep_flag[e] = enable ? 0 : 1;
  auto pair_of = [&](int pr, int& b, int& bp) {
    int bb = (int)((sqrtf(8.0f * (float)pr + 1.0f) + 1.0f) * 0.5f);
    while (bb * (bb - 1) / 2 > pr) --bb;
    while ((bb + 1) * bb / 2 <= pr) ++bb;
    bp = bb; b = pr - bb * (bb - 1) / 2;
  };
  if (!enable) {
    for (int p = tid; p < npairs; p += 256) { int b, bp; pair_of(p, b, bp); rep_list[p] = p; dup_ptr[p] = p; dup_bb[p] = b << 16 | bp; }
    if (tid == 0) { out[0] = npairs; dup_ptr[npairs] = npairs; }
    return;
  }
  for (int b = tid; b < B; b += 256) {
    int c = b;
    for (int b2 = 0; b2 < b; ++b2)
      if (radii[b2] == radii[b] && alpha[2 * b2] == alpha[2 * b] && alpha[2 * b2 + 1] == alpha[2 * b + 1] && beta[2 * b2] == beta[2 * b] &&
          beta[2 * b2 + 1] == beta[2 * b + 1]) { c = b2; break; }
    cls[b] = c;
  }
  __syncthreads();
  // per pair: its ball classes (packed) and its displacement, once - the quadratic search below then only compares
  int* key = cnt + npairs;                                   // [npairs] cls[b] << 16 | cls[bp]
  double* disp = (double*)(key + npairs + ((B + 3 * npairs) & 1));    // [npairs][d], 8-byte aligned
  for (int p = tid; p < npairs; p += 256) {
    int b, bp; pair_of(p, b, bp);
    key[p] = cls[b] << 16 | cls[bp];
    for (int i = 0; i < d; ++i) disp[p * d + i] = centers[bp * d + i] - centers[b * d + i];
  }
  __syncthreads();
  // displacements equal up to the rounding of the subtraction (a lattice with a pitch that is no binary fraction: 2.1 - 1.4 and
  // 1.4 - 0.7 differ in the last place): 32 ulp of the largest coordinate; an exact lattice matches bit for bit either way
  double cmax = 0.0;
  for (int e = 0; e < B * d; ++e) cmax = fmax(cmax, fabs(centers[e]));
  const double tol = 32.0 * 2.220446049250313e-16 * cmax;
  for (int p = tid; p < npairs; p += 256) {
    int r = p;
    const int kp = key[p];
    for (int q = 0; q < p; ++q) {
      if (key[q] != kp) continue;
      bool same = true;
      for (int i = 0; i < d; ++i) same = same && (fabs(disp[q * d + i] - disp[p * d + i]) <= tol);
      if (same) { r = q; break; }
    }
    rep[p] = r; cnt[p] = 0;
  }
  __syncthreads();
  if (tid == 0) {
    for (int p = 0; p < npairs; ++p) if (rep[p] != p) rep[p] = rep[rep[p]];     // (near-equality need not be transitive: resolve chains; rep[q], q < p, is final)
    for (int p = 0; p < npairs; ++p) cnt[rep[p]]++;
    int nrep = 0, pos = 0;
    for (int p = 0; p < npairs; ++p)
      if (rep[p] == p) { rep_list[nrep] = p; dup_ptr[nrep] = pos; pos += cnt[p]; cnt[p] = dup_ptr[nrep]; rep[p] = -nrep - 1; ++nrep;   // rep[p] < 0: class index
                         int b, bp; pair_of(p, b, bp); rep_flag[b * B + bp] = 1; }
    dup_ptr[nrep] = pos;
    out[0] = nrep;
    for (int p = 0; p < npairs; ++p) {           // pairs in ascending order: the representative is the first of its class
      const int r = rep[p] < 0 ? p : rep[p];
      int b, bp; pair_of(p, b, bp);
      dup_bb[cnt[r]++] = b << 16 | bp;
    }
  }
}

constexpr int FILL_SYM_THREADS = 1024;
constexpr int NB_TILE = 64;             // the factorisation's tile: diagonal 64 x 64 tiles are read whole
constexpr int FILL_SYM_MAXT = 8;           // pair-table elements prefetched per thread: H2lin <= 8 * 1024

// KT = pair-table elements each thread carries in registers from one combination to the next (KT * 512 >= H2); the loads are
// unconditional with a clamped index (a conditionally assigned register array was kept in scratch by hipcc)
template <int KT>
__global__ void __launch_bounds__(FILL_SYM_THREADS) k_fill_sym(int H, int U, int H2, int n_end, int B, int nb, int npairs,
                                                                const int* __restrict__ deg, const int* __restrict__ units,
                                                                const int* __restrict__ spos, const int* __restrict__ qchunk,
                                                                int terms_max, int pairs_max, const uint32_t* __restrict__ qptr,
                                                                const double* __restrict__ qcoef, const uint16_t* __restrict__ qidx,
                                                                const cplx* __restrict__ T, const cplx* __restrict__ tab,
                                                                cplx* __restrict__ A, long long lda, long long sys_stride,
                                                                const int* __restrict__ classes) {
  extern __shared__ char smem[];
  cplx* sT = (cplx*)smem;                                  // [H2] pair table of the current combination
  cplx* sQ = sT + H2;                                      // [2][n_end]: q of the row ball, q of the column ball
  double* sCoef = (double*)(sQ + 2 * n_end);               // [terms_max + 1]: the chunk's terms, then a dummy (0.0, index 0)
  uint32_t* sPtr = (uint32_t*)(sCoef + terms_max + 1);     // [2 pairs_max + 1], relative to the chunk's first term
  uint16_t* sIdx = (uint16_t*)(sPtr + 2 * pairs_max + 1);  // [terms_max + 1]
  const int tid = threadIdx.x;
  const int p0 = qchunk[blockIdx.x], p1 = qchunk[blockIdx.x + 1], npr = p1 - p0;
  const uint32_t t0 = qptr[2 * (size_t)p0], t1 = qptr[2 * (size_t)p1];
  for (uint32_t q = t0 + tid; q < t1; q += FILL_SYM_THREADS) { sCoef[q - t0] = qcoef[q]; sIdx[q - t0] = qidx[q]; }
  for (int e = tid; e <= 2 * npr; e += FILL_SYM_THREADS) sPtr[e] = qptr[2 * (size_t)p0 + e] - t0;
  if (tid == 0) { sCoef[t1 - t0] = 0.0; sIdx[t1 - t0] = 0; }
  // this thread's unit pair (fixed for the whole kernel)
  const bool active = tid < npr;
  const int pi = p0 + (active ? tid : 0);
  const int u = pi / U, v = pi - u * U;
  const int rh = units[2 * u], rp = units[2 * u + 1], ch = units[2 * v], cp = units[2 * v + 1];
  const bool r2 = rp != rh, c2 = cp != ch;                 // two rows / two columns in this block
  const int row_c = u, row_s = r2 ? U + spos[u] : 0, col_c = v, col_s = c2 ? U + spos[v] : 0;
  const int nrow = deg[rh], ncol = deg[ch];
  const double q2 = 0.70710678118654752440;
  // pair classes (k_pair_dedupe): a combination is (system, class); its block is contracted once from the representative pair's
  // table and stored to every pair of the class
  const int nrep = classes[0];
  const int* rep_list = classes + 4;
  const int* dup_ptr = rep_list + npairs;
  const int* dup_bb = dup_ptr + npairs + 1;
  const int ncomb = nrep * nb;
  // eight named registers instead of an array: hipcc kept every array form (plain, unrolled, compile-time indexed through
  // lambdas) in scratch memory
#define BIEM_TN_LIST(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BIEM_TN_DECL(k) cplx tn##k = make_double2(0.0, 0.0);
  BIEM_TN_LIST(BIEM_TN_DECL)
#define BIEM_TN_LOAD(k) if (k < KT) { const int l = k * FILL_SYM_THREADS + tid; tn##k = Tp_[l < H2 ? l : H2 - 1]; }
#define BIEM_TN_PUT(k) if (k < KT) { const int l = k * FILL_SYM_THREADS + tid; if (l < H2) sT[l] = tn##k; }
  auto pair_of = [&](int pr, int& b, int& bp) {           // pr-th upper pair (row ball b < column ball bp)
    int bb = (int)((sqrtf(8.0f * (float)pr + 1.0f) + 1.0f) * 0.5f);
    while (bb * (bb - 1) / 2 > pr) --bb;
    while ((bb + 1) * bb / 2 <= pr) ++bb;
    bp = bb; b = pr - bb * (bb - 1) / 2;
  };
  auto table_of = [&](int cb) -> const cplx* {
    const int s = cb / nrep, pr = rep_list[cb - s * nrep];
    int b, bp; pair_of(pr, b, bp);
    return T + ((size_t)s * B * B + (size_t)b * B + bp) * H2;
  };
  int comb = blockIdx.y;
  if (comb < ncomb) { const cplx* Tp_ = table_of(comb); BIEM_TN_LIST(BIEM_TN_LOAD) }
  for (; comb < ncomb; comb += gridDim.y) {
    const int s = comb / nrep, ci = comb - s * nrep, pr = rep_list[ci];
    int b, bp; pair_of(pr, b, bp);
    __syncthreads();                                       // the previous combination's readers are done (also orders the chunk loads)
    BIEM_TN_LIST(BIEM_TN_PUT)
    if (tid < 2 * n_end) {
      const int which = tid >= n_end, n = tid - which * n_end;
      const cplx* tb = tab + ((size_t)s * B + (which ? bp : b)) * 3 * n_end;
      sQ[tid] = cmul(tb[n], crecip(zsqrt(cmul(tb[n], tb[n_end + n]))));     // gj / sqrt(gj gh)
    }
    __syncthreads();
    if (comb + (int)gridDim.y < ncomb) { const cplx* Tp_ = table_of(comb + (int)gridDim.y); BIEM_TN_LIST(BIEM_TN_LOAD) }   // lands while this combination is contracted
    if (!active) continue;
    // Two term lists per unit pair (plan.hpp): A = (h,h'), B = (h,p'); the conjugate entries (p,p') and (p,h') have the same
    // coefficients with every table index replaced by its partner's, which the paired table layout keeps at index ^ 1.  So one
    // coefficient and one index read feed two chains; per step all reads of both lists are issued, then the four table reads,
    // then the FMAs.  A list that has run out reads the chunk's dummy term (coefficient 0, index 0).
    const uint32_t dummy = sPtr[2 * npr];                 // = number of terms of the chunk: the slot behind them
    uint32_t qa = sPtr[2 * tid], qb = sPtr[2 * tid + 1];
    const uint32_t ea = qb, eb = sPtr[2 * tid + 2];
    const uint32_t lm = (ea - qa) > (eb - qb) ? (ea - qa) : (eb - qb);
    double s0r = 0, s0i = 0, s1r = 0, s1i = 0, s2r = 0, s2i = 0, s3r = 0, s3i = 0;      // A, mirror of A, B, mirror of B
    for (uint32_t i = 0; i < lm; ++i) {
      const uint32_t ga = qa < ea ? qa : dummy, gb = qb < eb ? qb : dummy;
      const double ca = sCoef[ga], cb = sCoef[gb];
      const unsigned ia = sIdx[ga], ib = sIdx[gb];
      const cplx za = sT[ia], zam = sT[ia ^ 1u], zb = sT[ib], zbm = sT[ib ^ 1u];
      s0r = fma(ca, za.x, s0r); s0i = fma(ca, za.y, s0i);
      s1r = fma(ca, zam.x, s1r); s1i = fma(ca, zam.y, s1i);
      s2r = fma(cb, zb.x, s2r); s2i = fma(cb, zb.y, s2i);
      s3r = fma(cb, zbm.x, s3r); s3i = fma(cb, zbm.y, s3i);
      ++qa; ++qb;
    }
    // W^H S W on the 2 x 2 block: rows (h + p)/sqrt2, i (h - p)/sqrt2; columns (h' + p')/sqrt2, i (p' - h')/sqrt2
    // raw entries: (h,h') = A; (p,p') = mirror A and (h,p') = B, (p,h') = mirror B when both units are doubles; with a single
    // row unit the mirror of A is (h,p'), with a single column unit it is (p,h')
    const cplx mA = make_double2(s1r, s1i);
    cplx x00 = make_double2(s0r, s0i), x01 = (r2 && c2) ? make_double2(s2r, s2i) : mA, x10 = (r2 && c2) ? make_double2(s3r, s3i) : mA, x11 = mA;
    if (r2) {
      const cplx a0c = make_double2((x00.x + x10.x) * q2, (x00.y + x10.y) * q2), a1c = make_double2((x01.x + x11.x) * q2, (x01.y + x11.y) * q2);
      const cplx d0 = make_double2((x00.x - x10.x) * q2, (x00.y - x10.y) * q2), d1 = make_double2((x01.x - x11.x) * q2, (x01.y - x11.y) * q2);
      x00 = a0c; x01 = a1c; x10 = make_double2(-d0.y, d0.x); x11 = make_double2(-d1.y, d1.x);
    }
    if (c2) {
      const cplx a0c = make_double2((x00.x + x01.x) * q2, (x00.y + x01.y) * q2), d0 = make_double2((x01.x - x00.x) * q2, (x01.y - x00.y) * q2);
      const cplx a1c = make_double2((x10.x + x11.x) * q2, (x10.y + x11.y) * q2), d1 = make_double2((x11.x - x10.x) * q2, (x11.y - x10.y) * q2);
      x00 = a0c; x01 = make_double2(-d0.y, d0.x); x10 = a1c; x11 = make_double2(-d1.y, d1.x);
    }
    const cplx scale = cmul(sQ[nrow], sQ[n_end + ncol]);
    cplx* As = A + (size_t)s * sys_stride;
    const int e0 = dup_ptr[ci], e1 = dup_ptr[ci + 1];
    auto put = [&](int rslot, int cslot, cplx val) {
      const cplx w = cmul(val, scale);
      for (int e = e0; e < e1; ++e) {                        // every pair of the class (the representative first)
        const int bb = dup_bb[e];
        const int row = (bb >> 16) * H + rslot, col = (bb & 0xffff) * H + cslot;   // b < bp: strictly above the diagonal
        As[(size_t)row * lda + col] = w;
        if ((row >> 6) == (col >> 6)) As[(size_t)col * lda + row] = w;   // a diagonal 64 x 64 tile is read whole: mirror (A~ = A~^T)
      }
    };
    put(row_c, col_c, x00);
    if (c2) put(row_c, col_s, x01);
    if (r2) { put(row_s, col_c, x10); if (c2) put(row_s, col_s, x11); }
  }
#undef BIEM_TN_LIST
#undef BIEM_TN_DECL
#undef BIEM_TN_LOAD
#undef BIEM_TN_PUT
}

// ---------------------------------------------------------------------------------------------
// Symmetric fill, reduced-table form (default where its tables fit LDS).  Same work split as k_fill_sym above - a workgroup owns a
// chunk of unit pairs, one per thread, and loops over (class, system) combinations, the next combination's table prefetched into
// registers - but the contraction reads LDS without bank conflicts and half as much of it:
//   * the phase e^{i mu . phi} is common to all terms of an entry and leaves the sum (plan.hpp): ONE chain per list over the
//     reduced table T'[e] (a label and its conjugate partner share an entry), the conjugate entry is conj(phase) x the same sum;
//   * the term lists of the 64 unit pairs of a wave are stored transposed and padded to the wave's longest list: per step the wave
//     reads 64 consecutive coefficients and 64 consecutive 16-bit indices, all trip counts are wave-uniform (no divergence, no
//     clamped dummy terms);
//   * consecutive lanes are consecutive column units (n', m' = -n' .. 0) and T' is laid out degree-major, so the 64 table reads of a
//     step fall on (nearly) consecutive entries.
// The gather form above measured 51 % of its LDS cycles as bank conflicts and 42 bytes of LDS per term and lane (profiles/r02_pmc_summary.txt).
// ---------------------------------------------------------------------------------------------
// workgroup barrier that orders LDS traffic only (see k_fill_red)
__device__ inline void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
#ifdef BIEM_FILL_TRACE
__device__ unsigned long long g_fill_trace[8];
#define BIEM_FT(i) { if (tid == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ft_acc[i] += now_ - ft_t; ft_t = now_; } }
#else
#define BIEM_FT(i)
#endif
template <int KT, int NC>
__global__ void __launch_bounds__(FILL_SYM_THREADS) k_fill_red(int H, int U, int HR, int E, int NP, int n_end, int B, int nb, int npairs,
                                                                const int* __restrict__ deg, const int* __restrict__ units,
                                                                const int* __restrict__ spos, const int* __restrict__ rchunk,
                                                                const int* __restrict__ rcrow, const int* __restrict__ rwrow, int rows_max,
                                                                const double* __restrict__ rcoef, const uint16_t* __restrict__ ridx,
                                                                const uint16_t* __restrict__ rphsel,
                                                                const cplx* __restrict__ T,
                                                                cplx* __restrict__ A, long long lda, long long sys_stride,
                                                                const int* __restrict__ classes) {
  extern __shared__ char smem[];
  // NC = 2: TWO combinations per iteration - the coefficient and index reads of a group of rows feed both tables (a fifth less LDS
  // traffic per term: 21 instead of 26 bytes) and the barriers, the loop head and the list walk are paid once for two blocks
  cplx* sT = (cplx*)smem;                                  // [NC][HR = E + NP + 2 n_end] table rows of the current combinations: T', phases, q factors
  double* sCoef = (double*)(sT + (size_t)NC * HR);         // [rows_max][64]
  uint16_t* sIdx = (uint16_t*)(sCoef + (size_t)rows_max * 64);   // [rows_max / 4][64][4]
  // (no static __shared__ here: statics precede the dynamic region unpadded, 33 ints would leave every ds_read_b64 / b128 below
  // misaligned - replayed at 64 cycles per wave-instruction; measured: 143 instead of 46 ms per 256 systems of cfg 3)
  int* sW = (int*)(sIdx + (size_t)rows_max * 64);          // [33] row offsets of the waves' lists
  const int tid = threadIdx.x, lane = tid & 63, TH = blockDim.x;          // 1024 threads (one workgroup per CU) or 512 (two)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p0 = rchunk[blockIdx.x], p1 = rchunk[blockIdx.x + 1], npr = p1 - p0;
  const int r0 = rcrow[blockIdx.x], nrows = rcrow[blockIdx.x + 1] - r0;
  {
    const double* gc = rcoef + (size_t)r0 * 64;
    const uint16_t* gi = ridx + (size_t)r0 * 64;
    for (int q = tid; q < nrows * 64; q += TH) { sCoef[q] = gc[q]; sIdx[q] = gi[q]; }
    if (tid < 33) sW[tid] = rwrow[blockIdx.x * 33 + tid];
  }
  // this thread's unit pair (fixed for the whole kernel): slots, degrees, phase selectors and the offsets of its (up to) four
  // entries inside a block, so that a combination's stores need no 64-bit multiplications
  const bool active = tid < npr;
  const int pi = p0 + (active ? tid : 0);
  const int u = pi / U, v = pi - u * U;
  const int rh = units[2 * u], rp = units[2 * u + 1], ch = units[2 * v], cp = units[2 * v + 1];
  const bool r2 = rp != rh, c2 = cp != ch;                 // two rows / two columns in this block
  const int row_c = u, row_s = r2 ? U + spos[u] : 0, col_c = v, col_s = c2 ? U + spos[v] : 0;
  const int nrow = deg[rh], ncol = deg[ch];
  const unsigned selA = rphsel[2 * (size_t)pi], selB = rphsel[2 * (size_t)pi + 1];
  const long long o00 = (long long)row_c * lda + col_c, o01 = (long long)row_c * lda + col_s, o10 = (long long)row_s * lda + col_c,
                  o11 = (long long)row_s * lda + col_s;
  const double q2 = 0.70710678118654752440;
  const int nrep = classes[0];
  const int* dup_ptr = classes + 4 + npairs;
  const int* dup_bb = dup_ptr + npairs + 1;
  const int ncomb = nrep * nb;
#define BIEM_TN_LIST(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
  typedef double v2d_t __attribute__((ext_vector_type(2)));   // (a native vector: a struct cannot be a tied asm operand)
#define BIEM_TN_DECL(k) v2d_t tn##k = {0.0, 0.0};
  BIEM_TN_LIST(BIEM_TN_DECL)
#define BIEM_TM_DECL(k) v2d_t tm##k = {0.0, 0.0};
  BIEM_TN_LIST(BIEM_TM_DECL)                                 // (second combination of an iteration, NC = 2)
  // The prefetch loads are inline asm and their wait is the explicit one of BIEM_TN_CLAIM: hipcc's own wait for a VGPR load is a
  // vmcnt(0) wherever control flow joins, i.e. at the top of the loop, AFTER this combination's stores - every iteration would then
  // wait for the acknowledgement of its own stores (microseconds under a full HBM write queue, with the CU to itself).
#define BIEM_TN_LOAD(k) if (k < KT) { const int l = k * TH + tid; const cplx* a_ = Tp_ + (l < HR ? l : HR - 1); \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(tn##k) : "v"(a_) : "memory"); }
#define BIEM_TN_CLAIM(k) if (k < KT) asm volatile("s_waitcnt vmcnt(0)" : "+v"(tn##k) : : "memory");
#define BIEM_TN_PUT(k) if (k < KT) { const int l = k * TH + tid; if (l < HR) sT[l] = make_double2(tn##k.x, tn##k.y); }
#define BIEM_TM_LOAD(k) if (NC > 1 && k < KT) { const int l = k * TH + tid; const cplx* a_ = Tp_ + (l < HR ? l : HR - 1); \
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(tm##k) : "v"(a_) : "memory"); }
#define BIEM_TM_CLAIM(k) if (NC > 1 && k < KT) asm volatile("s_waitcnt vmcnt(0)" : "+v"(tm##k) : : "memory");
#define BIEM_TM_PUT(k) if (NC > 1 && k < KT) { const int l = k * TH + tid; if (l < HR) sT[HR + l] = make_double2(tm##k.x, tm##k.y); }
  // a combination is (system s, class ci); its table is the one of the class's first pair (dup_bb: b << 16 | bp, the representative first)
  auto table_of = [&](int cb) -> const cplx* {
    const int s = cb / nrep, bb = dup_bb[dup_ptr[cb - s * nrep]];
    return T + ((size_t)s * B * B + (size_t)(bb >> 16) * B + (bb & 0xffff)) * HR;
  };
  // iteration i of this workgroup takes the combinations comb and (NC = 2) comb + G, G = gridDim.y; the next one comb + NC G
  const int G = gridDim.y;
  int comb = blockIdx.y;
  if (comb < ncomb) { const cplx* Tp_ = table_of(comb); BIEM_TN_LIST(BIEM_TN_LOAD) }
  if (NC > 1 && comb + G < ncomb) { const cplx* Tp_ = table_of(comb + G); BIEM_TN_LIST(BIEM_TM_LOAD) }
  BIEM_TN_LIST(BIEM_TN_CLAIM)
  BIEM_TN_LIST(BIEM_TM_CLAIM)
#ifdef BIEM_FILL_TRACE
  unsigned long long ft_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ft_t = __builtin_amdgcn_s_memtime();
#endif
  for (; comb < ncomb; comb += NC * G) {
    const bool two = NC > 1 && comb + G < ncomb;           // (wave-uniform) a second combination in this iteration
    BIEM_FT(0)
    // Raw barriers with an LDS-only wait: __syncthreads() carries a workgroup-scope fence, which hipcc lowers to s_waitcnt vmcnt(0) -
    // every barrier would wait for the acknowledgement of the previous combination's global stores (nobody in the workgroup reads
    // them).  Only LDS traffic has to be ordered here: this wave's LDS reads / writes are complete at lgkmcnt(0).
    lds_barrier();                                         // the previous combination's readers are done (also orders the chunk loads)
    BIEM_FT(1)
    BIEM_TN_LIST(BIEM_TN_PUT)
    BIEM_TN_LIST(BIEM_TM_PUT)
    BIEM_FT(2)
    lds_barrier();
    BIEM_FT(4)
    if (comb + NC * G < ncomb) { const cplx* Tp_ = table_of(comb + NC * G); BIEM_TN_LIST(BIEM_TN_LOAD) }   // lands while this iteration is contracted
    if (NC > 1 && comb + (NC + 1) * G < ncomb) { const cplx* Tp_ = table_of(comb + (NC + 1) * G); BIEM_TN_LIST(BIEM_TM_LOAD) }
    // (a wave without unit pairs runs zero groups and joins the others at the claim of the prefetched registers below: every path
    // through the loop body must pass that point, or the compiler drains the memory queue again where the paths meet)
    const bool wave_on = wave * 64 < npr;
    // rows of list A: [ra, rb), of list B: [rb, re), all multiples of 4.  One loop over the groups of four rows, software
    // pipelined: the coefficient / index reads of group g + 1 are in flight while the four table reads of group g are waited for
    // (the chain index -> table entry -> fma is two dependent LDS round trips otherwise); at the A | B boundary (wave-uniform)
    // the accumulators are handed over.
    const int ga = wave_on ? sW[2 * wave] >> 2 : 0, gb = wave_on ? sW[2 * wave + 1] >> 2 : 0, ge = wave_on ? sW[2 * wave + 2] >> 2 : 0;
    double ar = 0.0, ai = 0.0, xr = 0.0, xi = 0.0;        // running sums; (xr, xi) keeps list A's once list B has started
    double br = 0.0, bi = 0.0, yr = 0.0, yi = 0.0;        // the same for the second combination
    if (ga < ge) {
      const double* cc = sCoef + lane;
      const uint2* ii = (const uint2*)sIdx + lane;
      uint2 pk = ii[(size_t)ga * 64];
      double c0 = cc[(size_t)(4 * ga) * 64], c1 = cc[(size_t)(4 * ga + 1) * 64], c2v = cc[(size_t)(4 * ga + 2) * 64], c3 = cc[(size_t)(4 * ga + 3) * 64];
      for (int g = ga; g < ge; ++g) {
        const int gn = g + 1 < ge ? g + 1 : g;             // (the last trip re-reads its own group: harmless)
        const cplx z0 = sT[pk.x & 0xffffu], z1 = sT[pk.x >> 16], z2 = sT[pk.y & 0xffffu], z3 = sT[pk.y >> 16];
        cplx w0, w1, w2, w3;
        if (NC > 1) { w0 = sT[HR + (pk.x & 0xffffu)]; w1 = sT[HR + (pk.x >> 16)]; w2 = sT[HR + (pk.y & 0xffffu)]; w3 = sT[HR + (pk.y >> 16)]; }
        const uint2 pkn = ii[(size_t)gn * 64];
        const double n0 = cc[(size_t)(4 * gn) * 64], n1 = cc[(size_t)(4 * gn + 1) * 64], n2 = cc[(size_t)(4 * gn + 2) * 64], n3 = cc[(size_t)(4 * gn + 3) * 64];
        if (g == gb) { xr = ar; xi = ai; ar = 0.0; ai = 0.0; yr = br; yi = bi; br = 0.0; bi = 0.0; }
        ar = fma(c0, z0.x, ar); ai = fma(c0, z0.y, ai);
        ar = fma(c1, z1.x, ar); ai = fma(c1, z1.y, ai);
        ar = fma(c2v, z2.x, ar); ai = fma(c2v, z2.y, ai);
        ar = fma(c3, z3.x, ar); ai = fma(c3, z3.y, ai);
        if (NC > 1) {
          br = fma(c0, w0.x, br); bi = fma(c0, w0.y, bi);
          br = fma(c1, w1.x, br); bi = fma(c1, w1.y, bi);
          br = fma(c2v, w2.x, br); bi = fma(c2v, w2.y, bi);
          br = fma(c3, w3.x, br); bi = fma(c3, w3.y, bi);
        }
        pk = pkn; c0 = n0; c1 = n1; c2v = n2; c3 = n3;
      }
    }
    BIEM_FT(5)
    // The prefetched table row is claimed HERE, before this combination's stores are issued: the wait then sees only the loads (issued
    // before the contraction) and the PREVIOUS combination's stores, which have had a whole iteration to drain.
    BIEM_TN_LIST(BIEM_TN_CLAIM)
    BIEM_TN_LIST(BIEM_TM_CLAIM)
    if (!active) continue;
    // the block of one combination from its two sums: phases, the 2 x 2 transform to real harmonics, q factors, stores
    auto finish = [&](const cplx* sTc, int cb, cplx RA, cplx RB) {
    const int s = cb / nrep, ci = cb - s * nrep;
    // entries of the 2 x 2 raw block: (h,h') = phase_A R_A, its conjugate entry conj(phase_A) R_A; (h,p') = phase_B R_B, (p,h') = conj(phase_B) R_B
    cplx phA = sTc[E + (selA >> 1)], phB = sTc[E + (selB >> 1)];
    if (selA & 1u) phA.y = -phA.y;
    if (selB & 1u) phB.y = -phB.y;
    const cplx mA = cmul(RA, make_double2(phA.x, -phA.y));
    cplx x00 = cmul(RA, phA), x01 = (r2 && c2) ? cmul(RB, phB) : mA, x10 = (r2 && c2) ? cmul(RB, make_double2(phB.x, -phB.y)) : mA, x11 = mA;
    if (r2) {
      const cplx a0c = make_double2((x00.x + x10.x) * q2, (x00.y + x10.y) * q2), a1c = make_double2((x01.x + x11.x) * q2, (x01.y + x11.y) * q2);
      const cplx d0 = make_double2((x00.x - x10.x) * q2, (x00.y - x10.y) * q2), d1 = make_double2((x01.x - x11.x) * q2, (x01.y - x11.y) * q2);
      x00 = a0c; x01 = a1c; x10 = make_double2(-d0.y, d0.x); x11 = make_double2(-d1.y, d1.x);
    }
    if (c2) {
      const cplx a0c = make_double2((x00.x + x01.x) * q2, (x00.y + x01.y) * q2), d0 = make_double2((x01.x - x00.x) * q2, (x01.y - x00.y) * q2);
      const cplx a1c = make_double2((x10.x + x11.x) * q2, (x10.y + x11.y) * q2), d1 = make_double2((x11.x - x10.x) * q2, (x11.y - x10.y) * q2);
      x00 = a0c; x01 = make_double2(-d0.y, d0.x); x10 = a1c; x11 = make_double2(-d1.y, d1.x);
    }
    const cplx scale = cmul(sTc[E + NP + nrow], sTc[E + NP + n_end + ncol]);
    x00 = cmul(x00, scale); x01 = cmul(x01, scale); x10 = cmul(x10, scale); x11 = cmul(x11, scale);
    cplx* As = A + (size_t)s * sys_stride;
    const int e0 = dup_ptr[ci], e1 = dup_ptr[ci + 1];
    for (int e = e0; e < e1; ++e) {                          // every pair of the class (the representative first)
      const int bb = dup_bb[e];
      const int rb0 = (bb >> 16) * H, cb0 = (bb & 0xffff) * H;     // b < bp: the block lies strictly above the diagonal
      cplx* Ab = As + (size_t)rb0 * lda + cb0;               // (wave-uniform: scalar arithmetic)
      Ab[o00] = x00;
      if (c2) Ab[o01] = x01;
      if (r2) { Ab[o10] = x10; if (c2) Ab[o11] = x11; }
      if (cb0 - rb0 - H < NB_TILE) {
        // a diagonal 64 x 64 tile is read whole by the factorisation: entries of this block that fall into one are mirrored (A~ = A~^T);
        // only blocks whose first column lies within 64 of their last row can reach one
        auto mirror = [&](int rslot, int cslot, cplx w) {
          const int row = rb0 + rslot, col = cb0 + cslot;
          if ((row >> 6) == (col >> 6)) As[(size_t)col * lda + row] = w;
        };
        mirror(row_c, col_c, x00);
        if (c2) mirror(row_c, col_s, x01);
        if (r2) { mirror(row_s, col_c, x10); if (c2) mirror(row_s, col_s, x11); }
      }
    }
    };
    if (gb < ge) finish(sT, comb, make_double2(xr, xi), make_double2(ar, ai));
    else finish(sT, comb, make_double2(ar, ai), make_double2(0.0, 0.0));
    if (two) {
      if (gb < ge) finish(sT + HR, comb + G, make_double2(yr, yi), make_double2(br, bi));
      else finish(sT + HR, comb + G, make_double2(br, bi), make_double2(0.0, 0.0));
    }
    BIEM_FT(6)
  }
#ifdef BIEM_FILL_TRACE
  if (tid == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_fill_trace[i], ft_acc[i]);
#endif
#undef BIEM_TN_LIST
#undef BIEM_TN_DECL
#undef BIEM_TN_LOAD
#undef BIEM_TN_PUT
#undef BIEM_TN_CLAIM
#undef BIEM_TM_DECL
#undef BIEM_TM_LOAD
#undef BIEM_TM_CLAIM
#undef BIEM_TM_PUT
}
#ifdef BIEM_FILL_TRACE
extern "C" int biem_debug_fill_trace(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fill_trace), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_fill_trace), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif

// ---------------------------------------------------------------------------------------------
// 2-D fills without term lists (tree a, any order).  Graf's theorem leaves ONE term per entry,
//     S(m, m') = i^{|m| + |mu| - |m'|} T[mu] / sqrt(2 pi),   mu = m' - m,   T[mu] = T'[|mu|] e^{i mu phi},   T'[j] = C_2 H_j(k|t|) / sqrt(2 pi)
// (the 1 / sqrt(2 pi) in front is the triple integral of three circular harmonics - the coefficient of the plan's one-term lists),
// so a block is a Toeplitz matrix in (m, m') up to signs: every thread evaluates its entries from the table row of the pair
// (k_pair_tables in reduced mode: T'[0 .. 2 n_end - 2], the phases e^{i j phi}, j >= 0, and the q factors of the two balls) read
// straight from global memory - consecutive lanes read consecutive entries, the row (at most 6 n_end complex numbers) stays in L1 /
// L2 - and stores them.  No LDS, no term lists: the plan of a 2-D order beyond kLists2dMax holds none (they would be H^2 = 47 M
// one-term entries at the reference's n_end = 3444, accuracy/accuracy_k_a.csv), and the kernels are bound by their stores.
// ---------------------------------------------------------------------------------------------
__device__ inline double sign_even(int e) { return ((e / 2) & 1) ? -1.0 : 1.0; }      // i^e for even e (either sign)

// symmetric form: one thread per unit pair (m, m'), 0 <= m, m' < n_end, of the block (b, bp), b < bp; grid (unit pairs / 256, combinations)
__global__ void __launch_bounds__(256) k_fill2d_sym(int n_end, int H, int B, int nb, int npairs, const cplx* __restrict__ T, cplx* __restrict__ A,
                                                     long long lda, long long sys_stride, const int* __restrict__ classes) {
  const int E = 2 * n_end - 1, HR = 2 * E + 2 * n_end, U = n_end;
  const int pi = blockIdx.x * 256 + threadIdx.x;
  if (pi >= U * U) return;
  const int m = pi / U, mp = pi - m * U;
  const bool r2 = m > 0, c2 = mp > 0;
  // slots of a ball's unknowns: cosine combination of unit m at m, its sine combination at U + m - 1
  const int row_c = m, row_s = U + m - 1, col_c = mp, col_s = U + mp - 1;
  const long long o00 = (long long)row_c * lda + col_c, o01 = (long long)row_c * lda + col_s, o10 = (long long)row_s * lda + col_c,
                  o11 = (long long)row_s * lda + col_s;
  const int muA = mp - m, jA = muA < 0 ? -muA : muA, jB = m + mp;
  const double sA = (muA >= 0 ? 1.0 : ((jA & 1) ? -1.0 : 1.0)) * kInvSqrt2Pi;     // i^{m + |mu| - m'} / sqrt(2 pi): sign 1 for m' >= m, (-1)^{m - m'} otherwise
  const double sB = ((m & 1) ? -1.0 : 1.0) * kInvSqrt2Pi;                          // (h, p'): mu = -(m + m'), i^{2 m}
  const double q2 = 0.70710678118654752440;
  const int nrep = classes[0];
  const int* dup_ptr = classes + 4 + npairs;
  const int* dup_bb = dup_ptr + npairs + 1;
  const int ncomb = nrep * nb;
  for (int comb = blockIdx.y; comb < ncomb; comb += gridDim.y) {
    const int s = comb / nrep, ci = comb - s * nrep;
    const int e0 = dup_ptr[ci], e1 = dup_ptr[ci + 1], bb0 = dup_bb[e0];
    const cplx* row = T + ((size_t)s * B * B + (size_t)(bb0 >> 16) * B + (bb0 & 0xffff)) * HR;
    const cplx tA = row[jA], tB = row[jB];
    cplx phA = row[E + jA], phB = row[E + jB];
    const cplx scale = cmul(row[2 * E + m], row[2 * E + n_end + mp]);
    if (muA < 0) phA.y = -phA.y;                          // T[mu] = T'[|mu|] e^{i mu phi}
    const cplx RA = cscale(tA, sA), RB = cscale(tB, sB);
    // raw 2 x 2 block of the units (m, -m) x (m', -m'): (h,h') = R_A e^{i mu_A phi}, (p,p') its conjugate-phase partner,
    // (h,p') = R_B e^{-i (m + m') phi}, (p,h') = R_B e^{+i (m + m') phi}
    const cplx mA = cmul(RA, make_double2(phA.x, -phA.y));
    cplx x00 = cmul(RA, phA), x01 = (r2 && c2) ? cmul(RB, make_double2(phB.x, -phB.y)) : mA, x10 = (r2 && c2) ? cmul(RB, phB) : mA, x11 = mA;
    if (r2) {
      const cplx a0c = make_double2((x00.x + x10.x) * q2, (x00.y + x10.y) * q2), a1c = make_double2((x01.x + x11.x) * q2, (x01.y + x11.y) * q2);
      const cplx d0 = make_double2((x00.x - x10.x) * q2, (x00.y - x10.y) * q2), d1 = make_double2((x01.x - x11.x) * q2, (x01.y - x11.y) * q2);
      x00 = a0c; x01 = a1c; x10 = make_double2(-d0.y, d0.x); x11 = make_double2(-d1.y, d1.x);
    }
    if (c2) {
      const cplx a0c = make_double2((x00.x + x01.x) * q2, (x00.y + x01.y) * q2), d0 = make_double2((x01.x - x00.x) * q2, (x01.y - x00.y) * q2);
      const cplx a1c = make_double2((x10.x + x11.x) * q2, (x10.y + x11.y) * q2), d1 = make_double2((x11.x - x10.x) * q2, (x11.y - x10.y) * q2);
      x00 = a0c; x01 = make_double2(-d0.y, d0.x); x10 = a1c; x11 = make_double2(-d1.y, d1.x);
    }
    x00 = cmul(x00, scale); x01 = cmul(x01, scale); x10 = cmul(x10, scale); x11 = cmul(x11, scale);
    cplx* As = A + (size_t)s * sys_stride;
    for (int e = e0; e < e1; ++e) {                          // every pair of the class (the representative first)
      const int bb = dup_bb[e];
      const int rb0 = (bb >> 16) * H, cb0 = (bb & 0xffff) * H;
      cplx* Ab = As + (size_t)rb0 * lda + cb0;
      Ab[o00] = x00;
      if (c2) Ab[o01] = x01;
      if (r2) { Ab[o10] = x10; if (c2) Ab[o11] = x11; }
      if (cb0 - rb0 - H < NB_TILE) {                         // entries inside a diagonal 64 x 64 tile are mirrored (read whole)
        auto mirror = [&](int rslot, int cslot, cplx w) {
          const int rr = rb0 + rslot, cc = cb0 + cslot;
          if ((rr >> 6) == (cc >> 6)) As[(size_t)cc * lda + rr] = w;
        };
        mirror(row_c, col_c, x00);
        if (c2) mirror(row_c, col_s, x01);
        if (r2) { mirror(row_s, col_c, x10); if (c2) mirror(row_s, col_s, x11); }
      }
    }
  }
}

// general form (reference / equilibrated scaling, natural order of the harmonics: m = 0 .. n-1, -(n-1) .. -1): one thread per entry
// (h, h') of the pair (b, bp), b < bp, writes block (b, bp) and - through T(-t) = (-1)^{n''} T(t) - block (bp, b); blockIdx.y >=
// npairs: the diagonal block of ball blockIdx.y - npairs.
__global__ void __launch_bounds__(256) k_fill2d(int n_end, int H, int B, int npairs, const cplx* __restrict__ T, const cplx* __restrict__ tab,
                                                 int scaling, cplx* __restrict__ A, long long lda, long long sys_stride) {
  const int E = 2 * n_end - 1, HR = 2 * E + 2 * n_end;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)H * H) return;
  const int h = (int)(e / H), hp = (int)(e - (long long)h * H);
  const int s = blockIdx.z;
  cplx* As = A + (size_t)s * sys_stride;
  if ((int)blockIdx.y >= npairs) {
    const int b = blockIdx.y - npairs;
    const int n = h < n_end ? h : H - h;
    const cplx* tb = tab + ((size_t)s * B + b) * 3 * n_end;
    cplx v = make_double2(0.0, 0.0);
    if (h == hp) v = scaling == BIEM_FILL_REFERENCE ? cmul(tb[n_end + n], tb[2 * n_end + n]) : make_double2(1.0, 0.0);
    As[((size_t)b * H + h) * lda + (size_t)b * H + hp] = v;
    return;
  }
  int bp = (int)((sqrtf(8.0f * (float)blockIdx.y + 1.0f) + 1.0f) * 0.5f);
  while (bp * (bp - 1) / 2 > (int)blockIdx.y) --bp;
  while ((bp + 1) * bp / 2 <= (int)blockIdx.y) ++bp;
  const int b = blockIdx.y - bp * (bp - 1) / 2;
  const int m = h < n_end ? h : h - H, mp = hp < n_end ? hp : hp - H;          // signed orders
  const int am = m < 0 ? -m : m, amp = mp < 0 ? -mp : mp, mu = mp - m, j = mu < 0 ? -mu : mu;
  const cplx* row = T + ((size_t)s * B * B + (size_t)b * B + bp) * HR;
  cplx ph = row[E + j];
  if (mu < 0) ph.y = -ph.y;
  const cplx raw = cscale(cmul(row[j], ph), sign_even(am + j - amp) * kInvSqrt2Pi);
  const cplx* tb = tab + ((size_t)s * B + b) * 3 * n_end;
  const cplx* tbp = tab + ((size_t)s * B + bp) * 3 * n_end;
  auto colfac = [&](const cplx* tball, int n) { return scaling == BIEM_FILL_REFERENCE ? tball[2 * n_end + n] : crecip(tball[n_end + n]); };
  As[((size_t)b * H + h) * lda + (size_t)bp * H + hp] = cmul(cmul(raw, tb[am]), colfac(tbp, amp));
  const cplx mm = cmul(cmul(raw, tbp[am]), colfac(tb, amp));
  As[((size_t)bp * H + h) * lda + (size_t)b * H + hp] = ((am + amp) & 1) ? make_double2(-mm.x, -mm.y) : mm;
}

// ---------------------------------------------------------------------------------------------
// Symmetric fill, systems in lanes (batches of >= 32 systems): lane = system, the wave walks the unit pairs of its chunk one
// after the other.  Every lane of a wave then runs the SAME term list: the coefficient and the table index of a term are
// wave-uniform (broadcast LDS reads), the pair-table row T[l][0..63] of the 64 systems is one contiguous 1-KiB load (tables
// stored Tt[pair][l][system]), all lanes have the same trip counts (the entry-per-lane form above runs each wave to its longest
// list: 59 % lane efficiency at n_end = 20) and no LDS gather, so no bank conflicts (58 % of its LDS cycles).  The pair tables
// are not staged in LDS at all (they are served by L1 / L2), which also removes the H2 ceiling of the LDS budget.
// A lane writes its 2 x 2 block into its own system's matrix: 16-byte stores 64 systems apart; consecutive unit pairs of a
// chunk are consecutive columns, so a 128-byte line of every system fills up within a few hundred cycles and merges in L2.
// q factors per (ball, degree, system) come transposed as well (Qt[b][n][system], k_qfactors_t).
// ---------------------------------------------------------------------------------------------
constexpr int FILL_SYS_THREADS = 256;

__global__ void k_qfactors_t(int n_end, int B, int nb, int nbp, const cplx* __restrict__ tab, cplx* __restrict__ Qt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;            // (b, n, system), system fastest
  if (i >= B * n_end * nbp) return;
  // i = ((g * B + b) * n_end + n) * 64 + lane
  const int lane = i & 63, r = i >> 6, n = r % n_end, gb = r / n_end, b = gb % B, g = gb / B;
  const int s = g * 64 + lane, sc = s < nb ? s : nb - 1;
  const cplx* tb = tab + ((size_t)sc * B + b) * 3 * n_end;
  Qt[i] = cmul(tb[n], crecip(zsqrt(cmul(tb[n], tb[n_end + n]))));    // gj / sqrt(gj gh)
}

__global__ void __launch_bounds__(FILL_SYS_THREADS) k_fill_sys(int H, int U, int H2, int n_end, int B, int nb, int nbp, int npairs,
                                                                const int* __restrict__ deg, const int* __restrict__ units,
                                                                const int* __restrict__ spos, const int* __restrict__ schunk,
                                                                int terms_max, int pairs_max, const uint32_t* __restrict__ qptr,
                                                                const double* __restrict__ qcoef, const uint16_t* __restrict__ qidx,
                                                                const cplx* __restrict__ Tt, const cplx* __restrict__ Qt,
                                                                cplx* __restrict__ A, long long lda, long long sys_stride, int abl_nostore) {
  extern __shared__ char smem[];
  double* sCoef = (double*)smem;                                    // [terms_max + 1]: the chunk's terms, then a dummy (0.0, row 0)
  uint32_t* sOff = (uint32_t*)(sCoef + terms_max + 1);              // [terms_max + 1]: row offset of the term's table entry, in elements
  uint32_t* sPtr = sOff + terms_max + 1;                            // [4 pairs_max + 1]
  int* sMeta = (int*)(sPtr + 4 * pairs_max + 1);                    // [3 pairs_max]: slots and degrees of the pair's units
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p0 = schunk[blockIdx.x], p1 = schunk[blockIdx.x + 1], npr = p1 - p0;
  const uint32_t t0 = qptr[4 * (size_t)p0], t1 = qptr[4 * (size_t)p1];
  for (uint32_t q = t0 + tid; q < t1; q += FILL_SYS_THREADS) { sCoef[q - t0] = qcoef[q]; sOff[q - t0] = (uint32_t)qidx[q] * 64u; }
  for (int e = tid; e <= 4 * npr; e += FILL_SYS_THREADS) sPtr[e] = qptr[4 * (size_t)p0 + e] - t0;
  if (tid == 0) { sCoef[t1 - t0] = 0.0; sOff[t1 - t0] = 0; }
  for (int e = tid; e < npr; e += FILL_SYS_THREADS) {
    const int pi = p0 + e, u = pi / U, v = pi - u * U;
    const int rh = units[2 * u], rp = units[2 * u + 1], ch = units[2 * v], cp = units[2 * v + 1];
    sMeta[3 * e] = u | ((rp != rh ? U + spos[u] : 0xffff) << 16);          // row slots: cosine, sine (0xffff: none)
    sMeta[3 * e + 1] = v | ((cp != ch ? U + spos[v] : 0xffff) << 16);      // column slots
    sMeta[3 * e + 2] = deg[rh] | (deg[ch] << 16);
  }
  __syncthreads();
  const double q2 = 0.70710678118654752440;
  const int ng = nbp >> 6;
  const int ncomb = npairs * ng;
  for (int comb = blockIdx.y; comb < ncomb; comb += gridDim.y) {
    const int g = comb / npairs, pr = comb - g * npairs;
    int bp = (int)((sqrtf(8.0f * (float)pr + 1.0f) + 1.0f) * 0.5f);
    while (bp * (bp - 1) / 2 > pr) --bp;
    while ((bp + 1) * bp / 2 <= pr) ++bp;
    const int b = pr - bp * (bp - 1) / 2;                              // row ball b < column ball bp
    const int sys = g * 64 + lane;
    const bool live = sys < nb;
    const cplx* __restrict__ Tl = Tt + ((size_t)g * npairs + pr) * H2 * 64 + lane;    // + row offset of a term
    const cplx* __restrict__ Qr = Qt + ((size_t)g * B + b) * n_end * 64 + lane;
    const cplx* __restrict__ Qc = Qt + ((size_t)g * B + bp) * n_end * 64 + lane;
    cplx* As = A + (size_t)(live ? sys : 0) * sys_stride;
    for (int e = wave; e < npr; e += FILL_SYS_THREADS / 64) {
      const uint32_t a0 = sPtr[4 * e], a1 = sPtr[4 * e + 1], a2 = sPtr[4 * e + 2], a3 = sPtr[4 * e + 3], a4 = sPtr[4 * e + 4];
      const uint32_t l0 = a1 - a0, l1 = a2 - a1, l2 = a3 - a2, l3 = a4 - a3;
      uint32_t lm = l0 > l1 ? l0 : l1; lm = lm > l2 ? lm : l2; lm = lm > l3 ? lm : l3;
      double s0r = 0, s0i = 0, s1r = 0, s1i = 0, s2r = 0, s2i = 0, s3r = 0, s3i = 0;
      // all bounds are wave-uniform.  Per step the four chains' coefficient / offset reads are issued, then the four table loads,
      // then the FMAs (one `if (i < len)` block per chain made hipcc wait for each load in turn: 175 ms per 256 systems);
      // a chain that has run out takes the chunk's dummy term (coefficient 0, row 0).  Two steps per trip: eight loads in flight.
      const uint32_t dummy = sPtr[4 * npr];
      uint32_t q0 = a0, q1 = a1, q2p = a2, q3 = a3;
      for (uint32_t i = 0; i < lm; i += 2) {
        const uint32_t g0 = q0 < a1 ? q0 : dummy, g1 = q1 < a2 ? q1 : dummy, g2 = q2p < a3 ? q2p : dummy, g3 = q3 < a4 ? q3 : dummy;
        const uint32_t h0 = q0 + 1 < a1 ? q0 + 1 : dummy, h1 = q1 + 1 < a2 ? q1 + 1 : dummy, h2 = q2p + 1 < a3 ? q2p + 1 : dummy, h3 = q3 + 1 < a4 ? q3 + 1 : dummy;
        const double c0 = sCoef[g0], c1 = sCoef[g1], c2v = sCoef[g2], c3 = sCoef[g3];
        const double d0 = sCoef[h0], d1 = sCoef[h1], d2v = sCoef[h2], d3 = sCoef[h3];
        const uint32_t o0 = sOff[g0], o1 = sOff[g1], o2 = sOff[g2], o3 = sOff[g3];
        const uint32_t r0 = sOff[h0], r1 = sOff[h1], r2o = sOff[h2], r3 = sOff[h3];
        const cplx z0 = Tl[o0], z1 = Tl[o1], z2 = Tl[o2], z3 = Tl[o3];
        const cplx w0 = Tl[r0], w1 = Tl[r1], w2 = Tl[r2o], w3 = Tl[r3];
        s0r = fma(c0, z0.x, s0r); s0i = fma(c0, z0.y, s0i);
        s1r = fma(c1, z1.x, s1r); s1i = fma(c1, z1.y, s1i);
        s2r = fma(c2v, z2.x, s2r); s2i = fma(c2v, z2.y, s2i);
        s3r = fma(c3, z3.x, s3r); s3i = fma(c3, z3.y, s3i);
        s0r = fma(d0, w0.x, s0r); s0i = fma(d0, w0.y, s0i);
        s1r = fma(d1, w1.x, s1r); s1i = fma(d1, w1.y, s1i);
        s2r = fma(d2v, w2.x, s2r); s2i = fma(d2v, w2.y, s2i);
        s3r = fma(d3, w3.x, s3r); s3i = fma(d3, w3.y, s3i);
        q0 += 2; q1 += 2; q2p += 2; q3 += 2;
      }
      const int mr = sMeta[3 * e], mc = sMeta[3 * e + 1], md = sMeta[3 * e + 2];
      const int row_c = mr & 0xffff, row_s = (mr >> 16) & 0xffff, col_c = mc & 0xffff, col_s = (mc >> 16) & 0xffff;
      const bool r2 = row_s != 0xffff, c2 = col_s != 0xffff;
      cplx x00 = make_double2(s0r, s0i), x01 = make_double2(s1r, s1i), x10 = make_double2(s2r, s2i), x11 = make_double2(s3r, s3i);
      if (r2) {
        const cplx a0c = make_double2((x00.x + x10.x) * q2, (x00.y + x10.y) * q2), a1c = make_double2((x01.x + x11.x) * q2, (x01.y + x11.y) * q2);
        const cplx d0 = make_double2((x00.x - x10.x) * q2, (x00.y - x10.y) * q2), d1 = make_double2((x01.x - x11.x) * q2, (x01.y - x11.y) * q2);
        x00 = a0c; x01 = a1c; x10 = make_double2(-d0.y, d0.x); x11 = make_double2(-d1.y, d1.x);
      }
      if (c2) {
        const cplx a0c = make_double2((x00.x + x01.x) * q2, (x00.y + x01.y) * q2), d0 = make_double2((x01.x - x00.x) * q2, (x01.y - x00.y) * q2);
        const cplx a1c = make_double2((x10.x + x11.x) * q2, (x10.y + x11.y) * q2), d1 = make_double2((x11.x - x10.x) * q2, (x11.y - x10.y) * q2);
        x00 = a0c; x01 = make_double2(-d0.y, d0.x); x10 = a1c; x11 = make_double2(-d1.y, d1.x);
      }
      const cplx scale = cmul(Qr[(md & 0xffff) * 64], Qc[(md >> 16) * 64]);
      if (live && !abl_nostore) {
        auto put = [&](int rslot, int cslot, cplx val) {
          const int row = b * H + rslot, col = bp * H + cslot;        // b < bp: strictly above the diagonal
          const cplx w = cmul(val, scale);
          As[(size_t)row * lda + col] = w;
          if ((row >> 6) == (col >> 6)) As[(size_t)col * lda + row] = w;
        };
        put(row_c, col_c, x00);
        if (c2) put(row_c, col_s, x01);
        if (r2) { put(row_s, col_c, x10); if (c2) put(row_s, col_s, x11); }
      }
    }
  }
}

// what k_fill_sym leaves: the identity diagonal blocks and the padding, again only where the factorisation reads
// (columns c >= 64 (r / 64) of row r); one workgroup per row
__global__ void __launch_bounds__(64) k_fill_sym_diag(int H, int N, int n_pad, cplx* __restrict__ A, long long lda, long long sys_stride, int no_pad) {
  const int r = blockIdx.x, s = blockIdx.y;
  cplx* row = A + (size_t)s * sys_stride + (size_t)r * lda;
  const int c0 = (r / 64) * 64;
  auto put = [&](int c) { row[c] = (r == c) ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0); };
  if (r >= N) { for (int c = c0 + threadIdx.x; c < n_pad; c += 64) put(c); return; }
  const int br = r / H, bend = (br + 1) * H;              // this row's own ball: columns [br H, bend)
  for (int c = (c0 > br * H ? c0 : br * H) + threadIdx.x; c < bend; c += 64) put(c);
  if (!no_pad) for (int c = N + threadIdx.x; c < n_pad; c += 64) put(c);
}

// bytes of one system the symmetric path writes and the factorisation reads: sum over rows of 16 * 64 (r / 64 + 1)
double fill_sym_bytes(int n_pad) {
  const double t = n_pad / 64;
  return 16.0 * 64.0 * 64.0 * t * (t + 1.0) * 0.5;
}

int launch_fill_sym(const biem_plan* p, int nb, int B, const double* d_k, const double* d_centers, int geom_batched, const double* d_tab,
                    double* d_A, long long lda, long long sys_stride, int n_pad, void* d_work, size_t work_bytes, hipStream_t st, bool no_padding,
                    const FillDedupe* dedupe) {
  const int H = p->H, N = B * H, U = (int)(p->units.size() / 2);
  if (nb <= 0 || B <= 0) return BIEM_OK;
  if (lda < n_pad || n_pad < N || n_pad % 64) { set_error("biem_fill (symmetric): lda / n_pad too small or n_pad not a multiple of 64"); return BIEM_ERR_ARG; }
  if (work_bytes < fill_workspace_bytes(p, nb, B)) { set_error("biem_fill: workspace too small"); return BIEM_ERR_ARG; }
  if (nb > 65535) { set_error("biem_fill (symmetric): at most 65535 systems per call"); return BIEM_ERR_ARG; }
  cplx* T = (cplx*)d_work;
  ProfScope ps(PK_FILL, st, (double)nb * fill_sym_bytes(n_pad));
  const bool direct2d = p->tree == TREE_A && (!p->lists_built || !getenv("BIEM_FILL_FORM"));
  if (!direct2d && 2 * p->n_end > kMaxRad) { set_error("n_end=%d exceeds the built table size", p->n_end); return BIEM_ERR_UNSUPPORTED; }
  if (direct2d && B > 1) {
    // 2-D: the list-free kernel (any order); pair classes as in the list forms
    if (!plan_2d_direct(p)) { set_error("biem_fill (symmetric): internal: 2-D table order"); return BIEM_ERR_ARG; }
    const int npairs = B * (B - 1) / 2;
    int* classes = (int*)((char*)d_work + fill_workspace_bytes(p, nb, B) - fill_dedupe_bytes(B));
    const char* mn = getenv("BIEM_FILL_DEDUPE_MIN");
    const bool on = dedupe && !geom_batched && nb >= (mn ? atoi(mn) : 8) && npairs <= kDedupeMaxPairs && B <= 65535 && !getenv("BIEM_FILL_NO_DEDUPE");
    const size_t shm_dd = on ? (size_t)(B + 3 * npairs + 2) * sizeof(int) + (size_t)npairs * p->d * sizeof(double) : 0;
    if (shm_dd > 48 * 1024) BIEM_HIPCHK(hipFuncSetAttribute((const void*)k_pair_dedupe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_dd));
    hipLaunchKernelGGL(k_pair_dedupe, dim3(1), dim3(256), shm_dd, st, B, p->d, npairs, d_centers,
                       on ? dedupe->radii : nullptr, on ? dedupe->alpha : nullptr, on ? dedupe->beta : nullptr, on ? 1 : 0, classes);
    BIEM_LAUNCHCHK();
    // (the pair tables of the class heads only: the serial radial recurrence of a pair's table is the expensive part of a 2-D fill)
    { const int rc = launch_pair_tables_red(p, nb, B, d_k, d_centers, geom_batched, d_tab, T, st, classes + 4 + 3 * npairs + 1); if (rc) return rc; }
    const long long ncomb = (long long)npairs * nb;
    const unsigned gx = (unsigned)(((long long)U * U + 255) / 256);
    long long gy = ncomb < 65535 ? ncomb : 65535;
    hipLaunchKernelGGL(k_fill2d_sym, dim3(gx, (unsigned)gy), dim3(256), 0, st, p->n_end, H, B, nb, npairs, T, (cplx*)d_A, lda, sys_stride, classes);
    BIEM_LAUNCHCHK();
  }
  // Two forms.  "entry" (one unit pair per lane, pair table in LDS) is the fast one wherever its tables fit LDS; "sys" (one
  // system per lane, pair tables from L2 / Infinity Cache: bound by the ~35-70 GB/s a CU gets from there, 150 vs 92 ms per 256
  // systems at cfg 3) has no ceiling on the order and takes over where the entry form does not fit.  BIEM_FILL_FORM forces one.
  const char* form = getenv("BIEM_FILL_FORM");
  size_t shm_entry = 0, shm_red = 0;
  // BIEM_FILL_FORM: "red" (default where it fits) = reduced-table entry form k_fill_red, "entry" = the gather form k_fill_sym it
  // replaced (kept for A/B runs), "sys" = systems in lanes
  const bool red_fits = fill_red_fits(p, &shm_red);
  const bool use_red = red_fits && !(form && (form[0] == 's' || form[0] == 'e'));
  const bool entry_fits = use_red || fill_sym_entry_fits(p, &shm_entry);
  const bool sys_form = form ? (form[0] == 's') : !entry_fits;
  if (!direct2d && B > 1 && sys_form) {
    // systems in lanes.  Workspace: Tt[groups][npairs][H2][64] then Qt[groups][B][n_end][64] (fill_workspace_bytes covers it)
    const int nbp = (nb + 63) / 64 * 64, npairs = B * (B - 1) / 2;
    const size_t need = ((size_t)npairs * p->H2 + (size_t)B * p->n_end) * nbp * sizeof(cplx);
    if (need > work_bytes) { set_error("biem_fill (symmetric, systems in lanes): workspace too small for %d systems", nb); return BIEM_ERR_ARG; }
    cplx* Qt = T + (size_t)npairs * p->H2 * nbp;
    const size_t shm = (size_t)(p->schunk_terms_max + 1) * 12 + (size_t)(4 * p->schunk_pairs_max + 1) * 4 + (size_t)p->schunk_pairs_max * 12 + 16;
    if (p->H2 > 65536) { set_error("biem_fill (symmetric, systems in lanes): n_end=%d has %d table labels, the 16-bit term indices hold 65536", p->n_end, p->H2); return BIEM_ERR_UNSUPPORTED; }
    if (shm > 64 * 1024 || (size_t)p->H2 * 64 >= (1ull << 32)) { set_error("biem_fill (symmetric, systems in lanes): a unit pair of n_end=%d has %d terms", p->n_end, p->schunk_terms_max); return BIEM_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(k_pair_tables, dim3(B * B, nb), dim3(64), 0, st, p->tree, p->d, p->n2, p->H2, p->Cd, p->d_labels2, p->d_deg2, B,
                       (const cplx*)d_k, d_centers, geom_batched, T, 0, nbp, nullptr, 0);
    const int nq = B * p->n_end * nbp;
    hipLaunchKernelGGL(k_qfactors_t, dim3((nq + 255) / 256), dim3(256), 0, st, p->n_end, B, nb, nbp, (const cplx*)d_tab, Qt);
    BIEM_LAUNCHCHK();
    const int nchunks = (int)p->schunk.size() - 1;
    const long long ncomb = (long long)npairs * (nbp / 64);
    long long gy = (16 * 256 + nchunks - 1) / nchunks;            // about 16 workgroups per CU in all, each looping over combinations
    { const char* e = getenv("BIEM_FILL_GY"); if (e && atoi(e) > 0) gy = atoi(e); }
    if (gy < 1) gy = 1;
    if (gy > ncomb) gy = ncomb;
    if (gy > 65535) gy = 65535;
    BIEM_HIPCHK(hipFuncSetAttribute((const void*)k_fill_sys, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(k_fill_sys, dim3(nchunks, (unsigned)gy), dim3(FILL_SYS_THREADS), shm, st, H, U, p->H2, p->n_end, B, nb, nbp, npairs, p->d_deg,
                       p->d_units, p->d_spos, p->d_schunk, p->schunk_terms_max, p->schunk_pairs_max, p->d_qptr, p->d_qcoef, p->d_qidx16,
                       (const cplx*)T, (const cplx*)Qt, (cplx*)d_A, lda, sys_stride, getenv("BIEM_ABL_FILL_NOSTORE") ? 1 : 0);
    BIEM_LAUNCHCHK();
  } else if (!direct2d && B > 1) {
    const int nchunks = use_red ? (int)p->rchunk.size() - 1 : (int)p->qchunk.size() - 1;
    const size_t shm = use_red ? shm_red : shm_entry;
    if (!entry_fits) {
      set_error("biem_fill (symmetric, one unit pair per lane): tables do not fit LDS (n_end=%d: H2=%d, chunk terms=%d)", p->n_end, p->H2, p->qchunk_terms_max);
      return BIEM_ERR_UNSUPPORTED;
    }
    const int npairs = B * (B - 1) / 2;
    const long long ncomb = (long long)npairs * nb;
    int* classes = (int*)((char*)d_work + fill_workspace_bytes(p, nb, B) - fill_dedupe_bytes(B));
    {
      // (a handful of systems: fewer, longer combinations would leave most of the chip without a workgroup - one N = 4064 system
      // filled in 0.49 instead of 0.17 ms with classes - so every pair stays on its own below 8 systems)
      const char* mn = getenv("BIEM_FILL_DEDUPE_MIN");                       // (tests: classes for small batches too)
      const bool on = dedupe && !geom_batched && nb >= (mn ? atoi(mn) : 8) && npairs <= kDedupeMaxPairs && B <= 65535 && !getenv("BIEM_FILL_NO_DEDUPE");
      const size_t shm_dd = on ? (size_t)(B + 3 * npairs + 2) * sizeof(int) + (size_t)npairs * p->d * sizeof(double) : 0;
      if (shm_dd > 48 * 1024) BIEM_HIPCHK(hipFuncSetAttribute((const void*)k_pair_dedupe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_dd));
      hipLaunchKernelGGL(k_pair_dedupe, dim3(1), dim3(256), shm_dd, st, B, p->d, npairs, d_centers,
                         on ? dedupe->radii : nullptr, on ? dedupe->alpha : nullptr, on ? dedupe->beta : nullptr, on ? 1 : 0, classes);
      BIEM_LAUNCHCHK();
    }
    // pair tables, of the class heads only (the fill reads no others)
    const int* rep_flag = classes + 4 + 3 * npairs + 1;
    if (use_red) {
      const int rc = launch_pair_tables_red(p, nb, B, d_k, d_centers, geom_batched, d_tab, T, st, rep_flag);
      if (rc) return rc;
    } else {
      hipLaunchKernelGGL(k_pair_tables, dim3(B * B, nb), dim3(64), 0, st, p->tree, p->d, p->n2, p->H2, p->Cd, p->d_labels2, p->d_deg2, B,
                         (const cplx*)d_k, d_centers, geom_batched, T, 0, 0, p->d_lin2, p->H2lin, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, nullptr, rep_flag);
      BIEM_LAUNCHCHK();
    }
    // enough workgroups to fill the chip a few times over, each with a long loop over combinations
    long long gy = (8 * 256 + nchunks - 1) / nchunks;
    if (use_red && p->red_waves == 16) {
      // one workgroup per CU at a time (its lists take the CU's LDS): the grid runs in rounds of 256 workgroups of about equal work, so a
      // grid of 8.1 rounds costs 9.  Among 6 .. 10 rounds' worth of workgroups take the count that wastes least of its last round
      // (cfg 3: 77 chunks x 27 = 8.12 rounds -> x 26 = 7.82: -10 % fill time)
      static int ncu_of[64] = {0};               // CUs per device, queried once
      int devid = 0, ncu = 256;
      if (hipGetDevice(&devid) == hipSuccess && devid >= 0 && devid < 64) {
        if (ncu_of[devid] == 0) { int v = 0; ncu_of[devid] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, devid) == hipSuccess && v > 0) ? v : 256; }
        ncu = ncu_of[devid];
      }
      double best = 1e30;
      for (long long g = (6LL * ncu + nchunks - 1) / nchunks; g <= (10LL * ncu) / nchunks + 1; ++g) {
        if (g < 1) continue;
        const double wgs = (double)nchunks * g, rounds = ceil(wgs / ncu);
        const double waste = rounds * ncu / wgs;
        if (waste < best - 1e-9 || (waste < best + 1e-9 && g > gy)) { best = waste; gy = g; }
      }
    }
    if (gy < 1) gy = 1;
    if (gy > ncomb) gy = ncomb;
    if (gy > 65535) gy = 65535;
    const int HR = p->E + p->NP + 2 * p->n_end;
    const int red_threads = 64 * p->red_waves;
    const int kt_need = use_red ? (HR + red_threads - 1) / red_threads : (p->H2lin + FILL_SYM_THREADS - 1) / FILL_SYM_THREADS;
#define BIEM_LAUNCH_FILL_RED(KT, NC)                                                                                                      \
  {                                                                                                                                       \
    BIEM_HIPCHK(hipFuncSetAttribute((const void*)k_fill_red<KT, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));              \
    hipLaunchKernelGGL((k_fill_red<KT, NC>), dim3(nchunks, (unsigned)gy), dim3(red_threads), shm, st, H, U, HR, p->E, p->NP, p->n_end, B, nb, npairs, \
                       p->d_deg, p->d_units, p->d_spos, p->d_rchunk, p->d_rcrow, p->d_rwrow, p->rchunk_rows_max, p->d_rcoef, p->d_ridx,   \
                       p->d_rphsel, T, (cplx*)d_A, lda, sys_stride, classes);                                                             \
  }
    if (use_red) {
      // two combinations per iteration (the plan reserves LDS for two table rows) unless BIEM_FILL_NC=1 at plan build (A / B runs)
      const bool nc2 = p->red_nc == 2;
      if (nc2) {
        if (kt_need <= 1) BIEM_LAUNCH_FILL_RED(1, 2)
        else if (kt_need <= 2) BIEM_LAUNCH_FILL_RED(2, 2)
        else if (kt_need <= 4) BIEM_LAUNCH_FILL_RED(4, 2)
        else BIEM_LAUNCH_FILL_RED(8, 2)
      } else {
        if (kt_need <= 1) BIEM_LAUNCH_FILL_RED(1, 1)
        else if (kt_need <= 2) BIEM_LAUNCH_FILL_RED(2, 1)
        else if (kt_need <= 4) BIEM_LAUNCH_FILL_RED(4, 1)
        else BIEM_LAUNCH_FILL_RED(8, 1)
      }
    } else
#define BIEM_LAUNCH_FILL_SYM(KT)                                                                                                          \
  {                                                                                                                                       \
    BIEM_HIPCHK(hipFuncSetAttribute((const void*)k_fill_sym<KT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));                  \
    hipLaunchKernelGGL(k_fill_sym<KT>, dim3(nchunks, (unsigned)gy), dim3(FILL_SYM_THREADS), shm, st, H, U, p->H2lin, p->n_end, B, nb, npairs, \
                       p->d_deg, p->d_units, p->d_spos, p->d_qchunk, p->qchunk_terms_max, p->qchunk_pairs_max, p->d_q2ptr, p->d_q2coef,   \
                       p->d_q2idx16, T, (const cplx*)d_tab, (cplx*)d_A, lda, sys_stride, classes);                                        \
  }
    if (kt_need <= 1) BIEM_LAUNCH_FILL_SYM(1)
    else if (kt_need <= 2) BIEM_LAUNCH_FILL_SYM(2)
    else if (kt_need <= 3) BIEM_LAUNCH_FILL_SYM(3)
    else if (kt_need <= 4) BIEM_LAUNCH_FILL_SYM(4)
    else if (kt_need <= 6) BIEM_LAUNCH_FILL_SYM(6)
    else BIEM_LAUNCH_FILL_SYM(8)
#undef BIEM_LAUNCH_FILL_SYM
#undef BIEM_LAUNCH_FILL_RED
    BIEM_LAUNCHCHK();
  }
  // (no_padding: the caller's solver never reads the identity padding - the LDS-resident path of small systems)
  hipLaunchKernelGGL(k_fill_sym_diag, dim3(no_padding ? N : n_pad, nb), dim3(64), 0, st, H, N, n_pad, (cplx*)d_A, lda, sys_stride, no_padding ? 1 : 0);
  BIEM_LAUNCHCHK();
  return BIEM_OK;
}

// ---------------------------------------------------------------------------------------------
// RHS projection  f[row][h] = sum_q g[row][q] W[q][h]   (ush.expand, _biem.py:627-639)
// block = RT rows x 256 columns; g rows staged in LDS, W streamed coalesced along h.
// ---------------------------------------------------------------------------------------------
template <int RT>
__global__ void __launch_bounds__(256) k_rhs_project(int H, int Q, int rows, int B, int nrhs, const cplx* __restrict__ g,
                                                      const cplx* __restrict__ W, cplx* __restrict__ f, long long sys_stride,
                                                      long long elem_stride, long long rhs_stride, const int* __restrict__ hpos) {
  extern __shared__ cplx sg[];   // [RT][QC]
  constexpr int QC = 256;
  int row0 = blockIdx.y * RT;
  int h = blockIdx.x * 256 + threadIdx.x;
  cplx acc[RT];
  for (int r = 0; r < RT; ++r) acc[r] = make_double2(0.0, 0.0);
  for (int q0 = 0; q0 < Q; q0 += QC) {
    int qn = min(QC, Q - q0);
    __syncthreads();
    for (int i = threadIdx.x; i < RT * QC; i += 256) {
      int r = i / QC, q = i % QC;
      sg[i] = (row0 + r < rows && q < qn) ? g[(size_t)(row0 + r) * Q + q0 + q] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    if (h < H) {
      for (int q = 0; q < qn; ++q) {
        cplx w = W[(size_t)(q0 + q) * H + h];
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = cfma(sg[r * QC + q], w, acc[r]);
      }
    }
  }
  if (h < H) {
    for (int r = 0; r < RT; ++r) {
      int row = row0 + r;
      if (row >= rows) break;
      long long b = row % B, sr = row / B, rr = sr % nrhs, s = sr / nrhs;     // row = (s*nrhs + rr)*B + b
      f[(size_t)s * sys_stride + ((size_t)b * H + (hpos ? hpos[h] : h)) * elem_stride + (size_t)rr * rhs_stride] = acc[r];
    }
  }
}

// Few rows (one system per call: B nrhs rows): the form above leaves the whole sum over q to (H / 256) x (rows / RT) workgroups - 4 at cfg 3,
// 435 us.  Here a workgroup takes 16 harmonics and splits the quadrature points over its 16 lane groups (partial sums reduced through
// LDS in a fixed order: deterministic), so H / 16 workgroups per row block share the stream of W.
template <int RT>
__global__ void __launch_bounds__(256) k_rhs_project_few(int H, int Q, int rows, int B, int nrhs, const cplx* __restrict__ g,
                                                          const cplx* __restrict__ W, cplx* __restrict__ f, long long sys_stride,
                                                          long long elem_stride, long long rhs_stride, const int* __restrict__ hpos) {
  extern __shared__ cplx sg[];   // [RT][QC]; afterwards the partial sums [16 slices][RT][16]
  constexpr int QC = 256;
  const int row0 = blockIdx.y * RT;
  const int hl = threadIdx.x & 15, qs = threadIdx.x >> 4;
  const int h = blockIdx.x * 16 + hl, hc = h < H ? h : H - 1;
  cplx acc[RT];
  for (int r = 0; r < RT; ++r) acc[r] = make_double2(0.0, 0.0);
  for (int q0 = 0; q0 < Q; q0 += QC) {
    const int qn = min(QC, Q - q0);
    __syncthreads();
    for (int i = threadIdx.x; i < RT * QC; i += 256) {
      const int r = i / QC, q = i % QC;
      sg[i] = (row0 + r < rows && q < qn) ? g[(size_t)(row0 + r) * Q + q0 + q] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    for (int q = qs; q < qn; q += 16) {
      const cplx w = W[(size_t)(q0 + q) * H + hc];
#pragma unroll
      for (int r = 0; r < RT; ++r) acc[r] = cfma(sg[r * QC + q], w, acc[r]);
    }
  }
  __syncthreads();
  for (int r = 0; r < RT; ++r) sg[(qs * RT + r) * 16 + hl] = acc[r];
  __syncthreads();
  if (threadIdx.x < RT * 16) {
    const int r = threadIdx.x >> 4, row = row0 + r;
    cplx t = make_double2(0.0, 0.0);
    for (int z = 0; z < 16; ++z) { const cplx v = sg[(z * RT + r) * 16 + hl]; t.x += v.x; t.y += v.y; }
    if (row < rows && h < H) {
      long long b = row % B, sr = row / B, rr = sr % nrhs, sy = sr / nrhs;     // row = (s*nrhs + rr)*B + b
      f[(size_t)sy * sys_stride + ((size_t)b * H + (hpos ? hpos[h] : h)) * elem_stride + (size_t)rr * rhs_stride] = t;
    }
  }
}

int launch_rhs_project(const biem_plan* p, int nb, int B, int nrhs, const double* d_g, double* d_f, long long sys_stride,
                       long long elem_stride, long long rhs_stride, hipStream_t st, bool slot_order) {
  if (nrhs < 1) { set_error("biem_rhs_project: nrhs < 1"); return BIEM_ERR_ARG; }
  int rows = nb * nrhs * B;
  if (rows <= 0) return BIEM_OK;
  constexpr int RT = 8;
  size_t shm = (size_t)RT * 256 * sizeof(cplx);
  ProfScope ps(PK_RHS, st, 8.0 * (double)rows * p->Q * p->H);
  if ((long long)((p->H + 255) / 256) * ((rows + RT - 1) / RT) < 128) {
    hipLaunchKernelGGL(k_rhs_project_few<RT>, dim3((p->H + 15) / 16, (rows + RT - 1) / RT), dim3(256), shm, st, p->H, p->Q, rows, B,
                       nrhs, (const cplx*)d_g, (const cplx*)p->d_W, (cplx*)d_f, sys_stride, elem_stride, rhs_stride,
                       slot_order ? p->d_hpos : nullptr);
    BIEM_LAUNCHCHK();
    return BIEM_OK;
  }
  hipLaunchKernelGGL(k_rhs_project<RT>, dim3((p->H + 255) / 256, (rows + RT - 1) / RT), dim3(256), shm, st, p->H, p->Q, rows, B,
                     nrhs, (const cplx*)d_g, (const cplx*)p->d_W, (cplx*)d_f, sys_stride, elem_stride, rhs_stride,
                     slot_order ? p->d_hpos : nullptr);
  BIEM_LAUNCHCHK();
  return BIEM_OK;
}

// ---------------------------------------------------------------------------------------------
// right-hand sides / solutions of the complex-symmetric form (the matrix itself comes from k_fill_sym above):
//   f~ = R W^H f  before the factorisation,   x = W R^-1 x~  after it,   R = diag(1 / sqrt(gj gh)) per (ball, degree).
// The right-hand-side columns hold a ball's harmonics in the internal slot order (biem_rhs_project with the plan's hpos: h of
// unit u in slot u, its partner p in slot U + spos[u]), so both transforms work in place on the two slots of a unit.
// ---------------------------------------------------------------------------------------------
__device__ inline cplx sym_r(const cplx* __restrict__ tball, int n_end, int n) {   // 1 / sqrt(gj gh)
  return crecip(zsqrt(cmul(tball[n], tball[n_end + n])));
}

__global__ void __launch_bounds__(256) k_sym_rhs(int H, int U, int n_end, int B, int nrhs, int n_pad, const int* __restrict__ units,
                                                  const int* __restrict__ spos, const int* __restrict__ deg, const cplx* __restrict__ tab,
                                                  cplx* __restrict__ A, long long lda, long long sys_stride, int inverse) {
  const int s = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;                 // (unit over all balls, rhs)
  if (t >= B * U * nrhs) return;
  const int q = t % nrhs, ru = t / nrhs, br = ru / U, ur = ru - br * U;
  const int rh = units[2 * ur], rp = units[2 * ur + 1];
  const cplx* ts = tab + ((size_t)s * B + br) * 3 * n_end;
  cplx* F = A + (size_t)s * sys_stride + n_pad + q;
  cplx* pc = F + (size_t)(br * H + ur) * lda;
  cplx* psn = F + (size_t)(br * H + U + (rp != rh ? spos[ur] : 0)) * lda;
  const double q2 = 0.70710678118654752440;
  if (!inverse) {
    const cplx r = sym_r(ts, n_end, deg[rh]);
    const cplx fh = *pc;
    if (rp == rh) { *pc = cmul(fh, r); return; }
    const cplx fp = *psn;
    const cplx a = make_double2((fh.x + fp.x) * q2, (fh.y + fp.y) * q2), d = make_double2((fh.x - fp.x) * q2, (fh.y - fp.y) * q2);
    *pc = cmul(a, r);
    *psn = cmul(make_double2(-d.y, d.x), r);                    // i (f_h - f_p) / sqrt2
  } else {
    const cplx g = zsqrt(cmul(ts[deg[rh]], ts[n_end + deg[rh]]));      // 1 / r
    const cplx yh = cmul(*pc, g);
    if (rp == rh) { *pc = yh; return; }
    const cplx yp = cmul(*psn, g);                              // x_h = (y_c - i y_s)/sqrt2, x_p = (y_c + i y_s)/sqrt2
    *pc = make_double2((yh.x + yp.y) * q2, (yh.y - yp.x) * q2);
    *psn = make_double2((yh.x - yp.y) * q2, (yh.y + yp.x) * q2);
  }
}

int launch_sym_rhs(const biem_plan* p, int nb, int B, int nrhs, int n_pad, const double* d_tab, double* d_A, long long lda,
                   long long sys_stride, bool inverse_on_solution, hipStream_t st) {
  const int U = (int)(p->units.size() / 2);
  if (nb <= 0 || B <= 0 || nrhs <= 0) return BIEM_OK;
  if (nb > 65535) { set_error("biem symmetric path: at most 65535 systems per call"); return BIEM_ERR_ARG; }
  ProfScope ps(PK_SWAP, st, 0.0);   // class 4: row interchanges in the LU, these transforms in the symmetric path
  hipLaunchKernelGGL(k_sym_rhs, dim3((B * U * nrhs + 255) / 256, nb), dim3(256), 0, st, p->H, U, p->n_end, B, nrhs, n_pad, p->d_units,
                     p->d_spos, p->d_deg, (const cplx*)d_tab, (cplx*)d_A, lda, sys_stride, inverse_on_solution ? 1 : 0);
  BIEM_LAUNCHCHK();
  return BIEM_OK;
}

// ---------------------------------------------------------------------------------------------
// density = x / (gh * blc)   (reference scaling of the unknown; single-ball shortcut _biem.py:673-690 with x = f)
// ---------------------------------------------------------------------------------------------
__global__ void k_density(int H, int n_end, int B, int nrhs, long long total, const int* __restrict__ deg, const cplx* __restrict__ x,
                          long long sys_stride, long long elem_stride, long long rhs_stride, const cplx* __restrict__ tab,
                          cplx* __restrict__ dens, const int* __restrict__ hpos) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // i = ((s*nrhs + r)*B + b)*H + h
  if (i >= total) return;
  int h = (int)(i % H);
  long long srb = i / H;
  long long b = srb % B, sr = srb / B, r = sr % nrhs, s = sr / nrhs;
  int n = deg[h];
  const cplx* t = tab + (size_t)(s * B + b) * 3 * n_end;
  cplx v = x[(size_t)s * sys_stride + ((size_t)b * H + (hpos ? hpos[h] : h)) * elem_stride + (size_t)r * rhs_stride];
  dens[i] = cmul(v, crecip(cmul(t[n_end + n], t[2 * n_end + n])));
}

int launch_density(const biem_plan* p, int nb, int B, int nrhs, const double* d_x, long long sys_stride, long long elem_stride,
                   long long rhs_stride, const double* d_tab, double* d_density, hipStream_t st, bool slot_order) {
  if (nrhs < 1) { set_error("biem_density: nrhs < 1"); return BIEM_ERR_ARG; }
  long long total = (long long)nb * nrhs * B * p->H;
  if (total <= 0) return BIEM_OK;
  hipLaunchKernelGGL(k_density, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p->H, p->n_end, B, nrhs, total, p->d_deg,
                     (const cplx*)d_x, sys_stride, elem_stride, rhs_stride, (const cplx*)d_tab, (cplx*)d_density, slot_order ? p->d_hpos : nullptr);
  BIEM_LAUNCHCHK();
  return BIEM_OK;
}

}  // namespace biem
