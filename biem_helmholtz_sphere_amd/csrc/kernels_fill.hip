// kernels_fill.hip -- special-function tables, matrix fill, RHS projection, density scaling (gfx950).
//
// Replaces the span _biem.py:627-639 (RHS) and _biem.py:694-792 (matrix) of the reference, which there is
// ~40 array-API ops and 3-5 full-size temporaries; here every matrix element is written exactly once.
#include "common.hpp"

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
                              const cplx* __restrict__ beta, int ab_batched, cplx* __restrict__ tab) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb * B) return;
  int s = i / B, b = i % B;
  const cplx kk = k[s];
  const double et = eta[s];
  double rho = radii[(geom_batched ? (size_t)s * B : 0) + b];
  cplx al = alpha[(ab_batched ? (size_t)s * B : 0) + b];
  cplx be = beta[(ab_batched ? (size_t)s * B : 0) + b];
  cplx J[kMaxRad + 3], Hh[kMaxRad + 3];
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
  if (p->n_end > kMaxRad) { set_error("n_end=%d exceeds the built table size %d", p->n_end, kMaxRad); return BIEM_ERR_UNSUPPORTED; }
  int total = nb * B;
  if (total <= 0) return BIEM_OK;
  ProfScope ps(PK_TABLES, st);
  hipLaunchKernelGGL(k_ball_tables, dim3((total + 63) / 64), dim3(64), 0, st, p->d, p->n_end, nb, B, (const cplx*)d_k, d_eta, d_radii,
                     geom_batched, (const cplx*)d_alpha, (const cplx*)d_beta, ab_batched, (cplx*)d_tab);
  BIEM_LAUNCHCHK();
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
                                                     const double* __restrict__ centers, int geom_batched, cplx* __restrict__ T) {
  __shared__ cplx sJ[kMaxRad * 2 + 6];
  __shared__ cplx sH[kMaxRad * 2 + 6];
  int pair = blockIdx.x, s = blockIdx.y;
  int b = pair / B, bp = pair % B;
  if (b >= bp) return;                       // the fill derives block (bp, b) from (b, bp)
  const double* cb = centers + ((geom_batched ? (size_t)s * B : 0) + b) * d;
  const double* cp = centers + ((geom_batched ? (size_t)s * B : 0) + bp) * d;
  double t[4];
  double r2 = 0.0;
  for (int i = 0; i < d; ++i) { t[i] = cb[i] - cp[i]; r2 += t[i] * t[i]; }
  double r = sqrt(r2);
  if (threadIdx.x == 0) radial_jh(d, n2 - 1, cscale(k[s], r), sJ, sH);
  __syncthreads();
  Dir dir = make_dir(tree, t);
  cplx* out = T + ((size_t)s * B * B + pair) * H2;
  for (int l = threadIdx.x; l < H2; l += 64) {
    double re, im;
    harmonic_single(tree, labels2[3 * l], labels2[3 * l + 1], labels2[3 * l + 2], dir, &re, &im);
    int n = deg2[l];
    out[l] = cmul(cscale(sH[n], Cd), make_double2(re, im));
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
                                                        long long lda, long long sys_stride) {
  extern __shared__ char smem[];
  cplx* sT = (cplx*)smem;                               // [H2] pair table T_{b,bp}
  cplx* sC = sT + H2;                                   // [H] column factors of the partner ball bp
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
      for (int l = tid; l < H2; l += FILL_THREADS) sT[l] = Tp[l];
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

size_t fill_workspace_bytes(const biem_plan* p, int nb, int B) { return (size_t)nb * B * B * p->H2 * sizeof(cplx); }

int launch_fill(const biem_plan* p, int nb, int B, const double* d_k, const double* d_centers, int geom_batched,
                const double* d_tab, int scaling, double* d_A, long long lda, long long sys_stride, int n_pad,
                void* d_work, size_t work_bytes, hipStream_t st) {
  const int H = p->H, N = B * H;
  if (nb <= 0 || B <= 0) return BIEM_OK;
  if (lda < (n_pad > N ? n_pad : N) || n_pad < N) { set_error("biem_fill: lda/n_pad too small"); return BIEM_ERR_ARG; }
  if (work_bytes < fill_workspace_bytes(p, nb, B)) { set_error("biem_fill: workspace too small"); return BIEM_ERR_ARG; }
  if (2 * p->n_end > kMaxRad) { set_error("n_end=%d exceeds the built table size", p->n_end); return BIEM_ERR_UNSUPPORTED; }
  if (scaling != BIEM_FILL_REFERENCE && scaling != BIEM_FILL_EQUILIBRATED) { set_error("biem_fill: bad scaling"); return BIEM_ERR_ARG; }
  cplx* T = (cplx*)d_work;
  ProfScope ps(PK_FILL, st, 16.0 * (double)nb * N * (double)N);
  if (B > 1) {
    hipLaunchKernelGGL(k_pair_tables, dim3(B * B, nb), dim3(64), 0, st, p->tree, p->d, p->n2, p->H2, p->Cd, p->d_labels2,
                       p->d_deg2, B, (const cplx*)d_k, d_centers, geom_batched, T);
    BIEM_LAUNCHCHK();
  }
  size_t shm = (size_t)(p->H2 + 2 * H) * sizeof(cplx) + (size_t)p->chunk_terms_max * 10 + (size_t)(p->chunk_ents_max + 1) * 4 + 16;
  if (shm > 160 * 1024 || (p->chunk_terms_max == 0 && p->coef.size() > 0)) {
    set_error("biem_fill: tables do not fit LDS (H2=%d, chunk terms=%d)", p->H2, p->chunk_terms_max);
    return BIEM_ERR_UNSUPPORTED;
  }
  BIEM_HIPCHK(hipFuncSetAttribute((const void*)k_fill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  const int nchunks = (int)p->chunk_ent.size() - 1;
  hipLaunchKernelGGL(k_fill, dim3(nchunks, (B + 1) / 2, nb), dim3(FILL_THREADS), shm, st, H, p->H2, p->n_end, B, p->d_deg, p->d_chunk_ent,
                     p->chunk_terms_max, p->chunk_ents_max, p->d_ptr, p->d_coef, p->d_tidx16, T, (const cplx*)d_tab, scaling,
                     (cplx*)d_A, lda, sys_stride);
  BIEM_LAUNCHCHK();
  if (n_pad > N) {
    hipLaunchKernelGGL(k_fill_pad, dim3(64, nb), dim3(256), 0, st, N, n_pad, (cplx*)d_A, lda, sys_stride);
    BIEM_LAUNCHCHK();
  }
  return BIEM_OK;
}

// ---------------------------------------------------------------------------------------------
// RHS projection  f[row][h] = sum_q g[row][q] W[q][h]   (ush.expand, _biem.py:627-639)
// block = RT rows x 256 columns; g rows staged in LDS, W streamed coalesced along h.
// ---------------------------------------------------------------------------------------------
template <int RT>
__global__ void __launch_bounds__(256) k_rhs_project(int H, int Q, int rows, int B, int nrhs, const cplx* __restrict__ g,
                                                      const cplx* __restrict__ W, cplx* __restrict__ f, long long sys_stride,
                                                      long long elem_stride, long long rhs_stride) {
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
      f[(size_t)s * sys_stride + ((size_t)b * H + h) * elem_stride + (size_t)rr * rhs_stride] = acc[r];
    }
  }
}

int launch_rhs_project(const biem_plan* p, int nb, int B, int nrhs, const double* d_g, double* d_f, long long sys_stride,
                       long long elem_stride, long long rhs_stride, hipStream_t st) {
  if (nrhs < 1) { set_error("biem_rhs_project: nrhs < 1"); return BIEM_ERR_ARG; }
  int rows = nb * nrhs * B;
  if (rows <= 0) return BIEM_OK;
  constexpr int RT = 8;
  size_t shm = (size_t)RT * 256 * sizeof(cplx);
  ProfScope ps(PK_RHS, st, 8.0 * (double)rows * p->Q * p->H);
  hipLaunchKernelGGL(k_rhs_project<RT>, dim3((p->H + 255) / 256, (rows + RT - 1) / RT), dim3(256), shm, st, p->H, p->Q, rows, B,
                     nrhs, (const cplx*)d_g, (const cplx*)p->d_W, (cplx*)d_f, sys_stride, elem_stride, rhs_stride);
  BIEM_LAUNCHCHK();
  return BIEM_OK;
}

// ---------------------------------------------------------------------------------------------
// complex-symmetric form of the equilibrated system (for the L D L^T path).
// With W the unitary map to real harmonics (pairs h <= p, conj Y_h = Y_p:  W e_h = (e_h + e_p)/sqrt2,  W e_p = i (e_p - e_h)/sqrt2)
// and R = diag(1 / sqrt(gj gh)) per (ball, degree),   A~ = R W^H M W R^{-1}   is complex symmetric  (G P symmetric, see
// tests/test_oracle_golden.py::test_translation_block_matrix_is_complex_symmetric_up_to_conjugate_pairing), f~ = R W^H f,
// and the solution of the original system is x = W R^{-1} x~.
// k_symmetrize: one thread per (row unit, column unit) 2 x 2 block of [M | F], in place; right-hand-side columns take the row
// transform only.  k_unsymmetrize: x from x~ in the right-hand-side columns.
// ---------------------------------------------------------------------------------------------
__device__ inline cplx sym_r(const cplx* __restrict__ tball, int n_end, int n) {   // 1 / sqrt(gj gh)
  return crecip(zsqrt(cmul(tball[n], tball[n_end + n])));
}

__global__ void __launch_bounds__(256) k_symmetrize(int H, int U, int n_end, int B, int nrhs, int n_pad, const int* __restrict__ units,
                                                     const int* __restrict__ deg, const cplx* __restrict__ tab, cplx* __restrict__ A,
                                                     long long lda, long long sys_stride) {
  const int s = blockIdx.z;
  const int cu = blockIdx.x * 256 + threadIdx.x;                // column unit over all balls, then the right-hand sides
  if (cu >= B * U + nrhs) return;
  for (int ru = blockIdx.y; ru < B * U; ru += gridDim.y) {      // row unit over all balls (grid.y is capped at 65535)
    const int br = ru / U, ur = ru - br * U;
    const int rh = units[2 * ur], rp = units[2 * ur + 1];
    const cplx* ts = tab + (size_t)s * B * 3 * n_end;
    const cplx rr = sym_r(ts + (size_t)br * 3 * n_end, n_end, deg[rh]);
    cplx* As = A + (size_t)s * sys_stride;
    cplx* row_h = As + (size_t)(br * H + rh) * lda;
    cplx* row_p = As + (size_t)(br * H + rp) * lda;
    const double q2 = 0.70710678118654752440;
    int ch, cp; cplx scale;
    if (cu < B * U) {
      const int bc = cu / U, uc = cu - bc * U;
      ch = bc * H + units[2 * uc]; cp = bc * H + units[2 * uc + 1];
      scale = cmul(rr, zsqrt(cmul(ts[(size_t)bc * 3 * n_end + deg[units[2 * uc]]], ts[(size_t)bc * 3 * n_end + n_end + deg[units[2 * uc]]])));   // r_row / r_col
    } else { ch = cp = n_pad + (cu - B * U); scale = rr; }
    // the L D L^T factorisation reads only the lower triangle and the diagonal 64 x 64 blocks: blocks wholly right of their
    // rows' diagonal blocks are neither transformed nor written (their memory keeps the untransformed M; never read)
    if (cu < B * U && (ch < cp ? ch : cp) >= ((br * H + (rh > rp ? rh : rp)) / 64 + 1) * 64) continue;
    cplx x00 = row_h[ch], x01 = row_h[cp], x10 = x00, x11 = x01;
    if (rp != rh) { x10 = row_p[ch]; x11 = row_p[cp]; }
    if (rp != rh) {   // rows: (h + p)/sqrt2, i (h - p)/sqrt2
      cplx a0 = make_double2((x00.x + x10.x) * q2, (x00.y + x10.y) * q2), a1 = make_double2((x01.x + x11.x) * q2, (x01.y + x11.y) * q2);
      cplx d0 = make_double2((x00.x - x10.x) * q2, (x00.y - x10.y) * q2), d1 = make_double2((x01.x - x11.x) * q2, (x01.y - x11.y) * q2);
      x00 = a0; x01 = a1; x10 = make_double2(-d0.y, d0.x); x11 = make_double2(-d1.y, d1.x);
    }
    if (cp != ch) {   // columns: (h + p)/sqrt2, i (p - h)/sqrt2
      cplx a0 = make_double2((x00.x + x01.x) * q2, (x00.y + x01.y) * q2), d0 = make_double2((x01.x - x00.x) * q2, (x01.y - x00.y) * q2);
      cplx a1 = make_double2((x10.x + x11.x) * q2, (x10.y + x11.y) * q2), d1 = make_double2((x11.x - x10.x) * q2, (x11.y - x10.y) * q2);
      x00 = a0; x01 = make_double2(-d0.y, d0.x); x10 = a1; x11 = make_double2(-d1.y, d1.x);
    }
    row_h[ch] = cmul(x00, scale);
    if (cp != ch) row_h[cp] = cmul(x01, scale);
    if (rp != rh) {
      row_p[ch] = cmul(x10, scale);
      if (cp != ch) row_p[cp] = cmul(x11, scale);
    }
  }
}

__global__ void __launch_bounds__(256) k_unsymmetrize(int H, int U, int n_end, int B, int nrhs, int n_pad, const int* __restrict__ units,
                                                       const int* __restrict__ deg, const cplx* __restrict__ tab, cplx* __restrict__ A,
                                                       long long lda, long long sys_stride) {
  const int s = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;                 // (row unit over all balls, rhs)
  if (t >= B * U * nrhs) return;
  const int q = t % nrhs, ru = t / nrhs, br = ru / U, ur = ru - br * U;
  const int rh = units[2 * ur], rp = units[2 * ur + 1];
  const cplx* ts = tab + (size_t)s * B * 3 * n_end + (size_t)br * 3 * n_end;
  const cplx g = zsqrt(cmul(ts[deg[rh]], ts[n_end + deg[rh]]));      // 1 / r
  cplx* As = A + (size_t)s * sys_stride + n_pad + q;
  cplx* ph = As + (size_t)(br * H + rh) * lda;
  cplx* pp = As + (size_t)(br * H + rp) * lda;
  const cplx yh = cmul(*ph, g);
  if (rp == rh) { *ph = yh; return; }
  const cplx yp = cmul(*pp, g);
  const double q2 = 0.70710678118654752440;                            // x_h = (y_h - i y_p)/sqrt2, x_p = (y_h + i y_p)/sqrt2
  *ph = make_double2((yh.x + yp.y) * q2, (yh.y - yp.x) * q2);
  *pp = make_double2((yh.x - yp.y) * q2, (yh.y + yp.x) * q2);
}

int launch_symmetrize(const biem_plan* p, int nb, int B, int nrhs, int n_pad, const double* d_tab, double* d_A, long long lda,
                      long long sys_stride, bool inverse_on_solution, hipStream_t st) {
  const int U = (int)(p->units.size() / 2);
  if (nb <= 0 || B <= 0) return BIEM_OK;
  if (nb > 65535) { set_error("biem symmetric path: at most 65535 systems per call"); return BIEM_ERR_ARG; }
  ProfScope ps(PK_SWAP, st, 0.0);   // class 4: row interchanges in the LU, this transform in the symmetric path
  if (!inverse_on_solution)
    hipLaunchKernelGGL(k_symmetrize, dim3((B * U + nrhs + 255) / 256, B * U < 65535 ? B * U : 65535, nb), dim3(256), 0, st, p->H, U, p->n_end, B, nrhs, n_pad,
                       p->d_units, p->d_deg, (const cplx*)d_tab, (cplx*)d_A, lda, sys_stride);
  else
    hipLaunchKernelGGL(k_unsymmetrize, dim3((B * U * nrhs + 255) / 256, nb), dim3(256), 0, st, p->H, U, p->n_end, B, nrhs, n_pad,
                       p->d_units, p->d_deg, (const cplx*)d_tab, (cplx*)d_A, lda, sys_stride);
  BIEM_LAUNCHCHK();
  return BIEM_OK;
}

// ---------------------------------------------------------------------------------------------
// density = x / (gh * blc)   (reference scaling of the unknown; single-ball shortcut _biem.py:673-690 with x = f)
// ---------------------------------------------------------------------------------------------
__global__ void k_density(int H, int n_end, int B, int nrhs, long long total, const int* __restrict__ deg, const cplx* __restrict__ x,
                          long long sys_stride, long long elem_stride, long long rhs_stride, const cplx* __restrict__ tab,
                          cplx* __restrict__ dens) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // i = ((s*nrhs + r)*B + b)*H + h
  if (i >= total) return;
  int h = (int)(i % H);
  long long srb = i / H;
  long long b = srb % B, sr = srb / B, r = sr % nrhs, s = sr / nrhs;
  int n = deg[h];
  const cplx* t = tab + (size_t)(s * B + b) * 3 * n_end;
  cplx v = x[(size_t)s * sys_stride + ((size_t)b * H + h) * elem_stride + (size_t)r * rhs_stride];
  dens[i] = cmul(v, crecip(cmul(t[n_end + n], t[2 * n_end + n])));
}

int launch_density(const biem_plan* p, int nb, int B, int nrhs, const double* d_x, long long sys_stride, long long elem_stride,
                   long long rhs_stride, const double* d_tab, double* d_density, hipStream_t st) {
  if (nrhs < 1) { set_error("biem_density: nrhs < 1"); return BIEM_ERR_ARG; }
  long long total = (long long)nb * nrhs * B * p->H;
  if (total <= 0) return BIEM_OK;
  hipLaunchKernelGGL(k_density, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p->H, p->n_end, B, nrhs, total, p->d_deg,
                     (const cplx*)d_x, sys_stride, elem_stride, rhs_stride, (const cplx*)d_tab, (cplx*)d_density);
  BIEM_LAUNCHCHK();
  return BIEM_OK;
}

}  // namespace biem
