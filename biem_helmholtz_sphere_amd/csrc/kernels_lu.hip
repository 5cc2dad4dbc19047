// kernels_lu.hip -- K4/K5: batched dense complex LU with partial pivoting + triangular solves on gfx950.
//
// Replaces batch_tensorsolve.btensorsolve -> linalg.solve (LAPACK zgesv) at reference _biem.py:797.
//
// Layout: every system is the augmented row-major matrix [M | F] with n_pad rows, n_cols = n_pad + nrhs columns and
// leading dimension lda (complex128 elements).  Right-looking blocked LU, panel width NB:
//   panel_load   M[j:, j:j+NB] -> P (column-major workspace, so the pivot search and the rank-1 updates are coalesced)
//   panel_factor unblocked LU with partial pivoting on P (one workgroup per system)
//   panel_store  P -> M (L21 stays in P: it is the A operand of the trailing zgemm in exactly the [k][i] order
//                the f64 MFMA A-fragment wants, so it is staged to LDS without a transpose)
//   swap         row interchanges on the columns outside the panel (coalesced: rows are contiguous)
//   trsm         U12 = L11^{-1} M[j:j+NB, j+NB:]   (includes the right-hand-side columns: forward elimination rides along)
//   gemm         M[j+NB:, j+NB:] -= L21 * U12      3M zgemm on v_mfma_f64_4x4x4_4b_f64 (k_gemm3m_pipe; K = 64 / 128 / 256)
// then a blocked back substitution with U.  All kernels are batched over systems (blockIdx.z / .y).
#include "common.hpp"
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace biem {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int NB = 64;    // panel width
constexpr int BS = 64;    // back-substitution block

int lu_npad(int N) { return ((N + NB - 1) / NB) * NB; }

static inline long long ldp_of(int n_pad) { return (long long)n_pad; }

size_t lu_workspace_bytes(int nb, int n_pad, int nrhs) {
  (void)nrhs;
  // four 64-column panels (two K = 128 blocks = one K = 256 update) + the 64 x 64 operand I - L11^{-1} of the MFMA triangular solve
  // + the tile map of the triangular (symmetric) update: one int per lower-triangle tile of the largest trailing matrix
  const size_t T = (size_t)n_pad / NB;
  // + two doubles per system: max |A| over what the symmetric factorisation reads and max |U| (a-posteriori growth check)
  // (the row-panel form of the symmetric path needs no panels, only (64 x 64 + 64) complex per system; one layout serves all)
  return (size_t)nb * (4 * NB * (size_t)ldp_of(n_pad) + (size_t)NB * NB) * sizeof(cplx) + ((T * (T + 1) / 2 + 63) / 64) * 64 * sizeof(int) +
         (size_t)nb * 2 * sizeof(double) + (size_t)nb * NB * sizeof(cplx);
}

// ---------------------------------------------------------------------------------------------
// a-posteriori element growth of the symmetric factorisation (bounded multipliers alone do not bound it):
// growth[s][0] = max |a_ij| over the part of A the factorisation reads, growth[s][1] = max |u_ij|, both as cabs1 = |re| + |im|.
// Bit patterns of non-negative doubles order like unsigned integers and every NaN pattern lies above the finite ones, so a
// 64-bit atomicMax keeps the maximum and a NaN sticks.  k_growth_check marks a system (info = -(n_pad + 1)) whose factor U grew
// by more than GROWTH_MAX over A, or holds a non-finite entry; the caller re-solves it with the pivoted LU.
// ---------------------------------------------------------------------------------------------
constexpr double GROWTH_MAX = 2.0e2;     // (1e3 with multipliers <= 2 in round 1, 2e2 with multipliers <= 10 in round 2; see NOPIV_REL for round 3)
__device__ inline double cabs1(cplx v) { return fabs(v.x) + fabs(v.y); }
__device__ inline double nan_max(double a, double b) { return !(b <= a) ? b : a; }       // NaN in b wins; NaN in a stays
__device__ inline void block_max_publish(double m, unsigned long long* dst) {              // 1-D blocks of whole waves
  __shared__ double sm_max[16];
  for (int o = 32; o > 0; o >>= 1) m = nan_max(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0) sm_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = sm_max[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = nan_max(m, sm_max[i]);
    // most workgroups cannot raise the maximum: a plain read filters them out (the slot only grows, so a stale value is safe)
    if (!(m <= __longlong_as_double((long long)*(volatile unsigned long long*)dst))) atomicMax(dst, (unsigned long long)__double_as_longlong(m));
  }
}
// max |A| over the lower triangle and the diagonal 64 x 64 blocks: 8 rows per workgroup, coalesced along the row
__global__ void __launch_bounds__(256) k_absmax_lower(const cplx* __restrict__ A, long long lda, long long sys_stride, int n_pad,
                                                       unsigned long long* __restrict__ growth) {
  const int s = blockIdx.y;
  const cplx* As = A + (size_t)s * sys_stride;
  double m = 0.0;
  for (int r = 0; r < 8; ++r) {
    const int i = blockIdx.x * 8 + r;
    if (i >= n_pad) break;
    const int cend = (i / NB + 1) * NB;
    for (int c = threadIdx.x; c < cend; c += 256) m = nan_max(m, cabs1(As[(size_t)i * lda + c]));
  }
  block_max_publish(m, growth + 2 * (size_t)s);
}
unsigned long long* lu_growth_slots(void* d_work, int nb, int n_pad) {
  const size_t T = (size_t)n_pad / NB;
  cplx* Winv = (cplx*)d_work + (size_t)nb * 4 * NB * (size_t)ldp_of(n_pad);
  int* tri = (int*)(Winv + (size_t)nb * NB * NB);
  return (unsigned long long*)(tri + ((T * (T + 1) / 2 + 63) / 64) * 64);
}
__global__ void k_growth_preset(int nb, unsigned long long* __restrict__ growth, double amax) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nb) return;
  growth[2 * s] = (unsigned long long)__double_as_longlong(amax);
  growth[2 * s + 1] = 0ULL;
}
int lu_growth_init(void* d_work, int nb, int n_pad, double amax, hipStream_t st) {
  if (nb <= 0) return BIEM_OK;
  hipLaunchKernelGGL(k_growth_preset, dim3((nb + 63) / 64), dim3(64), 0, st, nb, lu_growth_slots(d_work, nb, n_pad), amax);
  BIEM_LAUNCHCHK();
  return BIEM_OK;
}
__global__ void k_growth_check(int nb, int n_pad, const unsigned long long* __restrict__ growth, int* __restrict__ info, double limit) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nb) return;
  const double amax = __longlong_as_double((long long)growth[2 * s]), umax = __longlong_as_double((long long)growth[2 * s + 1]);
  if (!(umax <= limit * amax) && info[s] == 0) info[s] = -(n_pad + 1);
}

// ---------------------------------------------------------------------------------------------
// panel load / store (transposing copies through LDS)
// ---------------------------------------------------------------------------------------------
constexpr int TR = 32;   // rows per transpose tile
__global__ void __launch_bounds__(256) k_panel_load(const cplx* __restrict__ A, long long lda, long long sys_stride,
                                                     cplx* __restrict__ Pw, long long ldp, long long p_stride, int n_pad, int j) {
  __shared__ cplx tile[TR][NB + 1];
  const int s = blockIdx.y, i0 = j + blockIdx.x * TR;
  const cplx* As = A + (size_t)s * sys_stride;
  cplx* Ps = Pw + (size_t)s * p_stride;
  for (int idx = threadIdx.x; idx < TR * NB; idx += 256) {
    int r = idx / NB, c = idx % NB, gi = i0 + r;
    if (gi < n_pad) tile[r][c] = As[(size_t)gi * lda + j + c];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < TR * NB; idx += 256) {
    int c = idx / TR, r = idx % TR, gi = i0 + r;
    if (gi < n_pad) Ps[(size_t)c * ldp + gi] = tile[r][c];
  }
}

__global__ void __launch_bounds__(256) k_panel_store(cplx* __restrict__ A, long long lda, long long sys_stride,
                                                      const cplx* __restrict__ Pw, long long ldp, long long p_stride, int n_pad, int j) {
  __shared__ cplx tile[TR][NB + 1];
  const int s = blockIdx.y, i0 = j + blockIdx.x * TR;
  cplx* As = A + (size_t)s * sys_stride;
  const cplx* Ps = Pw + (size_t)s * p_stride;
  for (int idx = threadIdx.x; idx < TR * NB; idx += 256) {
    int c = idx / TR, r = idx % TR, gi = i0 + r;
    if (gi < n_pad) tile[r][c] = Ps[(size_t)c * ldp + gi];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < TR * NB; idx += 256) {
    int r = idx / NB, c = idx % NB, gi = i0 + r;
    if (gi < n_pad) As[(size_t)gi * lda + j + c] = tile[r][c];
  }
}

// ---------------------------------------------------------------------------------------------
// panel factorisation: LU with partial pivoting of the column-major panel, blocked in strips of PW columns.
//   k_panel_strip  (one 1024-thread workgroup per system): unblocked right-looking elimination confined to the strip's PW
//                  columns (pivot = max |re| + |im|, LAPACK izamax's cabs1, ties -> smallest row; the interchange is
//                  applied to all NB panel columns; the pass that writes column c+1 also searches it), then
//                  U_strip,right = L_strip^{-1} P[strip rows, right columns]
//   k_panel_update (all CUs): the rank-PW update of the rows below the strip on the right columns
// History (profiles/): unblocked sweep 2.15 ms per 6400 x 64 panel (NB^2/2 column passes through L2); strips inside one
// workgroup 0.89 ms, then f64-VALU bound: the rank-8 updates are 92 MFLOP on ONE CU, and 32 systems keep only 32 of 256
// CUs busy - hence the update runs as its own launch over the whole chip.
// ---------------------------------------------------------------------------------------------
constexpr int PW = 8;

// symmetric path: smallest accepted |diagonal| / |entry of its row|: every multiplier <= 10 (partial pivoting: <= 1).  It is the growth
// check (GROWTH_MAX) that bounds the error; with it in place the limit of 2 of round 1 only sent close-sphere systems at low k
// to the pivoted LU that the symmetric path solves to the same 1e-13 (profiles/r02_ldlt_fallback_survey.txt: a third -> a ninth of them)
// Round 3: multipliers <= 100.  The rejections of the close-sphere survey all sit at the first unknown of the second sphere (its
// monopole after the first sphere's elimination: pivot 1 - coupling^2) at LOW wavenumbers, with multipliers of 11 .. 77 (they
// saturate near 76 as k -> 0 for two unit spheres 0.04 apart) and a measured growth of 8 .. 45; the factorisation without
// interchanges solves every one of them to 4e-15 .. 1e-14 of the pivoted LU (NumPy emulation of this factorisation on the symmetric
// form, cond 33 .. 614).  It is the a-posteriori growth limit (200) that bounds the error; a limit of 10 on the multipliers only
// cost a fill and a pivoted LU (3 x the time) for systems the symmetric path solves to rounding.
constexpr double NOPIV_REL = 0.01;
constexpr int STRIP_CACHE_ROWS = 1024;   // rows of the strip kept in LDS (one per thread): 1024 x 8 x 16 B = 128 KiB

__global__ void __launch_bounds__(1024) k_panel_strip(cplx* __restrict__ Pw, long long ldp, long long p_stride, int n_pad, int j, int c0,
                                                       int second, int* __restrict__ ipiv, int* __restrict__ info) {
  // Thread t owns the fixed rows rs + t + 1024 k of the strip (rs = j + c0).  Its first row (k = 0) lives in LDS for the whole
  // strip - the strip is read and written once per column pass otherwise, and with every CU running one system's strip the
  // kernel is bound by that traffic (6.5 TB/s at 256 systems); the upper rows are the ones every pass touches longest.
  extern __shared__ cplx tile[];   // [PW][STRIP_CACHE_ROWS]: tile[q * 1024 + (row - rs)]
  __shared__ double sval[2][16];   // per-wave pivot candidates, double-buffered over columns
  __shared__ int sidx[2][16];
  __shared__ cplx sU[PW];          // current pivot row restricted to the strip
  __shared__ cplx sL[PW][PW];      // unit-lower strip triangle
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  cplx* Ps = Pw + (size_t)s * p_stride;
  const int rs = j + c0;
  cplx* Pc = Ps + (size_t)c0 * ldp;
  const unsigned ldp32 = (unsigned)ldp;
  const int my0 = rs + tid;                      // the cached row of this thread
  const bool have0 = my0 < n_pad;
  auto sget = [&](int q, int row) -> cplx { return row - rs < STRIP_CACHE_ROWS ? tile[q * STRIP_CACHE_ROWS + (row - rs)] : Pc[(unsigned)q * ldp32 + (unsigned)row]; };
  auto sput = [&](int q, int row, cplx v) {
    if (row - rs < STRIP_CACHE_ROWS) tile[q * STRIP_CACHE_ROWS + (row - rs)] = v; else Pc[(unsigned)q * ldp32 + (unsigned)row] = v;
  };

  auto publish = [&](double best, int bi, int buf) {
    for (int o = 32; o > 0; o >>= 1) {
      double ob = __shfl_down(best, o, 64);
      int oi = __shfl_down(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { sval[buf][wave] = best; sidx[buf][wave] = bi; }
  };
  auto decide = [&](int buf, int r0) -> int {   // every thread reduces the 16 candidates redundantly (no extra barrier)
    double b = sval[buf][0]; int ix = sidx[buf][0];
#pragma unroll
    for (int w = 1; w < 16; ++w) { double v = sval[buf][w]; int i2 = sidx[buf][w]; if (v > b || (v == b && i2 < ix)) { b = v; ix = i2; } }
    return ix == 0x7fffffff ? r0 : ix;          // all-NaN column: keep the diagonal
  };

  {  // stage the cached rows and search the strip's first column
    double best = -1.0; int bi = 0x7fffffff;
    if (have0) {
#pragma unroll
      for (int q = 0; q < PW; ++q) tile[q * STRIP_CACHE_ROWS + tid] = Pc[(unsigned)q * ldp32 + (unsigned)my0];
      const cplx v = tile[tid];
      const double a = fabs(v.x) + fabs(v.y);
      if (a > best) { best = a; bi = my0; }
    }
    for (int i = my0 + STRIP_CACHE_ROWS; i < n_pad; i += 1024) {
      const cplx v = Pc[(unsigned)i];
      const double a = fabs(v.x) + fabs(v.y);
      if (a > best) { best = a; bi = i; }
    }
    publish(best, bi, 0);
  }
  __syncthreads();
  int buf = 0;
  // the column loop is unrolled (cq = c - c0 is a compile-time index into the per-row register copy of the strip):
  // per row all PW columns are LOADED FIRST (independent requests in flight), then eliminated, then stored - the earlier
  // load/fma/store-per-element form was one dependent L2 round trip per element (no __restrict__ is possible here)
#pragma unroll
  for (int cq = 0; cq < PW; ++cq) {
    const int c = c0 + cq;
    const int r0 = j + c;
    const int p = decide(buf, r0);
    if (tid == 0) ipiv[(size_t)s * n_pad + r0] = p;
    if (tid < NB) {
      const bool mine = tid >= c0 && tid < c0 + PW;     // a strip column: rows may live in LDS
      cplx a = mine ? sget(tid - c0, r0) : Ps[(size_t)tid * ldp + r0];
      if (p != r0) {
        cplx b = mine ? sget(tid - c0, p) : Ps[(size_t)tid * ldp + p];
        if (mine) { sput(tid - c0, p, a); sput(tid - c0, r0, b); }
        else { Ps[(size_t)tid * ldp + p] = a; Ps[(size_t)tid * ldp + r0] = b; }
        a = b;
      }
      if (mine) sU[tid - c0] = a;   // row r0 after the interchange, strip columns
    }
    __syncthreads();
    const cplx piv = sU[cq];
    const bool singular = piv.x == 0.0 && piv.y == 0.0;
    if (singular && tid == 0 && info[s] == 0) info[s] = r0 + 1;
    const cplx rinv = singular ? make_double2(0.0, 0.0) : crecip(piv);
    cplx u[PW];
#pragma unroll
    for (int q = 0; q < PW; ++q) u[q] = sU[q];
    double best = -1.0; int bi = 0x7fffffff;
    if (have0 && my0 > r0) {                              // the cached row: LDS
      cplx v[PW];
#pragma unroll
      for (int q = cq; q < PW; ++q) v[q] = tile[q * STRIP_CACHE_ROWS + tid];
      if (!singular) {
        const cplx l = cmul(v[cq], rinv);
        v[cq] = l;
#pragma unroll
        for (int q = cq + 1; q < PW; ++q) v[q] = cfnma(l, u[q], v[q]);
      }
#pragma unroll
      for (int q = cq; q < PW; ++q) tile[q * STRIP_CACHE_ROWS + tid] = v[q];
      if (cq + 1 < PW) { double a = fabs(v[cq + 1 < PW ? cq + 1 : cq].x) + fabs(v[cq + 1 < PW ? cq + 1 : cq].y); if (a > best) { best = a; bi = my0; } }
    }
    for (int i = my0 + STRIP_CACHE_ROWS; i < n_pad; i += 1024) {     // the other rows: global memory (always below r0)
      cplx v[PW];
#pragma unroll
      for (int q = cq; q < PW; ++q) v[q] = Pc[(unsigned)q * ldp32 + (unsigned)i];
      if (!singular) {
        const cplx l = cmul(v[cq], rinv);
        v[cq] = l;
#pragma unroll
        for (int q = cq + 1; q < PW; ++q) v[q] = cfnma(l, u[q], v[q]);
      }
#pragma unroll
      for (int q = cq; q < PW; ++q) Pc[(unsigned)q * ldp32 + (unsigned)i] = v[q];
      if (cq + 1 < PW) { double a = fabs(v[cq + 1 < PW ? cq + 1 : cq].x) + fabs(v[cq + 1 < PW ? cq + 1 : cq].y); if (a > best) { best = a; bi = i; } }
    }
    if (cq + 1 < PW) { publish(best, bi, buf ^ 1); buf ^= 1; }
    __syncthreads();
  }
  // cached rows back to the panel
  if (have0) {
#pragma unroll
    for (int q = 0; q < PW; ++q) Pc[(unsigned)q * ldp32 + (unsigned)my0] = tile[q * STRIP_CACHE_ROWS + tid];
  }
  __syncthreads();
  const int nright = NB - (c0 + PW);
  if (nright <= 0) return;
  __shared__ cplx sL10[PW][PW];       // second strip of a pair: its rows x the first strip's columns
  if (tid < PW * PW) {
    int q = tid / PW, q2 = tid % PW;   // L[q][q2], q2 < q
    sL[q][q2] = (q2 < q) ? Ps[(size_t)(c0 + q2) * ldp + rs + q] : make_double2(0.0, 0.0);
    if (second) sL10[q][q2] = Ps[(size_t)(c0 - PW + q2) * ldp + rs + q];
  }
  __syncthreads();
  if (tid < nright) {
    cplx* colr = Ps + (size_t)(c0 + PW + tid) * ldp + rs;
    cplx x[PW];
#pragma unroll
    for (int q = 0; q < PW; ++q) x[q] = colr[q];
    if (second) {
      // the right columns have not received the first strip's update yet (it is applied together with this strip's as one
      // rank-16 update): bring this strip's rows up to date here, x -= L10 U0, U0 = the first strip's rows of this column
      cplx u0[PW];
#pragma unroll
      for (int q2 = 0; q2 < PW; ++q2) u0[q2] = colr[q2 - PW];
#pragma unroll
      for (int q = 0; q < PW; ++q)
#pragma unroll
        for (int q2 = 0; q2 < PW; ++q2) x[q] = cfnma(sL10[q][q2], u0[q2], x[q]);
    }
#pragma unroll
    for (int q = 1; q < PW; ++q)
#pragma unroll
      for (int q2 = 0; q2 < q; ++q2) x[q] = cfnma(sL[q][q2], x[q2], x[q]);
#pragma unroll
    for (int q = 0; q < PW; ++q) colr[q] = x[q];
  }
}

// rows below the strip, right columns:  P[cc][i] -= sum_q L[i][q] U[q][cc];  one thread per row, all CUs
template <int PWU>
__global__ void __launch_bounds__(256) k_panel_update(cplx* __restrict__ Pw, long long ldp, long long p_stride, int n_pad, int j, int c0,
                                                       int ncols) {
  // rank-PWU update of `ncols` (a multiple of 8) columns right of the PWU factored columns c0 .. c0+PWU-1, rows below them
  __shared__ cplx sUr[PWU][NB];
  const int s = blockIdx.y, tid = threadIdx.x;
  cplx* Ps = Pw + (size_t)s * p_stride;
  const int rs = j + c0;
  for (int e = tid; e < ncols * PWU; e += 256) {
    int t = e / PWU, q = e % PWU;
    sUr[q][t] = Ps[(size_t)(c0 + PWU + t) * ldp + rs + q];
  }
  __syncthreads();
  const int i = rs + PWU + blockIdx.x * 256 + tid;
  if (i >= n_pad) return;
  cplx l[PWU];
#pragma unroll
  for (int q = 0; q < PWU; ++q) l[q] = Ps[(size_t)(c0 + q) * ldp + i];
  // 8 independent loads, 8 PWU complex FMAs, 8 stores per group
  for (int t0 = 0; t0 < ncols; t0 += 8) {
    cplx v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = Ps[(size_t)(c0 + PWU + t0 + t) * ldp + i];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int q = 0; q < PWU; ++q) v[t] = cfnma(l[q], sUr[q][t0 + t], v[t]);
#pragma unroll
    for (int t = 0; t < 8; ++t) Ps[(size_t)(c0 + PWU + t0 + t) * ldp + i] = v[t];
  }
}

// ---------------------------------------------------------------------------------------------
// row interchanges outside the panel: thread per column, all NB swaps in order.
// columns: left part [0, j) when `left`, right part [j+NB, n_cols).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_swap(cplx* __restrict__ A, long long lda, long long sys_stride, int n_pad, int n_cols,
                                               int j, const int* __restrict__ ipiv, int left) {
  __shared__ int sp[NB];
  const int s = blockIdx.y;
  if (threadIdx.x < NB) sp[threadIdx.x] = ipiv[(size_t)s * n_pad + j + threadIdx.x];
  __syncthreads();
  int t = blockIdx.x * 256 + threadIdx.x;
  int col;
  if (left) { col = t < j ? t : t + NB; } else { col = t + j + NB; }
  if (col >= n_cols) return;
  cplx* As = A + (size_t)s * sys_stride + col;
  for (int c = 0; c < NB; ++c) {
    int p = sp[c];
    if (p != j + c) {
      cplx a = As[(size_t)(j + c) * lda], b = As[(size_t)p * lda];
      As[(size_t)(j + c) * lda] = b;
      As[(size_t)p * lda] = a;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// trailing update -- persistent zgemm on v_mfma_f64_4x4x4_4b_f64 (kernel below: k_gemm3m_pipe).
//
// MFMA form.  Measured on MI355X (tools/mfma_probe*.hip, profiles/r01_mfma_f64_*probe*.txt): the 16x16x4 f64 MFMA
// saturates at ~47-49 TFLOP/s (one per ~100 cycles per SIMD) at any occupancy, the 4-block 4x4x4 form issues every
// 16.3 cycles = 75-78 TFLOP/s with >= 48 independent accumulators.  The 4-block form multiplies A_blk (4x4) by B_blk (4x4)
// for blk = 0..3 (lane l: i|j = l&3, blk = (l>>2)&3, k = l>>4; D: j = l&3, blk, i = l>>4; CBSZ/ABID are not honoured
// for f64: profiles/r01_mfma_f64_4x4x4_layout.txt), so a 16x16 tile is built from 4 instructions whose A fragment holds
// the SAME 4-row block in all four slots (an LDS broadcast read): accumulator g = rows 4g..4g+3 x 16 columns, i.e.
// register g of the 16x16x4 result layout.  The f64 NEG bits (blgp bit 0 negates A) give acc = C - A*B directly.
//
// Memory schedule.  With K = NB the update is only 16 flop per byte of C traffic; a read-modify-write epilogue leaves
// every wave ~60 % of its cycles in s_waitcnt (profiles/r01_gemm_pmc.txt) because all workgroups hit HBM together and
// the MFMAs then idle.  Each workgroup is persistent and streams: the C tile is loaded in slices during the K-chunks,
// the final stores stay in flight while the next tile starts, and the operand stream runs ahead across tile boundaries.
//
// Tile order.  Tiles are numbered system-major, then bands of 8 tile-rows, then column-major inside a band, so 64
// consecutive tiles form an 8 x 8 block sharing 8 A- and 8 B-panels.  The workgroups that share blockIdx % 8 (one XCD
// under the observed round-robin placement; speed only) sweep one block together.
// (Superseded variants - 16x16x4 MFMA with RMW epilogue, 2-stage 4M and 3M kernels - are described with their numbers
// in DESIGN.md section 5; their sources are in the git history.)
// ---------------------------------------------------------------------------------------------

struct TileGrid {
  int ty_n, tx_n, per_sys, full_bands, ntiles;
  int row_begin, row_end, col_begin, col_end;   // C region updated by this launch
  int brow;                                     // first row of the B operand (U12 rows brow .. brow + K)
  // tiles of tile column `pcol_tx` deliver their result transposed into the panel workspace (the next panel to factor:
  // column-major P[c][row]) instead of the matrix, which saves that panel's transposing load; pout == nullptr: off
  cplx* pout; long long pout_ld, pout_stride; int pcol_tx;
  int tri;                                      // 1: only tiles with tx <= ty (square region, symmetric update); 2: only tx >= ty
  const int* tri_map; int tri_full;             // (ty << 16 | tx) of the first tri_full tiles of that order (the full bands)
  int blk_sh;                                   // log2 of the tiles per XCD block of the workgroup -> tile map: 6, or 3 for small launches
  unsigned long long per_sys_magic;             // ceil(2^40 / per_sys): t / per_sys = (t * magic) >> 40 for t < 2^25 (scalar multiply, no VALU division)
};

// the triangular order: lower triangle incl. the diagonal tiles in bands of 8 tile rows; band b (tile rows 8b .. 8b+hb-1) holds
// the columns 0 .. 8b+hb-1, column-major; column tx <= 8b has hb tiles, column 8b+q has hb-q.  A full band holds 64 b + 36
// tiles, 32 b^2 + 4 b tiles precede it - independent of the matrix size, so ONE table serves every launch of a factorisation.
__device__ __host__ inline void tri_decode_band(int r, int b, int hb, int& ty, int& tx) {
  int rr = r - (32 * b * b + 4 * b);
  if (rr < 8 * b * hb) { tx = rr / hb; ty = 8 * b + rr - tx * hb; }
  else {
    int rem = rr - 8 * b * hb, q = 0;
    while (rem >= hb - q) { rem -= hb - q; ++q; }
    tx = 8 * b + q; ty = 8 * b + q + rem;
  }
}
__global__ void k_tri_map(int* map, int n) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  int b = (int)((sqrtf(16.0f + 128.0f * (float)r) - 4.0f) * (1.0f / 64.0f));
  while (b > 0 && 32 * b * b + 4 * b > r) --b;
  while (32 * (b + 1) * (b + 1) + 4 * (b + 1) <= r) ++b;
  int ty, tx;
  tri_decode_band(r, b, 8, ty, tx);
  map[r] = ty << 16 | tx;
}

__device__ inline void tile_decode(const TileGrid& tg, int t, int& s, int& ty, int& tx) {
  s = (int)(((unsigned long long)(unsigned)t * tg.per_sys_magic) >> 40);
  int r = t - s * tg.per_sys;
  if (r >= tg.per_sys) { r -= tg.per_sys; ++s; }      // (never taken for t < 2^25; kept as a guard)
  if (tg.tri) {
    if (r < tg.tri_full) { const int v = tg.tri_map[r]; ty = v >> 16; tx = v & 0xffff; }
    else tri_decode_band(r, tg.full_bands, tg.ty_n - 8 * tg.full_bands, ty, tx);      // the partial last band
    if (tg.tri == 2) { const int t2 = ty; ty = tx; tx = t2; }                          // upper triangle: the mirror tile
    return;
  }
  int fb = tg.full_bands * 8 * tg.tx_n;
  if (r < fb) {
    int band = r / (8 * tg.tx_n), rr = r - band * 8 * tg.tx_n;
    tx = rr >> 3; ty = band * 8 + (rr & 7);
  } else {
    int rem = r - fb, h = tg.ty_n - tg.full_bands * 8;
    tx = rem / h; ty = tg.full_bands * 8 + rem - tx * h;
  }
}

constexpr int BM3 = 64, BN3 = 64;   // workgroup tile of the trailing update
constexpr int KC = 8;               // K rows per LDS stage (chunk)

// ---------------------------------------------------------------------------------------------
// trailing update, 3M form with a 3-stage LDS-DMA ring (product kernel).
// Evidence for the structure: the 2-stage kernels above run the 3M and the 4M arithmetic in the SAME time
// (444.7 vs 447.7 ms per 32-system step) - the update is bound by the latency of loads issued one chunk ahead, not by the
// MFMA pipe: hipcc drains vmcnt(0) at every __syncthreads() while an LDS-DMA is in flight and before any use of a
// VGPR-destination load.  Here every byte (A chunk, B chunk and the C slice of the chunk) arrives by LDS-DMA, each wave
// issues exactly NDMA instructions per chunk (addresses are clamped instead of masked, so the count is uniform), the
// barrier is a raw s_barrier and the waits are hand-counted: s_waitcnt vmcnt(NDMA) retires the group of the chunk about
// to be multiplied and leaves the next chunk's group in flight.  A full tile's 16 result stores also sit in the VM
// queue; the first two chunks after them wait vmcnt(NDMA + 16).
// Stage = A[8][64] + B[8][64] + C slice (UPC x 256 lanes) = 20 (K=128) or 24 KiB (K=64); 3 stages; 2 workgroups per CU.
// ---------------------------------------------------------------------------------------------
template <int N> __device__ inline void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// LDS fragment read outside the compiler's memory model: hipcc orders every ds_read it can see behind ALL pending LDS-DMA
// (s_waitcnt vmcnt(0)), which would drain the ring's prefetches; the consumer issues lds_wait() + sched_barrier itself.
__device__ inline cplx lds_read16(const cplx* p) {
  cplx v;
  unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) cplx*)p;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
// acc += a*b / acc -= a*b on the 4-block f64 MFMA (the f64 NEG bit, blgp bit 0, negates A).
// (An inline-asm form with the accumulator tied "+v" was tried to stop hipcc from rotating accumulators through the
// register file; it produced wrong results on gfx950 even with hazard padding, and the rolled chunk loop made it
// unnecessary - the builtin is the only form used.)
__device__ inline void mfma_acc(double& acc, double a, double b) { acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0); }
__device__ inline void mfma_acc_neg(double& acc, double a, double b) { acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 1); }
// keeps hipcc from moving VALU work into the MFMA block (and the MFMAs out of it)
__device__ inline void mfma_fence() { __builtin_amdgcn_sched_barrier(0); }
__device__ inline void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);   // keep MFMAs behind the wait (hipcc moves register-only instructions across asm)
}

// What the ISA of earlier attempts taught (all measured):
//  * hipcc puts s_waitcnt vmcnt(0) in front of every ds_read it can see while an LDS-DMA is pending (even with one
//    __shared__ array per stage) -> fragment reads are inline asm with an explicit lgkmcnt wait;
//  * unrolling the chunk loop (3 stage copies) made hipcc rotate the 48 accumulators through the register file and copy
//    them back with ~100-200 v_mov_b64 per chunk (they share the SIMD's vector issue port with the MFMAs) -> one rolled
//    chunk loop with a run-time stage offset;
//  * per-lane 64-bit address arithmetic for 5 DMAs per chunk cost ~250 VALU instructions -> wave-uniform scalar bases plus
//    per-lane 32-bit offsets that are constant for the whole kernel.
#define BIEM_PRIO_M() __builtin_amdgcn_s_setprio(1)
#define BIEM_PRIO_O() __builtin_amdgcn_s_setprio(3)
#ifdef BIEM_GEMM_TRACE
// diagnostic build only (tools/gemm_trace.cpp): wave 0 of the first 4 workgroups stamps (all 4 waves) s_memtime at 7 points of each of its
// first 64 chunks into LDS and dumps them at exit (no VM traffic inside the loop, the hand-counted vmcnt waits stay valid)
__device__ unsigned long long g_gemm_trace[16][64][8];
#ifdef BIEM_TR_STAMPS
#define BIEM_TR(i) { if (lane == 0 && tr_n < 64) s_tr[(wave * 64 + tr_n) * 8 + (i)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define BIEM_TR_NEXT() { ++tr_n; }
#else
#define BIEM_TR(i)
#define BIEM_TR_NEXT()
#endif
#else
#define BIEM_TR(i)
#define BIEM_TR_NEXT()
#endif
template <int KD>
__global__ void __launch_bounds__(256, 2) k_gemm3m_pipe(cplx* __restrict__ A, long long lda, long long sys_stride,
                                                         const cplx* __restrict__ Pw, long long ldp, long long p_stride,
                                                         TileGrid tg) {
  const int n_pad = tg.row_end, n_cols = tg.col_end;
  constexpr int NCH = KD / KC;               // 8, 16 or 32 K-chunks per tile
  constexpr int UPC = NCH >= 16 ? 1 : 16 / NCH;   // C units (one complex per lane) per chunk that carries C: 1 or 2
  constexpr int NCC = 16 / UPC;              // chunks that carry C units: the first NCC of a tile (all of them for K <= 128)
#if defined(BIEM_ABL_NOCDMA)                  // timing ablation: no C-slice DMA in the fused (interior, K = 128) path
  constexpr int NDMA = 4;
#elif defined(BIEM_ABL_ONLYCDMA)              // timing ablation: only the C-slice DMA
  constexpr int NDMA = 1;
#else
  constexpr int NDMA = 4 + UPC;              // LDS-DMA instructions per wave per chunk
#endif
  constexpr int AST = 68;                    // A row stride in LDS: +4 elements (64 B) so the broadcast A-fragment reads of
                                             // two k-rows in one ds_read_b128 lane group hit different banks
  constexpr int BOF = KC * AST;              // B block offset inside a stage
  constexpr int COF = BOF + KC * 64;         // C-slice offset
  constexpr int STG = COF + UPC * 256;       // complex elements per stage
  __shared__ cplx ring[3 * STG];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA bases (M0) and tile offsets stay on the SALU
  const int l3 = lane & 3, l15 = lane & 15, l4 = lane >> 4;
  const int w = blockIdx.x, nblk = gridDim.x >> 3, xl = w & 7;
  int q = (w >> 3) - nblk;
  auto next_tile = [&]() -> int {
    for (;;) {
      q += nblk;
      int base = ((q >> tg.blk_sh) * 8 + xl) << tg.blk_sh;
      if (base >= tg.ntiles) return -1;
      int t = base + (q & ((1 << tg.blk_sh) - 1));
      if (t < tg.ntiles) return t;
    }
  };
  int t = next_tile();
  if (t < 0) return;

  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  int cs, cty, ctx;
  tile_decode(tg, t, cs, cty, ctx);

  const unsigned offA0 = (unsigned)(((size_t)(wave) * ldp + lane) * sizeof(cplx));
  const unsigned offA1 = (unsigned)(((size_t)(wave + 4) * ldp + lane) * sizeof(cplx));
  const unsigned offB0 = (unsigned)(((size_t)(wave) * lda + lane) * sizeof(cplx));
  const unsigned offB1 = (unsigned)(((size_t)(wave + 4) * lda + lane) * sizeof(cplx));
  // wave w owns rows 16w .. 16w+15 of the 64 x 64 tile and all 64 columns: 4 broadcast A fragments (row quads g) and 4 B
  // fragments (column groups n) per k4-step instead of the 8 + 2 of a 32 x 32 wave tile: 16 instead of 20 fragment reads
  // and 3M operand sums per chunk.  C unit u = 4 n + g: rows 16w + 4g + (lane >> 4), columns 16n + (lane & 15).
  const unsigned offC = (unsigned)(((size_t)(wave * 16 + l4) * lda + l15) * sizeof(cplx));
  // Producer state: the DMA stream runs two chunks ahead of the multiplication and crosses tile boundaries on its own.
  // Interior tiles use running scalar bases (pA, pB advance by a constant per chunk; pC = tile origin + a 16-entry
  // pattern); edge tiles recompute clamped per-lane addresses (rare).
  const long long strideA = (long long)KC * ldp * (long long)sizeof(cplx);
  const long long strideB = (long long)KC * lda * (long long)sizeof(cplx);
  int p_s = cs, p_ty = cty, p_tx = ctx, p_ch = 0;       // tile / chunk the next DMA group belongs to
  bool p_interior = false, p_valid = true;
  int p_tiles = 0, c_tiles = 0;                          // tiles started by the producer / finished by the consumer
  const char *pA = nullptr, *pB = nullptr, *pC = nullptr;
  int n_new = NDMA;                                      // size of the newest DMA group in flight (K = 256: 5 with a C unit, 4 without)
  auto producer_tile = [&]() {                             // (re)compute the bases for chunk 0 of tile (p_s, p_ty, p_tx)
    const int r0 = tg.row_begin + p_ty * BM3, c0 = tg.col_begin + p_tx * BN3;
    p_interior = r0 + BM3 <= n_pad && c0 + BN3 <= n_cols;
    pA = (const char*)(Pw + ((size_t)p_s * p_stride + r0));
    pB = (const char*)(A + ((size_t)p_s * sys_stride + (size_t)tg.brow * lda + c0));
    pC = (const char*)(A + ((size_t)p_s * sys_stride + (size_t)r0 * lda + c0));
    p_ch = 0;
    ++p_tiles;
  };
  producer_tile();
  // issue the DMA group of the producer's current chunk into stage st (exactly NDMA instructions, all lanes active), advance
  auto issue_dma = [&](int st) {
    cplx* S = ring + st * STG;
    if (p_interior) {
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(pA + offA0), (lds_ptr_t)(S + wave * AST), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(pA + offA1), (lds_ptr_t)(S + (wave + 4) * AST), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(pB + offB0), (lds_ptr_t)(S + BOF + wave * 64), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(pB + offB1), (lds_ptr_t)(S + BOF + (wave + 4) * 64), 16, 0, 0);
      if (NCC == NCH || p_ch < NCC) {
#pragma unroll
        for (int i = 0; i < UPC; ++i) {
          const int u = p_ch * UPC + i;
          const long long dC = ((long long)(4 * (u & 3)) * lda + (u >> 2) * 16) * (long long)sizeof(cplx);
          __builtin_amdgcn_global_load_lds((glb_ptr_t)(pC + dC + offC), (lds_ptr_t)(S + COF + i * 256 + wave * 64), 16, 0, 0);
        }
      }
    } else {
      // edge tile: clamp instead of masking (the instruction count must stay uniform)
      const cplx* Ps = Pw + (size_t)p_s * p_stride;
      const cplx* As = A + (size_t)p_s * sys_stride;
      const int r0 = tg.row_begin + p_ty * BM3, c0 = tg.col_begin + p_tx * BN3;
      const int ar = min(r0 + lane, n_pad - 1), bc = min(c0 + lane, n_cols - 1);
#pragma unroll
      for (int r = 0; r < 2; ++r)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(Ps + (size_t)(p_ch * KC + wave + 4 * r) * ldp + ar),
                                         (lds_ptr_t)(S + (wave + 4 * r) * AST), 16, 0, 0);
#pragma unroll
      for (int r = 0; r < 2; ++r)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(As + (size_t)(tg.brow + p_ch * KC + wave + 4 * r) * lda + bc),
                                         (lds_ptr_t)(S + BOF + (wave + 4 * r) * 64), 16, 0, 0);
      if (NCC == NCH || p_ch < NCC) {
#pragma unroll
        for (int i = 0; i < UPC; ++i) {
          const int u = p_ch * UPC + i;
          const int row = min(r0 + wave * 16 + 4 * (u & 3) + l4, n_pad - 1);
          const int col = min(c0 + (u >> 2) * 16 + l15, n_cols - 1);
          __builtin_amdgcn_global_load_lds((glb_ptr_t)(As + (size_t)row * lda + col),
                                           (lds_ptr_t)(S + COF + i * 256 + wave * 64), 16, 0, 0);
        }
      }
    }
    n_new = (NCC == NCH || p_ch < NCC) ? NDMA : NDMA - UPC;      // VM instructions of the group just issued
  };
  auto advance = [&]() {
    pA += strideA; pB += strideB;
    if (++p_ch == NCH) {                                   // producer moves on to the next tile of this workgroup
      int tn = next_tile();
      if (tn >= 0) { tile_decode(tg, tn, p_s, p_ty, p_tx); producer_tile(); }
      else p_valid = false;
    }
  };

  double N1[4][4], P2[4][4], N3[4][4];       // [column group n][row quad g]; flat index = C unit u = 4 n + g
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int g = 0; g < 4; ++g) { N1[n][g] = 0.0; P2[n][g] = 0.0; N3[n][g] = 0.0; }

#ifdef BIEM_TR_STAMPS
  __shared__ unsigned long long s_tr[4 * 64 * 8];
  int tr_n = 0;
  for (int i = tid; i < 4 * 64 * 8; i += 256) s_tr[i] = 0;
  __syncthreads();
#endif
  auto issue = [&](int st) { issue_dma(st); advance(); };
  issue(0);
  issue(1);
  int st = 0;                 // stage of the chunk about to be multiplied
  int stores_pending = 0;     // 0: none, 1: 16 stores of a full tile were issued after the groups in flight, 2: unknown count
  // per-lane LDS offsets of the fragments inside a stage (elements)
  const int fbo = BOF + l4 * 64 + l15;                    // + k4*4*64 + n*16
  const int fao = l4 * AST + wave * 16 + l3;              // + k4*4*AST + 4g
  // A VALU instruction issued while the SIMD partner (the other workgroup's wave) streams MFMAs costs ~28 cycles even at
  // priority 3 (tools/mfma_valu_mix: 8 alone, 101 at equal priority; SALU and LDS instructions are unaffected).  So the
  // phase between two MFMA blocks holds no VALU work at all: the fragment addresses of the NEXT chunk, the 3M operand sums
  // and the C-slice additions are all issued inside this wave's own MFMA block, in the shadow of its MFMAs.
  typedef const __attribute__((address_space(3))) cplx* lds_cptr_t;
  unsigned aA = (unsigned)(size_t)(lds_cptr_t)(ring + fao), aB = (unsigned)(size_t)(lds_cptr_t)(ring + fbo),
           aC = (unsigned)(size_t)(lds_cptr_t)(ring + COF + tid);                      // stage 0
  // the producer is exactly one tile ahead whenever the consumer finishes a tile (it switches at the consumer's chunk
  // NCH-3 and not again before chunk NCH-3 of the next tile): its current coordinates are the consumer's next tile
  for (;;) {
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
      // retire this chunk's DMA group (mine), then meet the other waves: their groups have landed too and nobody still
      // reads the stage the next group is about to overwrite
      BIEM_TR(0)
      // (the newest group holds n_new instructions: NDMA, or NDMA - UPC for the chunks of a K = 256 tile without a C unit)
      const bool small_grp = NCC != NCH && n_new != NDMA;
      if (__builtin_expect(stores_pending == 0 && p_valid, 1)) {
        if (small_grp) wait_vmcnt<NDMA - UPC>(); else wait_vmcnt<NDMA>();
      } else if (!p_valid) {
        wait_vmcnt<0>();                                   // tail of this workgroup's work: no further groups are issued
      } else if (stores_pending == 2 && c == 0) {
        wait_vmcnt<0>();
      } else if (stores_pending == 1 && c < 2) {
        if (small_grp) wait_vmcnt<NDMA - UPC + 16>(); else wait_vmcnt<NDMA + 16>();
      } else {
        if (small_grp) wait_vmcnt<NDMA - UPC>(); else wait_vmcnt<NDMA>();
      }
      BIEM_TR(1)
#ifndef BIEM_ABL_NOBARRIER
      __builtin_amdgcn_s_barrier();
#endif
      __builtin_amdgcn_sched_barrier(0);
      BIEM_TR(2)
      const int st2 = st >= 1 ? st - 1 : 2;            // (st + 2) % 3
      // interior chunks put their DMA group between the fragment reads and the lgkmcnt wait (below):
      // the VMEM issue (~100 cycles per instruction with 8 waves' groups in flight) then runs under the LDS latency
#ifdef BIEM_ABL_NODMA      // timing ablation: no DMA at all (compute-only period)
      const bool fused = false;
      if (p_valid) advance();
#else
      const bool fused = p_valid && p_interior;
      if (p_valid && !fused) issue(st2);
#endif
      __builtin_amdgcn_sched_barrier(0);
      BIEM_TR(3)
      // fragments of both k4-steps and the C units of this chunk: 20 + UPC ds_read_b128 and their lgkmcnt wait in ONE asm
      // statement - hipcc may copy an asm output right after the statement, i.e. before a separate wait (that was the
      // cause of percent-level errors in an earlier build); byte offsets: k4*4352 + g*64 (A), k4*4096 + n*256 (B)
      cplx fb[2][4], fa[2][4], cv[UPC];   // [k4][column group of 16], [k4][row quad]
      unsigned m0_keep;                    // M0 is compiler-reserved: the fused statements save and restore it
      {
        if (UPC == 1 && fused && (NCC == NCH || p_ch < NCC)) {
          typedef __attribute__((address_space(3))) cplx* lds_cplx_t;
          cplx* S2 = ring + st2 * STG;
          const unsigned mA = (unsigned)(size_t)(lds_cplx_t)(S2 + wave * AST), mB = (unsigned)(size_t)(lds_cplx_t)(S2 + BOF + wave * 64);
          const long long dC = ((long long)(4 * (p_ch & 3)) * lda + (p_ch >> 2) * 16) * (long long)sizeof(cplx);
#ifdef BIEM_ABL_CHOT      // timing ablation: the C slice comes from an L2-resident address (the panel workspace)
          const char* pCc = (const char*)Pw + (dC & 0xfffff);
#else
          const char* pCc = pC + dC;
#endif
          asm volatile(
#ifndef BIEM_ABL_NOLDS
              "ds_read_b128 %[b0], %[aB]\n\tds_read_b128 %[b1], %[aB] offset:256\n\tds_read_b128 %[b2], %[aB] offset:512\n\tds_read_b128 %[b3], %[aB] offset:768\n\t"
              "ds_read_b128 %[b4], %[aB] offset:4096\n\tds_read_b128 %[b5], %[aB] offset:4352\n\tds_read_b128 %[b6], %[aB] offset:4608\n\tds_read_b128 %[b7], %[aB] offset:4864\n\t"
              "ds_read_b128 %[a0], %[aA]\n\tds_read_b128 %[a1], %[aA] offset:64\n\tds_read_b128 %[a2], %[aA] offset:128\n\tds_read_b128 %[a3], %[aA] offset:192\n\t"
              "ds_read_b128 %[a4], %[aA] offset:4352\n\tds_read_b128 %[a5], %[aA] offset:4416\n\tds_read_b128 %[a6], %[aA] offset:4480\n\tds_read_b128 %[a7], %[aA] offset:4544\n\t"
              "ds_read_b128 %[c0], %[aC]\n\t"
#endif
#ifndef BIEM_ABL_ONLYCDMA
              "s_mov_b32 %[keep], m0\n\t"
              "s_mov_b32 m0, %[mA]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[oA0], %[pA]\n\t"
              "s_add_u32 m0, %[mA], 4352\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[oA1], %[pA]\n\t"
              "s_mov_b32 m0, %[mB]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[oB0], %[pB]\n\t"
              "s_add_u32 m0, %[mB], 4096\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[oB1], %[pB]\n\t"
#endif
#ifndef BIEM_ABL_NOCDMA
              "s_add_u32 m0, %[mB], 8192\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[oC], %[pC]\n\t"
#endif
              "s_mov_b32 m0, %[keep]\n\t"
              "s_waitcnt lgkmcnt(0)"
              : [keep] "=&s"(m0_keep), [a0] "=&v"(fa[0][0]), [a1] "=&v"(fa[0][1]), [a2] "=&v"(fa[0][2]), [a3] "=&v"(fa[0][3]), [a4] "=&v"(fa[1][0]),
                [a5] "=&v"(fa[1][1]), [a6] "=&v"(fa[1][2]), [a7] "=&v"(fa[1][3]), [b0] "=&v"(fb[0][0]), [b1] "=&v"(fb[0][1]),
                [b2] "=&v"(fb[0][2]), [b3] "=&v"(fb[0][3]), [b4] "=&v"(fb[1][0]), [b5] "=&v"(fb[1][1]), [b6] "=&v"(fb[1][2]),
                [b7] "=&v"(fb[1][3]), [c0] "=&v"(cv[0])
              : [aA] "v"(aA), [aB] "v"(aB), [aC] "v"(aC), [mA] "s"(mA), [mB] "s"(mB), [oA0] "v"(offA0), [oA1] "v"(offA1), [oB0] "v"(offB0),
                [oB1] "v"(offB1), [oC] "v"(offC), [pA] "s"(pA), [pB] "s"(pB), [pC] "s"(pCc)
              : "memory", "scc");
          __builtin_amdgcn_sched_barrier(0);
          n_new = NDMA;
          advance();
        } else if (UPC == 1 && fused) {     // K = 256, producer chunk >= 16: no C unit in this group
          typedef __attribute__((address_space(3))) cplx* lds_cplx_t;
          cplx* S2 = ring + st2 * STG;
          const unsigned mA = (unsigned)(size_t)(lds_cplx_t)(S2 + wave * AST), mB = (unsigned)(size_t)(lds_cplx_t)(S2 + BOF + wave * 64);
          asm volatile(
#ifndef BIEM_ABL_NOLDS
              "ds_read_b128 %[b0], %[aB]\n\tds_read_b128 %[b1], %[aB] offset:256\n\tds_read_b128 %[b2], %[aB] offset:512\n\tds_read_b128 %[b3], %[aB] offset:768\n\t"
              "ds_read_b128 %[b4], %[aB] offset:4096\n\tds_read_b128 %[b5], %[aB] offset:4352\n\tds_read_b128 %[b6], %[aB] offset:4608\n\tds_read_b128 %[b7], %[aB] offset:4864\n\t"
              "ds_read_b128 %[a0], %[aA]\n\tds_read_b128 %[a1], %[aA] offset:64\n\tds_read_b128 %[a2], %[aA] offset:128\n\tds_read_b128 %[a3], %[aA] offset:192\n\t"
              "ds_read_b128 %[a4], %[aA] offset:4352\n\tds_read_b128 %[a5], %[aA] offset:4416\n\tds_read_b128 %[a6], %[aA] offset:4480\n\tds_read_b128 %[a7], %[aA] offset:4544\n\t"
              "ds_read_b128 %[c0], %[aC]\n\t"
#endif
#ifndef BIEM_ABL_ONLYCDMA
              "s_mov_b32 %[keep], m0\n\t"
              "s_mov_b32 m0, %[mA]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[oA0], %[pA]\n\t"
              "s_add_u32 m0, %[mA], 4352\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[oA1], %[pA]\n\t"
              "s_mov_b32 m0, %[mB]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[oB0], %[pB]\n\t"
              "s_add_u32 m0, %[mB], 4096\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[oB1], %[pB]\n\t"
#endif
              "s_mov_b32 m0, %[keep]\n\t"
              "s_waitcnt lgkmcnt(0)"
              : [keep] "=&s"(m0_keep), [a0] "=&v"(fa[0][0]), [a1] "=&v"(fa[0][1]), [a2] "=&v"(fa[0][2]), [a3] "=&v"(fa[0][3]), [a4] "=&v"(fa[1][0]),
                [a5] "=&v"(fa[1][1]), [a6] "=&v"(fa[1][2]), [a7] "=&v"(fa[1][3]), [b0] "=&v"(fb[0][0]), [b1] "=&v"(fb[0][1]),
                [b2] "=&v"(fb[0][2]), [b3] "=&v"(fb[0][3]), [b4] "=&v"(fb[1][0]), [b5] "=&v"(fb[1][1]), [b6] "=&v"(fb[1][2]),
                [b7] "=&v"(fb[1][3]), [c0] "=&v"(cv[0])
              : [aA] "v"(aA), [aB] "v"(aB), [aC] "v"(aC), [mA] "s"(mA), [mB] "s"(mB), [oA0] "v"(offA0), [oA1] "v"(offA1), [oB0] "v"(offB0),
                [oB1] "v"(offB1), [pA] "s"(pA), [pB] "s"(pB)
              : "memory", "scc");
          __builtin_amdgcn_sched_barrier(0);
          n_new = NDMA - UPC;
          advance();
        } else if (UPC == 2 && fused) {
          // the K = 64 form: two C units per chunk (slots mB + 8192 and mB + 12288)
          typedef __attribute__((address_space(3))) cplx* lds_cplx_t;
          cplx* S2 = ring + st2 * STG;
          const unsigned mA = (unsigned)(size_t)(lds_cplx_t)(S2 + wave * AST), mB = (unsigned)(size_t)(lds_cplx_t)(S2 + BOF + wave * 64);
          const int u0 = p_ch * 2, u1 = u0 + 1;
          const char* pC0 = pC + ((long long)(4 * (u0 & 3)) * lda + (u0 >> 2) * 16) * (long long)sizeof(cplx);
          const char* pC1 = pC + ((long long)(4 * (u1 & 3)) * lda + (u1 >> 2) * 16) * (long long)sizeof(cplx);
          asm volatile(
              "ds_read_b128 %[b0], %[aB]\n\tds_read_b128 %[b1], %[aB] offset:256\n\tds_read_b128 %[b2], %[aB] offset:512\n\tds_read_b128 %[b3], %[aB] offset:768\n\t"
              "ds_read_b128 %[b4], %[aB] offset:4096\n\tds_read_b128 %[b5], %[aB] offset:4352\n\tds_read_b128 %[b6], %[aB] offset:4608\n\tds_read_b128 %[b7], %[aB] offset:4864\n\t"
              "ds_read_b128 %[a0], %[aA]\n\tds_read_b128 %[a1], %[aA] offset:64\n\tds_read_b128 %[a2], %[aA] offset:128\n\tds_read_b128 %[a3], %[aA] offset:192\n\t"
              "ds_read_b128 %[a4], %[aA] offset:4352\n\tds_read_b128 %[a5], %[aA] offset:4416\n\tds_read_b128 %[a6], %[aA] offset:4480\n\tds_read_b128 %[a7], %[aA] offset:4544\n\t"
              "ds_read_b128 %[c0], %[aC]\n\tds_read_b128 %[c1], %[aC] offset:4096\n\t"
              "s_mov_b32 %[keep], m0\n\t"
              "s_mov_b32 m0, %[mA]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[oA0], %[pA]\n\t"
              "s_add_u32 m0, %[mA], 4352\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[oA1], %[pA]\n\t"
              "s_mov_b32 m0, %[mB]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[oB0], %[pB]\n\t"
              "s_add_u32 m0, %[mB], 4096\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[oB1], %[pB]\n\t"
              "s_add_u32 m0, %[mB], 8192\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[oC], %[pC0]\n\t"
              "s_add_u32 m0, %[mB], 12288\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[oC], %[pC1]\n\t"
              "s_mov_b32 m0, %[keep]\n\t"
              "s_waitcnt lgkmcnt(0)"
              : [keep] "=&s"(m0_keep), [a0] "=&v"(fa[0][0]), [a1] "=&v"(fa[0][1]), [a2] "=&v"(fa[0][2]), [a3] "=&v"(fa[0][3]), [a4] "=&v"(fa[1][0]),
                [a5] "=&v"(fa[1][1]), [a6] "=&v"(fa[1][2]), [a7] "=&v"(fa[1][3]), [b0] "=&v"(fb[0][0]), [b1] "=&v"(fb[0][1]),
                [b2] "=&v"(fb[0][2]), [b3] "=&v"(fb[0][3]), [b4] "=&v"(fb[1][0]), [b5] "=&v"(fb[1][1]), [b6] "=&v"(fb[1][2]),
                [b7] "=&v"(fb[1][3]), [c0] "=&v"(cv[0]), [c1] "=&v"(cv[UPC - 1])
              : [aA] "v"(aA), [aB] "v"(aB), [aC] "v"(aC), [mA] "s"(mA), [mB] "s"(mB), [oA0] "v"(offA0), [oA1] "v"(offA1), [oB0] "v"(offB0),
                [oB1] "v"(offB1), [oC] "v"(offC), [pA] "s"(pA), [pB] "s"(pB), [pC0] "s"(pC0), [pC1] "s"(pC1)
              : "memory", "scc");
          __builtin_amdgcn_sched_barrier(0);
          n_new = NDMA;
          advance();
        } else if constexpr (UPC == 1) {
          asm volatile(
              "ds_read_b128 %[b0], %[aB]\n\tds_read_b128 %[b1], %[aB] offset:256\n\tds_read_b128 %[b2], %[aB] offset:512\n\tds_read_b128 %[b3], %[aB] offset:768\n\t"
              "ds_read_b128 %[b4], %[aB] offset:4096\n\tds_read_b128 %[b5], %[aB] offset:4352\n\tds_read_b128 %[b6], %[aB] offset:4608\n\tds_read_b128 %[b7], %[aB] offset:4864\n\t"
              "ds_read_b128 %[a0], %[aA]\n\tds_read_b128 %[a1], %[aA] offset:64\n\tds_read_b128 %[a2], %[aA] offset:128\n\tds_read_b128 %[a3], %[aA] offset:192\n\t"
              "ds_read_b128 %[a4], %[aA] offset:4352\n\tds_read_b128 %[a5], %[aA] offset:4416\n\tds_read_b128 %[a6], %[aA] offset:4480\n\tds_read_b128 %[a7], %[aA] offset:4544\n\t"
              "ds_read_b128 %[c0], %[aC]\n\t"
              "s_waitcnt lgkmcnt(0)"
              : [a0] "=&v"(fa[0][0]), [a1] "=&v"(fa[0][1]), [a2] "=&v"(fa[0][2]), [a3] "=&v"(fa[0][3]), [a4] "=&v"(fa[1][0]),
                [a5] "=&v"(fa[1][1]), [a6] "=&v"(fa[1][2]), [a7] "=&v"(fa[1][3]), [b0] "=&v"(fb[0][0]), [b1] "=&v"(fb[0][1]),
                [b2] "=&v"(fb[0][2]), [b3] "=&v"(fb[0][3]), [b4] "=&v"(fb[1][0]), [b5] "=&v"(fb[1][1]), [b6] "=&v"(fb[1][2]),
                [b7] "=&v"(fb[1][3]), [c0] "=&v"(cv[0])
              : [aA] "v"(aA), [aB] "v"(aB), [aC] "v"(aC)
              : "memory");
        } else {
          asm volatile(
              "ds_read_b128 %[b0], %[aB]\n\tds_read_b128 %[b1], %[aB] offset:256\n\tds_read_b128 %[b2], %[aB] offset:512\n\tds_read_b128 %[b3], %[aB] offset:768\n\t"
              "ds_read_b128 %[b4], %[aB] offset:4096\n\tds_read_b128 %[b5], %[aB] offset:4352\n\tds_read_b128 %[b6], %[aB] offset:4608\n\tds_read_b128 %[b7], %[aB] offset:4864\n\t"
              "ds_read_b128 %[a0], %[aA]\n\tds_read_b128 %[a1], %[aA] offset:64\n\tds_read_b128 %[a2], %[aA] offset:128\n\tds_read_b128 %[a3], %[aA] offset:192\n\t"
              "ds_read_b128 %[a4], %[aA] offset:4352\n\tds_read_b128 %[a5], %[aA] offset:4416\n\tds_read_b128 %[a6], %[aA] offset:4480\n\tds_read_b128 %[a7], %[aA] offset:4544\n\t"
              "ds_read_b128 %[c0], %[aC]\n\tds_read_b128 %[c1], %[aC] offset:4096\n\t"
              "s_waitcnt lgkmcnt(0)"
              : [a0] "=&v"(fa[0][0]), [a1] "=&v"(fa[0][1]), [a2] "=&v"(fa[0][2]), [a3] "=&v"(fa[0][3]), [a4] "=&v"(fa[1][0]),
                [a5] "=&v"(fa[1][1]), [a6] "=&v"(fa[1][2]), [a7] "=&v"(fa[1][3]), [b0] "=&v"(fb[0][0]), [b1] "=&v"(fb[0][1]),
                [b2] "=&v"(fb[0][2]), [b3] "=&v"(fb[0][3]), [b4] "=&v"(fb[1][0]), [b5] "=&v"(fb[1][1]), [b6] "=&v"(fb[1][2]),
                [b7] "=&v"(fb[1][3]), [c0] "=&v"(cv[0]), [c1] "=&v"(cv[UPC - 1])
              : [aA] "v"(aA), [aB] "v"(aB), [aC] "v"(aC)
              : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      BIEM_TR(4)
      // 3M operand sums and the C-slice additions BEFORE the MFMA block (measured: issued inside the block, in the shadow of
      // this wave's own MFMAs, the FP64 adds cost more - they share the FP64 pipe with the MFMAs and the C additions then
      // wait for accumulators in flight: 57.6 vs 63.1 TFLOP/s; only the integer address work below is free in there).
      double fbs[2][4], fas[2][4];
#pragma unroll
      for (int k4 = 0; k4 < 2; ++k4) {
#ifdef BIEM_ABL_NOSUMS
#pragma unroll
        for (int n = 0; n < 4; ++n) fbs[k4][n] = fb[k4][n].x;
#pragma unroll
        for (int g = 0; g < 4; ++g) fas[k4][g] = fa[k4][g].x;
#else
#pragma unroll
        for (int n = 0; n < 4; ++n) fbs[k4][n] = fb[k4][n].x + fb[k4][n].y;
#pragma unroll
        for (int g = 0; g < 4; ++g) fas[k4][g] = fa[k4][g].x + fa[k4][g].y;
#endif
      }
      // this chunk's C units (u = c*UPC + i -> column group u>>2, row quad u&3) join their accumulators.  Which
      // accumulator that is depends on c: a switch over c made hipcc merge all 32 accumulators through v_mov_b64 copies
      // behind the MFMA block (~1300 stalled cycles per chunk, found with tools/gemm_trace), an fma(value, sel_u, acc_u) over
      // all units cost 32 FP64 VALU instructions that compete with the MFMAs for the FP64 pipe.  A dynamically indexed
      // register array compiles to s_set_gpr_idx + v_mov (indirect VGPR addressing): 3 FP64 adds per unit.
#ifndef BIEM_ABL_NOCADD
      if (NCC == NCH || c < NCC) {
#pragma unroll
        for (int i = 0; i < UPC; ++i) {
          (&N1[0][0])[c * UPC + i] += cv[i].x;
          (&N3[0][0])[c * UPC + i] += cv[i].x + cv[i].y;
        }
      }
#endif
      // the MFMA block runs at low priority, everything else at high (the partner's SALU / LDS / VMEM phase slips between
      // this wave's MFMAs); the next chunk's fragment addresses are computed in its shadow
      mfma_fence();
      BIEM_PRIO_M();
      mfma_fence();
      BIEM_TR(5)
      {
        const int stn = st == 2 ? 0 : st + 1;
        const unsigned sbase = (unsigned)(size_t)(lds_cptr_t)(ring + stn * STG);     // wave-uniform
        aA = sbase + (unsigned)(fao * (int)sizeof(cplx));
        aB = sbase + (unsigned)(fbo * (int)sizeof(cplx));
        aC = sbase + (unsigned)((COF + tid) * (int)sizeof(cplx));
      }
#ifdef BIEM_ABL_NOMFMA     // timing ablation: data movement only
#pragma unroll
      for (int k4 = 0; k4 < 2; ++k4) {
#pragma unroll
        for (int n = 0; n < 4; ++n) asm volatile("" ::"v"(fb[k4][n].x), "v"(fb[k4][n].y), "v"(fbs[k4][n]));
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" ::"v"(fa[k4][g].x), "v"(fa[k4][g].y), "v"(fas[k4][g]));
      }
#else
#pragma unroll
      for (int k4 = 0; k4 < 2; ++k4) {
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int g = 0; g < 4; ++g) mfma_acc_neg(N1[n][g], fa[k4][g].x, fb[k4][n].x);
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int g = 0; g < 4; ++g) mfma_acc(P2[n][g], fa[k4][g].y, fb[k4][n].y);
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int g = 0; g < 4; ++g) mfma_acc_neg(N3[n][g], fas[k4][g], fbs[k4][n]);
      }
#endif
      mfma_fence();
      BIEM_PRIO_O();
      BIEM_TR(6)
      __builtin_amdgcn_sched_barrier(0);
      BIEM_TR(7)
      BIEM_TR_NEXT()
      st = st == 2 ? 0 : st + 1;
      if (c == 1) stores_pending = 0;
    }
    // tile finished: Cr' = N1 + P2, Ci' = N3 - N1 + P2; plain stores stay in flight while the next tile starts.
    // Full tiles address their 16 stores as (scalar tile origin + scalar unit offset) + the constant per-lane 32-bit offset
    // the C-slice DMA uses: no VALU address arithmetic (the generic form cost ~10 VALU instructions per store, three of them
    // integer multiplies, in the phase where the SIMD partner streams MFMAs: 10 % of the kernel, tools/gemm_trace ablation)
    cplx* Cs = A + (size_t)cs * sys_stride;
    const int row0 = tg.row_begin + cty * BM3, col0 = tg.col_begin + ctx * BN3;
    const bool full = row0 + BM3 <= n_pad && col0 + BN3 <= n_cols;
    if (full && tg.pout != nullptr && ctx == tg.pcol_tx) {
      // P[(16 n + l15)][row0 + 16 w + 4 g + l4]: 64-byte runs (4 rows) per lane quad; the 64-column panel tiles are always full
      cplx* Po = tg.pout + (size_t)cs * tg.pout_stride + (size_t)l15 * tg.pout_ld + (row0 + wave * 16 + l4);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const cplx v = make_double2(N1[n][g] + P2[n][g], N3[n][g] - N1[n][g] + P2[n][g]);
          Po[(size_t)(16 * n) * tg.pout_ld + 4 * g] = v;
          N1[n][g] = 0.0; P2[n][g] = 0.0; N3[n][g] = 0.0;
        }
      }
    } else if (full) {
      char* tb = (char*)(Cs + (size_t)row0 * lda + col0);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const long long du = ((long long)(4 * g) * lda + n * 16) * (long long)sizeof(cplx);
#ifdef BIEM_ABL_NOEPI
          asm volatile("" ::"v"(N1[n][g]), "v"(P2[n][g]), "v"(N3[n][g]));   // keep the MFMAs alive
#elif defined(BIEM_ABL_ONESTORE)
          { const cplx v = make_double2(N1[n][g] + P2[n][g], N3[n][g] - N1[n][g] + P2[n][g]);
            if (n == 3 && g == 3) *(cplx*)(tb + du + offC) = v; else asm volatile("" ::"v"(v.x), "v"(v.y)); }
#elif defined(BIEM_ABL_NOSTORE)
          { const cplx v = make_double2(N1[n][g] + P2[n][g], N3[n][g] - N1[n][g] + P2[n][g]);
            asm volatile("" ::"v"(v.x), "v"(v.y)); }
#else
          const cplx v = make_double2(N1[n][g] + P2[n][g], N3[n][g] - N1[n][g] + P2[n][g]);
          *(cplx*)(tb + du + offC) = v;
#endif
          N1[n][g] = 0.0; P2[n][g] = 0.0; N3[n][g] = 0.0;
        }
      }
    } else {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int col = col0 + n * 16 + l15;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = row0 + wave * 16 + 4 * g + l4;
          const cplx v = make_double2(N1[n][g] + P2[n][g], N3[n][g] - N1[n][g] + P2[n][g]);
          if (col < n_cols && row < n_pad) Cs[(size_t)row * lda + col] = v;
          N1[n][g] = 0.0; P2[n][g] = 0.0; N3[n][g] = 0.0;
        }
      }
    }
    if (++c_tiles == p_tiles) break;                     // the producer started no further tile: done
    stores_pending = full ? 1 : 2;
    cs = p_s; cty = p_ty; ctx = p_tx;
  }
#ifdef BIEM_TR_STAMPS
  __syncthreads();
  if (blockIdx.x < 4) {     // 4 workgroups x 4 waves
    if (lane == 0) s_tr[wave * 512 + 7] = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);     // HW_ID[15:0]
    __syncthreads();
    for (int i = tid; i < 4 * 64 * 8; i += 256) g_gemm_trace[blockIdx.x * 4 + (i >> 9)][(i >> 3) & 63][i & 7] = s_tr[i];
  }
#endif
}

// C[row_begin:row_end, col_begin:col_end] -= P[0:kd]^T (rows of the region) * M[brow:brow+kd, cols of the region]
static int launch_gemm_stream(hipStream_t st, int nb, cplx* A, long long lda, long long sys_stride, const cplx* Pw, long long ldp,
                               long long p_stride, int row_begin, int row_end, int col_begin, int col_end, int brow, int kd,
                               int prof_class = PK_GEMM, double prof_work = -1.0, cplx* pout = nullptr, long long pout_ld = 0,
                               long long pout_stride = 0, int pcol_tx = 0, const int* tri_map = nullptr, bool upper = false) {
  const int tri = tri_map != nullptr ? (upper ? 2 : 1) : 0;
  const int rrows = row_end - row_begin, rcols = col_end - col_begin;
  if (rrows <= 0 || rcols <= 0) return BIEM_OK;
  TileGrid tg;
  tg.pout = pout; tg.pout_ld = pout_ld; tg.pout_stride = pout_stride; tg.pcol_tx = pcol_tx; tg.tri = tri;
  tg.ty_n = (rrows + BM3 - 1) / BM3; tg.tx_n = (rcols + BN3 - 1) / BN3;
  tg.per_sys = tri ? tg.ty_n * (tg.ty_n + 1) / 2 : tg.ty_n * tg.tx_n; tg.full_bands = tg.ty_n / 8; tg.ntiles = tg.per_sys * nb;
  tg.tri_map = tri_map; tg.tri_full = 32 * tg.full_bands * tg.full_bands + 4 * tg.full_bands;
  // Workgroups with the same blockIdx % 8 (one XCD) sweep blocks of 64 consecutive tiles together (shared operand panels in that
  // XCD's L2).  A small launch - one system, or the last groups of a factorisation - would leave most workgroups without a
  // tile that way (63 tiles: all in block 0, i.e. on the 8 workgroups of one label, 8 tiles each in sequence: 204 us for a
  // K = 192 tile row of one N = 4064 system): blocks of 8 tiles then.
  // Up to 512 tiles every tile has its own workgroup: tile = blockIdx (a block of ONE tile per label and round) - with blocks of 8 a
  // 33-tile launch (the strip of one system half-way through its factorisation) ran on the workgroups of 5 labels, several of them
  // taking two tiles in sequence while three quarters of the grid had none: 29.5 us per K = 64 strip launch instead of 13.
  tg.blk_sh = tg.ntiles <= 512 ? 0 : tg.ntiles < 2048 ? 3 : 6;
  tg.per_sys_magic = ((1ULL << 40) + (unsigned long long)tg.per_sys - 1) / (unsigned long long)tg.per_sys;
  if (tg.ntiles >= (1 << 25)) { set_error("biem_lu: more than 2^25 tiles in one update launch"); return BIEM_ERR_ARG; }   // unreachable: 2^25 tiles are 2 TB of matrix
  tg.row_begin = row_begin; tg.row_end = row_end; tg.col_begin = col_begin; tg.col_end = col_end; tg.brow = brow;
  const int cap = 512;                         // persistent grid: 2 workgroups per CU
  int want = (tg.ntiles + 7) / 8 * 8;          // one workgroup per tile up to the cap, multiple of 8
  int grid = want < cap ? want : cap;
  ProfScope ps(prof_class, st, prof_work >= 0.0 ? prof_work : 8.0 * (double)nb * (tri ? (double)tg.per_sys * BM3 * BN3 : rrows * (double)rcols) * kd);
  if (kd == 64)
    hipLaunchKernelGGL(k_gemm3m_pipe<64>, dim3(grid), dim3(256), 0, st, A, lda, sys_stride, Pw, ldp, p_stride, tg);
  else if (kd == 256)
    hipLaunchKernelGGL(k_gemm3m_pipe<256>, dim3(grid), dim3(256), 0, st, A, lda, sys_stride, Pw, ldp, p_stride, tg);
  else if (kd == 192)
    hipLaunchKernelGGL(k_gemm3m_pipe<192>, dim3(grid), dim3(256), 0, st, A, lda, sys_stride, Pw, ldp, p_stride, tg);
  else
    hipLaunchKernelGGL(k_gemm3m_pipe<128>, dim3(grid), dim3(256), 0, st, A, lda, sys_stride, Pw, ldp, p_stride, tg);
  return BIEM_OK;
}

// W = I - L11^{-1} for the unit-lower 64 x 64 diagonal block of a panel, stored [k][i] (the MFMA A-operand order), so that
//   U12 = L11^{-1} A12 = A12 - W A12
// runs on the streaming zgemm (C -= A*B with B = C's own rows; a tile reads all of its 64 x 128 block before it stores).
// One 64-thread workgroup per system; thread c owns column c of the inverse (forward substitution in LDS).
__global__ void __launch_bounds__(64) k_inv_l11(const cplx* __restrict__ Pj, long long ldp, long long p_stride, int j,
                                                 cplx* __restrict__ Winv) {
  extern __shared__ cplx sm[];
  cplx* sLT = sm;               // sLT[k][r] = L[r][k]
  cplx* sW = sm + NB * NB;      // sW[r][c]
  const int s = blockIdx.x, c = threadIdx.x;
  const cplx* Ps = Pj + (size_t)s * p_stride;
  for (int k = 0; k < NB; ++k) {
    sLT[k * NB + c] = Ps[(size_t)k * ldp + j + c];
    sW[k * NB + c] = (k == c) ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0);
  }
  __syncthreads();
  for (int r = 1; r < NB; ++r) {
    cplx acc = make_double2(0.0, 0.0);
    for (int k = 0; k < r; ++k) acc = cfma(sLT[k * NB + r], sW[k * NB + c], acc);
    if (r > c) sW[r * NB + c] = make_double2(-acc.x, -acc.y);
  }
  __syncthreads();
  cplx* Wo = Winv + (size_t)s * NB * NB;
  for (int k = 0; k < NB; ++k) {
    cplx v = sW[c * NB + k];                                   // inverse[i = c][k]
    Wo[k * NB + c] = (c > k) ? make_double2(-v.x, -v.y) : make_double2(0.0, 0.0);
  }
}

// apply the row interchanges of the second panel of a block to the stored multipliers of the first (P columns 0..NB-1)
__global__ void __launch_bounds__(64) k_swap_p(cplx* __restrict__ Pw, long long ldp, long long p_stride, int n_pad, int j, int ncols,
                                                const int* __restrict__ ipiv) {
  // the interchanges of the panel at column j on the multipliers of the EARLIER panels of the same K = 256 group
  // (workspace columns 0 .. ncols-1): their trailing update is still to come and needs the rows in their final order
  const int s = blockIdx.x;
  for (int c = threadIdx.x; c < ncols; c += 64) {
    cplx* col = Pw + (size_t)s * p_stride + (size_t)c * ldp;
    for (int q = 0; q < NB; ++q) {
      int p = ipiv[(size_t)s * n_pad + j + q];
      if (p != j + q) { cplx a = col[j + q]; col[j + q] = col[p]; col[p] = a; }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// back substitution with U (row-major), block size BS
// ---------------------------------------------------------------------------------------------
// The right-hand sides are addressed as F[s * f_stride + row * ldf + q]: the augmented columns of the matrix itself
// (F = A + n_pad, ldf = lda, f_stride = sys_stride) in the fused solve, a separate array in biem_lu_solve.
// value of lane i (WAVE-UNIFORM i) in every lane: two v_readlane_b32 through the scalar file instead of the LDS crossbar round trip
// of ds_bpermute (__shfl) - these broadcasts sit on the dependent chain of the elimination / substitution steps
__device__ inline double lane_bcast(double v, int i) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), i), hi = __builtin_amdgcn_readlane(__double2hiint(v), i);
  return __hiloint2double(hi, lo);
}
// diagonal block: x = U[jr:jr+BS, jr:jr+BS]^{-1} y, one 64-thread workgroup per (system, rhs)
__global__ void __launch_bounds__(64) k_back_diag(const cplx* __restrict__ A, long long lda, long long sys_stride, cplx* __restrict__ F,
                                                   long long ldf, long long f_stride, int jr) {
  // the 64 x 64 block goes through LDS once (coalesced rows): read from global memory element by element inside the 64 dependent
  // steps it cost 24 us per block; one wave, so the steps need no barrier - x_c travels by a lane broadcast
  __shared__ cplx sU[BS][BS + 1];
  const int s = blockIdx.x, q = blockIdx.y, r = threadIdx.x;
  const cplx* Ub = A + (size_t)s * sys_stride + (size_t)jr * lda + jr;
  for (int rr = 0; rr < BS; ++rr) sU[rr][r] = Ub[(size_t)rr * lda + r];
  cplx* Fq = F + (size_t)s * f_stride + q;
  cplx y = Fq[(size_t)(jr + r) * ldf];
  __syncthreads();
  const cplx inv = crecip(sU[r][r]);           // every lane its own diagonal entry, once
  for (int c = BS - 1; c >= 0; --c) {
    const cplx t = cmul(y, inv);               // lane c holds x_c
    const cplx xc = make_double2(lane_bcast(t.x, c), lane_bcast(t.y, c));
    if (r == c) y = xc;
    if (r < c) y = cfnma(sU[r][c], xc, y);
  }
  Fq[(size_t)(jr + r) * ldf] = y;
}

// forward counterpart (stored factors, biem_lu_solve): the 64 interchanges of the panel at column j on the right-hand side, then
// y = L11^{-1} f with the unit-lower diagonal block; one 64-thread workgroup per (system, rhs)
__global__ void __launch_bounds__(64) k_fwd_diag(const cplx* __restrict__ A, long long lda, long long sys_stride, const int* __restrict__ ipiv,
                                                  int n_pad, cplx* __restrict__ F, long long ldf, long long f_stride, int j) {
  __shared__ cplx sx;
  const int s = blockIdx.x, q = blockIdx.y, r = threadIdx.x;
  cplx* Fq = F + (size_t)s * f_stride + q;
  if (r == 0) {
    for (int c = 0; c < NB; ++c) {
      const int p = ipiv[(size_t)s * n_pad + j + c];
      if (p != j + c) { const cplx a = Fq[(size_t)(j + c) * ldf], b = Fq[(size_t)p * ldf]; Fq[(size_t)(j + c) * ldf] = b; Fq[(size_t)p * ldf] = a; }
    }
  }
  __syncthreads();
  const cplx* Lrow = A + (size_t)s * sys_stride + (size_t)(j + r) * lda + j;
  cplx y = Fq[(size_t)(j + r) * ldf];
  for (int c = 0; c < NB - 1; ++c) {
    if (r == c) sx = y;
    __syncthreads();
    if (r > c) y = cfnma(Lrow[c], sx, y);
    __syncthreads();
  }
  Fq[(size_t)(j + r) * ldf] = y;
}

// rows [row_begin, row_end): y[i] -= M[i, jr:jr+64] . x[jr:jr+64]; one wave per row (back substitution: the rows above the
// solved block with M = U; forward substitution with stored factors: the rows below the panel with M = L)
constexpr int BACK_ROWS = 16;   // rows per workgroup (4 per wave)
// With `info` given (row form of the symmetric path) the pass also checks the entries it reads, u_ic of the strips right of the diagonal
// blocks: |u_ic|^2 <= inv_rel2 |u_ii|^2 (every multiplier l_ci = u_ic / u_ii within 1 / rel; NaN-safe) else info = -(first row of the
// 64-row panel + 1), and max |u_ii u_ic| into the growth slot.
__global__ void __launch_bounds__(256) k_back_update(const cplx* __restrict__ A, long long lda, long long sys_stride, cplx* __restrict__ F,
                                                      long long ldf, long long f_stride, int nrhs, int jr, int row_begin, int row_end,
                                                      int* __restrict__ info = nullptr, unsigned long long* __restrict__ growth = nullptr,
                                                      double inv_rel2 = 0.0) {
  // The 64 solution values are strided by ldf in memory (one cache line each): they are gathered ONCE per workgroup into LDS
  // instead of once per row
  __shared__ cplx sx[BS];
  const int s = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const cplx* As = A + (size_t)s * sys_stride;
  cplx* Fs = F + (size_t)s * f_stride;
  // this wave's rows of the block column: loaded once, used by the checks and by every right-hand side
  cplx u[BACK_ROWS / 4];
  const int i0 = row_begin + blockIdx.x * BACK_ROWS + wave * (BACK_ROWS / 4);
#pragma unroll
  for (int k = 0; k < BACK_ROWS / 4; ++k) u[k] = (i0 + k < row_end) ? As[(size_t)(i0 + k) * lda + jr + lane] : make_double2(0.0, 0.0);
  if (info != nullptr) {
    double um2 = 0.0;
    bool badm = false;
#pragma unroll
    for (int k = 0; k < BACK_ROWS / 4; ++k) {
      if (i0 + k >= row_end) break;
      const cplx d = As[(size_t)(i0 + k) * lda + i0 + k];
      const double m2 = u[k].x * u[k].x + u[k].y * u[k].y, d2 = d.x * d.x + d.y * d.y;
      if (!(m2 <= inv_rel2 * d2)) badm = true;
      um2 = nan_max(um2, m2 * d2);
    }
    block_max_publish(sqrt(um2), growth + 2 * (size_t)s + 1);
    if (badm && info[s] == 0) info[s] = -((i0 / NB) * NB + 1);
  }
  for (int q = 0; q < nrhs; ++q) {
    if (q > 0) __syncthreads();
    if (threadIdx.x < BS) sx[threadIdx.x] = Fs[(size_t)(jr + threadIdx.x) * ldf + q];
    __syncthreads();
    const cplx x = sx[lane];
#pragma unroll
    for (int k = 0; k < BACK_ROWS / 4; ++k) {
      const int i = i0 + k;
      if (i >= row_end) break;
      const cplx v = cmul(u[k], x);
      double vr = v.x, vi = v.y;
      for (int o = 32; o > 0; o >>= 1) { vr += __shfl_down(vr, o, 64); vi += __shfl_down(vi, o, 64); }
      if (lane == 0) {
        cplx* y = Fs + (size_t)i * ldf + q;
        cplx t = *y;
        t.x -= vr; t.y -= vi;
        *y = t;
      }
    }
  }
}

// One block step of the back substitution with the STORED inverses of the diagonal blocks (few systems per call: the chain of
// 2 n / 64 dependent launches is what one system per call waits for).  The factorisation keeps W_b = I - U_bb^{-T} of every panel
// (stored [k][i] = delta_ki - (U_bb^{-1})[k][i]), so  x_b = U_bb^{-1} y_b = y_b - sum_{i >= k} W_b[k][i] y_i  is a 64 x 64 product that
// every workgroup of the update forms for itself - no 64-step triangular solve (k_back_diag: 12 us) and one launch per block instead
// of two.  y_b must not be overwritten while other workgroups read it: the solution goes to X[(s nrhs + q) n_pad + row] and is copied
// back at the end (k_rhs_compact).  The update of the rows above (and the checks of the entries it reads) is k_back_update's.
__global__ void __launch_bounds__(256) k_back_step(const cplx* __restrict__ A, long long lda, long long sys_stride, cplx* __restrict__ F,
                                                    long long ldf, long long f_stride, const cplx* __restrict__ Wall, long long w_stride,
                                                    cplx* __restrict__ X, int n_pad, int nrhs, int jr, int* __restrict__ info,
                                                    unsigned long long* __restrict__ growth, double inv_rel2) {
  __shared__ cplx sx[BS], syb[BS];
  const int s = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const cplx* As = A + (size_t)s * sys_stride;
  cplx* Fs = F + (size_t)s * f_stride;
  const cplx* Wb = Wall + (size_t)s * w_stride + (size_t)(jr / NB) * NB * NB;
  cplx u[BACK_ROWS / 4];
  const int i0 = blockIdx.x * BACK_ROWS + wave * (BACK_ROWS / 4);
#pragma unroll
  for (int k = 0; k < BACK_ROWS / 4; ++k) u[k] = (i0 + k < jr) ? As[(size_t)(i0 + k) * lda + jr + lane] : make_double2(0.0, 0.0);
  // this wave's 16 rows of W_b (lanes along i): loaded once, used by every right-hand side
  cplx wr[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) { const int k = wave * 16 + kk; wr[kk] = lane >= k ? Wb[k * NB + lane] : make_double2(0.0, 0.0); }
  // (all loads of the prologue are issued together: the right-hand side's block, the diagonal entries of the checks)
  cplx ynext = threadIdx.x < BS ? Fs[(size_t)(jr + threadIdx.x) * ldf] : make_double2(0.0, 0.0);
  cplx dg[BACK_ROWS / 4];
#pragma unroll
  for (int k = 0; k < BACK_ROWS / 4; ++k) dg[k] = (i0 + k < jr) ? As[(size_t)(i0 + k) * lda + i0 + k] : make_double2(1.0, 0.0);
  for (int q = 0; q < nrhs; ++q) {
    __syncthreads();
    if (threadIdx.x < BS) syb[threadIdx.x] = ynext;
    if (q + 1 < nrhs && threadIdx.x < BS) ynext = Fs[(size_t)(jr + threadIdx.x) * ldf + q + 1];
    __syncthreads();
    const cplx yl = syb[lane];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const cplx v = cmul(wr[kk], yl);
      double vr = v.x, vi = v.y;
      for (int o = 32; o > 0; o >>= 1) { vr += __shfl_down(vr, o, 64); vi += __shfl_down(vi, o, 64); }
      if (lane == 0) { const int k = wave * 16 + kk; sx[k] = make_double2(syb[k].x - vr, syb[k].y - vi); }
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < BS) X[((size_t)s * nrhs + q) * n_pad + jr + threadIdx.x] = sx[threadIdx.x];
    const cplx x = sx[lane];
#pragma unroll
    for (int k = 0; k < BACK_ROWS / 4; ++k) {
      const int i = i0 + k;
      if (i >= jr) break;
      const cplx v = cmul(u[k], x);
      double vr = v.x, vi = v.y;
      for (int o = 32; o > 0; o >>= 1) { vr += __shfl_down(vr, o, 64); vi += __shfl_down(vi, o, 64); }
      if (lane == 0) {
        cplx* y = Fs + (size_t)i * ldf + q;
        cplx t = *y;
        t.x -= vr; t.y -= vi;
        *y = t;
      }
    }
  }
  {   // the checks of the entries this workgroup read, behind the arithmetic the next launch waits for
    double um2 = 0.0;
    bool badm = false;
#pragma unroll
    for (int k = 0; k < BACK_ROWS / 4; ++k) {
      if (i0 + k >= jr) break;
      const double m2 = u[k].x * u[k].x + u[k].y * u[k].y, d2 = dg[k].x * dg[k].x + dg[k].y * dg[k].y;
      if (!(m2 <= inv_rel2 * d2)) badm = true;
      um2 = nan_max(um2, m2 * d2);
    }
    block_max_publish(sqrt(um2), growth + 2 * (size_t)s + 1);
    if (badm && info[s] == 0) info[s] = -((i0 / NB) * NB + 1);
  }
}

__global__ void k_zero_int(int* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

// ---------------------------------------------------------------------------------------------
// symmetric path, few right-hand sides: the forward elimination of the right-hand-side columns as matrix-vector work
// (a 64 x 64 MFMA tile per 64 rows is almost empty for one column, and the triangular solve needs no inverse).
// ---------------------------------------------------------------------------------------------
// the 64 rows of the panel at column j (workspace columns pc ..): y = L11^{-1} (f - L[rows, 0:kd] y_prev), kd = pc = the columns
// of the group's earlier panels (rows jg .. jg+kd).  One 64-thread workgroup per (system, right-hand side).
__global__ void __launch_bounds__(64) k_rhs_panel(cplx* __restrict__ A, long long lda, long long sys_stride, const cplx* __restrict__ Pw,
                                                   long long ldp, long long p_stride, int n_pad, int j, int jg, int pc) {
  __shared__ cplx sy[4 * NB];
  __shared__ cplx sx;
  const int s = blockIdx.x, q = blockIdx.y, r = threadIdx.x;
  cplx* F = A + (size_t)s * sys_stride + n_pad + q;
  const cplx* Pr = Pw + (size_t)s * p_stride + j + r;            // row j + r of the workspace, column k at Pr[k * ldp]
  for (int k = r; k < pc; k += NB) sy[k] = F[(size_t)(jg + k) * lda];
  __syncthreads();
  cplx y = F[(size_t)(j + r) * lda];
  for (int k = 0; k < pc; ++k) y = cfnma(Pr[(size_t)k * ldp], sy[k], y);
  for (int c = 0; c < NB - 1; ++c) {
    if (r == c) sx = y;
    __syncthreads();
    if (r > c) y = cfnma(Pr[(size_t)(pc + c) * ldp], sx, y);
    __syncthreads();
  }
  F[(size_t)(j + r) * lda] = y;
}

// rows below a group: f[i] -= L[i, 0:kd] y[jg : jg+kd]; a workgroup takes 64 rows, its four waves a quarter of the kd terms each
// (one thread per row over all kd terms left one system's update on n / 256 workgroups with 256 dependent loads per thread:
// 22 us per launch at N = 4064); the kd values of y in LDS
constexpr int RHS_UPD_ROWS = 64;
__global__ void __launch_bounds__(256) k_rhs_update(cplx* __restrict__ A, long long lda, long long sys_stride, const cplx* __restrict__ Pw,
                                                     long long ldp, long long p_stride, int n_pad, int row_begin, int jg, int kd) {
  __shared__ cplx sy[4 * NB];
  __shared__ cplx part[3][64];
  const int s = blockIdx.y, q = blockIdx.z;
  cplx* F = A + (size_t)s * sys_stride + n_pad + q;
  for (int k = threadIdx.x; k < kd; k += 256) sy[k] = F[(size_t)(jg + k) * lda];
  __syncthreads();
  const int lane = threadIdx.x & 63, kq = threadIdx.x >> 6;
  const int i = row_begin + blockIdx.x * RHS_UPD_ROWS + lane, ic = i < n_pad ? i : n_pad - 1;
  const cplx* Pr = Pw + (size_t)s * p_stride + ic;
  const int k0 = (kd >> 2) * kq, k1 = kq == 3 ? kd : k0 + (kd >> 2);       // kd is a multiple of 4 here (64 .. 256)
  cplx a0 = make_double2(0.0, 0.0), a1 = a0, a2 = a0, a3 = a0;
  int k = k0;
  for (; k + 3 < k1; k += 4) {
    a0 = cfma(Pr[(size_t)k * ldp], sy[k], a0);
    a1 = cfma(Pr[(size_t)(k + 1) * ldp], sy[k + 1], a1);
    a2 = cfma(Pr[(size_t)(k + 2) * ldp], sy[k + 2], a2);
    a3 = cfma(Pr[(size_t)(k + 3) * ldp], sy[k + 3], a3);
  }
  for (; k < k1; ++k) a0 = cfma(Pr[(size_t)k * ldp], sy[k], a0);
  const cplx sum = make_double2((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y));
  if (kq > 0) part[kq - 1][lane] = sum;
  __syncthreads();
  if (kq == 0 && i < n_pad) {
    cplx f = F[(size_t)i * lda];
    f.x -= (sum.x + part[0][lane].x) + (part[1][lane].x + part[2][lane].x);
    f.y -= (sum.y + part[0][lane].y) + (part[1][lane].y + part[2][lane].y);
    F[(size_t)i * lda] = f;
  }
}

// ---------------------------------------------------------------------------------------------
// panel factorisation without interchanges (symmetric path): the 64 x 64 diagonal block is factored by one workgroup per
// system (k_diag_nopiv, which also forms X = U11^{-1}); the rows below are L21 = A21 X, one thread per row over all CUs
// (k_panel_l21): 160 column passes per panel through HBM instead of the strips' 576, no per-column barriers.
// The acceptance test of the pivoted-path strips carries over unchanged: |pivot| >= rel * max |column below, updated| is
// |l_ic| <= 1 / rel for every multiplier; a violation marks the system (info = -(row + 1) of the panel's first row).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_diag_nopiv(cplx* __restrict__ Pj, long long ldp, long long p_stride, int n_pad, int j,
                                                     cplx* __restrict__ Xinv, int* __restrict__ ipiv, int* __restrict__ info, double rel,
                                                     unsigned long long* __restrict__ growth) {
  // 256 threads: row r = tid & 63, part = tid >> 6 (one wave each).  Elimination: the four parts share the trailing columns of a
  // step (c2 = c+1+part, +4, ...).  Inverse: X = U11^{-1} by anti-diagonals d = c - k (entries of one d are independent), the
  // sum over m of an entry split over the four parts and reduced through LDS.
  __shared__ cplx a[NB][NB + 1];     // a[c][r]: column c, row r (as in the workspace)
  __shared__ cplx x[NB][NB + 1];     // x[k][c] = (U11^{-1})[k][c]
  __shared__ cplx red[4][NB];
  __shared__ int bad;
  const int s = blockIdx.x, tid = threadIdx.x, r = tid & 63, part = tid >> 6;
  cplx* Ps = Pj + (size_t)s * p_stride + j;
  if (tid == 0) bad = 0;
  for (int c = part; c < NB; c += 4) { a[c][r] = Ps[(size_t)c * ldp + r]; x[c][r] = make_double2(0.0, 0.0); }
  if (part == 0) ipiv[(size_t)s * n_pad + j + r] = j + r;
  __syncthreads();
  for (int c = 0; c < NB; ++c) {
    const cplx piv = a[c][c];
    const double pa = fabs(piv.x) + fabs(piv.y);
    cplx l = make_double2(0.0, 0.0);
    if (r > c) {
      const cplx v = a[c][r];
      if (part == 0 && !(pa >= rel * (fabs(v.x) + fabs(v.y)))) bad = 1;
      l = cmul(v, crecip(piv));
      for (int c2 = c + 1 + part; c2 < NB; c2 += 4) a[c2][r] = cfnma(l, a[c2][c], a[c2][r]);
    } else if (r == c && part == 0 && !(pa > 0.0)) bad = 1;
    __syncthreads();                       // every part has read a[c][r]; the trailing columns are updated
    if (r > c && part == 0) a[c][r] = l;   // column c is not read again inside this loop
  }
  __syncthreads();
  // X: thread (k = r, part) works on the entry (k, k + d) of anti-diagonal d
  for (int d = 0; d < NB; ++d) {
    const int k = r, c = r + d;
    cplx acc = make_double2(0.0, 0.0);
    if (c < NB)
      for (int m = k + 1 + part; m <= c; m += 4) acc = cfma(a[m][k], x[m][c], acc);     // U[k][m] x[m][c]
    red[part][r] = acc;
    __syncthreads();
    if (part == 0 && c < NB) {
      cplx sum = red[0][r];
      sum.x += red[1][r].x + red[2][r].x + red[3][r].x; sum.y += red[1][r].y + red[2][r].y + red[3][r].y;
      cplx rhs = (d == 0) ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0);
      rhs.x -= sum.x; rhs.y -= sum.y;
      x[k][c] = cmul(rhs, crecip(a[k][k]));
    }
    __syncthreads();
  }
  double um = 0.0;                                   // U11 = the entries on and above the diagonal (row r <= column c)
  for (int c = part; c < NB; c += 4) { Ps[(size_t)c * ldp + r] = a[c][r]; if (r <= c) um = nan_max(um, cabs1(a[c][r])); }
  block_max_publish(um, growth + 2 * (size_t)s + 1);
  cplx* Xo = Xinv + (size_t)s * NB * NB;
  for (int k = part; k < NB; k += 4) Xo[k * NB + r] = x[k][r];
  if (tid == 0 && bad && info[s] == 0) info[s] = -(j + 1);
}

__global__ void __launch_bounds__(256) k_panel_l21(cplx* __restrict__ Pj, long long ldp, long long p_stride, int n_pad, int j,
                                                    const cplx* __restrict__ Xinv, int* __restrict__ info, double rel) {
  extern __shared__ cplx sX[];       // [k][c], 64 x 64
  const int s = blockIdx.y, tid = threadIdx.x;
  const cplx* Xs = Xinv + (size_t)s * NB * NB;
  for (int e = tid; e < NB * NB; e += 256) sX[e] = Xs[e];
  __syncthreads();
  const int i = j + NB + blockIdx.x * 256 + tid;
  if (i >= n_pad) return;
  cplx* Pr = Pj + (size_t)s * p_stride + i;
  double lmax = 0.0;
  // columns 32..63 first (they need a_0..a_63 and overwrite a_32..a_63), then 0..31 (a_0..a_31 only)
#pragma unroll 1
  for (int half = 1; half >= 0; --half) {
    const int cb = half * 32;
    cplx acc[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) acc[q] = make_double2(0.0, 0.0);
    if (half) {
#pragma unroll 4
      for (int k = 0; k < 32; ++k) {            // full 32 columns
        const cplx ak = Pr[(size_t)k * ldp];
#pragma unroll
        for (int q = 0; q < 32; ++q) acc[q] = cfma(ak, sX[k * NB + cb + q], acc[q]);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {            // triangular part: X[k][c] = 0 for c < k
      const cplx ak = Pr[(size_t)(cb + kk) * ldp];
#pragma unroll
      for (int q = 0; q < 32; ++q)
        if (q >= kk) acc[q] = cfma(ak, sX[(cb + kk) * NB + cb + q], acc[q]);
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      Pr[(size_t)(cb + q) * ldp] = acc[q];
      const double v = fabs(acc[q].x) + fabs(acc[q].y);
      if (!(v <= lmax)) lmax = v;                 // NaN-safe: a non-finite multiplier makes lmax NaN and marks the system
    }
  }
  if (!(lmax * rel <= 1.0) && info[s] == 0) info[s] = -(j + 1);
}

// U rows of a factored panel from its multipliers, symmetric path: U[j+i][c] = d_i L[c][i] for the columns c right of the panel
// (A = L D L^T, so U = D L^T needs no triangular solve and no pending updates).  P is column-major: both sides are contiguous in c.
// One workgroup: 256 columns x 16 of the panel's 64 rows; it also publishes max |U| of its entries (growth check).
__global__ void __launch_bounds__(256) k_u_from_l(cplx* __restrict__ A, long long lda, long long sys_stride, const cplx* __restrict__ Pj,
                                                   long long ldp, long long p_stride, int n_pad, int j, unsigned long long* __restrict__ growth) {
  const int s = blockIdx.z;
  const int c = j + NB + blockIdx.x * 256 + threadIdx.x;
  double um = 0.0;
  if (c < n_pad) {
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
      const int i = blockIdx.y * 16 + q;
      const cplx* Pi = Pj + (size_t)s * p_stride + (size_t)i * ldp;
      const cplx u = cmul(Pi[j + i], Pi[c]);
      A[(size_t)s * sys_stride + (size_t)(j + i) * lda + c] = u;
      um = nan_max(um, cabs1(u));
    }
  }
  block_max_publish(um, growth + 2 * (size_t)s + 1);
}

int launch_lu_factor_solve(int nb, int n_pad, int nrhs, double* d_A, long long lda, long long sys_stride, int* d_ipiv,
                           int* d_info, void* d_work, size_t work_bytes, hipStream_t st, bool keep_multipliers, bool symmetric,
                           bool amax_ready) {
  if (nb <= 0 || n_pad <= 0) return BIEM_OK;
  if (n_pad % NB) { set_error("biem_lu: n_pad=%d is not a multiple of %d (use biem_lu_npad)", n_pad, NB); return BIEM_ERR_ARG; }
  if (nrhs < 0 || lda < n_pad + nrhs) { set_error("biem_lu: lda < n_pad + nrhs"); return BIEM_ERR_ARG; }
  if (nb > 65535 || nrhs > 65535) { set_error("biem_lu: at most 65535 systems / right-hand sides per call (got %d / %d)", nb, nrhs); return BIEM_ERR_ARG; }
  if (work_bytes < lu_workspace_bytes(nb, n_pad, nrhs)) { set_error("biem_lu: workspace too small"); return BIEM_ERR_ARG; }
  cplx* A = (cplx*)d_A;
  cplx* Pw = (cplx*)d_work;
  const long long ldp = ldp_of(n_pad), p_stride = 4LL * NB * ldp;
  const int n_cols = n_pad + nrhs;
  hipLaunchKernelGGL(k_zero_int, dim3((nb + 63) / 64), dim3(64), 0, st, d_info, nb);
  const size_t strip_lds = (size_t)PW * STRIP_CACHE_ROWS * sizeof(cplx);
  BIEM_HIPCHK(hipFuncSetAttribute((const void*)k_panel_strip, hipFuncAttributeMaxDynamicSharedMemorySize, (int)strip_lds));
  // (per call, not once per process: the attribute belongs to the current device)
  BIEM_HIPCHK(hipFuncSetAttribute((const void*)k_inv_l11, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * NB * NB * sizeof(cplx))));
  BIEM_HIPCHK(hipFuncSetAttribute((const void*)k_panel_l21, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(NB * NB * sizeof(cplx))));

  // status of the update launches (the lambdas below cannot return it themselves): the first failure is kept and returned
  int gemm_rc = BIEM_OK;
  auto gemm = [&](auto&&... a) { const int r = launch_gemm_stream(a...); if (r != BIEM_OK && gemm_rc == BIEM_OK) gemm_rc = r; };
  cplx* Winv = Pw + (size_t)nb * p_stride;      // 64 x 64 per system: I - L11^{-1} (LU) / U11^{-1} (symmetric path)
  int* const tri_map_ws = (int*)(Winv + (size_t)nb * NB * NB);      // tile map of the triangular updates, then the growth slots
  unsigned long long* growth = lu_growth_slots(d_work, nb, n_pad);  // [nb][2]: max |A|, max |U| (symmetric path)
  // BIEM_LDLT_PIVOT_REL (tests): acceptance threshold of the diagonal pivots (multipliers <= 1 / threshold); 1e30 rejects every system
  double nopiv = 0.0;
  double growth_max = GROWTH_MAX;     // BIEM_LDLT_GROWTH_MAX (tests): accepted max |U| / max |A|
  if (symmetric) {
    const char* e = getenv("BIEM_LDLT_PIVOT_REL"); nopiv = e ? atof(e) : NOPIV_REL; if (!(nopiv > 0.0)) nopiv = NOPIV_REL;
    const char* g = getenv("BIEM_LDLT_GROWTH_MAX"); if (g && atof(g) > 0.0) growth_max = atof(g);
  }
  // factor the 64-column panel at column j, multipliers into P columns [pc, pc + NB)
  auto panel = [&](int j, int pc, bool in_workspace = false) {
    cplx* Pj = Pw + (size_t)pc * ldp;
    const int rows = n_pad - j;
    ProfScope ps(PK_PANEL, st, 4.0 * (double)nb * rows * NB * NB);
    if (!in_workspace)     // (panels b, c, d of a group arrive in the workspace straight from the update that produced them)
      hipLaunchKernelGGL(k_panel_load, dim3((rows + TR - 1) / TR, nb), dim3(256), 0, st, A, lda, sys_stride, Pj, ldp, p_stride, n_pad, j);
    if (symmetric) {
      // no interchanges: diagonal block in one workgroup per system, then L21 = A21 U11^{-1} over all CUs
      hipLaunchKernelGGL(k_diag_nopiv, dim3(nb), dim3(256), 0, st, Pj, ldp, p_stride, n_pad, j, Winv, d_ipiv, d_info, nopiv, growth);
      if (rows > NB)
        hipLaunchKernelGGL(k_panel_l21, dim3((rows - NB + 255) / 256, nb), dim3(256), NB * NB * sizeof(cplx), st, Pj, ldp, p_stride, n_pad, j,
                           Winv, d_info, nopiv);
    } else
    // strips in pairs: after the first strip only the second strip's 8 columns are updated (rank 8); the columns right of
    // the pair get both strips' updates as ONE rank-16 pass (336 instead of 504 column passes per panel through HBM)
    for (int c0 = 0; c0 < NB; c0 += 2 * PW) {
      hipLaunchKernelGGL(k_panel_strip, dim3(nb), dim3(1024), strip_lds, st, Pj, ldp, p_stride, n_pad, j, c0, 0, d_ipiv, d_info);
      int below = n_pad - (j + c0 + PW);
      if (below > 0)
        hipLaunchKernelGGL(k_panel_update<PW>, dim3((below + 255) / 256, nb), dim3(256), 0, st, Pj, ldp, p_stride, n_pad, j, c0, PW);
      hipLaunchKernelGGL(k_panel_strip, dim3(nb), dim3(1024), strip_lds, st, Pj, ldp, p_stride, n_pad, j, c0 + PW, 1, d_ipiv, d_info);
      below = n_pad - (j + c0 + 2 * PW);
      const int ncols = NB - (c0 + 2 * PW);
      if (ncols > 0 && below > 0)
        hipLaunchKernelGGL(k_panel_update<2 * PW>, dim3((below + 255) / 256, nb), dim3(256), 0, st, Pj, ldp, p_stride, n_pad, j, c0, ncols);
    }
    // back to the row-major matrix: everything (factors for the caller) or only the 64 rows of the diagonal block - U11 is
    // all the rest of the solve reads from these columns (the trailing updates take L21 from the panel workspace)
    const int srows = keep_multipliers ? rows : (rows < NB ? rows : NB);
    hipLaunchKernelGGL(k_panel_store, dim3((srows + TR - 1) / TR, nb), dim3(256), 0, st, A, lda, sys_stride, Pj, ldp, p_stride,
                       keep_multipliers ? n_pad : j + srows, j);
    if (pc > 0 && !symmetric) hipLaunchKernelGGL(k_swap_p, dim3(nb), dim3(64), 0, st, Pw, ldp, p_stride, n_pad, j, pc, d_ipiv);
  };
  // the panel's row interchanges on the columns right of it
  auto swap_right = [&](int j) {
    const int rcols = n_cols - (j + NB);
    if (rcols <= 0) return;
    ProfScope ps(PK_SWAP, st, 64.0 * (double)nb * NB * rcols);
    hipLaunchKernelGGL(k_swap, dim3((rcols + 255) / 256, nb), dim3(256), 0, st, A, lda, sys_stride, n_pad, n_cols, j, d_ipiv, 0);
  };
  // U row block: M[j:j+NB, j+NB:] <- L11^{-1} M[j:j+NB, j+NB:]
  auto trsm = [&](int j, int pc, int col_begin = -1) {
    if (col_begin < 0) col_begin = j + NB;
    const int rcols = n_cols - col_begin;
    if (rcols <= 0) return;
    const double work = 4.0 * (double)nb * NB * NB * rcols;
    {
      ProfScope ps(PK_TRSM, st, 0.0);
      hipLaunchKernelGGL(k_inv_l11, dim3(nb), dim3(64), 2 * NB * NB * sizeof(cplx), st, Pw + (size_t)pc * ldp, ldp, p_stride, j, Winv);
    }
    // A operand W[k][i], i = row - j: hand the kernel the base shifted by -j rows (only rows j .. j+63 are addressed)
    gemm(st, nb, A, lda, sys_stride, Winv - j, NB, (long long)NB * NB, j, j + NB, col_begin, n_cols, j, NB, PK_TRSM, work);
  };
  // symmetric path: the panel's U rows over the matrix columns by transposition, over the right-hand sides by the solve
  const bool rhs_gemv = nrhs > 0 && nrhs <= 8;   // few right-hand sides: matrix-vector kernels instead of nearly empty MFMA tiles
  auto u_rows_sym = [&](int j, int jg, int pc) {
    const int rcols = n_pad - (j + NB);
    if (rcols > 0) {
      ProfScope ps(PK_TRSM, st, 0.0);
      hipLaunchKernelGGL(k_u_from_l, dim3((rcols + 255) / 256, NB / 16, nb), dim3(256), 0, st, A, lda, sys_stride, Pw + (size_t)pc * ldp, ldp, p_stride, n_pad, j, growth);
    }
    if (rhs_gemv) {
      ProfScope ps(PK_TRSM, st, 0.0);
      hipLaunchKernelGGL(k_rhs_panel, dim3(nb, nrhs), dim3(64), 0, st, A, lda, sys_stride, Pw, ldp, p_stride, n_pad, j, jg, pc);
    } else if (nrhs > 0) {
      if (pc > 0) gemm(st, nb, A, lda, sys_stride, Pw, ldp, p_stride, j, j + NB, n_pad, n_cols, jg, pc, PK_OTHER);
      trsm(j, pc, n_pad);
    }
  };

  if (symmetric) {
    // the tile map of the triangular updates (full bands of the largest trailing matrix), behind the panels and W
    int* tri_map = tri_map_ws;
    {
      // growth check, part 1: max |A| over what will be read (amax_ready: the caller's fill has already stored it), max |U| = 0
      if (!amax_ready) {
        hipLaunchKernelGGL(k_zero_int, dim3((4 * nb + 63) / 64), dim3(64), 0, st, (int*)growth, 4 * nb);
        ProfScope ps(PK_SWAP, st, 0.0);
        hipLaunchKernelGGL(k_absmax_lower, dim3((n_pad + 7) / 8, nb), dim3(256), 0, st, A, lda, sys_stride, n_pad, growth);
      }
    }
    {
      const int T = n_pad / NB, fb = T / 8, n_map = 32 * fb * fb + 4 * fb;
      if (n_map > 0) hipLaunchKernelGGL(k_tri_map, dim3((n_map + 255) / 256), dim3(256), 0, st, tri_map, n_map);
    }
    // A = L D L^T without interchanges (the caller guarantees a complex-symmetric matrix; a rejected diagonal is reported in
    // info).  Same four-panel groups and the same kernels; what changes: no interchanges; a panel's U rows are its transposed,
    // D-scaled multipliers; only the right-hand-side columns of those rows take pending updates and the triangular solve; the
    // K = 256 update runs over the lower triangle of tiles (and the right-hand sides): half the flops of the LU.
    for (int J = 0; J < n_pad; J += 4 * NB) {
      panel(J, 0); u_rows_sym(J, J, 0);
      for (int q = 1; q < 4; ++q) {
        const int jq = J + q * NB;
        if (jq >= n_pad) break;
        // the panel's columns: all pending updates of the group (K = 64 q) for all rows below, delivered into the workspace
        gemm(st, nb, A, lda, sys_stride, Pw, ldp, p_stride, jq, n_pad, jq, jq + NB, J, q * NB, PK_OTHER, -1.0,
                           Pw + (size_t)(q * NB) * ldp, ldp, p_stride, 0);
        panel(jq, q * NB, true);
        // right-hand sides of the panel's 64 rows: pending updates, then the solve; matrix columns: transposition
        u_rows_sym(jq, J, q * NB);
      }
      if (J + 4 * NB >= n_pad) break;
      gemm(st, nb, A, lda, sys_stride, Pw, ldp, p_stride, J + 4 * NB, n_pad, J + 4 * NB, n_pad, J, 4 * NB, PK_GEMM, -1.0,
                         nullptr, 0, 0, 0, tri_map);
      if (rhs_gemv) {
        ProfScope ps(PK_OTHER, st, 0.0);
        hipLaunchKernelGGL(k_rhs_update, dim3((n_pad - (J + 4 * NB) + RHS_UPD_ROWS - 1) / RHS_UPD_ROWS, nb, nrhs), dim3(256), 0, st, A, lda, sys_stride, Pw, ldp,
                           p_stride, n_pad, J + 4 * NB, J, 4 * NB);
      } else if (nrhs > 0) {
        gemm(st, nb, A, lda, sys_stride, Pw, ldp, p_stride, J + 4 * NB, n_pad, n_pad, n_cols, J, 4 * NB, PK_OTHER);
      }
    }
  } else

  {
    // three-level schedule.  Group = four 64-column panels a, b | c, d (workspace columns 0, 64 | 128, 192):
    //   block E = (a, b): a: factor, interchanges, U row block; its K = 64 update goes only to the 64 columns panel b consists
    //      of; b: factor, interchanges (also on a's stored multipliers); only now - with the rows in their final order - a's
    //      update of the 64 rows of b's U block, then b's U row block.
    //   E's K = 128 update is applied only where block O needs it: O's 128 columns before O is factored (T1, all rows), and the
    //      64 U rows of c and of d right of O after the respective panel's interchanges (T2c; for d fused with c's own update
    //      of those rows into one K = 192 pass).
    //   block O = (c, d): the same as E, its interchanges also applied to E's multipliers in the workspace.
    //   ONE K = 256 update of everything below and right of the group with [L_a L_b L_c L_d] x [U_a; U_b; U_c; U_d]: the
    //   trailing matrix is read and written once per 256 columns instead of once per 128 (77 vs 65 TFLOP/s for the update
    //   itself: no C slices, C additions or tile stores in the second half of a tile's K loop).
    for (int J = 0; J < n_pad; J += 4 * NB) {
      panel(J, 0); swap_right(J); trsm(J, 0);
      if (J + NB >= n_pad) break;                        // odd tail: nothing below the panel, forward elimination done
      gemm(st, nb, A, lda, sys_stride, Pw, ldp, p_stride, J + NB, n_pad, J + NB, J + 2 * NB, J, NB, PK_OTHER, -1.0,
                         Pw + (size_t)NB * ldp, ldp, p_stride, 0);
      panel(J + NB, NB, true); swap_right(J + NB);
      gemm(st, nb, A, lda, sys_stride, Pw, ldp, p_stride, J + NB, J + 2 * NB, J + 2 * NB, n_cols, J, NB, PK_OTHER);
      trsm(J + NB, NB);
      if (J + 2 * NB >= n_pad) break;
      // T1: E's update of panel c's columns, ALL rows below E (the rows c's pivot search ranges over must be in one state);
      // the result goes straight into the workspace.  d's columns wait: they take E's and c's updates in one K = 192 pass.
      const int c_end = J + 3 * NB < n_pad ? J + 3 * NB : n_pad;
      gemm(st, nb, A, lda, sys_stride, Pw, ldp, p_stride, J + 2 * NB, n_pad, J + 2 * NB, c_end, J, 2 * NB, PK_OTHER, -1.0,
                         Pw + (size_t)(2 * NB) * ldp, ldp, p_stride, 0);
      panel(J + 2 * NB, 2 * NB, true); swap_right(J + 2 * NB);
      // T2c: E's update of c's 64 U rows right of c - only now, after c's interchanges: rows that an interchange can exchange
      // must carry the same updates, and the rows below still wait for the K = 256 update
      gemm(st, nb, A, lda, sys_stride, Pw, ldp, p_stride, J + 2 * NB, c_end, c_end, n_cols, J, 2 * NB, PK_OTHER);
      trsm(J + 2 * NB, 2 * NB);
      if (J + 3 * NB >= n_pad) break;
      // d's columns: E's and c's updates in one K = 192 pass, delivered into the workspace
      gemm(st, nb, A, lda, sys_stride, Pw, ldp, p_stride, J + 3 * NB, n_pad, J + 3 * NB, J + 4 * NB, J, 3 * NB, PK_OTHER, -1.0,
                         Pw + (size_t)(3 * NB) * ldp, ldp, p_stride, 0);
      panel(J + 3 * NB, 3 * NB, true); swap_right(J + 3 * NB);
      // T2d and c's update of d's U rows in one K = 192 pass: [L_a L_b L_c] x [U_a; U_b; U_c]
      gemm(st, nb, A, lda, sys_stride, Pw, ldp, p_stride, J + 3 * NB, J + 4 * NB, J + 4 * NB, n_cols, J, 3 * NB, PK_OTHER);
      trsm(J + 3 * NB, 3 * NB);
      gemm(st, nb, A, lda, sys_stride, Pw, ldp, p_stride, J + 4 * NB, n_pad, J + 4 * NB, n_cols, J, 4 * NB);
    }
  }
  if (symmetric) hipLaunchKernelGGL(k_growth_check, dim3((nb + 63) / 64), dim3(64), 0, st, nb, n_pad, growth, d_info, growth_max);
  BIEM_LAUNCHCHK();
  if (gemm_rc != BIEM_OK) return gemm_rc;
  if (nrhs > 0) {
    ProfScope ps(PK_BACK, st, 4.0 * (double)nb * n_pad * (double)n_pad * nrhs);
    for (int jr = n_pad - BS; jr >= 0; jr -= BS) {
      hipLaunchKernelGGL(k_back_diag, dim3(nb, nrhs), dim3(64), 0, st, A, lda, sys_stride, A + n_pad, lda, sys_stride, jr);
      if (jr > 0)
        hipLaunchKernelGGL(k_back_update, dim3((jr + BACK_ROWS - 1) / BACK_ROWS, nb), dim3(256), 0, st, A, lda, sys_stride, A + n_pad, lda,
                           sys_stride, nrhs, jr, 0, jr);
    }
    BIEM_LAUNCHCHK();
  }
  return BIEM_OK;
}

// ---------------------------------------------------------------------------------------------
// Symmetric path in ROW form (what biem_solve_ldlt runs):  A = U^T U  with U = D^{1/2} L^T upper triangular, the complex-symmetric
// analogue of the Cholesky factorisation (no conjugation, principal complex square roots of the pivots; same pivots, same
// multipliers l_ci = u_ic / u_ii and same acceptance test as the L D L^T form it replaces).  Why this form: with A = U^T U the
// trailing update  A22 -= U12^T U12  takes BOTH zgemm operands from the same 64-row strip of the row-major matrix
// (A-operand[k][i] = U12[k][row i], B-operand[k][c] = U12[k][col c]), which is also exactly what the back substitution reads.
// So the factorisation works in place on the upper triangle: no column-major panel workspace, no transposing panel load / store,
// no transposed GEMM epilogue, no separate "U rows from L" pass - a panel is two passes over its strip instead of about six.
//   k_diag_utu_reg (one workgroup per system, defined with the small-system kernel below): the 64 x 64 diagonal block: pivots d,
//               U11 = D^{-1/2} (D L11^T), W = I - U11^{-T}, multiplier test inside the block
//   strip:      U12 = U11^{-T} A12 = A12 - W A12 in place on the streaming zgemm (K = 64, B operand = the strip's own rows; the
//               right-hand-side columns are columns of the strip: forward elimination rides along).  A one-thread-per-column
//               VALU form with the triangle of U11^{-T} from the scalar cache or LDS was 5x slower (292 vs 53 ms per 256 systems)
//   checks:     multiplier test |u_ic| <= 100 |u_ii| and growth max |u_ii u_ic| of the strip entries are taken where the entries
//               are read anyway: in the back substitution (k_back_update)
//   in-group:   the next panel's 64 rows take the group's pending updates (K = 64 q) for all columns right of them
//   K = 256:    one update of the UPPER triangle of tiles below the group (TileGrid.tri = 2), right-hand sides by k_rhs_update
// Only the upper triangle and the diagonal 64 x 64 tiles of A are read.  Growth check as in the L D L^T form, with moduli:
// max |d_i l_ci| = max |u_ii u_ic| against max |a_ij| over the part read.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_absmax_upper(const cplx* __restrict__ A, long long lda, long long sys_stride, int n_pad,
                                                       unsigned long long* __restrict__ growth) {
  const int s = blockIdx.y;
  const cplx* As = A + (size_t)s * sys_stride;
  double m = 0.0;
  for (int r = 0; r < 8; ++r) {
    const int i = blockIdx.x * 8 + r;
    if (i >= n_pad) break;
    for (int c = (i / NB) * NB + threadIdx.x; c < n_pad; c += 256) { const cplx v = As[(size_t)i * lda + c]; m = nan_max(m, sqrt(v.x * v.x + v.y * v.y)); }
  }
  block_max_publish(m, growth + 2 * (size_t)s);
}

// Back substitution of the row form, one launch per 64-row block (bottom up), one 1024-thread workgroup per system:
//   y_R -= U[R, C] x_C over all solved columns C right of the block - the 16 waves stream 4 rows each across the strip, 64 columns
//   per step, partial sums per lane and ONE reduction per row at the end - then the 64 x 64 triangular solve U[R,R] x_R = y_R in
//   the same launch (diagonal block in LDS, one wave).  Reads U exactly once in long contiguous runs (the column-block form
//   k_back_update re-launches per 64 columns with 16-KiB workgroups: 31 vs 84 GB / 5 TB/s = 17 ms per 256 systems at cfg 3).
// The pass also takes the checks of the strip entries (see k_back_update).  NQ right-hand sides per pass.
template <int NQ>
__global__ void __launch_bounds__(1024) k_back_row(const cplx* __restrict__ A, long long lda, long long sys_stride, cplx* __restrict__ Y,
                                                    int nrhs, int n_pad, int q0, int nq, int ib, int do_checks,
                                                    int* __restrict__ info, unsigned long long* __restrict__ growth, double inv_rel2) {
  // Y[s][q][row]: the right-hand sides / solutions in a compact copy (in the augmented matrix they sit one row stride apart:
  // gathering 64 of them per step from there cost more than the four 1-KiB row loads they are multiplied with)
  __shared__ cplx sU[NB][NB + 1];
  __shared__ cplx sy[NB][NQ];
  __shared__ double sm_max2[16];
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const cplx* As = A + (size_t)s * sys_stride;
  cplx* Ys = Y + (size_t)s * nrhs * n_pad;
  const int rb = ib * NB;
  for (int e = tid; e < NB * NB; e += 1024) { const int r = e >> 6, c = e & 63; sU[r][c] = As[(size_t)(rb + r) * lda + rb + c]; }
  cplx acc[4][NQ];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[k][q] = make_double2(0.0, 0.0);
  double d2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { const cplx d = As[(size_t)(rb + 4 * wave + k) * lda + rb + 4 * wave + k]; d2[k] = d.x * d.x + d.y * d.y; }
  double um2 = 0.0;
  bool badm = false;
  const cplx* Ur = As + (size_t)(rb + 4 * wave) * lda + lane;
#pragma unroll 2
  for (int c0 = rb + NB; c0 < n_pad; c0 += NB) {
    cplx x[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) x[q] = q < nq ? Ys[(size_t)(q0 + q) * n_pad + c0 + lane] : make_double2(0.0, 0.0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const cplx u = Ur[(size_t)k * lda + c0];
      if (do_checks) {
        const double m2 = u.x * u.x + u.y * u.y;
        if (!(m2 <= inv_rel2 * d2[k])) badm = true;
        um2 = nan_max(um2, m2 * d2[k]);
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) acc[k][q] = cfma(u, x[q], acc[k][q]);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      double vr = acc[k][q].x, vi = acc[k][q].y;
      for (int o = 32; o > 0; o >>= 1) { vr += __shfl_down(vr, o, 64); vi += __shfl_down(vi, o, 64); }
      if (lane == 0) {
        cplx y = q < nq ? Ys[(size_t)(q0 + q) * n_pad + rb + 4 * wave + k] : make_double2(0.0, 0.0);
        y.x -= vr; y.y -= vi;
        sy[4 * wave + k][q] = y;
      }
    }
  if (do_checks) {
    double m = sqrt(um2);
    for (int o = 32; o > 0; o >>= 1) m = nan_max(m, __shfl_down(m, o, 64));
    if (lane == 0) sm_max2[wave] = m;
    if (badm && info[s] == 0) info[s] = -(rb + 1);
  }
  __syncthreads();
  if (do_checks && tid == 0) {
    double m = sm_max2[0];
    for (int w = 1; w < 16; ++w) m = nan_max(m, sm_max2[w]);
    unsigned long long* dst = growth + 2 * (size_t)s + 1;
    if (!(m <= __longlong_as_double((long long)*(volatile unsigned long long*)dst))) atomicMax(dst, (unsigned long long)__double_as_longlong(m));
  }
  if (wave == 0) {            // the triangular solve of the block: lane = row, wave-synchronous
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q >= nq) break;
      cplx y = sy[lane][q];
      for (int c = NB - 1; c >= 0; --c) {
        if (lane == c) y = cmul(y, crecip(sU[c][c]));
        const double xr = lane_bcast(y.x, c), xi = lane_bcast(y.y, c);
        if (lane < c) y = cfnma(sU[lane][c], make_double2(xr, xi), y);
      }
      Ys[(size_t)(q0 + q) * n_pad + rb + lane] = y;
    }
  }
}

// right-hand-side columns of the augmented matrix <-> compact Y[s][q][row]
__global__ void __launch_bounds__(256) k_rhs_compact(cplx* __restrict__ A, long long lda, long long sys_stride, cplx* __restrict__ Y, int nrhs,
                                                      int n_pad, int to_matrix) {
  const int s = blockIdx.z, q = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_pad) return;
  cplx* f = A + (size_t)s * sys_stride + (size_t)r * lda + n_pad + q;
  cplx* y = Y + ((size_t)s * nrhs + q) * n_pad + r;
  if (to_matrix) *f = *y; else *y = *f;
}

// ---------------------------------------------------------------------------------------------
// Small systems (cfg 1: N = 72): the whole augmented system in LDS, one workgroup per system, ONE launch for factorisation,
// forward elimination, checks and back substitution - the blocked path above spends its time in per-panel launches there
// (4096 systems of N = 72: 5.2 of 6.8 ms in diagonal-block kernels that run one 64 x 64 block per workgroup).
// Same factorisation A = U^T U on the upper triangle (rows n .. of an identity-padded system are skipped), same acceptance tests
// and info codes; U is written back to the upper triangle.  n <= 128 rows, nrhs <= 8 and n + nrhs <= 128 (two 64-column lane slots; packed upper triangle of LDS).
// ---------------------------------------------------------------------------------------------
// 1 / d on the critical path of an elimination step: hardware reciprocal estimate + two Newton steps (4 FMAs) instead of the
// IEEE division sequence (~12 dependent instructions); relative error ~1e-16 for normal d
__device__ inline double fast_recip(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}
constexpr int SMALL_N_MAX = 128;           // and n + nrhs <= 128 (two 64-column lane slots), packed store within the LDS
constexpr int SMALL_RHS_MAX = 8;
constexpr int SMALL_THREADS = 512;
// LDS of k_small_utu: packed upper triangle with the right-hand sides appended to each row, then 1/a_cc and 1/sqrt(a_cc) per row
// and two rows of multipliers
static inline size_t small_utu_lds(int n, int nrhs) { return ((size_t)n * (n + 1) / 2 + (size_t)n * nrhs + 4 * (size_t)n) * sizeof(cplx); }
// The matrix lives in registers during the elimination: wave w owns rows w, w + 8, ... (KR of them), lane l columns l and l + 64
// (TWO); a finished row (row c + 1 after step c) is published once to the packed LDS store, which the other waves read it from
// and which the back substitution and the write-back then use.  One barrier per step, no read-modify-write through LDS.
template <int KR, bool TWO>
__global__ void __launch_bounds__(SMALL_THREADS, (KR <= 9 ? 4 : 2)) k_small_utu(cplx* __restrict__ A, long long lda, long long sys_stride, int n, int n_pad, int nrhs,
                                                              int* __restrict__ info, unsigned long long* __restrict__ growth, double rel, int amax_ready) {
  // row r of the packed store: columns r .. n-1 of the matrix, then the nrhs right-hand sides; element (r, c) at off(r) + c, nc = n + nrhs
  extern __shared__ cplx sa[];
  __shared__ int bad_row;
  constexpr int NW = SMALL_THREADS / 64;
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, nc = n + nrhs;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar: the row tests below become scalar branches
  auto off = [&](int r) { return r * nc - (r * (r - 1)) / 2 - r; };
  cplx* ipiv = sa + (size_t)n * (n + 1) / 2 + (size_t)n * nrhs;            // 1 / a_cc
  cplx* isq = ipiv + n;                                                     // 1 / sqrt(a_cc)
  cplx* lrow = isq + n;                                                     // [2][n]: the multipliers a_cj / a_cc of the current row
  cplx* As = A + (size_t)s * sys_stride;
  if (tid == 0) bad_row = -1;
  const int j0 = lane, j1 = lane + 64;
  const long long g0 = j0 < n ? j0 : n_pad + (j0 - n), g1 = j1 < n ? j1 : n_pad + (j1 - n);     // global columns of the two slots
  cplx a0[KR], a1[KR];
  double am = 0.0;                               // (squares; the root is taken once)
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const int i = w + NW * k;
    a0[k] = a1[k] = make_double2(0.0, 0.0);
    if (i < n) {
      const cplx* src = As + (size_t)i * lda;
      if (j0 >= i && j0 < nc) { a0[k] = src[g0]; if (j0 < n) am = nan_max(am, a0[k].x * a0[k].x + a0[k].y * a0[k].y); }
      if (TWO && j1 >= i && j1 < nc) { a1[k] = src[g1]; if (j1 < n) am = nan_max(am, a1[k].x * a1[k].x + a1[k].y * a1[k].y); }
    }
  }
  // A finished row i: its wave publishes it (packed store), 1 / a_ii (ipiv) and the multipliers a_ij / a_ii (lrow[i & 1]);
  // the reciprocal is computed here ONCE per row (plain 1 / |d|^2 form: the systems are equilibrated, |a_ii| = O(1); an
  // overflow would surface as inf / NaN in the growth test)
  auto publish = [&](int i, const cplx& r0, const cplx& r1) {
    cplx d;
    if (!TWO || i < 64) { d.x = lane_bcast(r0.x, i & 63); d.y = lane_bcast(r0.y, i & 63); }
    else { d.x = lane_bcast(r1.x, i & 63); d.y = lane_bcast(r1.y, i & 63); }
    const double rr = fast_recip(d.x * d.x + d.y * d.y);
    const cplx ip = make_double2(d.x * rr, -d.y * rr);
    cplx* ri = sa + off(i);
    cplx* lr = lrow + (i & 1) * n;
    if (j0 >= i && j0 < nc) { ri[j0] = r0; if (j0 < n) lr[j0] = cmul(r0, ip); }
    if (TWO && j1 >= i && j1 < nc) { ri[j1] = r1; if (j1 < n) lr[j1] = cmul(r1, ip); }
    if (lane == 0) ipiv[i] = ip;
  };
  if (w == 0) publish(0, a0[0], a1[0]);          // row 0 is final from the start
  if (!amax_ready) block_max_publish(sqrt(am), growth + 2 * (size_t)s);
  // Elimination in the D L^T form (row c stays unscaled: a_ij -= (a_ci / a_cc) a_cj); U = D^{-1/2} (D L^T) at the write-back.
  double um = 0.0;
  for (int c = 0; c < n; ++c) {
    __syncthreads();                            // row c has been published
    const cplx* rc = sa + off(c);
    const cplx* lr = lrow + (c & 1) * n;
    const cplx u0 = rc[(j0 >= c && j0 < nc) ? j0 : c];
    const cplx u1 = TWO ? rc[(j1 >= c && j1 < nc) ? j1 : c] : make_double2(0.0, 0.0);
    if (w == ((c + 1 + NW / 2) & (NW - 1))) {
      // acceptance tests on row c, once (by a wave that does not publish the next row): multipliers |a_cj| <= |piv| / rel;
      // growth: |a_cj| is the D L^T entry
      const cplx piv = rc[c];
      const double pa = fabs(piv.x) + fabs(piv.y);
      const bool in0 = j0 >= c && j0 < n, in1 = TWO && j1 >= c && j1 < n;
      const double v0 = in0 ? fabs(u0.x) + fabs(u0.y) : 0.0, v1 = in1 ? fabs(u1.x) + fabs(u1.y) : 0.0;
      if ((in0 && j0 > c && !(pa >= rel * v0)) || (in1 && j1 > c && !(pa >= rel * v1)) || !(pa > 0.0)) atomicMax(&bad_row, n - 1 - c);
      if (in0) um = nan_max(um, u0.x * u0.x + u0.y * u0.y);
      if (in1) um = nan_max(um, u1.x * u1.x + u1.y * u1.y);
    }
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int i = w + NW * k;
      if (i > c && i < n) {
        const cplx f = lr[i];
        if (k < 64 / NW) a0[k] = cfnma(f, u0, a0[k]);         // (rows from 64 on have nothing in columns 0 .. 63)
        if (TWO) a1[k] = cfnma(f, u1, a1[k]);
        if (i == c + 1) publish(i, a0[k], a1[k]);             // this row is final now
      }
    }
  }
  __syncthreads();
  for (int r = tid; r < n; r += SMALL_THREADS) isq[r] = crecip(zsqrt(sa[off(r) + r]));
  um = sqrt(um);
  block_max_publish(um, growth + 2 * (size_t)s + 1);
  // back substitution (D L^T) x = y': x_c = (y'_c - sum_{j > c} a_cj x_j) / a_cc, column oriented, one barrier per step:
  // y'_c is final when step c starts; thread i < c takes a_ic x_c off y'_i; x_c = y'_c / a_cc is formed again at the write-back
  for (int c = n - 1; c > 0; --c) {
    __syncthreads();
    const cplx* rc = sa + off(c);
    const cplx ip = ipiv[c];
    for (int i = tid; i < c; i += SMALL_THREADS) {
      cplx* ri = sa + off(i);
      const cplx aic = ri[c];
      for (int q = 0; q < nrhs; ++q) ri[n + q] = cfnma(aic, cmul(rc[n + q], ip), ri[n + q]);
    }
  }
  __syncthreads();
  for (int r = w; r < n; r += SMALL_THREADS / 64) {
    cplx* dstg = As + (size_t)r * lda;
    const cplx* src = sa + off(r);
    const cplx sc = isq[r], ip = ipiv[r];
    for (int c = r + lane; c < nc; c += 64) {
      if (c < n) dstg[c] = cmul(src[c], sc);                  // U = D^{-1/2} (D L^T)
      else dstg[n_pad + (c - n)] = cmul(src[c], ip);         // the solution
    }
  }
  if (tid == 0 && bad_row >= 0 && info[s] == 0) info[s] = -(((n - 1 - bad_row) / NB) * NB + 1);
}

// ---------------------------------------------------------------------------------------------
// The diagonal 64 x 64 block of a panel, register-resident like k_small_utu (same elimination, same publication of finished
// rows): lane l of the wave that owns row i holds a_il and, in the second slot, column l of the identity carried through the
// elimination - [A11 | I] -> [D L^T | L^-1] - so the inverse the strip needs, U11^{-T} = D^{-1/2} L^{-1}, comes out of the lanes
// that the 64-column matrix block leaves idle.  Replaced an LDS form (64 steps of read-modify-write through LDS with a complex
// division per thread, then 64 two-barrier steps for the inverse; git history): 125 -> ~50 us per launch of one workgroup per CU
// (cfg 3: 56.4 -> 49.1 ms per 256-system step for strips + diagonal blocks; cfg 2: 61.9 -> 71.7 k systems/s, same box).
// Writes U11 into the upper triangle of the block and W = I - U11^{-T} as W[k][i] (the A-operand order of the streaming zgemm).
// ---------------------------------------------------------------------------------------------
#ifdef BIEM_DIAG_TRACE
// diagnostic build only (tools/diag_trace.cpp): lane 0 of every wave of workgroup 0 stamps s_memtime at 4 points of each step
__device__ unsigned long long g_diag_trace[16][66][4];
#define BIEM_DT(step, i) { if (blockIdx.x == 0 && lane == 0) g_diag_trace[w][step][i] = __builtin_amdgcn_s_memtime(); }
#else
#define BIEM_DT(step, i)
#endif
constexpr int DIAG_LDS_CPLX = 2 * (NB * (NB + 1) / 2) + 2 * NB + NB + 2 * NB;          // packed U rows, packed L^-1 rows, multipliers [2][64], 1 / sqrt(d), combined rows [2][64]
#ifndef BIEM_DIAG_THREADS
#define BIEM_DIAG_THREADS 1024
#endif
constexpr int DIAG_THREADS = BIEM_DIAG_THREADS;         // 16 waves x 4 rows: the step is bound by the instructions a wave issues for its rows
__global__ void __launch_bounds__(DIAG_THREADS) k_diag_utu_reg(cplx* __restrict__ A, long long lda, long long sys_stride, int j,
                                                                 cplx* __restrict__ Wt, long long w_stride, int* __restrict__ info, double rel,
                                                                 unsigned long long* __restrict__ growth) {
  extern __shared__ cplx sd[];
  __shared__ int bad;
  constexpr int NW = DIAG_THREADS / 64, KR = NB / NW;
  cplx* su = sd;                                   // (r, c), c >= r, at uoff(r) + c
  cplx* sy = su + NB * (NB + 1) / 2;               // (i, k), k <= i, at yoff(i) + k
  cplx* lrow = sy + NB * (NB + 1) / 2;             // [2][64] multipliers a_cj / a_cc of the current row
  cplx* isq = lrow + 2 * NB;                       // 1 / sqrt(d_r)
  auto uoff = [](int r) { return r * NB - (r * (r - 1)) / 2 - r; };
  auto yoff = [](int i) { return (i * (i + 1)) / 2; };
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar: the row tests below become scalar branches
  cplx* Ab = A + (size_t)s * sys_stride + (size_t)j * lda + j;
  if (tid == 0) bad = 0;
  BIEM_DT(64, 0)
  cplx a0[KR], y1[KR];
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const int i = w + NW * k;
    a0[k] = lane >= i ? Ab[(size_t)i * lda + lane] : make_double2(0.0, 0.0);
    y1[k] = make_double2(lane == i ? 1.0 : 0.0, 0.0);
  }
  // A finished row i is published three times: packed rows su (D L^T) and sy (L^-1) for the write-back, and for the elimination ONE
  // combined vector comb[i & 1]: lane l <= i: (L^-1)_il (1 at l == i), lane l > i: a_il - the row-i operand of BOTH updates of a later
  // row r > i (its D L^T part lives in lanes >= r, its L^-1 part needs lanes <= i, and (L^-1)_il = 0 for l > i) - plus the multipliers
  // a_il / d_i in lrow[i & 1].  The step is LDS-bandwidth bound (tools/diag_trace.cpp: every wave reading the U row, the L^-1 row and a
  // broadcast multiplier per owned row = 96 reads of 1 KB per step, 650 of 1760 traced cycles): one row read instead of two, and waves /
  // rows that are finished read nothing.  (Multipliers taken from a vector through v_readlane instead of broadcast reads: slower,
  // 39 -> 48 us.)
  cplx* comb = isq + NB;                           // [2][64]
  auto publish = [&](int i, const cplx& r0, const cplx& r1) {
    cplx d;
    d.x = __shfl(r0.x, i, 64); d.y = __shfl(r0.y, i, 64);
    const double rr = fast_recip(d.x * d.x + d.y * d.y);
    const cplx ip = make_double2(d.x * rr, -d.y * rr);
    comb[(i & 1) * NB + lane] = make_double2(lane > i ? r0.x : r1.x, lane > i ? r0.y : r1.y);   // (by value: a conditional on the references selects an address and puts the rows into scratch)
    if (lane >= i) { su[uoff(i) + lane] = r0; lrow[(i & 1) * NB + lane] = cmul(r0, ip); }
    if (lane <= i) sy[yoff(i) + lane] = r1;
  };
  if (w == 0) publish(0, a0[0], y1[0]);
  double um = 0.0;
  BIEM_DT(64, 1)
  for (int c = 0; c < NB; ++c) {
    BIEM_DT(c, 0)
    __syncthreads();                            // row c has been published
    BIEM_DT(c, 1)
    const bool accept = w == ((c + 1 + NW / 2) & (NW - 1));   // acceptance tests on row c, once, by a wave that does not publish the next row
    if (accept) {
      const cplx* rc = su + uoff(c);
      const cplx piv = rc[c], ur = rc[lane >= c ? lane : c];
      const double pa = fabs(piv.x) + fabs(piv.y);
      if ((lane > c && !(pa >= rel * (fabs(ur.x) + fabs(ur.y)))) || !(pa > 0.0)) bad = 1;
      if (lane >= c) um = nan_max(um, ur.x * ur.x + ur.y * ur.y);
    }
    if (w + NW * (KR - 1) > c) {                // (a wave whose rows are all finished only takes the barriers)
    const cplx u0 = comb[(c & 1) * NB + lane];
    const cplx u1 = lane <= c ? u0 : make_double2(0.0, 0.0);
    const cplx* lr = lrow + (c & 1) * NB;
    // this wave's multipliers: all LDS reads issued together (inside the branches each would be waited for in turn)
    cplx fk[KR];
#pragma unroll
    for (int k = 0; k < KR; ++k) { const int i = w + NW * k; fk[k] = make_double2(0.0, 0.0); if (i > c) fk[k] = lr[i]; }
#ifdef BIEM_DIAG_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BIEM_DT(c, 2)
#endif
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int i = w + NW * k;
      if (i > c) {
        a0[k] = cfnma(fk[k], u0, a0[k]);
        y1[k] = cfnma(fk[k], u1, y1[k]);
        if (i == c + 1) { publish(i, a0[k], y1[k]); BIEM_DT(c, 3) }
      }
    }
    }
  }
  BIEM_DT(64, 2)
  __syncthreads();
  if (tid < NB) isq[tid] = crecip(zsqrt(su[uoff(tid) + tid]));
  block_max_publish(sqrt(um), growth + 2 * (size_t)s + 1);       // (its barrier also orders isq)
  // U11 = D^{-1/2} (D L^T) into the upper triangle of the block (lanes along the row)
  for (int r = w; r < NB; r += NW)
    if (lane >= r) Ab[(size_t)r * lda + lane] = cmul(su[uoff(r) + lane], isq[r]);
  // W[k][i] = delta_ki - (U11^{-T})[i][k] = delta_ki - L^-1[i][k] / sqrt(d_i), k <= i (lanes along i)
  cplx* Wo = Wt + (size_t)s * w_stride;
  for (int k = w; k < NB; k += NW) {
    cplx v = make_double2(0.0, 0.0);
    if (k <= lane) { const cplx xt = cmul(sy[yoff(lane) + k], isq[lane]); v = make_double2((k == lane ? 1.0 : 0.0) - xt.x, -xt.y); }
    Wo[k * NB + lane] = v;
  }
  BIEM_DT(64, 3)
  if (tid == 0 && bad && info[s] == 0) info[s] = -(j + 1);
}

// ---------------------------------------------------------------------------------------------
// The same diagonal block, FOUR pivots per barrier: wave w owns the four consecutive rows 4w .. 4w+3.  tools/diag_trace.cpp showed
// the one-pivot-per-barrier form above to be a chain of latencies, not of work: per pivot a barrier, an LDS round trip, a lane
// broadcast of the pivot (another LDS round trip), a reciprocal (rcp + two Newton steps) and the multiplier products - about a
// dozen dependent FP64 instructions of ~20 cycles each plus ~400 cycles of LDS / barrier, ~1000 cycles where the arithmetic of a
// step needs 250.  Here a block of four finished rows is published at once: the waves behind it apply the four rows (rank-4
// update of their own four rows), and the wave that owns the next four rows then factors them on its own - the ten entries of its
// 4 x 4 diagonal sub-block are broadcast ONCE (ten independent lane broadcasts in flight together), every lane runs the 4 x 4
// elimination on them redundantly (pivots, reciprocals and the six multipliers inside the block as wave-uniform values: no further
// broadcast), and the vector updates of the rows follow.  16 barriers and 16 broadcast round trips instead of 64 each.
// Same arithmetic per entry as the form above (same order of the rank-1 updates), same acceptance tests, same outputs.
// ---------------------------------------------------------------------------------------------
constexpr int DIAGB_LDS_CPLX = 2 * (NB * (NB + 1) / 2) + NB + 2 * 2 * 4 * NB + 16;   // packed U rows, packed L^-1 rows, 1 / sqrt(d), combined rows and multipliers [2][4][64] each, the 4 x 4 sub-block
__global__ void __launch_bounds__(1024) k_diag_utu_blk(cplx* __restrict__ A, long long lda, long long sys_stride, int j,
                                                        cplx* __restrict__ Wt, long long w_stride, int* __restrict__ info, double rel,
                                                        unsigned long long* __restrict__ growth) {
  extern __shared__ cplx sd[];
  __shared__ int bad;
  constexpr int NW = 16, KR = 4;
  cplx* su = sd;                                   // (r, c), c >= r, at uoff(r) + c
  cplx* sy = su + NB * (NB + 1) / 2;               // (i, k), k <= i, at yoff(i) + k
  cplx* isq = sy + NB * (NB + 1) / 2;              // 1 / sqrt(d_r)
  cplx* cmb = isq + NB;                            // [2][4][64]: lane l <= i: (L^-1)_il, lane l > i: a_il of the published row i
  cplx* mul = cmb + 2 * 4 * NB;                    // [2][4][64]: a_il / d_i
  cplx* dsc = mul + 2 * 4 * NB;                    // [4][4]: the owner's 4 x 4 diagonal sub-block on its way to all lanes
  auto uoff = [](int r) { return r * NB - (r * (r - 1)) / 2 - r; };
  auto yoff = [](int i) { return (i * (i + 1)) / 2; };
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  cplx* Ab = A + (size_t)s * sys_stride + (size_t)j * lda + j;
  if (tid == 0) bad = 0;
  cplx a0[KR], y1[KR];
#pragma unroll
  for (int r = 0; r < KR; ++r) {
    const int i = 4 * w + r;
    a0[r] = lane >= i ? Ab[(size_t)i * lda + lane] : make_double2(0.0, 0.0);
    y1[r] = make_double2(lane == i ? 1.0 : 0.0, 0.0);
  }
  // in-wave factorisation of this wave's four rows (all earlier blocks applied), then their publication
  auto factor_block = [&]() {
    const int i0 = 4 * w;
    // the sub-block through LDS: four predicated writes, ten broadcast reads, one round trip (a wave's LDS operations complete in
    // order).  Twenty ds_bpermute with a single source lane took ~1000 cycles (tools/diag_trace.cpp).
    cplx D[KR][KR];
    const int cl = lane - i0;
#pragma unroll
    for (int r = 0; r < KR; ++r) if (cl >= r && cl < KR) dsc[r * KR + cl] = a0[r];
    // (lanes exchange data here without a workgroup barrier: the wave-scope fences keep hipcc from reading the entries before the
    // other lanes' stores, or forwarding this lane's own store - it did, and every system failed the pivot test)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int r = 0; r < KR; ++r)
#pragma unroll
      for (int c = r; c < KR; ++c) D[r][c] = dsc[r * KR + c];
#ifdef BIEM_DIAG_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BIEM_DT(16 + w, 0)
#endif
    cplx ip[KR], m[KR][KR];
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      const double rr = fast_recip(D[r][r].x * D[r][r].x + D[r][r].y * D[r][r].y);
      const cplx dc = make_double2(D[r][r].x, -D[r][r].y);
      ip[r] = make_double2(dc.x * rr, dc.y * rr);
      // multipliers as (a conj d) / |d|^2: the product runs beside the reciprocal instead of behind it (two levels off the chain)
#pragma unroll
      for (int c = r + 1; c < KR; ++c) { const cplx t = cmul(D[r][c], dc); m[r][c] = make_double2(t.x * rr, t.y * rr); }
#pragma unroll
      for (int k = r + 1; k < KR; ++k)
#pragma unroll
        for (int c = k; c < KR; ++c) D[k][c] = cfnma(m[r][k], D[r][c], D[k][c]);
    }
#ifdef BIEM_DIAG_TRACE
    asm volatile("" :: "v"(ip[3].x), "v"(ip[3].y));
    BIEM_DT(16 + w, 1)
#endif
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      const cplx yr = lane <= i0 + r ? y1[r] : make_double2(0.0, 0.0);
#pragma unroll
      for (int k = r + 1; k < KR; ++k) { a0[k] = cfnma(m[r][k], a0[r], a0[k]); y1[k] = cfnma(m[r][k], yr, y1[k]); }
    }
    cplx* cb = cmb + (w & 1) * 4 * NB;
    cplx* mb = mul + (w & 1) * 4 * NB;
#pragma unroll
    for (int r = 0; r < KR; ++r) {
      const int i = i0 + r;
      cb[r * NB + lane] = make_double2(lane > i ? a0[r].x : y1[r].x, lane > i ? a0[r].y : y1[r].y);
      mb[r * NB + lane] = cmul(a0[r], ip[r]);
      if (lane >= i) su[uoff(i) + lane] = a0[r];
      if (lane <= i) sy[yoff(i) + lane] = y1[r];
    }
    BIEM_DT(16 + w, 2)
  };
  if (w == 0) factor_block();
  double um = 0.0;
  for (int b = 0; b < NW; ++b) {
    BIEM_DT(b, 0)
    __syncthreads();                            // rows 4b .. 4b+3 have been published
    BIEM_DT(b, 1)
    if (w == ((b + 1 + NW / 2) & (NW - 1))) {   // acceptance tests on the four rows, once, by a wave far from the chain
#pragma unroll
      for (int r = 0; r < KR; ++r) {
        const int c = 4 * b + r;
        const cplx* rc = su + uoff(c);
        const cplx piv = rc[c], ur = rc[lane >= c ? lane : c];
        const double pa = fabs(piv.x) + fabs(piv.y);
        if ((lane > c && !(pa >= rel * (fabs(ur.x) + fabs(ur.y)))) || !(pa > 0.0)) bad = 1;
        if (lane >= c) um = nan_max(um, ur.x * ur.x + ur.y * ur.y);
      }
    }
    if (w > b) {
      if (w == b + 1) __builtin_amdgcn_s_setprio(3);          // the next block's owner is the dependent chain
      const cplx* cb = cmb + (b & 1) * 4 * NB;
      const cplx* mb = mul + (b & 1) * 4 * NB;
      // (the reads of row r + 1 are issued before the arithmetic of row r: four exposed LDS round trips per block step otherwise)
      cplx nu = cb[lane], nf[KR];
#pragma unroll
      for (int k = 0; k < KR; ++k) nf[k] = mb[4 * w + k];
#pragma unroll
      for (int r = 0; r < KR; ++r) {
        const int c = 4 * b + r;
        const cplx u0 = nu;
        cplx fk[KR];
#pragma unroll
        for (int k = 0; k < KR; ++k) fk[k] = nf[k];
        if (r + 1 < KR) {
          nu = cb[(r + 1) * NB + lane];
#pragma unroll
          for (int k = 0; k < KR; ++k) nf[k] = mb[(r + 1) * NB + 4 * w + k];
        }
        const cplx u1 = lane <= c ? u0 : make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < KR; ++k) { a0[k] = cfnma(fk[k], u0, a0[k]); y1[k] = cfnma(fk[k], u1, y1[k]); }
      }
#ifdef BIEM_DIAG_TRACE
      asm volatile("" :: "v"(a0[0].x), "v"(a0[3].y), "v"(y1[3].x));
      BIEM_DT(b, 2)
#endif
      if (w == b + 1) { factor_block(); __builtin_amdgcn_s_setprio(0); }
    }
  }
  __syncthreads();
  if (tid < NB) isq[tid] = crecip(zsqrt(su[uoff(tid) + tid]));
  block_max_publish(sqrt(um), growth + 2 * (size_t)s + 1);       // (its barrier also orders isq)
  for (int r = w; r < NB; r += NW)
    if (lane >= r) Ab[(size_t)r * lda + lane] = cmul(su[uoff(r) + lane], isq[r]);
  cplx* Wo = Wt + (size_t)s * w_stride;
  for (int k = w; k < NB; k += NW) {
    cplx v = make_double2(0.0, 0.0);
    if (k <= lane) { const cplx xt = cmul(sy[yoff(lane) + k], isq[lane]); v = make_double2((k == lane ? 1.0 : 0.0) - xt.x, -xt.y); }
    Wo[k * NB + lane] = v;
  }
  if (tid == 0 && bad && info[s] == 0) info[s] = -(j + 1);
}

bool sym_small_path(int n_active, int nrhs) {
  return n_active > 0 && n_active <= SMALL_N_MAX && nrhs <= SMALL_RHS_MAX && n_active + nrhs <= 128 && small_utu_lds(n_active, nrhs) <= 160 * 1024 - 2048 &&
         !getenv("BIEM_NO_SMALL_PATH");
}

int launch_sym_factor_solve(int nb, int n_pad, int nrhs, double* d_A, long long lda, long long sys_stride, int* d_info, void* d_work,
                            size_t work_bytes, hipStream_t st, bool amax_ready, int n_active) {
  if (nb <= 0 || n_pad <= 0) return BIEM_OK;
  if (n_active <= 0 || n_active > n_pad) n_active = n_pad;        // rows n_active .. n_pad-1: identity padding (the caller's promise)
  if (n_pad % NB) { set_error("biem_sym: n_pad=%d is not a multiple of %d (use biem_lu_npad)", n_pad, NB); return BIEM_ERR_ARG; }
  if (nrhs < 0 || lda < n_pad + nrhs) { set_error("biem_sym: lda < n_pad + nrhs"); return BIEM_ERR_ARG; }
  if (nb > 65535 || nrhs > 65535) { set_error("biem_sym: at most 65535 systems / right-hand sides per call (got %d / %d)", nb, nrhs); return BIEM_ERR_ARG; }
  if (work_bytes < lu_workspace_bytes(nb, n_pad, nrhs)) { set_error("biem_sym: workspace too small"); return BIEM_ERR_ARG; }
  cplx* A = (cplx*)d_A;
  const int n_cols = n_pad + nrhs;
  cplx* Wt = (cplx*)d_work + (size_t)nb * 4 * NB * (size_t)ldp_of(n_pad);      // same place as the 64 x 64 block of the other paths
  int* tri_map = (int*)(Wt + (size_t)nb * NB * NB);
  unsigned long long* growth = lu_growth_slots(d_work, nb, n_pad);
  hipLaunchKernelGGL(k_zero_int, dim3((nb + 63) / 64), dim3(64), 0, st, d_info, nb);
  double nopiv = NOPIV_REL, growth_max = GROWTH_MAX;
  { const char* e = getenv("BIEM_LDLT_PIVOT_REL"); if (e && atof(e) > 0.0) nopiv = atof(e);
    const char* g = getenv("BIEM_LDLT_GROWTH_MAX"); if (g && atof(g) > 0.0) growth_max = atof(g); }
  if (sym_small_path(n_active, nrhs)) {
    // the whole system fits LDS: one launch does everything
    if (!amax_ready) hipLaunchKernelGGL(k_zero_int, dim3((4 * nb + 63) / 64), dim3(64), 0, st, (int*)growth, 4 * nb);   // max|A| is measured by the kernel
    ProfScope ps(PK_PANEL, st, 0.0);
    const size_t shm = small_utu_lds(n_active, nrhs);
    const bool two = n_active + nrhs > 64;
    const int kr = (n_active + SMALL_THREADS / 64 - 1) / (SMALL_THREADS / 64);
#define BIEM_SMALL(KR, TWO)                                                                                                         \
  {                                                                                                                                  \
    BIEM_HIPCHK(hipFuncSetAttribute((const void*)k_small_utu<KR, TWO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));        \
    hipLaunchKernelGGL((k_small_utu<KR, TWO>), dim3(nb), dim3(SMALL_THREADS), shm, st, A, lda, sys_stride, n_active, n_pad, nrhs, d_info, \
                       growth, nopiv, amax_ready ? 1 : 0);                                                                              \
  }
    if (!two) { if (kr <= 4) BIEM_SMALL(4, false) else BIEM_SMALL(8, false) }
    else if (kr <= 8) BIEM_SMALL(8, true)
    else if (kr <= 9) BIEM_SMALL(9, true)
    else if (kr <= 12) BIEM_SMALL(12, true)
    else BIEM_SMALL(16, true)
#undef BIEM_SMALL
    hipLaunchKernelGGL(k_growth_check, dim3((nb + 63) / 64), dim3(64), 0, st, nb, n_pad, growth, d_info, growth_max);
    BIEM_LAUNCHCHK();
    return BIEM_OK;
  }
  {
    const int T = n_pad / NB, fb = T / 8, n_map = 32 * fb * fb + 4 * fb;
    if (n_map > 0) hipLaunchKernelGGL(k_tri_map, dim3((n_map + 255) / 256), dim3(256), 0, st, tri_map, n_map);
  }
  if (!amax_ready) {
    hipLaunchKernelGGL(k_zero_int, dim3((4 * nb + 63) / 64), dim3(64), 0, st, (int*)growth, 4 * nb);
    ProfScope ps(PK_SWAP, st, 0.0);
    hipLaunchKernelGGL(k_absmax_upper, dim3((n_pad + 7) / 8, nb), dim3(256), 0, st, A, lda, sys_stride, n_pad, growth);
  }
  int gemm_rc = BIEM_OK;
  auto gemm = [&](auto&&... a) { const int r = launch_gemm_stream(a...); if (r != BIEM_OK && gemm_rc == BIEM_OK) gemm_rc = r; };
  const bool rhs_gemv = nrhs > 0 && nrhs <= 8;
  // Few systems (the column-block form of the back substitution below): every panel's W = I - U11^{-T} is kept - in the panel region of
  // the workspace, which the row form does not use: n_pad x 64 complex per system, then the solutions (nrhs x n_pad) - and the back
  // substitution multiplies by the stored inverses instead of solving with the diagonal blocks (k_back_step).
  const char* bf = getenv("BIEM_BACK_FORM");
  const bool col_form = bf ? bf[0] == 'c' || bf[0] == 's' : nb <= 64;
  // (every workgroup of a block step forms x_b for itself from the 64 KB inverse: a latency trade that pays for a handful of systems -
  // at 64 systems of cfg 4 the back substitution went from 4.4 to 11.5 ms with it; BIEM_BACK_FORM=step forces it, =col the two-launch form)
  const bool keep_w = col_form && nrhs > 0 && nrhs <= 3 * NB && (bf ? bf[0] == 's' : nb <= 8);
  cplx* Wall = (cplx*)d_work;
  const long long wall_stride = (long long)n_pad * NB;
  cplx* Xsol = Wall + (size_t)nb * wall_stride;
  BIEM_HIPCHK(hipFuncSetAttribute((const void*)k_diag_utu_reg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(DIAG_LDS_CPLX * sizeof(cplx))));
  BIEM_HIPCHK(hipFuncSetAttribute((const void*)k_diag_utu_blk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(DIAGB_LDS_CPLX * sizeof(cplx))));
  const char* dform = getenv("BIEM_DIAG_FORM");                  // step: one pivot per barrier (the A/B of the tests); default: four
  const bool diag_blk = !(dform && dform[0] == 's');
  auto panel = [&](int j) {
    cplx* Wp = keep_w ? Wall + (size_t)(j / NB) * NB * NB : Wt;
    const long long w_stride = keep_w ? wall_stride : (long long)NB * NB;
    {
      ProfScope ps(PK_PANEL, st, 0.0);
      if (diag_blk) hipLaunchKernelGGL(k_diag_utu_blk, dim3(nb), dim3(1024), DIAGB_LDS_CPLX * sizeof(cplx), st, A, lda, sys_stride, j, Wp, w_stride, d_info, nopiv, growth);
      else hipLaunchKernelGGL(k_diag_utu_reg, dim3(nb), dim3(DIAG_THREADS), DIAG_LDS_CPLX * sizeof(cplx), st, A, lda, sys_stride, j, Wp, w_stride, d_info, nopiv, growth);
    }
    // A operand W[k][i], i = row - j: the base shifted by -j rows (only rows j .. j+63 are addressed)
    if (n_cols > j + NB)
      gemm(st, nb, A, lda, sys_stride, Wp - j, NB, w_stride, j, j + NB, j + NB, n_cols, j, NB, PK_PANEL, 8.0 * (double)nb * (n_cols - j - NB) * NB * NB);
  };
  // (A fused form - the diagonal block updated alone, then ONE pass U12 = C - [(X P^T) | W] [Q ; C] with K = 64 (q + 1) over the strip
  // instead of the pending-update pass and the solve pass - was built and measured in round 3: these passes run at the zgemm
  // pipeline's rate per K-chunk like the bulk update (0.445 / 0.79 / 1.22 ms for K = 64 / 128 / 192 at cfg 3), not at a bandwidth
  // limit, so the same K-chunks in fewer passes gain 2.5 % of panel + in-group time at cfg 3, nothing at cfg 5, and lose 27 % at
  // cfg 4 and 30 % for one system per call (three more small launches per panel).  Not kept)
  // second stream + two events for the right-hand sides' update beside the K = 256 update (below); the stream lives per device
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // Opt-in (BIEM_RHS_SIDE_STREAM=1): measured +0.7 % (cfg 3) and +2 % (cfg 5) of the step in alternating runs on one box, but on
  // another box the K = 256 launches it co-runs with stretched by 6 % (cfg 5: 1643 -> 1612 systems/s) - the update kernel is tuned to
  // have the CUs to itself - and one system per call pays the two cross-stream dependencies per group (cfg 4: 6.8 -> 7.0 ms).
  { const char* es = getenv("BIEM_RHS_SIDE_STREAM");
    const bool want = es != nullptr && es[0] == '1';
    if (rhs_gemv && n_pad > 4 * NB && want) {
      static hipStream_t side_of[64] = {nullptr};
      int devid = 0;
      if (hipGetDevice(&devid) == hipSuccess && devid >= 0 && devid < 64) {
        if (side_of[devid] == nullptr && hipStreamCreateWithFlags(&side_of[devid], hipStreamNonBlocking) != hipSuccess) side_of[devid] = nullptr;
        side = side_of[devid];
      }
      if (side != nullptr && (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess ||
                              hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess)) side = nullptr;
    } }
  struct EvGuard { hipEvent_t &a, &b; ~EvGuard() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } ev_guard{ev_fork, ev_join};
  for (int J = 0; J < n_pad; J += 4 * NB) {
    const cplx* strip = A + (size_t)J * lda;        // both operands of this group's updates: rows J .. of the matrix itself
    panel(J);
    for (int q = 1; q < 4; ++q) {
      const int jq = J + q * NB;
      if (jq >= n_pad) break;
      // the next panel's 64 rows: all pending updates of the group (K = 64 q), every column right of them incl. the right-hand sides
      gemm(st, nb, A, lda, sys_stride, strip, lda, sys_stride, jq, jq + NB, jq, n_cols, J, q * NB, PK_OTHER);
      panel(jq);
    }
    if (J + 4 * NB >= n_pad) break;
    // The right-hand sides' update of the rows below the group streams the group's strips once (bandwidth, a few KB of LDS, few
    // registers); the K = 256 update of the matrix is bound by the matrix pipe and touches no right-hand-side column: the two run
    // side by side - the small kernel on a second stream between two events, joined before the next panel (whose strip solve reads
    // the right-hand-side columns).
    const bool beside = rhs_gemv && side != nullptr;
    if (beside) {
      BIEM_HIPCHK(hipEventRecord(ev_fork, st));
      BIEM_HIPCHK(hipStreamWaitEvent(side, ev_fork, 0));
      // (not in the stage times of biem_profile_*: its interval overlaps the update's; rocprofv3 shows the kernel)
      hipLaunchKernelGGL(k_rhs_update, dim3((n_pad - (J + 4 * NB) + RHS_UPD_ROWS - 1) / RHS_UPD_ROWS, nb, nrhs), dim3(256), 0, side, A, lda, sys_stride, strip, lda,
                         sys_stride, n_pad, J + 4 * NB, J, 4 * NB);
      BIEM_HIPCHK(hipEventRecord(ev_join, side));
    }
    gemm(st, nb, A, lda, sys_stride, strip, lda, sys_stride, J + 4 * NB, n_pad, J + 4 * NB, n_pad, J, 4 * NB, PK_GEMM, -1.0, nullptr, 0, 0, 0,
         tri_map, true);
    if (beside) {
      BIEM_HIPCHK(hipStreamWaitEvent(st, ev_join, 0));
    } else if (rhs_gemv) {
      ProfScope ps(PK_OTHER, st, 0.0);
      hipLaunchKernelGGL(k_rhs_update, dim3((n_pad - (J + 4 * NB) + RHS_UPD_ROWS - 1) / RHS_UPD_ROWS, nb, nrhs), dim3(256), 0, st, A, lda, sys_stride, strip, lda,
                         sys_stride, n_pad, J + 4 * NB, J, 4 * NB);
    } else if (nrhs > 0) {
      gemm(st, nb, A, lda, sys_stride, strip, lda, sys_stride, J + 4 * NB, n_pad, n_pad, n_cols, J, 4 * NB, PK_OTHER);
    }
  }
  BIEM_LAUNCHCHK();
  if (gemm_rc != BIEM_OK) return gemm_rc;
  {
    // back substitution (k_back_row, bottom block row first); its pass over U also takes the multiplier / growth checks of the
    // strip entries (nrhs == 0: one pass for the checks alone); right-hand sides in groups of up to 8
    ProfScope ps(PK_BACK, st, 4.0 * (double)nb * n_pad * (double)n_pad * nrhs);
    const double inv_rel2 = 1.0 / (nopiv * nopiv);
    cplx* Y = (cplx*)d_work;         // the panel region of the workspace is free in the row form: room for 4 * 64 right-hand sides per system
    // Few systems: the row form has one workgroup per system and 64-row block (a quarter of the CUs busy at 64 systems); the
    // column-block form spreads a system's rows over workgroups (cfg 4, N = 4064: 64 systems 6.1 -> 5.1 ms, 8 systems 5.7 -> 2.6 ms,
    // one system per call 22.8 -> 20.6 ms; at 256+ systems the row form wins: it reads U once in long runs).  BIEM_BACK_FORM=row|col|step forces one.
    if (keep_w) {
      for (int jr = n_pad - BS; jr >= 0; jr -= BS)
        hipLaunchKernelGGL(k_back_step, dim3(jr > 0 ? (jr + BACK_ROWS - 1) / BACK_ROWS : 1, nb), dim3(256), 0, st, A, lda, sys_stride, A + n_pad, lda,
                           sys_stride, Wall, wall_stride, Xsol, n_pad, nrhs, jr, d_info, growth, inv_rel2);
      hipLaunchKernelGGL(k_rhs_compact, dim3((n_pad + 255) / 256, nrhs, nb), dim3(256), 0, st, A, lda, sys_stride, Xsol, nrhs, n_pad, 1);
      BIEM_LAUNCHCHK();
      hipLaunchKernelGGL(k_growth_check, dim3((nb + 63) / 64), dim3(64), 0, st, nb, n_pad, growth, d_info, growth_max);
      BIEM_LAUNCHCHK();
      return BIEM_OK;
    }
    if (nrhs > 4 * NB || (col_form && nrhs > 0)) {
      // (also: more right-hand sides than the compact copy of the row form holds) the column-block form on the augmented columns, same checks
      for (int jr = n_pad - BS; jr >= 0; jr -= BS) {
        hipLaunchKernelGGL(k_back_diag, dim3(nb, nrhs), dim3(64), 0, st, A, lda, sys_stride, A + n_pad, lda, sys_stride, jr);
        if (jr > 0)
          hipLaunchKernelGGL(k_back_update, dim3((jr + BACK_ROWS - 1) / BACK_ROWS, nb), dim3(256), 0, st, A, lda, sys_stride, A + n_pad, lda,
                             sys_stride, nrhs, jr, 0, jr, d_info, growth, inv_rel2);
      }
      BIEM_LAUNCHCHK();
      hipLaunchKernelGGL(k_growth_check, dim3((nb + 63) / 64), dim3(64), 0, st, nb, n_pad, growth, d_info, growth_max);
      BIEM_LAUNCHCHK();
      return BIEM_OK;
    }
    if (nrhs > 0) hipLaunchKernelGGL(k_rhs_compact, dim3((n_pad + 255) / 256, nrhs, nb), dim3(256), 0, st, A, lda, sys_stride, Y, nrhs, n_pad, 0);
    int q0 = 0;
    do {
      const int nq = nrhs - q0 > 8 ? 8 : nrhs - q0;
      const int chk = q0 == 0 ? 1 : 0;
      for (int ib = n_pad / NB - 1; ib >= 0; --ib) {
        if (nq <= 1) hipLaunchKernelGGL(k_back_row<1>, dim3(nb), dim3(1024), 0, st, A, lda, sys_stride, Y, nrhs, n_pad, q0, nq, ib, chk, d_info, growth, inv_rel2);
        else if (nq == 2) hipLaunchKernelGGL(k_back_row<2>, dim3(nb), dim3(1024), 0, st, A, lda, sys_stride, Y, nrhs, n_pad, q0, nq, ib, chk, d_info, growth, inv_rel2);
        else if (nq <= 4) hipLaunchKernelGGL(k_back_row<4>, dim3(nb), dim3(1024), 0, st, A, lda, sys_stride, Y, nrhs, n_pad, q0, nq, ib, chk, d_info, growth, inv_rel2);
        else hipLaunchKernelGGL(k_back_row<8>, dim3(nb), dim3(1024), 0, st, A, lda, sys_stride, Y, nrhs, n_pad, q0, nq, ib, chk, d_info, growth, inv_rel2);
      }
      q0 += nq;
    } while (q0 < nrhs);
    if (nrhs > 0) hipLaunchKernelGGL(k_rhs_compact, dim3((n_pad + 255) / 256, nrhs, nb), dim3(256), 0, st, A, lda, sys_stride, Y, nrhs, n_pad, 1);
    BIEM_LAUNCHCHK();
  }
  hipLaunchKernelGGL(k_growth_check, dim3((nb + 63) / 64), dim3(64), 0, st, nb, n_pad, growth, d_info, growth_max);
  BIEM_LAUNCHCHK();
  return BIEM_OK;
}

// Solve with the stored factors of launch_lu_factor_solve(keep_multipliers = true): the multipliers of a panel are stored in the
// row order its own 64 interchanges left (later panels' interchanges are not applied to them), so the forward substitution
// interleaves interchanges and eliminations panel by panel; L D L^T factors are the case ipiv = identity, U = D L^T.
int launch_lu_solve(int nb, int n_pad, int nrhs, const double* d_LU, long long lda, long long sys_stride, const int* d_ipiv, double* d_B,
                    long long ldb, long long b_stride, hipStream_t st) {
  if (nb <= 0 || n_pad <= 0 || nrhs <= 0) return BIEM_OK;
  if (n_pad % NB) { set_error("biem_lu_solve: n_pad=%d is not a multiple of %d (use biem_lu_npad)", n_pad, NB); return BIEM_ERR_ARG; }
  if (lda < n_pad || ldb < nrhs) { set_error("biem_lu_solve: lda < n_pad or ldb < nrhs"); return BIEM_ERR_ARG; }
  if (nb > 65535 || nrhs > 65535) { set_error("biem_lu_solve: at most 65535 systems / right-hand sides per call (got %d / %d)", nb, nrhs); return BIEM_ERR_ARG; }
  const cplx* A = (const cplx*)d_LU;
  cplx* F = (cplx*)d_B;
  for (int j = 0; j < n_pad; j += NB) {
    hipLaunchKernelGGL(k_fwd_diag, dim3(nb, nrhs), dim3(64), 0, st, A, lda, sys_stride, d_ipiv, n_pad, F, ldb, b_stride, j);
    const int below = n_pad - (j + NB);
    if (below > 0)
      hipLaunchKernelGGL(k_back_update, dim3((below + BACK_ROWS - 1) / BACK_ROWS, nb), dim3(256), 0, st, A, lda, sys_stride, F, ldb, b_stride,
                         nrhs, j, j + NB, n_pad);
  }
  for (int jr = n_pad - BS; jr >= 0; jr -= BS) {
    hipLaunchKernelGGL(k_back_diag, dim3(nb, nrhs), dim3(64), 0, st, A, lda, sys_stride, F, ldb, b_stride, jr);
    if (jr > 0)
      hipLaunchKernelGGL(k_back_update, dim3((jr + BACK_ROWS - 1) / BACK_ROWS, nb), dim3(256), 0, st, A, lda, sys_stride, F, ldb, b_stride,
                         nrhs, jr, 0, jr);
  }
  BIEM_LAUNCHCHK();
  return BIEM_OK;
}

// ---------------------------------------------------------------------------------------------
// microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 (confirms the FP64 matrix peak the roofline is priced against)
// ---------------------------------------------------------------------------------------------
// V = 0: v_mfma_f64_16x16x4_f64 (2048 flops, 32 cycles);  V = 1: v_mfma_f64_4x4x4_4b_f64, the instruction of k_gemm3m_pipe (512 flops,
// 16 cycles).  Both price at 32 flops per cycle and SIMD.
template <int V>
__global__ void __launch_bounds__(256) k_bench_mfma(int iters, double* sink) {
  v4d acc[8];
  double acc1[16];
  for (int i = 0; i < 8; ++i) acc[i] = (v4d){0, 0, 0, 0};
  for (int i = 0; i < 16; ++i) acc1[i] = 0.0;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
    if (V == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[i], 0, 0, 0);
    }
  }
  double sacc = 0.0;
  for (int i = 0; i < 8; ++i) sacc += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) sacc += acc1[i];
  if (sacc == 123.456) sink[0] = sacc;
}

int bench_mfma_f64(int iters, double* tflops, hipStream_t st, int variant) {
  double* sink = nullptr;
  BIEM_HIPCHK(hipMalloc((void**)&sink, 16));
  hipEvent_t e0, e1;
  BIEM_HIPCHK(hipEventCreate(&e0));
  BIEM_HIPCHK(hipEventCreate(&e1));
  const int blocks = 256 * 2;   // 2 workgroups of 4 waves per CU -> 2 waves per SIMD
  auto launch = [&](int n) {
    if (variant == 1) hipLaunchKernelGGL(k_bench_mfma<1>, dim3(blocks), dim3(256), 0, st, n, sink);
    else hipLaunchKernelGGL(k_bench_mfma<0>, dim3(blocks), dim3(256), 0, st, n, sink);
  };
  launch(16);   // warm-up
  BIEM_HIPCHK(hipEventRecord(e0, st));
  launch(iters);
  BIEM_HIPCHK(hipEventRecord(e1, st));
  BIEM_HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  BIEM_HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  const double per_iter = variant == 1 ? 16.0 * (2.0 * 4 * 4 * 4 * 4) : 8.0 * (2.0 * 16 * 16 * 4);
  double flops = (double)blocks * 4.0 * (double)iters * per_iter;
  *tflops = flops / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(sink);
  return BIEM_OK;
}

#ifdef BIEM_GEMM_TRACE
__global__ void k_trace_fill(double* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = 1e-3 * (double)((i * 2654435761ull >> 7) & 1023) / 1024.0 - 5e-4;
}
// one trailing update of an (n x n, K = kd) region of nb systems on synthetic data; returns the stamps and the launch time
extern "C" int biem_debug_gemm(int nb, int n, int kd, int reps, unsigned long long* trace_out, float* ms_out) {
  const long long lda = n + 8, ldp = n + 256;   // the panel workspace is indexed by absolute row
  cplx *A = nullptr, *P = nullptr;
  const size_t na = (size_t)nb * (n + 256) * lda, np = (size_t)nb * 256 * ldp;
  if (hipMalloc((void**)&A, na * sizeof(cplx)) != hipSuccess) return 1;
  if (hipMalloc((void**)&P, np * sizeof(cplx)) != hipSuccess) return 1;
  hipLaunchKernelGGL(k_trace_fill, dim3(2048), dim3(256), 0, 0, (double*)A, na * 2);
  hipLaunchKernelGGL(k_trace_fill, dim3(2048), dim3(256), 0, 0, (double*)P, np * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch_gemm_stream(0, nb, A, lda, (long long)(n + 256) * lda, P, ldp, 256 * ldp, 256, 256 + n, 0, n, 0, kd);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r)
    launch_gemm_stream(0, nb, A, lda, (long long)(n + 256) * lda, P, ldp, 256 * ldp, 256, 256 + n, 0, n, 0, kd);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); *ms_out = ms / reps;
  hipMemcpyFromSymbol(trace_out, HIP_SYMBOL(g_gemm_trace), sizeof(unsigned long long) * 16 * 64 * 8);
  hipFree(A); hipFree(P);
  return 0;
}
#endif
#ifdef BIEM_DIAG_TRACE
extern "C" int biem_debug_diag(int reps, unsigned long long* trace_out, float* us_out) {
  const int n = 1024; const long long lda = n + 8;
  cplx *A = nullptr, *W = nullptr; int* info = nullptr; unsigned long long* growth = nullptr;
  if (hipMalloc((void**)&A, (size_t)n * lda * sizeof(cplx)) != hipSuccess || hipMalloc((void**)&W, NB * NB * sizeof(cplx)) != hipSuccess ||
      hipMalloc((void**)&info, 64) != hipSuccess || hipMalloc((void**)&growth, 64) != hipSuccess) return 1;
  std::vector<cplx> h((size_t)n * lda);
  for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) { const int a = r < c ? r : c, b = r < c ? c : r; h[(size_t)r * lda + c] = make_double2(r == c ? 3.0 : 0.3 * sin(0.37 * a + 1.1 * b), r == c ? 0.4 : 0.2 * cos(0.9 * a - 0.3 * b)); }
  hipMemcpy(A, h.data(), h.size() * sizeof(cplx), hipMemcpyHostToDevice);
  hipMemset(info, 0, 64); hipMemset(growth, 0, 64);
  hipFuncSetAttribute((const void*)k_diag_utu_reg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(DIAG_LDS_CPLX * sizeof(cplx)));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const bool blk = getenv("BIEM_DIAG_FORM") == nullptr;
  hipFuncSetAttribute((const void*)k_diag_utu_blk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(DIAGB_LDS_CPLX * sizeof(cplx)));
  for (int r = 0; r < 3; ++r) {
    if (blk) hipLaunchKernelGGL(k_diag_utu_blk, dim3(1), dim3(1024), DIAGB_LDS_CPLX * sizeof(cplx), 0, A, lda, 0, 64 * r, W, (long long)NB * NB, info, 0.01, growth);
    else hipLaunchKernelGGL(k_diag_utu_reg, dim3(1), dim3(DIAG_THREADS), DIAG_LDS_CPLX * sizeof(cplx), 0, A, lda, 0, 64 * r, W, (long long)NB * NB, info, 0.01, growth);
  }
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) {
    if (blk) hipLaunchKernelGGL(k_diag_utu_blk, dim3(1), dim3(1024), DIAGB_LDS_CPLX * sizeof(cplx), 0, A, lda, 0, 64 * (3 + r % 12), W, (long long)NB * NB, info, 0.01, growth);
    else hipLaunchKernelGGL(k_diag_utu_reg, dim3(1), dim3(DIAG_THREADS), DIAG_LDS_CPLX * sizeof(cplx), 0, A, lda, 0, 64 * (3 + r % 12), W, (long long)NB * NB, info, 0.01, growth);
  }
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); *us_out = ms * 1e3f / reps;
  hipMemcpyFromSymbol(trace_out, HIP_SYMBOL(g_diag_trace), sizeof(unsigned long long) * 16 * 66 * 4);
  hipFree(A); hipFree(W); hipFree(info); hipFree(growth);
  return 0;
}
#endif
#ifdef BIEM_GEMM_TRACE
// one small update launch, as a single system sees it: C[r0 : r0 + rows, 0 : n] -= P^T M[brow ..], `reps` launches back to back; cold = 1:
// every launch takes another row strip (the matrix is far larger than the caches), cold = 0: the same one
extern "C" int biem_debug_gemm_strip(int n, int kd, int rows, int reps, int cold, float* us_out, int extra_cols, int col0) {
  const long long lda = n + 8, ldp = n + 256;
  cplx *A = nullptr, *P = nullptr;
  const size_t na = (size_t)(n + 256) * lda, np = (size_t)256 * ldp;
  if (hipMalloc((void**)&A, na * sizeof(cplx)) != hipSuccess) return 1;
  if (hipMalloc((void**)&P, np * sizeof(cplx)) != hipSuccess) return 1;
  hipLaunchKernelGGL(k_trace_fill, dim3(2048), dim3(256), 0, 0, (double*)A, na * 2);
  hipLaunchKernelGGL(k_trace_fill, dim3(2048), dim3(256), 0, 0, (double*)P, np * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nstrips = (n - rows) / 64;
  for (int r = 0; r < 3; ++r) launch_gemm_stream(0, 1, A, lda, 0, P, ldp, 0, 256, 256 + rows, col0, n + extra_cols, 0, kd);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) {
    const int r0 = 256 + (cold ? 64 * ((r * 7) % nstrips) : 0);
    launch_gemm_stream(0, 1, A, lda, 0, P, ldp, 0, r0, r0 + rows, col0, n + extra_cols, cold ? r0 : 0, kd);
  }
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); *us_out = ms * 1e3f / reps;
  hipFree(A); hipFree(P);
  return 0;
}
#endif

}  // namespace biem
