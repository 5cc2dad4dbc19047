// common.hpp -- small helpers shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "plan.hpp"
#include "../../include/biem_mi355.h"

namespace biem {

typedef double2 cplx;  // .x = re, .y = im

__host__ __device__ inline cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__host__ __device__ inline cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__host__ __device__ inline cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__host__ __device__ inline cplx cscale(cplx a, double s) { return make_double2(a.x * s, a.y * s); }
// a + b*c
__host__ __device__ inline cplx cfma(cplx b, cplx c, cplx a) {
  return make_double2(fma(b.x, c.x, fma(-b.y, c.y, a.x)), fma(b.x, c.y, fma(b.y, c.x, a.y)));
}
// a - b*c
__host__ __device__ inline cplx cfnma(cplx b, cplx c, cplx a) {
  return make_double2(fma(-b.x, c.x, fma(b.y, c.y, a.x)), fma(-b.x, c.y, fma(-b.y, c.x, a.y)));
}
// 1/a with scaling (Smith) so that huge/tiny components do not overflow prematurely
__host__ __device__ inline cplx crecip(cplx a) {
  if (fabs(a.x) >= fabs(a.y)) {
    double r = a.y / a.x, den = a.x + a.y * r;
    return make_double2(1.0 / den, -r / den);
  } else {
    double r = a.x / a.y, den = a.x * r + a.y;
    return make_double2(r / den, -1.0 / den);
  }
}
__host__ __device__ inline cplx cdiv(cplx a, cplx b) { return cmul(a, crecip(b)); }

#define BIEM_HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { biem::set_error("%s failed: %s", #x, hipGetErrorString(e_)); return BIEM_ERR_HIP; } } while (0)
#define BIEM_LAUNCHCHK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { biem::set_error("kernel launch failed at %s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e_)); return BIEM_ERR_HIP; } } while (0)

const char* last_error();

// ---- optional HIP-event profiler (thread-local; off unless biem_profile_begin was called on this thread) ----
enum ProfClass { PK_TABLES = 0, PK_FILL, PK_RHS, PK_PANEL, PK_SWAP, PK_TRSM, PK_GEMM, PK_BACK, PK_OTHER, PK_COUNT };
struct Profiler {
  bool on = false;
  struct Rec { hipEvent_t a, b; int cls; };
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  double work[PK_COUNT] = {0};          // algorithmic flops (GEMM/TRSM/PANEL) or bytes (FILL) per class
  long long launches[PK_COUNT] = {0};
  hipEvent_t get();
};
Profiler& profiler();
struct ProfScope {
  Profiler& p; hipStream_t st; hipEvent_t a = nullptr; int cls;
  ProfScope(int cls_, hipStream_t st_, double work = 0.0) : p(profiler()), st(st_), cls(cls_) {
    if (!p.on) return;
    a = p.get();
    (void)hipEventRecord(a, st);
    p.work[cls] += work; p.launches[cls] += 1;
  }
  ~ProfScope() {
    if (!p.on || !a) return;
    hipEvent_t b = p.get();
    (void)hipEventRecord(b, st);
    p.recs.push_back({a, b, cls});
  }
};

// launch wrappers implemented in the .hip files (all stream-ordered, no syncs)
int launch_radial(int d, int nmax, int count, const double* d_x, double* d_out, hipStream_t st);
int launch_radial_c(int d, int nmax, int count, const double* d_z, double* d_out, hipStream_t st);
int launch_harmonics(const biem_plan* p, int count, const double* d_u, double* d_Y, hipStream_t st);
int launch_ball_tables(const biem_plan* p, int nb, int B, const double* d_k, const double* d_eta, const double* d_radii,
                       int geom_batched, const double* d_alpha, const double* d_beta, int ab_batched, double* d_tab, hipStream_t st);
// slot_order: harmonic h of a ball goes to / comes from the plan's internal slot hpos[h] (symmetric path) instead of position h
int launch_rhs_project(const biem_plan* p, int nb, int B, int nrhs, const double* d_g, double* d_f, long long sys_stride,
                       long long elem_stride, long long rhs_stride, hipStream_t st, bool slot_order = false);
size_t fill_workspace_bytes(const biem_plan* p, int nb, int B);
int launch_fill(const biem_plan* p, int nb, int B, const double* d_k, const double* d_centers, int geom_batched,
                const double* d_tab, int scaling, double* d_A, long long lda, long long sys_stride, int n_pad,
                void* d_work, size_t work_bytes, hipStream_t st);
int launch_density(const biem_plan* p, int nb, int B, int nrhs, const double* d_x, long long sys_stride, long long elem_stride,
                   long long rhs_stride, const double* d_tab, double* d_density, hipStream_t st, bool slot_order = false);
// the complex-symmetric form A~ = R W^H M W R^-1 in the plan's internal slot order, written only where the U^T U factorisation in row
// form (launch_sym_factor_solve) reads it (UPPER triangle + diagonal 64 x 64 tiles); fill_sym_bytes = those bytes per system
// FillDedupe (optional): per-ball radii [B], alpha [B], beta [B] (complex) shared by all systems of the call.  Ball pairs with the
// same displacement vector (to the rounding of the subtraction) and the same (radius, alpha, beta) on either side have IDENTICAL blocks of A~ (translation invariance of
// (S|R)): the block is contracted once and stored to every such pair (lattices of equal spheres: cfg 3 has 24 distinct blocks
// among its 120 pairs).  nullptr: every pair on its own (batched geometry or per-system alpha / beta).
struct FillDedupe { const double* radii; const double* alpha; const double* beta; };
int launch_fill_sym(const biem_plan* p, int nb, int B, const double* d_k, const double* d_centers, int geom_batched, const double* d_tab,
                    double* d_A, long long lda, long long sys_stride, int n_pad, void* d_work, size_t work_bytes, hipStream_t st, bool no_padding = false,
                    const FillDedupe* dedupe = nullptr);
double fill_sym_bytes(int n_pad);
int launch_uscat(const biem_plan* p, int nb, int B, int P, const double* d_k, const double* d_eta, const double* d_centers,
                 const double* d_radii, int geom_batched, const double* d_density, const double* d_points, int flags,
                 double* d_out, void* d_work, size_t work_bytes, hipStream_t st);
int lu_npad(int N);
size_t lu_workspace_bytes(int nb, int n_pad, int nrhs);
int launch_lu_factor_solve(int nb, int n_pad, int nrhs, double* d_A, long long lda, long long sys_stride, int* d_ipiv,
                           int* d_info, void* d_work, size_t work_bytes, hipStream_t st, bool keep_multipliers = true,
                           bool symmetric = false, bool amax_ready = false);
// complex-symmetric A = U^T U in row form on the upper triangle (kernels_lu.hip), fused with the solve of the augmented columns
bool sym_small_path(int n_active, int nrhs);      // whether launch_sym_factor_solve takes its one-launch LDS-resident path
// (n_active: rows n_active .. n_pad-1 are identity padding; systems of at most 128 active rows run in one LDS-resident launch)
int launch_sym_factor_solve(int nb, int n_pad, int nrhs, double* d_A, long long lda, long long sys_stride, int* d_info, void* d_work,
                            size_t work_bytes, hipStream_t st, bool amax_ready = false, int n_active = 0);
// where the symmetric factorisation keeps max |A|, max |U| per system inside its workspace (unsigned 64-bit patterns of doubles)
unsigned long long* lu_growth_slots(void* d_work, int nb, int n_pad);
// preset the slots for a caller that knows (a lower bound of) max |A|: growth[s] = (amax, 0); then pass amax_ready = true
int lu_growth_init(void* d_work, int nb, int n_pad, double amax, hipStream_t st);
// right-hand-side columns (slot order): f~ = R W^H f in place; inverse_on_solution: x = W R^-1 x~
int launch_sym_rhs(const biem_plan* p, int nb, int B, int nrhs, int n_pad, const double* d_tab, double* d_A, long long lda,
                   long long sys_stride, bool inverse_on_solution, hipStream_t st);
int launch_lu_solve(int nb, int n_pad, int nrhs, const double* d_LU, long long lda, long long sys_stride, const int* d_ipiv, double* d_B,
                    long long ldb, long long b_stride, hipStream_t st);
int bench_mfma_f64(int iters, double* tflops, hipStream_t st, int variant = 0);

}  // namespace biem
