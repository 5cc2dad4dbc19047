// abi.cpp -- extern "C" entry points of libbiem_mi355.so (declared in include/biem_mi355.h).
#include "common.hpp"
#include <cstdlib>
#include <cstring>
#include <new>

using namespace biem;

namespace biem {
Profiler& profiler() { static thread_local Profiler p; return p; }
hipEvent_t Profiler::get() {
  hipEvent_t e = nullptr;
  if (!pool.empty()) { e = pool.back(); pool.pop_back(); return e; }
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace biem

#define NEED(p, what) do { if (!(p)) { set_error("%s: null %s", __func__, what); return BIEM_ERR_ARG; } } while (0)
#define NEED_DEV(pl) do { NEED(pl, "plan"); if ((pl)->device < 0) { set_error("%s: plan not uploaded to a device (biem_plan_upload)", __func__); return BIEM_ERR_ARG; } } while (0)

extern "C" {

int biem_version(void) { return 200; }   // 0.2.0

// hash of the sources this library was built from (set by _build.py; the loader compares it with the checkout's)
#ifndef BIEM_SRC_HASH
#define BIEM_SRC_HASH "unknown"
#endif
extern const char biem_src_hash_marker[] = "BIEM_SRC_HASH=" BIEM_SRC_HASH ";";
const char* biem_build_id(void) { return BIEM_SRC_HASH; }

const char* biem_last_error(void) { return last_error(); }

int biem_device_count(int* n) {
  NEED(n, "n");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess || c <= 0) { *n = 0; set_error("no HIP device visible: %s", hipGetErrorString(e)); return BIEM_ERR_NO_DEVICE; }
  *n = c;
  return BIEM_OK;
}

int biem_plan_create_host(int tree, int n_end, biem_plan** plan) {
  NEED(plan, "plan");
  biem_plan* p = new (std::nothrow) biem_plan();
  if (!p) { set_error("out of host memory"); return BIEM_ERR_ALLOC; }
  int rc = BIEM_OK;
  try {
    rc = plan_build_host(p, tree, n_end);
  } catch (const std::bad_alloc&) {
    set_error("out of host memory building tables for n_end=%d", n_end);
    rc = BIEM_ERR_ALLOC;
  }
  if (rc != BIEM_OK) { delete p; *plan = nullptr; return rc; }
  *plan = p;
  return BIEM_OK;
}

int biem_plan_upload(biem_plan* plan) { NEED(plan, "plan"); return plan_upload(plan); }

int biem_plan_create(int tree, int n_end, biem_plan** plan) {
  int n = 0;
  int rc = biem_device_count(&n);
  if (rc != BIEM_OK) return rc;
  rc = biem_plan_create_host(tree, n_end, plan);
  if (rc != BIEM_OK) return rc;
  rc = plan_upload(*plan);
  if (rc != BIEM_OK) { plan_free(*plan); *plan = nullptr; }
  return rc;
}

int biem_plan_destroy(biem_plan* plan) { if (plan) plan_free(plan); return BIEM_OK; }

int biem_plan_info(const biem_plan* plan, int* d, int* n_harm, int* n_quad, int* n_harm2, long long* n_terms) {
  NEED(plan, "plan");
  if (d) *d = plan->d;
  if (n_harm) *n_harm = plan->H;
  if (n_quad) *n_quad = plan->Q;
  if (n_harm2) *n_harm2 = plan->H2;
  if (n_terms) *n_terms = (long long)plan->coef.size();
  return BIEM_OK;
}

int biem_plan_labels(const biem_plan* plan, int* h_labels, int* h_deg) {
  NEED(plan, "plan");
  if (h_labels) memcpy(h_labels, plan->labels.data(), plan->labels.size() * sizeof(int));
  if (h_deg) memcpy(h_deg, plan->deg.data(), plan->deg.size() * sizeof(int));
  return BIEM_OK;
}

int biem_plan_symmetric_order(const biem_plan* plan, int* h_partner, int* h_slot) {
  NEED(plan, "plan");
  const int U = (int)(plan->units.size() / 2);
  for (int u = 0; u < U; ++u) {
    const int h = plan->units[2 * u], p = plan->units[2 * u + 1];
    if (h_partner) { h_partner[h] = p; h_partner[p] = h; }
  }
  if (h_slot) memcpy(h_slot, plan->hpos.data(), plan->hpos.size() * sizeof(int));
  return BIEM_OK;
}

int biem_plan_fill_info(const biem_plan* plan, int* n_units, int* n_phases, int* reduced_ok, int* lds_rows, int* gather_ok) {
  NEED(plan, "plan");
  if (n_units) *n_units = plan->E;
  if (n_phases) *n_phases = plan->NP;
  if (reduced_ok) *reduced_ok = plan->red_lists_ok ? 1 : 0;
  if (lds_rows) *lds_rows = plan->rchunk_rows_max;
  if (gather_ok) *gather_ok = (plan->pair_lists_ok && plan->qchunk.size() > 1) ? 1 : 0;
  return BIEM_OK;
}

int biem_plan_quadrature(const biem_plan* plan, double* h_y, double* h_w) {
  NEED(plan, "plan");
  if (h_y) memcpy(h_y, plan->qy.data(), plan->qy.size() * sizeof(double));
  if (h_w) memcpy(h_w, plan->qw.data(), plan->qw.size() * sizeof(double));
  return BIEM_OK;
}

int biem_plan_projection(const biem_plan* plan, double* h_W) {
  NEED(plan, "plan"); NEED(h_W, "h_W");
  memcpy(h_W, plan->W.data(), plan->W.size() * sizeof(double));
  return BIEM_OK;
}

int biem_plan_terms(const biem_plan* plan, long long* h_ptr, double* h_coef, int* h_tidx) {
  NEED(plan, "plan");
  if (!plan->lists_built) { set_error("biem_plan_terms: a 2-D plan of order n_end > %d holds no term lists (one Graf term per entry, evaluated directly)", kLists2dMax); return BIEM_ERR_UNSUPPORTED; }
  if (h_ptr) for (size_t i = 0; i < plan->ptr.size(); ++i) h_ptr[i] = (long long)plan->ptr[i];
  if (h_coef) memcpy(h_coef, plan->coef.data(), plan->coef.size() * sizeof(double));
  if (h_tidx) memcpy(h_tidx, plan->tidx.data(), plan->tidx.size() * sizeof(int));
  return BIEM_OK;
}

int biem_radial(int d, int nmax, int count, const double* d_x, double* d_out, void* stream) {
  NEED(d_x, "d_x"); NEED(d_out, "d_out");
  return launch_radial(d, nmax, count, d_x, d_out, (hipStream_t)stream);
}

int biem_radial_complex(int d, int nmax, int count, const double* d_z, double* d_out, void* stream) {
  NEED(d_z, "d_z"); NEED(d_out, "d_out");
  return launch_radial_c(d, nmax, count, d_z, d_out, (hipStream_t)stream);
}

int biem_harmonics(const biem_plan* plan, int count, const double* d_u, double* d_Y, void* stream) {
  NEED_DEV(plan); NEED(d_u, "d_u"); NEED(d_Y, "d_Y");
  return launch_harmonics(plan, count, d_u, d_Y, (hipStream_t)stream);
}

int biem_ball_tables(const biem_plan* plan, int nb, int B, const double* d_k, const double* d_eta, const double* d_radii,
                     int geom_batched, const double* d_alpha, const double* d_beta, int ab_batched, double* d_tab, void* stream) {
  NEED_DEV(plan); NEED(d_k, "d_k"); NEED(d_eta, "d_eta"); NEED(d_radii, "d_radii"); NEED(d_alpha, "d_alpha"); NEED(d_beta, "d_beta"); NEED(d_tab, "d_tab");
  return launch_ball_tables(plan, nb, B, d_k, d_eta, d_radii, geom_batched, d_alpha, d_beta, ab_batched, d_tab, (hipStream_t)stream);
}

int biem_rhs_project(const biem_plan* plan, int nb, int B, int nrhs, const double* d_g, double* d_f, long long sys_stride,
                     long long elem_stride, long long rhs_stride, void* stream) {
  NEED_DEV(plan); NEED(d_g, "d_g"); NEED(d_f, "d_f");
  return launch_rhs_project(plan, nb, B, nrhs, d_g, d_f, sys_stride, elem_stride, rhs_stride, (hipStream_t)stream);
}

size_t biem_fill_workspace_bytes(const biem_plan* plan, int nb, int B) { return plan ? fill_workspace_bytes(plan, nb, B) : 0; }

int biem_fill(const biem_plan* plan, int nb, int B, const double* d_k, const double* d_centers, int geom_batched,
              const double* d_tab, int scaling, double* d_A, long long lda, long long sys_stride, int n_pad, void* d_work,
              size_t work_bytes, void* stream) {
  NEED_DEV(plan); NEED(d_k, "d_k"); NEED(d_centers, "d_centers"); NEED(d_tab, "d_tab"); NEED(d_A, "d_A");
  if (B > 1) NEED(d_work, "d_work");
  if (scaling == BIEM_FILL_SYMMETRIC)
    return launch_fill_sym(plan, nb, B, d_k, d_centers, geom_batched, d_tab, d_A, lda, sys_stride, n_pad, d_work, work_bytes, (hipStream_t)stream);
  return launch_fill(plan, nb, B, d_k, d_centers, geom_batched, d_tab, scaling, d_A, lda, sys_stride, n_pad, d_work, work_bytes,
                     (hipStream_t)stream);
}

int biem_lu_npad(int N) { return lu_npad(N); }
size_t biem_lu_workspace_bytes(int nb, int n_pad, int nrhs) { return lu_workspace_bytes(nb, n_pad, nrhs); }

int biem_lu_factor_solve(int nb, int n_pad, int nrhs, double* d_A, long long lda, long long sys_stride, int* d_ipiv, int* d_info,
                         void* d_work, size_t work_bytes, void* stream) {
  NEED(d_A, "d_A"); NEED(d_ipiv, "d_ipiv"); NEED(d_info, "d_info"); NEED(d_work, "d_work");
  // BIEM_LU_DISCARD_FACTORS=1 (tests): solve exactly as the fused path does, without storing the multipliers back
  const char* e = getenv("BIEM_LU_DISCARD_FACTORS");
  return launch_lu_factor_solve(nb, n_pad, nrhs, d_A, lda, sys_stride, d_ipiv, d_info, d_work, work_bytes, (hipStream_t)stream,
                                !(e && e[0] == '1'));
}

int biem_ldlt_factor_solve(int nb, int n_pad, int nrhs, double* d_A, long long lda, long long sys_stride, int* d_ipiv, int* d_info,
                           void* d_work, size_t work_bytes, void* stream) {
  NEED(d_A, "d_A"); NEED(d_ipiv, "d_ipiv"); NEED(d_info, "d_info"); NEED(d_work, "d_work");
  const char* e = getenv("BIEM_LU_DISCARD_FACTORS");
  return launch_lu_factor_solve(nb, n_pad, nrhs, d_A, lda, sys_stride, d_ipiv, d_info, d_work, work_bytes, (hipStream_t)stream,
                                !(e && e[0] == '1'), /*symmetric=*/true);
}

int biem_lu_factor(int nb, int n_pad, double* d_A, long long lda, long long sys_stride, int* d_ipiv, int* d_info, void* d_work,
                   size_t work_bytes, void* stream) {
  NEED(d_A, "d_A"); NEED(d_ipiv, "d_ipiv"); NEED(d_info, "d_info"); NEED(d_work, "d_work");
  return launch_lu_factor_solve(nb, n_pad, 0, d_A, lda, sys_stride, d_ipiv, d_info, d_work, work_bytes, (hipStream_t)stream, true, false);
}

int biem_ldlt_factor(int nb, int n_pad, double* d_A, long long lda, long long sys_stride, int* d_ipiv, int* d_info, void* d_work,
                     size_t work_bytes, void* stream) {
  NEED(d_A, "d_A"); NEED(d_ipiv, "d_ipiv"); NEED(d_info, "d_info"); NEED(d_work, "d_work");
  return launch_lu_factor_solve(nb, n_pad, 0, d_A, lda, sys_stride, d_ipiv, d_info, d_work, work_bytes, (hipStream_t)stream, true, true);
}

int biem_lu_solve(int nb, int n_pad, int nrhs, const double* d_LU, long long lda, long long sys_stride, const int* d_ipiv, double* d_B,
                  long long ldb, long long b_stride, void* stream) {
  NEED(d_LU, "d_LU"); NEED(d_ipiv, "d_ipiv"); NEED(d_B, "d_B");
  return launch_lu_solve(nb, n_pad, nrhs, d_LU, lda, sys_stride, d_ipiv, d_B, ldb, b_stride, (hipStream_t)stream);
}

int biem_sym_factor_solve(int nb, int n_pad, int nrhs, double* d_A, long long lda, long long sys_stride, int* d_info, void* d_work,
                          size_t work_bytes, void* stream) {
  NEED(d_A, "d_A"); NEED(d_info, "d_info"); NEED(d_work, "d_work");
  return launch_sym_factor_solve(nb, n_pad, nrhs, d_A, lda, sys_stride, d_info, d_work, work_bytes, (hipStream_t)stream, false);
}

int biem_density(const biem_plan* plan, int nb, int B, int nrhs, const double* d_x, long long sys_stride, long long elem_stride,
                 long long rhs_stride, const double* d_tab, double* d_density, void* stream) {
  NEED_DEV(plan); NEED(d_x, "d_x"); NEED(d_tab, "d_tab"); NEED(d_density, "d_density");
  return launch_density(plan, nb, B, nrhs, d_x, sys_stride, elem_stride, rhs_stride, d_tab, d_density, (hipStream_t)stream);
}

size_t biem_uscat_workspace_bytes(const biem_plan* plan, int nb, int B) {
  return plan ? (size_t)nb * B * plan->H * sizeof(cplx) : 0;
}

int biem_uscat(const biem_plan* plan, int nb, int B, int P, const double* d_k, const double* d_eta, const double* d_centers,
               const double* d_radii, int geom_batched, const double* d_density, const double* d_points, int flags, double* d_out,
               void* d_work, size_t work_bytes, void* stream) {
  NEED_DEV(plan); NEED(d_k, "d_k"); NEED(d_eta, "d_eta"); NEED(d_centers, "d_centers"); NEED(d_radii, "d_radii");
  NEED(d_density, "d_density"); NEED(d_points, "d_points"); NEED(d_out, "d_out"); NEED(d_work, "d_work");
  return launch_uscat(plan, nb, B, P, d_k, d_eta, d_centers, d_radii, geom_batched, d_density, d_points, flags, d_out, d_work,
                      work_bytes, (hipStream_t)stream);
}

// ---- whole path -----------------------------------------------------------------------------
namespace {
struct SolveLayout {
  int N, n_pad, chunk;
  long long lda, sys_stride;
  size_t off_tab, off_A, off_T, off_P, off_ipiv, total;
};
inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
SolveLayout make_layout(const biem_plan* p, int nb, int B, int nrhs, int chunk) {
  SolveLayout L;
  L.N = B * p->H;
  L.n_pad = lu_npad(L.N);
  L.lda = L.n_pad + ((nrhs + 7) / 8) * 8;   // right-hand side columns; rows stay 128-byte aligned (8 complex128)
  L.sys_stride = (long long)L.n_pad * L.lda;
  if (chunk <= 0) {
    // resident matrices per chunk: as many as fit ~24 GiB, at most nb
    size_t per = (size_t)L.sys_stride * 16 + (size_t)B * B * p->H2 * 16 + lu_workspace_bytes(1, L.n_pad, nrhs);
    size_t fit = ((size_t)24 << 30) / (per ? per : 1);
    chunk = (int)(fit < 1 ? 1 : (fit > (size_t)nb ? (size_t)nb : fit));
  }
  if (chunk > nb) chunk = nb;
  if (chunk > 32768) chunk = 32768;            // grid y / z dimensions of the per-system kernels
  L.chunk = chunk;
  size_t o = 0;
  L.off_tab = o; o = align256(o + (size_t)nb * B * 3 * p->n_end * 16);
  L.off_A = o; o = align256(o + (size_t)chunk * L.sys_stride * 16);
  L.off_T = o; o = align256(o + fill_workspace_bytes(p, chunk, B));
  L.off_P = o; o = align256(o + lu_workspace_bytes(chunk, L.n_pad, nrhs));
  L.off_ipiv = o; o = align256(o + (size_t)chunk * L.n_pad * sizeof(int));
  L.total = o;
  return L;
}
}  // namespace

size_t biem_solve_workspace_bytes(const biem_plan* plan, int nb, int B, int nrhs, int chunk) {
  if (!plan || nb <= 0 || B <= 0 || nrhs < 1) return 0;
  return make_layout(plan, nb, B, nrhs, chunk).total;
}

static int solve_impl(const biem_plan* plan, int nb, int B, int nrhs, const double* d_k, const double* d_eta, const double* d_centers,
               const double* d_radii, int geom_batched, const double* d_alpha, const double* d_beta, int ab_batched,
               const double* d_g, double* d_density, int* d_info, int chunk, void* d_work, size_t work_bytes, void* stream,
               bool symmetric) {
  NEED_DEV(plan); NEED(d_k, "d_k"); NEED(d_eta, "d_eta"); NEED(d_centers, "d_centers"); NEED(d_radii, "d_radii");
  NEED(d_alpha, "d_alpha"); NEED(d_beta, "d_beta"); NEED(d_g, "d_g"); NEED(d_density, "d_density"); NEED(d_info, "d_info"); NEED(d_work, "d_work");
  if (nb <= 0 || B <= 0) return BIEM_OK;
  if (nrhs < 1) { set_error("biem_solve: nrhs < 1"); return BIEM_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  SolveLayout L = make_layout(plan, nb, B, nrhs, chunk);
  if (work_bytes < L.total) { set_error("biem_solve: workspace too small (%zu < %zu)", work_bytes, L.total); return BIEM_ERR_ARG; }
  char* w = (char*)d_work;
  double* tab = (double*)(w + L.off_tab);
  double* A = (double*)(w + L.off_A);
  void* T = w + L.off_T;
  void* Pw = w + L.off_P;
  int* ipiv = (int*)(w + L.off_ipiv);
  const int H = plan->H, d = plan->d, Q = plan->Q;
  int rc = launch_ball_tables(plan, nb, B, d_k, d_eta, d_radii, geom_batched, d_alpha, d_beta, ab_batched, tab, st);
  if (rc) return rc;
  for (int s0 = 0; s0 < nb; s0 += L.chunk) {
    const int c = (nb - s0 < L.chunk) ? nb - s0 : L.chunk;
    const double* ks = d_k + 2 * (size_t)s0;   // complex128 per system
    const double* cen = d_centers + (geom_batched ? (size_t)s0 * B * d : 0);
    const double* tb = tab + (size_t)s0 * B * 3 * plan->n_end * 2;
    // right-hand side into column n_pad of the augmented matrix (padded rows: zero via fill_pad? -> set explicitly below)
    // small systems are solved in one LDS-resident launch that never reads the padding rows / columns: they are not written then
    const bool small = symmetric && sym_small_path(L.N, nrhs);
    // blocks of ball pairs with the same displacement and the same (radius, alpha, beta) on either side are contracted once
    const FillDedupe dd = {d_radii, d_alpha, d_beta};
    if (symmetric)
      rc = launch_fill_sym(plan, c, B, ks, cen, geom_batched, tb, A, L.lda, L.sys_stride, L.n_pad, T, fill_workspace_bytes(plan, c, B), st, small,
                           (!geom_batched && !ab_batched) ? &dd : nullptr);
    else
      rc = launch_fill(plan, c, B, ks, cen, geom_batched, tb, BIEM_FILL_EQUILIBRATED, A, L.lda, L.sys_stride, L.n_pad, T,
                       fill_workspace_bytes(plan, c, B), st);
    if (rc) return rc;
    if (!small)
      BIEM_HIPCHK(hipMemset2DAsync(A + (size_t)L.n_pad * 2, (size_t)L.lda * 16, 0, (size_t)(L.lda - L.n_pad) * 16,
                                   (size_t)L.n_pad * c, st));
    rc = launch_rhs_project(plan, c, B, nrhs, d_g + (size_t)s0 * nrhs * B * Q * 2, A + (size_t)L.n_pad * 2, L.sys_stride, L.lda, 1, st, symmetric);
    if (rc) return rc;
    bool amax_ready = false;
    if (symmetric) {
      rc = launch_sym_rhs(plan, c, B, nrhs, L.n_pad, tb, A, L.lda, L.sys_stride, false, st);
      if (rc) return rc;
      // growth check of the factorisation: max |A~| >= 1 (its diagonal is exactly 1), so 1 is a valid, conservative reference
      // (a system is handed to the pivoted LU when max |u_ii u_ic| exceeds GROWTH_MAX = 200 x this lower bound); saves a pass over the matrix
      rc = lu_growth_init(Pw, c, L.n_pad, 1.0, st);
      if (rc) return rc;
      amax_ready = true;
    }
    if (symmetric)
      rc = launch_sym_factor_solve(c, L.n_pad, nrhs, A, L.lda, L.sys_stride, d_info + s0, Pw, lu_workspace_bytes(c, L.n_pad, nrhs), st, amax_ready, L.N);
    else
      rc = launch_lu_factor_solve(c, L.n_pad, nrhs, A, L.lda, L.sys_stride, ipiv, d_info + s0, Pw, lu_workspace_bytes(c, L.n_pad, nrhs), st,
                                  /*keep_multipliers=*/false, false, false);   // the fused path only needs the solution
    if (rc) return rc;
    if (symmetric) {
      rc = launch_sym_rhs(plan, c, B, nrhs, L.n_pad, tb, A, L.lda, L.sys_stride, true, st);
      if (rc) return rc;
    }
    rc = launch_density(plan, c, B, nrhs, A + (size_t)L.n_pad * 2, L.sys_stride, L.lda, 1, tb, d_density + (size_t)s0 * nrhs * B * H * 2, st, symmetric);
    if (rc) return rc;
  }
  return BIEM_OK;
}

int biem_solve(const biem_plan* plan, int nb, int B, int nrhs, const double* d_k, const double* d_eta, const double* d_centers,
               const double* d_radii, int geom_batched, const double* d_alpha, const double* d_beta, int ab_batched,
               const double* d_g, double* d_density, int* d_info, int chunk, void* d_work, size_t work_bytes, void* stream) {
  return solve_impl(plan, nb, B, nrhs, d_k, d_eta, d_centers, d_radii, geom_batched, d_alpha, d_beta, ab_batched, d_g, d_density, d_info,
                    chunk, d_work, work_bytes, stream, false);
}

int biem_solve_ldlt(const biem_plan* plan, int nb, int B, int nrhs, const double* d_k, const double* d_eta, const double* d_centers,
                    const double* d_radii, int geom_batched, const double* d_alpha, const double* d_beta, int ab_batched,
                    const double* d_g, double* d_density, int* d_info, int chunk, void* d_work, size_t work_bytes, void* stream) {
  return solve_impl(plan, nb, B, nrhs, d_k, d_eta, d_centers, d_radii, geom_batched, d_alpha, d_beta, ab_batched, d_g, d_density, d_info,
                    chunk, d_work, work_bytes, stream, true);
}

int biem_profile_begin(void) {
  Profiler& p = profiler();
  for (auto& r : p.recs) { p.pool.push_back(r.a); p.pool.push_back(r.b); }
  p.recs.clear();
  for (int i = 0; i < PK_COUNT; ++i) { p.work[i] = 0.0; p.launches[i] = 0; }
  p.on = true;
  return BIEM_OK;
}

int biem_profile_end(double* ms, double* work, long long* launches) {
  Profiler& p = profiler();
  p.on = false;
  double acc[PK_COUNT] = {0};
  for (auto& r : p.recs) {
    (void)hipEventSynchronize(r.b);
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) acc[r.cls] += t;
    p.pool.push_back(r.a); p.pool.push_back(r.b);
  }
  p.recs.clear();
  for (int i = 0; i < PK_COUNT; ++i) {
    if (ms) ms[i] = acc[i];
    if (work) work[i] = p.work[i];
    if (launches) launches[i] = p.launches[i];
  }
  return BIEM_OK;
}

int biem_bench_mfma_f64(int iters, double* tflops, void* stream) {
  NEED(tflops, "tflops");
  return bench_mfma_f64(iters, tflops, (hipStream_t)stream);
}
int biem_bench_mfma_f64_ex(int iters, int variant, double* tflops, void* stream) {
  NEED(tflops, "tflops");
  if (iters <= 0 || (variant != 0 && variant != 1)) { set_error("biem_bench_mfma_f64_ex: iters > 0, variant 0 or 1"); return BIEM_ERR_ARG; }
  return bench_mfma_f64(iters, tflops, (hipStream_t)stream, variant);
}

}  // extern "C"
