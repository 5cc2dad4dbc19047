/*
 * biem_mi355.h  --  C ABI of libbiem_mi355.so: the biem() dense assembly-and-solve hot path on MI355X.
 *
 * The reference (ultrasphere-dev/biem-helmholtz-sphere v1.2.0) is pure Python and has no FFI; the
 * boundary it offers for this path is the Python signature of biem() (src/biem_helmholtz_sphere/
 * _biem.py:453-469).  Every entry point below replaces a span of that function (or of biem_u) and
 * cites it.  The host layer (biem_helmholtz_sphere_amd/_biem.py) binds these with ctypes; what a
 * maintainer of the reference would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no C++/torch types; all sizes are int / long long / size_t; complex128 = 2 doubles.
 *   - pointers named d_* are DEVICE pointers (HIP, gfx950) owned by the caller; h_* are host pointers.
 *   - every function returns a status (0 = BIEM_OK) and never throws; text via biem_last_error().
 *   - all device work is enqueued on the caller-supplied hipStream_t (passed as void*), nothing
 *     synchronises unless stated; the library keeps no global mutable state (error text is
 *     thread-local).  Workspaces are caller-allocated; sizes come from the *_bytes functions.
 *   - `nb` = number of independent (k, eta, incidence) systems in the call (the batch "..." of
 *     _biem.py:80-91), B = balls, H = harmonics of degree < n_end per ball, N = B*H,
 *     Npad = N rounded up (biem_lu_npad) - the LU works on an identity-padded Npad x Npad system.
 *   - order of the harm axis: see biem_plan_labels (this project's choice; the reference's order is
 *     decided inside un-vendored ush.flatten_harmonics and is not pinned by any fixture).
 */
#ifndef BIEM_MI355_H
#define BIEM_MI355_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes */
#define BIEM_OK 0
#define BIEM_ERR_ARG 1          /* bad argument (shape, null pointer, flag) */
#define BIEM_ERR_HIP 2          /* a HIP runtime call failed */
#define BIEM_ERR_UNSUPPORTED 3  /* coordinate tree / size not built */
#define BIEM_ERR_ALLOC 4
#define BIEM_ERR_NO_DEVICE 5    /* no gfx950 device visible */

/* coordinate trees (ultrasphere.create_from_branching_types strings) */
#define BIEM_TREE_A 0    /* "a"   d=2 */
#define BIEM_TREE_BA 1   /* "ba"  d=3, polar axis x0 */
#define BIEM_TREE_BBA 2  /* "bba" d=4 */
#define BIEM_TREE_CAA 3  /* "caa" d=4, Hopf-type: x0 = r cos t0 cos t1, x1 = r cos t0 sin t1, x2 = r sin t0 cos t2, x3 = r sin t0 sin t2 */

/* fill scalings */
#define BIEM_FILL_REFERENCE 0    /* A = blc_{n'} * { diag(alpha h + beta k h') | (S|R)^T (alpha j + beta k j') }  (_biem.py:745-792) */
#define BIEM_FILL_EQUILIBRATED 1 /* M = I + (S|R)^T (alpha j+beta k j')_row / (alpha h+beta k h')_col : what the LU factors */
#define BIEM_FILL_SYMMETRIC 2    /* A~ = R W^H M W R^-1, complex symmetric (W: unitary map to real harmonics, R = diag(1/sqrt(gj gh))):
                                    what the symmetric path factors.  Rows / columns of a ball in the internal slot order of
                                    biem_plan_symmetric_order; ONLY the upper triangle and the diagonal 64 x 64 tiles are written
                                    (n_pad must be a multiple of 64, lda >= n_pad); everything else is left untouched.  That is the input of
                                    biem_sym_factor_solve (row form, upper triangle) ONLY: biem_ldlt_factor / biem_ldlt_factor_solve read the
                                    LOWER triangle, which this fill does not write */

/* uscat flags */
#define BIEM_USCAT_FAR_FIELD 1
#define BIEM_USCAT_PER_BALL 2
#define BIEM_USCAT_KIND_INNER 4
#define BIEM_USCAT_POINTS_BATCHED 8 /* points carry their own batch axis (expand_x=False, _biem.py:879-883) */

typedef struct biem_plan biem_plan; /* opaque: host+device tables of one (tree, n_end) */

int biem_version(void);
/* hash of the csrc/ + include/ sources the library was built from (the Python loader rebuilds a library whose hash differs) */
const char* biem_build_id(void);
const char* biem_last_error(void);
/* number of visible HIP devices; BIEM_ERR_NO_DEVICE if none */
int biem_device_count(int* n);

/* ---- plan: k- and geometry-independent tables (replaces the table side of ush.expand /
 *      ush.harmonics_translation_coef / ush.index_array_harmonics, _biem.py:627,651,697,720) ---- */
/* host-only tables (no GPU needed): labels, quadrature, projection matrix, translation terms */
int biem_plan_create_host(int tree, int n_end, biem_plan** plan);
/* upload the tables to the current HIP device (idempotent) */
int biem_plan_upload(biem_plan* plan);
/* = create_host + upload */
int biem_plan_create(int tree, int n_end, biem_plan** plan);
int biem_plan_destroy(biem_plan* plan);
/* d, H (degree < n_end), Q quadrature points per ball, H2 (degree < 2 n_end - 1), number of translation terms */
int biem_plan_info(const biem_plan* plan, int* d, int* n_harm, int* n_quad, int* n_harm2, long long* n_terms);
/* h_labels[H][3]: a:(m,0,0)  ba:(n,m,0)  bba:(n,l,m)  caa:(n,m1,m2);  h_deg[H]: degree n */
int biem_plan_labels(const biem_plan* plan, int* h_labels, int* h_deg);
/* the real-harmonic form used by the symmetric path: h_partner[h] = p with conj Y_h = Y_p (p == h: a real harmonic); h_slot[h] =
 * internal position of harmonic h among its ball's H unknowns in that path: for a unit (h <= p) the "cosine" combination
 * (Y_h + Y_p)/sqrt2 sits in slot h_slot[h], the "sine" combination i (Y_p - Y_h)/sqrt2 in slot h_slot[p] */
int biem_plan_symmetric_order(const biem_plan* plan, int* h_partner /*[H]*/, int* h_slot /*[H]*/);
/* which forms of the symmetric fill this plan's tables admit (host information, no device needed): n_units = label units E of the
 * reduced pair table (a degree < 2 n_end - 1 label and its conjugate partner share an entry), n_phases = distinct azimuthal phases,
 * reduced_ok = 1 if every term list is of one kind and one phase and a wave's lists fit LDS (the default fill kernel), lds_rows =
 * term rows (64 lanes each) of its largest chunk, gather_ok = 1 if the gather form it replaced fits */
int biem_plan_fill_info(const biem_plan* plan, int* n_units, int* n_phases, int* reduced_ok, int* lds_rows, int* gather_ok);
/* unit vectors y[Q][d] and weights w[Q] of the boundary-data rule (ush.expand(n=n_end)) */
int biem_plan_quadrature(const biem_plan* plan, double* h_y, double* h_w);
/* projection matrix W[Q][H] (complex128, host copy):  f_h = sum_q W[q][h] g(y_q),  W = w_q conj(Y_h(y_q)) */
int biem_plan_projection(const biem_plan* plan, double* h_W);
/* translation terms in CSR over entries e = h*H + h' (row h = test index, column h' = unknown):
 * (S|R)_{h'->h}(t) = sum_{p in [ptr[e],ptr[e+1])} coef[p] * T[tidx[p]],  T[l] = C_d h_{n''}(k|t|) Y_l(t^), l over labels of degree < 2 n_end - 1 */
int biem_plan_terms(const biem_plan* plan, long long* h_ptr /*[H*H+1]*/, double* h_coef, int* h_tidx);

/* ---- special functions on the device, exposed for parity tests (ultrasphere.shn1 / potential_coef) ---- */
/* z[i][0..nmax] (j) and [nmax+1 .. 2nmax+1] (y): d-dimensional spherical Bessel functions at x[i] */
int biem_radial(int d, int nmax, int count, const double* d_x, double* d_out /*[count][2][nmax+1]*/, void* stream);
/* complex arguments z[i] (complex128): out[i][0][n] = z_n regular, out[i][1][n] = h_n = j_n + i y_n outgoing (complex128);
 * h is computed directly (closed forms / modified Bessel functions), never as the cancelling sum j + i y */
int biem_radial_complex(int d, int nmax, int count, const double* d_z /*[count] c128*/, double* d_out /*[count][2][nmax+1] c128*/,
                        void* stream);
/* Y[p][h] (complex128) at directions d_u[p][d] (need not be normalised) for all labels of degree < n_end */
int biem_harmonics(const biem_plan* plan, int count, const double* d_u, double* d_Y /*[count][H] c128*/, void* stream);

/* ---- wavenumbers: every d_k below is [nb] COMPLEX128 (re, im interleaved), the reference's k of dtype result_type(k, complex)
 *      (gui.py:296-301 passes complex k); systems with Im k == 0 take the real-argument special functions. ---- */
/* ---- per-ball tables (ush.harmonics_regular_singular_component + potential_coef, _biem.py:723-789) ----
 * d_tab[nb][B][3][n_end] complex128: [0] gj = alpha j_n + beta k j_n', [1] gh = alpha h_n + beta k h_n', [2] blc_n = dlc - i eta slc
 * geometry arrays are [nb][B]... when geom_batched != 0, else [B]... shared by all systems; alpha/beta likewise (ab_batched). */
int biem_ball_tables(const biem_plan* plan, int nb, int B, const double* d_k /*c128*/, const double* d_eta,
                     const double* d_radii, int geom_batched, const double* d_alpha /*c128*/, const double* d_beta /*c128*/,
                     int ab_batched, double* d_tab, void* stream);

/* ---- right-hand side (ush.expand, _biem.py:627-639): the incident field stays a Python callable; only its samples
 *      g[s][r][b][q] = (-alpha u_in - beta d_n u_in)(c_b + rho_b y_q) cross the ABI; r = 0..nrhs-1 are right-hand sides that
 *      share system s (incidences on which k, eta, the geometry and alpha/beta do not depend: factor once, solve many).
 *      f element (s, r, b, h) is written to d_f[s*sys_stride + (b*H + h)*elem_stride + r*rhs_stride]  (complex128 units) ---- */
int biem_rhs_project(const biem_plan* plan, int nb, int B, int nrhs, const double* d_g /*[nb][nrhs][B][Q] c128*/, double* d_f,
                     long long sys_stride, long long elem_stride, long long rhs_stride, void* stream);

/* ---- matrix fill (ush.harmonics_translation_coef + the where/create_diagonal/moveaxis block, _biem.py:694-792) ----
 * Writes the N x N matrix of every system, row-major [b][h][b'][h'] with leading dimension lda (complex128 units),
 * system s at d_A + 2*s*sys_stride doubles.  When lda > N / rows_pad > N the caller gets an identity-padded
 * Npad x Npad block (columns >= Npad are left untouched: that is where the LU keeps right-hand sides).
 * d_tab from biem_ball_tables.  Workspace: biem_fill_workspace_bytes. */
size_t biem_fill_workspace_bytes(const biem_plan* plan, int nb, int B);
int biem_fill(const biem_plan* plan, int nb, int B, const double* d_k /*c128*/, const double* d_centers /*[nb or 1][B][d]*/,
              int geom_batched, const double* d_tab, int scaling, double* d_A, long long lda, long long sys_stride,
              int n_pad, void* d_work, size_t work_bytes, void* stream);

/* ---- dense complex LU (batch_tensorsolve.btensorsolve -> linalg.solve, _biem.py:797) ----
 * Right-looking blocked LU with partial (row) pivoting in groups of four 64-column panels; trailing updates are 3M zgemm
 * (K = 64 / 128 inside a group, one K = 256 update per group) on v_mfma_f64_4x4x4_4b_f64.
 * The system is the augmented row-major [A | F]: Npad rows, Npad + nrhs columns, leading dimension lda >= Npad + nrhs.
 * biem_lu_factor_solve overwrites F with the solution (forward elimination rides in the trailing update, then a
 * blocked back substitution); A is overwritten by U and the un-permuted multipliers. */
int biem_lu_npad(int N);
size_t biem_lu_workspace_bytes(int nb, int n_pad, int nrhs);   /* panels of a group + 64 x 64 block per system + tile map of the symmetric path */
int biem_lu_factor_solve(int nb, int n_pad, int nrhs, double* d_A, long long lda, long long sys_stride,
                         int* d_ipiv /*[nb][n_pad]*/, int* d_info /*[nb]*/, void* d_work, size_t work_bytes, void* stream);

/* Factor now, solve later (the reference's btensorsolve has no such split: linalg.solve factors again for every call, _biem.py:797).
 * biem_lu_factor leaves in A the factor U (on and above the diagonal) and the multipliers (below), d_ipiv the interchanges:
 * row j was exchanged with row ipiv[j] >= j when column j was eliminated; the multipliers of a 64-column panel are stored in the
 * row order that panel's own interchanges left (LINPACK style: later interchanges are NOT applied to earlier multipliers), which
 * is what biem_lu_solve expects.  biem_lu_solve overwrites B (row-major [Npad][ldb] per system, nrhs <= ldb columns, system s at
 * d_B + 2*s*b_stride doubles; rows N..Npad-1 of an identity-padded system must hold zeros) with the solutions; it needs no
 * workspace and may be called any number of times.  d_info as in biem_lu_factor_solve. */
int biem_lu_factor(int nb, int n_pad, double* d_A, long long lda, long long sys_stride, int* d_ipiv /*[nb][n_pad]*/, int* d_info /*[nb]*/,
                   void* d_work, size_t work_bytes /* biem_lu_workspace_bytes(nb, n_pad, 0) */, void* stream);
int biem_lu_solve(int nb, int n_pad, int nrhs, const double* d_LU, long long lda, long long sys_stride, const int* d_ipiv,
                  double* d_B, long long ldb, long long b_stride, void* stream);
/* the same split for complex-symmetric systems (see biem_ldlt_factor_solve below): A <- L (unit lower, below the diagonal) and
 * U = D L^T (on and above), ipiv = identity; solve with biem_lu_solve.  d_info[s] < 0: rejected, use biem_lu_factor. */
int biem_ldlt_factor(int nb, int n_pad, double* d_A, long long lda, long long sys_stride, int* d_ipiv, int* d_info, void* d_work,
                     size_t work_bytes, void* stream);

/* Complex-SYMMETRIC systems (A = A^T, not Hermitian): A = L D L^T with the diagonal as pivots, no interchanges.  Same layout,
 * workspace and kernels as biem_lu_factor_solve; a panel's U rows are its transposed multipliers and the K = 256 updates run
 * over the lower triangle of tiles only (half the flops).  Only the lower triangle (and the diagonal 64 x 64 blocks) of A is
 * read.  d_info[s] = -(row+1), `row` = first row of the 64-column panel in which a diagonal entry was below 0.01 x the largest
 * entry of its (updated) column, i.e. a multiplier exceeded 100 (partial pivoting guarantees 1; 10 until round 2; or NaN);
 * d_info[s] = -(Npad+1): a-posteriori growth check failed, max |U| > 200 max |A| (|.| = |re| + |im|, A = the part read) or a
 * non-finite entry in U: the result of that system is not to be trusted and the caller re-solves it with biem_lu_factor_solve. */
int biem_ldlt_factor_solve(int nb, int n_pad, int nrhs, double* d_A, long long lda, long long sys_stride,
                           int* d_ipiv /*[nb][n_pad], identity on return*/, int* d_info /*[nb]*/, void* d_work, size_t work_bytes,
                           void* stream);

/* The same systems in ROW form, which is what biem_solve_ldlt runs: A = U^T U, U upper triangular (the complex-symmetric analogue
 * of Cholesky: U = D^{1/2} L^T with the L, D above; principal complex square roots, no conjugation, no interchanges).  Both
 * operands of every trailing update come from a 64-row strip of the row-major matrix itself, so the factorisation works in place
 * on the UPPER triangle: only the upper triangle and the diagonal 64 x 64 tiles of A are read; on return they hold U, the
 * augmented columns the solutions.  Same workspace size, d_info codes (rejected multiplier: -(first row of the 64-row panel + 1);
 * growth: -(Npad + 1)) and acceptance tests as biem_ldlt_factor_solve. */
int biem_sym_factor_solve(int nb, int n_pad, int nrhs, double* d_A, long long lda, long long sys_stride, int* d_info /*[nb]*/,
                          void* d_work, size_t work_bytes, void* stream);

/* density[s][r][b][h] = x / (gh * blc): the reference's `density` from the equilibrated unknowns (also the
 * single-ball shortcut _biem.py:648-691 with x = f).  x element (s, r, i) at d_x[s*sys_stride + i*elem_stride + r*rhs_stride]. */
int biem_density(const biem_plan* plan, int nb, int B, int nrhs, const double* d_x, long long sys_stride, long long elem_stride,
                 long long rhs_stride, const double* d_tab, double* d_density /*[nb][nrhs][B][H] c128*/, void* stream);

/* ---- field evaluation (biem_u, _biem.py:822-977) ----
 * d_points[d][P] (or [d][P][nb] with BIEM_USCAT_POINTS_BATCHED); out[P][nb] or [P][nb][B] (per ball), complex128.
 * Workspace: biem_uscat_workspace_bytes (holds density * blc). */
size_t biem_uscat_workspace_bytes(const biem_plan* plan, int nb, int B);
int biem_uscat(const biem_plan* plan, int nb, int B, int P, const double* d_k /*c128*/, const double* d_eta,
               const double* d_centers, const double* d_radii, int geom_batched, const double* d_density,
               const double* d_points, int flags, double* d_out, void* d_work, size_t work_bytes, void* stream);

/* ---- one call for the whole path: ball tables + fill (equilibrated) + LU + density, systems processed in
 *      chunks of `chunk` resident matrices (0 = choose); every system is factored once for its nrhs right-hand sides.
 *      d_g [nb][nrhs][B][Q] as in biem_rhs_project, d_density [nb][nrhs][B][H]. ---- */
size_t biem_solve_workspace_bytes(const biem_plan* plan, int nb, int B, int nrhs, int chunk);
int biem_solve(const biem_plan* plan, int nb, int B, int nrhs, const double* d_k /*c128*/, const double* d_eta, const double* d_centers,
               const double* d_radii, int geom_batched, const double* d_alpha, const double* d_beta, int ab_batched,
               const double* d_g, double* d_density, int* d_info, int chunk, void* d_work, size_t work_bytes, void* stream);

/* The same through the complex-symmetric form of the system: with W the unitary map to real harmonics and
 * R = diag(1/sqrt(gj gh)), R W^H M W R^-1 is complex symmetric (the reference solves the general system with LU,
 * _biem.py:797; the symmetry is a property of (S|R), tests/test_oracle_golden.py).  Symmetric fill (upper triangle), U^T U
 * factorisation without interchanges (biem_sym_factor_solve), back-transform, density.  d_info[s] < 0: a diagonal pivot was rejected (see biem_ldlt_factor_solve) - the caller
 * re-solves those systems with biem_solve.  Same arguments and workspace as biem_solve.
 * With geom_batched = 0 and ab_batched = 0 the fill contracts the block of ball pairs that share their displacement vector and
 * their (radius, alpha, beta) on either side once and stores it to each of them (identical blocks by translation invariance;
 * displacements are matched to the rounding of the subtraction, so on an exactly representable lattice the densities are bit for bit
 * those of the pair-by-pair fill, otherwise equal to rounding); systems of at most 128 unknowns (N + nrhs <= 128) are factorised and
 * solved by one launch. */
int biem_solve_ldlt(const biem_plan* plan, int nb, int B, int nrhs, const double* d_k /*c128*/, const double* d_eta,
                    const double* d_centers, const double* d_radii, int geom_batched, const double* d_alpha, const double* d_beta,
                    int ab_batched, const double* d_g, double* d_density, int* d_info, int chunk, void* d_work, size_t work_bytes,
                    void* stream);

/* ---- per-kernel-class timing with HIP events on the launch stream (thread-local; used by bench.py for the
 *      live `roofline` figures).  Between begin and end every launch of the calling thread is bracketed by two
 *      events; end synchronises on them and returns, per class, elapsed ms, algorithmic work and launch count.
 *      Classes: 0 tables, 1 fill (work = bytes written), 2 rhs, 3 panel, 4 swap (LU row interchanges; symmetrising transform of the LDL^T path), 5 trsm, 6 gemm = the K=256 trailing
 *      updates of the four-panel groups (work = real flops, 8 per complex multiply-add), 7 back substitution, 8 the K=64 and
 *      K=128 updates inside a group. */
#define BIEM_PROFILE_CLASSES 9
int biem_profile_begin(void);
int biem_profile_end(double* ms /*[9]*/, double* work /*[9]*/, long long* launches /*[9]*/);

/* ---- microbenchmarks used by bench.py / DESIGN.md (peak checks, not part of the path) ---- */
/* issues `iters` dependent-free v_mfma_f64_16x16x4_f64 per wave on every SIMD; returns achieved FLOP/s */
int biem_bench_mfma_f64(int iters, double* tflops, void* stream);
/* the same with a choice of instruction: variant 0 v_mfma_f64_16x16x4_f64, variant 1 v_mfma_f64_4x4x4_4b_f64 (the one the trailing
 * update uses).  What a pure MFMA stream SUSTAINS on the box it runs on (1 MI355X, 2 waves per SIMD, 2-second runs: 47.3 and
 * 75.3 TFLOP/s) - bench.py reports variant 1 beside the 78.6 TFLOP/s the roofline is priced against */
int biem_bench_mfma_f64_ex(int iters, int variant, double* tflops, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BIEM_MI355_H */
