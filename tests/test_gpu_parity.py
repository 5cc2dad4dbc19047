"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the reference's goldens.

Tolerances (complex fp64): BASELINE north_star asks <= 1e-10 relative on u_scat; component checks are tighter.
"""
import csv
import ctypes as C
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import biem_oracle as O  # noqa: E402  (test infrastructure: the checker)


@pytest.fixture(scope="module")
def amd():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import biem_helmholtz_sphere_amd as amd

    return amd


@pytest.fixture(scope="module")
def lib(amd):
    from biem_helmholtz_sphere_amd import _lib as L

    return L.load(), L


def _dev(a, dtype=torch.float64):
    return torch.as_tensor(np.array(a), device="cuda").to(dtype).contiguous()


# ---------------------------------------------------------------------------- special functions
@pytest.mark.parametrize("d", [2, 3, 4])
def test_radial_complex_device_vs_scipy(lib, d):
    """Complex arguments (complex wavenumber): regular z_n and outgoing h_n against SciPy's Amos routines, RELATIVE error -
    h_n is computed directly, so it stays accurate where j and y are e^{2 Im z} times larger (parity unpinned by the
    reference: no fixture has complex k; SciPy is the checker)."""
    l, L = lib
    nmax = 50
    zs = np.array([0.3 + 0.1j, 1.3 + 0.25j, 1.9 + 1.0j, 2.5 + 0.01j, 5 + 3j, 12 + 0.5j, 20 + 8j, 40 + 0.2j, 3 - 0.4j, 0.05 + 0.02j,
                   8 + 15j, 60 + 30j, 1e-3 + 1e-3j, 2.0 + 1e-9j, 7.0 + 0.0j])
    z = _dev(zs, torch.complex128)
    out = torch.zeros((len(zs), 2, nmax + 1), dtype=torch.complex128, device="cuda")
    L.check(l.biem_radial_complex(d, nmax, len(zs), z.data_ptr(), out.data_ptr(), None))
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for i, zv in enumerate(zs):
        j, h, _, _ = O.radial_h(nmax, d, zv if zv.imag != 0 else zv.real)
        ok = np.isfinite(h) & (np.abs(h) < 1e290) & (np.abs(j) > 1e-290)
        assert np.max(np.abs(out[i, 0][ok] / j[ok] - 1)) < 5e-12, (d, zv)
        assert np.max(np.abs(out[i, 1][ok] / h[ok] - 1)) < 5e-12, (d, zv)


@pytest.mark.parametrize("d", [2, 3, 4])
def test_radial_device_vs_scipy(lib, d):
    l, L = lib
    nmax = 60
    xs = np.concatenate([np.geomspace(0.05, 150, 80), [1.0, 2.404825557695773, 3.141592653589793, 136.0]])
    x = _dev(xs)
    out = torch.zeros((len(xs), 2, nmax + 1), dtype=torch.float64, device="cuda")
    L.check(l.biem_radial(d, nmax, len(xs), x.data_ptr(), out.data_ptr(), None))
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for i, xv in enumerate(xs):
        j, y, _, _ = O.radial(nmax, d, xv)
        ok = np.isfinite(y) & (np.abs(y) < 1e290)
        env = np.abs(j + 1j * y)[ok]
        assert np.max(np.abs(out[i, 0][ok] - j[ok]) / env) < 2e-12 * max(1.0, xv / 10), (d, xv)
        assert np.max(np.abs(out[i, 1][ok] - y[ok]) / env) < 2e-12 * max(1.0, xv / 10), (d, xv)
        # small regular functions keep RELATIVE accuracy (they scale the matrix rows)
        small = ok & (np.arange(nmax + 1) > xv) & (np.abs(j) > 1e-280)
        if small.any():
            assert np.max(np.abs(out[i, 0][small] / j[small] - 1)) < 1e-11, (d, xv)


@pytest.mark.parametrize("tree,n_end", [("a", 9), ("ba", 7), ("bba", 5), ("caa", 6)])
def test_harmonics_device_vs_oracle(lib, tree, n_end):
    l, L = lib
    plan = C.c_void_p()
    L.check(l.biem_plan_create(L.TREE_IDS[tree], n_end, C.byref(plan)))
    tr = O.tree(tree)
    rng = np.random.default_rng(0)
    u = rng.normal(size=(50, tr.d))
    u[0] = 0; u[0, 0] = 1.0           # poles
    u[1] = 0; u[1, 0] = -2.0
    u[2] = 0; u[2, 1] = 1.5
    H = tr.n_harm(n_end)
    Y = torch.zeros((50, H), dtype=torch.complex128, device="cuda")
    L.check(l.biem_harmonics(plan, 50, _dev(u).data_ptr(), Y.data_ptr(), None))
    torch.cuda.synchronize()
    Yo = tr.harmonics(u / np.linalg.norm(u, axis=-1, keepdims=True), n_end).T
    assert np.abs(Y.cpu().numpy() - Yo).max() < 1e-12
    l.biem_plan_destroy(plan)


# ---------------------------------------------------------------------------- fill
def _rand_geometry(rng, B, d, rmin=0.4, rmax=1.0, gap=1.15):
    cen, rad = [], []
    while len(cen) < B:
        c = rng.uniform(-4, 4, size=d)
        r = rng.uniform(rmin, rmax)
        if all(np.linalg.norm(c - c2) > gap * (r + r2) for c2, r2 in zip(cen, rad)):
            cen.append(c)
            rad.append(r)
    return np.array(cen), np.array(rad)


@pytest.mark.parametrize("tree,n_end,B", [("a", 6, 3), ("ba", 5, 3), ("bba", 4, 2), ("ba", 9, 2)])
def test_fill_reference_scaling_vs_oracle(amd, tree, n_end, B):
    """`matrix` attribute (reference scaling, _biem.py:745-792) element-wise against the oracle, batch of 2 k's."""
    rng = np.random.default_rng(5)
    tr = O.tree(tree)
    cen, rad = _rand_geometry(rng, B, tr.d)
    ks = np.array([0.9, 2.3])
    eta = np.array([1.0, 0.6])
    alpha, beta = 1.0 + 0.25j, 0.4 - 0.1j
    c = amd.create_from_branching_types(tree)
    calc = amd.biem(c, centers=_dev(cen)[None].expand(2, B, tr.d), radii=_dev(rad)[None].expand(2, B), k=_dev(ks), eta=_dev(eta),
                    n_end=n_end, alpha=alpha, beta=beta)
    assert calc.density is None
    M = calc.matrix.cpu().numpy()
    H = tr.n_harm(n_end)
    assert M.shape == (2, B, H, B, H)
    for s in range(2):
        A, _ = O.assemble(tr, n_end, ks[s], eta[s], cen, rad, np.full(B, alpha), np.full(B, beta))
        scale = np.abs(A).max(axis=(2, 3), keepdims=True) * 0 + np.abs(A)  # elementwise where nonzero
        err = np.abs(M[s] - A)
        # relative to the larger of the entry itself and 1e-14 * the row-block scale
        ref = np.maximum(np.abs(A), 1e-30)
        nz = np.abs(A) > 1e-200
        assert np.max(err[nz] / ref[nz]) < 5e-11, (tree, s)
        assert np.all(M[s][~nz] == 0)


# ---------------------------------------------------------------------------- LU
# n_pad = 1, 2, 3, 4 panels (one group of the K = 256 schedule and its tails), 5, 6, 7 (a group plus every tail), 9, 16
@pytest.mark.parametrize("N,nb,nrhs", [(64, 3, 1), (72, 2, 2), (150, 2, 3), (200, 2, 1), (300, 2, 2), (380, 1, 9), (440, 2, 1),
                                       (576, 1, 3), (1000, 2, 1)])
def test_lu_factor_solve_vs_numpy(lib, N, nb, nrhs):
    l, L = lib
    rng = np.random.default_rng(N)
    npad = l.biem_lu_npad(N)
    lda = npad + nrhs
    A = np.zeros((nb, npad, lda), dtype=np.complex128)
    As = rng.normal(size=(nb, N, N)) + 1j * rng.normal(size=(nb, N, N))
    Fs = rng.normal(size=(nb, N, nrhs)) + 1j * rng.normal(size=(nb, N, nrhs))
    A[:, :N, :N] = As
    for i in range(N, npad):
        A[:, i, i] = 1.0
    A[:, :N, npad:] = Fs
    dA = _dev(A, torch.complex128)
    ipiv = torch.zeros((nb, npad), dtype=torch.int32, device="cuda")
    info = torch.ones(nb, dtype=torch.int32, device="cuda")
    wb = l.biem_lu_workspace_bytes(nb, npad, nrhs)
    work = torch.empty(wb, dtype=torch.uint8, device="cuda")
    L.check(l.biem_lu_factor_solve(nb, npad, nrhs, dA.data_ptr(), lda, npad * lda, ipiv.data_ptr(), info.data_ptr(), work.data_ptr(), wb, None))
    torch.cuda.synchronize()
    X = dA.cpu().numpy()[:, :N, npad:]
    assert (info.cpu().numpy() == 0).all()
    for s in range(nb):
        Xo = np.linalg.solve(As[s], Fs[s])
        # backward-error style check and forward check (Gaussian matrices: cond ~ N)
        res = np.abs(As[s] @ X[s] - Fs[s]).max() / (np.abs(As[s]).sum(axis=1).max() * np.abs(X[s]).max())
        assert res < 1e-13, (N, s, res)
        assert np.abs(X[s] - Xo).max() / np.abs(Xo).max() < 1e-9
    if N > npad - 1:
        return
    # padded rows stay zero in the solution
    assert np.abs(dA.cpu().numpy()[:, N:, npad:]).max() == 0


def test_lu_singular_reports_info(lib):
    l, L = lib
    N = 64
    A = np.zeros((1, N, N + 1), dtype=np.complex128)
    A[0, :, :N] = np.eye(N)
    A[0, 10, 10] = 0.0
    dA = _dev(A, torch.complex128)
    ipiv = torch.zeros((1, N), dtype=torch.int32, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    wb = l.biem_lu_workspace_bytes(1, N, 1)
    work = torch.empty(wb, dtype=torch.uint8, device="cuda")
    L.check(l.biem_lu_factor_solve(1, N, 1, dA.data_ptr(), N + 1, N * (N + 1), ipiv.data_ptr(), info.data_ptr(), work.data_ptr(), wb, None))
    torch.cuda.synchronize()
    assert int(info.cpu()[0]) == 11


# ---------------------------------------------------------------------------- end to end
def _oracle_case(tree, cen, rad, k, n_end, eta, alpha, beta, direction, x):
    uin, ugr = O.plane_wave(k, direction)
    res = O.solve_biem(tree, centers=cen, radii=rad, k=k, n_end=n_end, eta=eta, alpha=alpha, beta=beta, uin=uin, uin_grad=ugr)
    return res, O.uscat(res, x)


@pytest.mark.parametrize(
    "tree,B,n_end,alpha,beta",
    [("a", 4, 10, 1.0, 0.0), ("a", 3, 8, 1.0, 1.0), ("ba", 2, 6, 1.0, 0.0), ("ba", 3, 7, 0.0, 1.0),
     ("ba", 3, 6, 1.0 + 0.5j, 0.3 - 0.2j), ("bba", 2, 4, 1.0, 0.0), ("bba", 2, 4, 1.0, 1.0), ("caa", 2, 4, 1.0, 0.5)],
)
def test_biem_end_to_end_vs_oracle(amd, tree, B, n_end, alpha, beta):
    rng = np.random.default_rng(11)
    tr = O.tree(tree)
    d = tr.d
    cen, rad = _rand_geometry(rng, B, d)
    k, eta = 1.4, 1.3
    direction = rng.normal(size=d)
    x = rng.normal(size=(7, d)) * 6.0
    x = x[[all(np.linalg.norm(p - c) > r for c, r in zip(cen, rad)) for p in x]]
    res, uo = _oracle_case(tree, cen, rad, k, n_end, eta, alpha, beta, direction, x)
    c = amd.create_from_branching_types(tree)
    uin, ugr = amd.plane_wave(k=_dev(k), direction=_dev(direction))
    calc = amd.biem(c, centers=_dev(cen), radii=_dev(rad), k=_dev(k), eta=_dev(eta), n_end=n_end, alpha=alpha, beta=beta,
                    uin=uin, uin_grad=ugr)
    dens = calc.density.cpu().numpy()
    assert dens.shape == res.density.shape
    # density: relative to the per-degree scale (entries span many decades)
    assert np.max(np.abs(dens - res.density) / (np.abs(res.density) + 1e-12 * np.abs(res.density).max())) < 1e-8
    u = calc.uscat(_dev(x.T)).cpu().numpy()
    assert np.max(np.abs(u - uo) / np.abs(uo)) < 1e-10
    # per-ball and far field
    upb = calc.uscat(_dev(x.T), per_ball=True).cpu().numpy()
    assert np.max(np.abs(upb - O.uscat(res, x, per_ball=True))) < 1e-10 * np.abs(uo).max()
    xf = x / np.linalg.norm(x, axis=-1, keepdims=True)
    uf = calc.uscat(_dev(xf.T), far_field=True).cpu().numpy()
    ufo = O.uscat(res, xf, far_field=True)
    assert np.max(np.abs(uf - ufo) / np.abs(ufo).max()) < 1e-10


def test_single_ball_shortcut_and_force_matrix(amd):
    tr = O.tree("ba")
    cen, rad = np.array([[0.3, -0.2, 0.1]]), np.array([0.8])
    k, n_end = 2.0, 8
    x = np.array([[2.0, 0.5, -1.0], [0.0, 3.0, 0.0]])
    res, uo = _oracle_case("ba", cen, rad, k, n_end, 1.0, 1.0, 0.5, [1.0, 0.2, 0.0], x)
    c = amd.create_from_branching_types("ba")
    uin, ugr = amd.plane_wave(k=_dev(k), direction=_dev([1.0, 0.2, 0.0]))
    kw = dict(centers=_dev(cen), radii=_dev(rad), k=_dev(k), n_end=n_end, alpha=1.0, beta=0.5, uin=uin, uin_grad=ugr)
    a = amd.biem(c, **kw)
    b = amd.biem(c, force_matrix=True, **kw)
    assert a.matrix is None and b.matrix is not None
    for calc in (a, b):
        assert np.max(np.abs(calc.uscat(_dev(x.T)).cpu().numpy() - uo) / np.abs(uo)) < 1e-10


def test_uscat_nan_mask_and_inner_kind(amd):
    c = amd.create_from_branching_types("a")
    uin, _ = amd.plane_wave(k=_dev(1.0), direction=_dev([1.0, 0.0]))
    cen = np.array([[0.0, 2.0], [0.0, -2.0]])
    calc = amd.biem(c, centers=_dev(cen), radii=_dev([1.0, 1.0]), k=_dev(1.0), n_end=5, uin=uin)
    x = _dev(np.array([[0.0, 0.0], [0.0, 2.2], [5.0, 0.0]]).T)
    u = calc.uscat(x).cpu().numpy()
    assert np.isfinite(u[0]) and np.isnan(u[1]) and np.isfinite(u[2])
    assert np.isnan(calc.uscat(x, per_ball=True).cpu().numpy()[1]).all()
    with pytest.raises(ValueError):
        amd.BIEMResultCalculator(c=c, centers=calc.centers, radii=calc.radii, k=calc.k, n_end=5, eta=calc.eta, kind="outer").uscat(x)


# ---------------------------------------------------------------------------- goldens straight through the product
def test_readme_doctest_numpy_in_numpy_out(amd):
    """README.md:117-124 of the reference, verbatim call pattern with NumPy arrays."""
    xp = np
    c = amd.create_from_branching_types("ba")
    uin, uin_grad = amd.plane_wave(k=xp.asarray(1.0), direction=xp.asarray((1.0, 0.0, 0.0)))
    calc = amd.biem(c, uin=uin, uin_grad=uin_grad, k=xp.asarray(1.0), n_end=6, eta=xp.asarray(1.0),
                    centers=xp.asarray(((0.0, 2.0, 0.0), (0.0, -2.0, 0.0))), radii=xp.asarray((1.0, 1.0)), kind="outer")
    u = calc.uscat(xp.asarray((0.0, 0.0, 0.0)))
    assert isinstance(u, np.ndarray)
    assert complex(xp.round(u, 6)) == pytest.approx(-0.741333 - 0.669657j, abs=1e-12)
    # 17-digit golden, accuracy_k_ba.csv:52
    assert abs(complex(u) - (-0.74133301331334 - 0.6696574197988229j)) < 1e-12


def test_golden_rows_through_gpu(amd, golden_dir):
    def run(tree, n_end, k_op, cen):
        c = amd.create_from_branching_types(tree)
        d = c.c_ndim
        e0 = np.zeros(d); e0[0] = 1.0
        uin, _ = amd.plane_wave(k=_dev(1.0), direction=_dev(e0))      # incident k = 1 quirk (cli.py:238-244)
        calc = amd.biem(c, centers=_dev(cen), radii=_dev(np.ones(len(cen))), k=_dev(k_op), eta=_dev(1.0), n_end=n_end, uin=uin)
        return complex(calc.uscat(_dev(np.zeros(d))).cpu().numpy())

    n = 0
    with open(os.path.join(golden_dir, "jascome_output.csv")) as f:
        for r in csv.DictReader(f):
            bt, n_end = r["branching_types"], int(r["n_end"])
            if n_end <= 6:
                tol = 2e-12 if n_end == 6 else 1e-12        # triplet drift of the reference itself at n_end = 6
                assert abs(run(bt, n_end, 1.0, O.grid_centers(0, O.tree(bt).d)) - complex(r["uscat"])) < tol, (bt, n_end)
                n += 1
    with open(os.path.join(golden_dir, "accuracy_k_ba.csv")) as f:
        for i, r in enumerate(csv.DictReader(f)):
            n_end = int(r["n_end"])
            if i % 23 == 0 and n_end <= 30:
                d = O.tree(r["branching_types"]).d
                assert abs(run(r["branching_types"], n_end, float(r["k"]), O.grid_centers(0, d)) - complex(r["uscat"])) < 1e-11, (r["branching_types"], n_end, r["k"])
                n += 1
    half = {4: 1, 16: 2, 64: 4}
    with open(os.path.join(golden_dir, "accuracy_n_balls_a.csv")) as f:
        for r in csv.DictReader(f):
            nbal, n_end = int(r["n_balls"]), int(r["n_end"])
            if nbal in half and n_end in (3, 13, 32, 64) and nbal * (2 * n_end - 1) <= 2100:
                assert abs(run("a", n_end, 1.0, O.grid_centers(half[nbal], 2)) - complex(r["uscat"])) < 1e-11, (nbal, n_end)
                n += 1
    assert n > 57


def test_batch_of_wavenumbers_matches_one_by_one(amd):
    """cfg-3 style batch (k sweep, shared geometry) == the same systems solved one at a time; oracle at two of them."""
    c = amd.create_from_branching_types("ba")
    cen = O.grid_centers(1, 3)
    B = len(cen)
    ks = np.linspace(0.5, 3.0, 5)
    dirs = np.tile(np.array([[1.0], [0.0], [0.0]]), (1, 5))
    uin, _ = amd.plane_wave(k=_dev(ks), direction=_dev(dirs))
    calc = amd.biem(c, centers=_dev(cen)[None], radii=_dev(np.ones(B))[None], k=_dev(ks), n_end=6, uin=uin, chunk=2)
    x = np.array([[0.0, 0.0, 0.0], [9.0, 1.0, 0.5]])
    u = calc.uscat(_dev(x.T)).cpu().numpy()           # (P, nb)
    assert u.shape == (2, 5)
    for s in (0, 3):
        res, uo = _oracle_case("ba", cen, np.ones(B), ks[s], 6, 1.0, 1.0, 0.0, [1.0, 0.0, 0.0], x)
        assert np.max(np.abs(u[:, s] - uo) / np.abs(uo)) < 1e-10


def test_resident_bytes_cap_of_the_default_chunk(amd, monkeypatch):
    """BIEM_MAX_RESIDENT_BYTES bounds the workspace biem() takes when `chunk` is left to it (by default 85 % of the memory the process can
    get): a cap of two systems' worth streams the batch through two resident matrices - same densities, bit for bit."""
    from biem_helmholtz_sphere_amd import _biem as impl

    c = amd.create_from_branching_types("ba")
    cen = O.grid_centers(1, 3)
    ks = np.linspace(0.5, 3.0, 5)
    dirs = np.tile(np.array([[1.0], [0.0], [0.0]]), (1, 5))
    uin, _ = amd.plane_wave(k=_dev(ks), direction=_dev(dirs))
    kw = dict(centers=_dev(cen)[None], radii=_dev(np.ones(4))[None], k=_dev(ks), n_end=6, uin=uin)
    ref = amd.biem(c, **kw).density
    plan = impl._plan("ba", 6, torch.device("cuda", 0))
    lib_ = impl.L.load()
    per = int(lib_.biem_solve_workspace_bytes(plan.handle, 1, 4, 1, 1))
    monkeypatch.setenv("BIEM_MAX_RESIDENT_BYTES", str(2 * per + per // 2))
    peak0 = torch.cuda.max_memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    got = amd.biem(c, **kw).density
    assert torch.equal(got, ref)
    assert torch.cuda.max_memory_allocated() - base < 3 * per + (8 << 20), (torch.cuda.max_memory_allocated() - base, per)
    del peak0


def test_mfma_f64_rate_is_reported(lib):
    l, L = lib
    t = C.c_double()
    L.check(l.biem_bench_mfma_f64(20000, C.byref(t), None))
    print("v_mfma_f64_16x16x4_f64 issue-rate microbenchmark: %.1f TFLOP/s" % t.value)
    assert 20.0 < t.value < 200.0


# ---------------------------------------------------------------------------- wider API surface
def test_sweep_driver_reproduces_reference_csv(amd, golden_dir, tmp_path):
    """The jascome-style sweep writes the reference's schema and its values match the committed file (n_end <= 5)."""
    from biem_helmholtz_sphere_amd import sweep

    out = tmp_path / "jascome_output.csv"
    sweep.main(["jascome", "--out", str(out), "--types", "a,ba,bpa,bba,bpbpa,caa", "--n-end-max", "5"])
    with open(out) as f:
        mine = {(r["branching_types"], int(r["n_end"])): complex(r["uscat"]) for r in csv.DictReader(f)}
    with open(os.path.join(golden_dir, "jascome_output.csv")) as f:
        hdr = f.readline()
    assert open(out).readline() == hdr
    n = 0
    with open(os.path.join(golden_dir, "jascome_output.csv")) as f:
        for r in csv.DictReader(f):
            key = (r["branching_types"], int(r["n_end"]))
            if key in mine:
                assert abs(mine[key] - complex(r["uscat"])) < 1e-12, key
                n += 1
    assert n == 30


def test_point_source_incident_field(amd):
    """point_source (reference :391-450): incident field h_n(k |x - s|) through the HIP radial kernel; vs the oracle."""
    k, src, n = 1.3, np.array([0.5, -4.0, 1.0]), 1
    cen = np.array([[0.0, 1.5, 0.0], [0.2, -1.4, 0.3]])
    rad = np.array([0.9, 0.7])
    uo, go = O.point_source(k, src, n)
    res = O.solve_biem("ba", centers=cen, radii=rad, k=k, n_end=7, alpha=1.0, beta=0.3, uin=uo, uin_grad=go)
    x = np.array([[3.0, 0.0, 0.5], [0.0, 0.0, 4.0]])
    ref = O.uscat(res, x)
    c = amd.create_from_branching_types("ba")
    uin, ugr = amd.point_source(k=_dev(k), source=_dev(src), n=n)
    calc = amd.biem(c, centers=_dev(cen), radii=_dev(rad), k=_dev(k), n_end=7, alpha=1.0, beta=0.3, uin=uin, uin_grad=ugr)
    u = calc.uscat(_dev(x.T)).cpu().numpy()
    assert np.max(np.abs(u - ref) / np.abs(ref)) < 1e-10


def test_float32_inputs_and_per_ball_alpha_beta_batch(amd):
    """float32 arrays (reference tests/conftest.py:54-56) give complex64 results; per-ball alpha/beta with a k batch."""
    c = amd.create_from_branching_types("a")
    cen = np.array([[0.0, 2.0], [0.0, -2.0], [4.0, 0.5]])
    rad = np.array([1.0, 0.8, 0.5])
    ks = np.array([0.7, 1.9])
    alpha = np.array([[1.0, 0.0, 1.0 + 0.5j], [1.0, 0.0, 1.0 + 0.5j]])
    beta = np.array([[0.0, 1.0, 0.2], [0.0, 1.0, 0.2]])
    dirs = np.tile(np.array([[1.0], [0.4]]), (1, 2))
    uin, ugr = amd.plane_wave(k=_dev(ks), direction=_dev(dirs))
    calc = amd.biem(c, centers=_dev(cen)[None], radii=_dev(rad)[None], k=_dev(ks), n_end=9, alpha=_dev(alpha, torch.complex128),
                    beta=_dev(beta, torch.complex128), uin=uin, uin_grad=ugr)
    x = np.array([[6.0, 1.0], [-3.0, 0.0]])
    u = calc.uscat(_dev(x.T)).cpu().numpy()
    for s in range(2):
        uo, go = O.plane_wave(ks[s], [1.0, 0.4])
        res = O.solve_biem("a", centers=cen, radii=rad, k=ks[s], n_end=9, alpha=alpha[s], beta=beta[s], uin=uo, uin_grad=go)
        ref = O.uscat(res, x)
        assert np.max(np.abs(u[:, s] - ref) / np.abs(ref)) < 1e-10
    # float32 in -> complex64 out, same values to single precision
    uin32, _ = amd.plane_wave(k=_dev(1.0, torch.float32), direction=_dev([1.0, 0.0], torch.float32))
    c32 = amd.biem(c, centers=_dev(cen, torch.float32), radii=_dev(rad, torch.float32), k=_dev(1.0, torch.float32), n_end=6, uin=uin32)
    assert c32.density.dtype == torch.complex64
    u32 = c32.uscat(_dev(x.T, torch.float32))
    assert u32.dtype == torch.complex64
    uo, _ = O.plane_wave(1.0, [1.0, 0.0])
    ref = O.uscat(O.solve_biem("a", centers=cen, radii=rad, k=1.0, n_end=6, uin=uo), x)
    assert np.max(np.abs(u32.cpu().numpy() - ref) / np.abs(ref)) < 1e-5


def test_factor_once_solve_many_incidences(amd, lib):
    """Incidences on which the operator does not depend ride as right-hand sides of one factorisation:
    k of shape (1, K), directions of shape (d, R, 1) -> density (R, K, B, H); equal to solving every (r, k) on its own."""
    c = amd.create_from_branching_types("ba")
    cen = np.array([[0.0, 1.6, 0.2], [0.3, -1.5, 0.0], [2.9, 0.1, -0.4]])
    rad = np.array([1.0, 0.8, 0.6])
    ks = np.array([[0.9, 2.1]])                                   # (1, K)
    ang = np.array([0.0, 0.7, 2.0])
    dirs = np.stack([np.cos(ang), np.sin(ang), 0.3 * np.ones(3)])[:, :, None]      # (d, R, 1)
    uin, ugr = amd.plane_wave(k=_dev(ks), direction=_dev(dirs))
    l, L = lib
    L.check(l.biem_profile_begin())
    calc = amd.biem(c, centers=_dev(cen)[None, None], radii=_dev(rad)[None, None], k=_dev(ks), n_end=7, alpha=1.0, beta=0.4,
                    uin=uin, uin_grad=ugr)
    ms, work, launches = (C.c_double * 9)(), (C.c_double * 9)(), (C.c_longlong * 9)()
    L.check(l.biem_profile_end(ms, work, launches))
    tiles = l.biem_lu_npad(3 * 49) // 64
    per_system = 16.0 * 64 * 64 * tiles * (tiles + 1) / 2             # bytes the symmetric fill writes: lower triangle + diagonal tiles
    assert launches[1] == 1 and work[1] == 2 * per_system             # ONE fill of K = 2 systems, not R*K = 6
    dens = calc.density
    assert tuple(dens.shape) == (3, 2, 3, 49)
    x = np.array([[5.0, 0.5, 0.2], [-3.0, 2.0, 1.0]])
    u = calc.uscat(_dev(x.T)).cpu().numpy()                        # (P, R, K)
    assert u.shape == (2, 3, 2)
    for r in range(3):
        for s in range(2):
            uo, go = O.plane_wave(ks[0, s], dirs[:, r, 0])
            res = O.solve_biem("ba", centers=cen, radii=rad, k=ks[0, s], n_end=7, alpha=1.0, beta=0.4, uin=uo, uin_grad=go)
            ref = O.uscat(res, x)
            assert np.max(np.abs(u[:, r, s] - ref) / np.abs(ref)) < 1e-10, (r, s)
            assert np.max(np.abs(dens[r, s].cpu().numpy() - res.density)) < 1e-9 * np.abs(res.density).max()


# ---------------------------------------------------------------------------- complex wavenumbers
@pytest.mark.parametrize("tree, cen", [("a", [[0.0, 1.6], [0.3, -1.5], [2.9, 0.2]]), ("ba", [[0.0, 1.6, 0.2], [0.3, -1.5, 0.0], [2.9, 0.1, -0.4]]),
                                       ("bba", [[0.0, 1.6, 0.2, 0.1], [0.3, -1.5, 0.0, 0.0]])])
def test_complex_wavenumber_vs_oracle(amd, tree, cen):
    """k with Im k > 0 (absorbing medium; the reference's GUI passes complex k, gui.py:296-301): Robin rows, batch of
    complex and real k in one call, near field, far field and per-ball output against the oracle (SciPy complex Bessel)."""
    c = amd.create_from_branching_types(tree)
    cen = np.array(cen)
    d = cen.shape[1]
    rad = np.array([1.0, 0.8, 0.6])[: len(cen)]
    ks = np.array([1.3 + 0.25j, 0.9 + 0.0j, 2.2 + 0.6j])
    n_end = 9 if d < 4 else 7
    dv = np.zeros(d); dv[0] = 0.6; dv[1] = 0.8
    uin, ugr = amd.plane_wave(k=_dev(ks, torch.complex128), direction=_dev(np.repeat(dv[:, None], 3, 1)))
    calc = amd.biem(c, centers=_dev(cen)[None], radii=_dev(rad)[None], k=_dev(ks, torch.complex128), n_end=n_end, alpha=1.0, beta=0.35,
                    eta=_dev(np.array([1.0, 2.0, 0.5])), uin=uin, uin_grad=ugr)
    assert calc.k.is_complex()
    xs = np.array([[4.0, 0.5] + [0.2] * (d - 2), [-3.0, 2.0] + [0.1] * (d - 2), [0.7, -4.2] + [0.0] * (d - 2)])
    u = calc.uscat(_dev(xs.T)).cpu().numpy()                               # [P, K]
    uf = calc.uscat(_dev(xs.T), far_field=True).cpu().numpy()
    upb = calc.uscat(_dev(xs.T), per_ball=True).cpu().numpy()              # [P, K, B]
    assert np.max(np.abs(upb.sum(-1) - u)) < 1e-12 * np.abs(u).max()
    for i, k in enumerate(ks):
        kk = k if k.imag != 0 else k.real
        uo, go = O.plane_wave(kk, dv)
        res = O.solve_biem(tree, centers=cen, radii=rad, k=kk, n_end=n_end, alpha=1.0, beta=0.35, eta=[1.0, 2.0, 0.5][i], uin=uo, uin_grad=go)
        ref, reff = O.uscat(res, xs), O.uscat(res, xs, far_field=True)
        assert np.max(np.abs(u[:, i] - ref) / np.abs(ref)) < 1e-10, (tree, k)
        assert np.max(np.abs(uf[:, i] - reff) / np.abs(reff)) < 1e-10, (tree, k)
        dn = calc.density[i].cpu().numpy()
        assert np.max(np.abs(dn - res.density)) < 1e-9 * np.abs(res.density).max(), (tree, k)


def test_point_source_complex_wavenumber(amd):
    k = 1.1 + 0.3j
    src = np.array([0.2, -0.1, 3.0])
    u, g = amd.point_source(k=_dev(np.array(k), torch.complex128), source=_dev(src), n=2)
    uo, go = O.point_source(k, src, 2)
    x = np.array([[1.0, 0.5, -0.2], [0.0, 2.0, 1.0], [-1.5, 0.3, 0.4]])
    assert np.max(np.abs(u(_dev(x.T)).cpu().numpy() - uo(x)) / np.abs(uo(x))) < 1e-11
    gg = g(_dev(x.T)).cpu().numpy().T
    assert np.max(np.abs(gg - go(x))) < 1e-11 * np.abs(go(x)).max()


def test_many_small_systems_are_chunked(amd):
    """cfg 1 geometry, 40 000 wavenumbers in one call: more systems than a grid dimension holds, so the solve is split into
    resident chunks; first, middle and last system against the oracle."""
    c = amd.create_from_branching_types("ba")
    cen = np.array([[0.0, 2.0, 0.0], [0.0, -2.0, 0.0]])
    ks = np.linspace(0.5, 3.0, 40000)
    dirs = np.zeros((3, len(ks))); dirs[0] = 1.0
    uin, _ = amd.plane_wave(k=_dev(ks), direction=_dev(dirs))
    calc = amd.biem(c, centers=_dev(cen)[None], radii=_dev(np.ones(2))[None], k=_dev(ks), n_end=6, uin=uin)
    u = calc.uscat(_dev(np.zeros((3, 1)))).cpu().numpy()[0]
    assert u.shape == (40000,) and np.all(np.isfinite(u))
    for i in (0, 20000, 39999):
        uo, _ = O.plane_wave(ks[i], [1.0, 0.0, 0.0])
        res = O.solve_biem("ba", centers=cen, radii=np.ones(2), k=ks[i], n_end=6, uin=uo)
        assert abs(u[i] - O.uscat(res, np.zeros((1, 3)))[0]) < 1e-11 * abs(u[i]), i


@pytest.mark.parametrize("bt,d,n_end,B", [("ba", 3, 6, 2), ("ba", 3, 4, 5), ("a", 2, 9, 5), ("ba", 3, 3, 9), ("ba", 3, 6, 3), ("ba", 3, 5, 5), ("a", 2, 13, 5)])
def test_small_systems_one_launch_path(amd, bt, d, n_end, B, monkeypatch):
    """Systems of at most 128 unknowns (N + nrhs <= 128) are factorised and solved in ONE launch with the whole system in LDS
    (k_small_utu), larger ones by the blocked row form: both on the same inputs (BIEM_NO_SMALL_PATH=1 forces the blocked one) and
    against the oracle.  N = 72 (cfg 1), 80, 85, 81, 108, 125 (2 right-hand sides would no longer fit: 1 here) and 125 in 2-D -
    one to two 64-row panels of the blocked path, every register layout of the small one."""
    c = amd.create_from_branching_types(bt)
    rng = np.random.default_rng(B * 7 + n_end)
    cen = np.zeros((B, d)); cen[:, 0] = 2.6 * np.arange(B); cen[:, 1:] = 0.3 * rng.normal(size=(B, d - 1))
    rad = 0.7 + 0.3 * rng.random(B)
    ks = np.array([0.9, 1.7, 2.4])
    dirs = np.zeros((d, len(ks))); dirs[0] = 1.0
    x = 10.0 + rng.normal(size=(4, d))

    def run():
        uin, ugr = amd.plane_wave(k=_dev(ks), direction=_dev(dirs))
        calc = amd.biem(c, centers=_dev(cen)[None], radii=_dev(rad)[None], k=_dev(ks), n_end=n_end, alpha=1.0, beta=0.5j, uin=uin, uin_grad=ugr)
        return calc.uscat(_dev(x.T)).cpu().numpy()

    u_small = run()
    monkeypatch.setenv("BIEM_NO_SMALL_PATH", "1")
    u_blocked = run()
    assert np.max(np.abs(u_small - u_blocked)) < 1e-12 * np.max(np.abs(u_blocked))
    for i, k in enumerate(ks):
        e0 = np.zeros(d); e0[0] = 1.0
        uo, go = O.plane_wave(k, e0)
        res = O.solve_biem(bt, centers=cen, radii=rad, k=k, n_end=n_end, alpha=1.0, beta=0.5j, uin=uo, uin_grad=go)
        ref = O.uscat(res, x)
        assert np.max(np.abs(u_small[:, i] - ref)) < 1e-10 * np.max(np.abs(ref)), i


@pytest.mark.parametrize("bt,d,n_end,B", [("ba", 3, 7, 3), ("a", 2, 11, 4), ("bpa", 3, 5, 2), ("ba", 3, 20, 2), ("bba", 4, 6, 3), ("bpbpa", 4, 5, 2), ("bba", 4, 10, 2),
                                          ("caa", 4, 6, 3), ("caa", 4, 12, 2), ("caa", 4, 1, 2)])
def test_uscat_point_per_lane_matches_generic(amd, bt, d, n_end, B, monkeypatch):
    """Near-field evaluation for kind = "outer" and the far field on trees a / ba / bpa / bba / bpbpa / caa run one point per lane with recurrences for
    h_n, the polar factors and e^{i m phi} (k_uscat_fast); BIEM_USCAT_GENERIC=1 forces the harmonic-by-harmonic kernel.  Same results (1e-12), incl. per_ball,
    points given per system, complex k, NaN inside the balls and a point count that is not a multiple of the workgroup size."""
    c = amd.create_from_branching_types(bt)
    rng = np.random.default_rng(n_end + B)
    cen = np.zeros((B, d)); cen[:, 0] = 2.7 * np.arange(B); cen[:, 1:] = 0.4 * rng.normal(size=(B, d - 1))
    rad = 0.6 + 0.4 * rng.random(B)
    ks = np.array([0.8, 1.9 + 0.2j, 3.1])
    dirs = np.zeros((d, len(ks))); dirs[1] = 1.0
    uin, _ = amd.plane_wave(k=_dev(ks, torch.complex128), direction=_dev(dirs))
    calc = amd.biem(c, centers=_dev(cen)[None], radii=_dev(rad)[None], k=_dev(ks, torch.complex128), n_end=n_end, uin=uin)
    x = 4.0 * rng.normal(size=(333, d)); x[:, 0] += 2.0
    x[0] = cen[0] + 0.3 * rad[0] * np.eye(d)[0]                       # inside ball 0 -> NaN
    x[1] = cen[B - 1] + 1.0000001 * rad[B - 1] * np.eye(d)[1]         # just outside
    xs = np.repeat(x[:, :, None], len(ks), axis=2) + 0.01 * np.arange(len(ks))   # [P, d, K] per-system points
    xs[0] = x[0][:, None]

    def run():
        return (calc.uscat(_dev(x.T)).cpu().numpy(), calc.uscat(_dev(x.T), per_ball=True).cpu().numpy(),
                calc.uscat(_dev(np.transpose(xs, (1, 0, 2))), expand_x=False).cpu().numpy(),
                calc.uscat(_dev(x.T), far_field=True).cpu().numpy(), calc.uscat(_dev(x.T), far_field=True, per_ball=True).cpu().numpy())

    fast = run()
    monkeypatch.setenv("BIEM_USCAT_GENERIC", "1")
    gen = run()
    for i, (f, g) in enumerate(zip(fast, gen)):
        assert f.shape == g.shape
        assert np.array_equal(np.isnan(f.real), np.isnan(g.real))
        if i < 3:
            assert np.isnan(f[0]).all() and not np.isnan(f[1]).any()
        else:
            assert not np.isnan(f.real).any()                              # the far field is defined everywhere
        ok = ~np.isnan(g.real)
        assert np.max(np.abs(f[ok] - g[ok])) < 1e-12 * np.max(np.abs(g[ok])), i


@pytest.mark.parametrize("bt,d,n_end,kval", [("a", 2, 9, 1.7), ("a", 2, 40, 30.0), ("ba", 3, 7, 2.0 + 0.3j), ("ba", 3, 20, 25.0), ("ba", 3, 48, 11.0),
                                             ("bpa", 3, 6, 3.0), ("bba", 4, 6, 2.2), ("bba", 4, 14, 9.0 - 0.4j), ("caa", 4, 7, 2.0), ("caa", 4, 12, 14.0)])
def test_uscat_inner_point_per_lane_matches_generic(amd, bt, d, n_end, kval, monkeypatch):
    """kind = "inner": the per-lane kernel keeps j_0 .. j_{n_end-1}(k r) of its point in an LDS row (backward recurrence, once per
    ball) and walks the harmonics as for the exterior.  Same results as the harmonic-by-harmonic kernel (1e-12 of the largest
    value), with k r on both sides of n_end, a complex wavenumber, the centre of the ball, points outside (NaN), per_ball and a
    point count that is no multiple of the workgroup."""
    c = amd.create_from_branching_types(bt)
    rng = np.random.default_rng(n_end)
    cen, rad = 0.3 * rng.normal(size=(1, d)), np.array([1.3])
    ks = np.array([kval, 0.6 * kval])
    dirs = np.zeros((d, len(ks))); dirs[0] = 1.0
    uin, ugr = amd.plane_wave(k=_dev(ks, torch.complex128), direction=_dev(dirs))
    calc = amd.biem(c, centers=_dev(cen)[None], radii=_dev(rad)[None], k=_dev(ks, torch.complex128), n_end=n_end, alpha=1.0, beta=0.2,
                    uin=uin, uin_grad=ugr, kind="inner")
    x = rng.normal(size=(150, d))
    x *= (rad[0] * rng.uniform(0.01, 0.999, size=150) / np.linalg.norm(x, axis=1))[:, None]
    x += cen[0]
    x[0] = cen[0]                                                     # the centre
    x[1] = cen[0] + 1.2 * rad[0] * np.eye(d)[0]                       # outside -> NaN
    x[2] = cen[0] + 0.5 * rad[0] * np.eye(d)[d - 1]                   # on an axis (degenerate azimuth)

    def run():
        return calc.uscat(_dev(x.T)).cpu().numpy(), calc.uscat(_dev(x.T), per_ball=True).cpu().numpy()

    fast = run()
    monkeypatch.setenv("BIEM_USCAT_GENERIC", "1")
    gen = run()
    for f, g in zip(fast, gen):
        assert f.shape == g.shape and np.array_equal(np.isnan(f.real), np.isnan(g.real))
        assert np.isnan(f[1]).all() and not np.isnan(f[0]).any() and not np.isnan(f[2:]).any()
        ok = ~np.isnan(g.real)
        assert np.max(np.abs(f[ok] - g[ok])) < 1e-12 * np.max(np.abs(g[ok]))


def test_pair_classes_at_the_top_of_the_2d_range(amd, monkeypatch):
    """Pair classes match displacements within 32 ulp of the largest coordinate; the block of a class is its representative's, so a
    member's table is evaluated at a displacement off by up to that much: a phase error of k x (32 ulp) in H_n(k|t|) e^{i mu phi}.  The
    worst case the build supports is 2-D at k |t| ~ 1e4: a 2 x 3 lattice with a pitch that is no binary fraction (displacements
    equal to rounding only) at k = 1024 (k |t| up to 1.1e4), n_end = 152, with classes (forced for this small batch) and without -
    densities agree to 1e-10 of their largest entry, the tolerance of the whole path (measured: 1e-12)."""
    c = amd.create_from_branching_types("a")
    gx, gy = np.meshgrid(np.arange(2) * 3.7 + 0.1, np.arange(3) * 3.7 - 0.3, indexing="ij")
    cen = np.stack([gx.ravel(), gy.ravel()], -1)
    ks = np.array([1024.0, 724.0773439350247])
    dirs = np.zeros((2, 2)); dirs[0] = 1.0
    dens = []
    for off in (False, True):
        monkeypatch.setenv("BIEM_FILL_DEDUPE_MIN", "1")
        if off:
            monkeypatch.setenv("BIEM_FILL_NO_DEDUPE", "1")
        else:
            monkeypatch.delenv("BIEM_FILL_NO_DEDUPE", raising=False)
        uin, _ = amd.plane_wave(k=_dev(ks), direction=_dev(dirs))
        dens.append(amd.biem(c, centers=_dev(cen)[None], radii=_dev(np.ones(6))[None], k=_dev(ks), n_end=152, uin=uin).density.cpu().numpy())
    err = np.abs(dens[0] - dens[1]).max(axis=(1, 2)) / np.abs(dens[1]).max(axis=(1, 2))
    assert np.all(np.isfinite(dens[0])) and err.max() < 1e-10, err


def test_symmetric_fill_pair_classes(amd, monkeypatch):
    """Ball pairs with the same displacement vector and the same (radius, alpha, beta) on either side share their block of the
    symmetric matrix: it is contracted once and stored to every pair of the class (k_pair_dedupe).  A 3 x 2 lattice where the classes
    are cut three ways - all balls alike; two radii; two Robin coefficients - each against the oracle, and bit for bit against the
    same call with BIEM_FILL_NO_DEDUPE=1 (the copies are copies).  (Classes are used from 8 systems per call on; BIEM_FILL_DEDUPE_MIN=1
    here, and one case with 9 wavenumbers at the default.)"""
    monkeypatch.setenv("BIEM_FILL_DEDUPE_MIN", "1")
    c = amd.create_from_branching_types("ba")
    gx, gy = np.meshgrid(np.arange(3) * 3.0, np.arange(2) * 3.0, indexing="ij")
    cen = np.stack([gx.ravel(), gy.ravel(), np.zeros(6)], -1)
    ks = np.array([0.8, 2.1])
    dirs = np.zeros((3, 2)); dirs[0] = 1.0; dirs[2] = 0.3
    x = np.array([[10.0, 1.0, 2.0], [-4.0, 5.0, -1.0], [3.0, -6.0, 0.5]])
    cases = [
        (np.ones(6), 1.0, 0.4),
        (np.array([1.0, 0.7, 1.0, 0.7, 1.0, 1.0]), 1.0, 0.4),
        (np.ones(6), np.array([[1.0, 1.0, 2.0 + 0.5j, 1.0, 2.0 + 0.5j, 1.0]]), np.array([[0.4, 0.4, 0.4, 0.1, 0.4, 0.4]])),
    ]
    for rad, alpha, beta in cases:
        out = []
        for off in (False, True):
            if off:
                monkeypatch.setenv("BIEM_FILL_NO_DEDUPE", "1")
            else:
                monkeypatch.delenv("BIEM_FILL_NO_DEDUPE", raising=False)
            uin, ugr = amd.plane_wave(k=_dev(ks), direction=_dev(dirs))
            a = alpha if np.isscalar(alpha) else _dev(alpha, torch.complex128)
            b = beta if np.isscalar(beta) else _dev(beta, torch.complex128)
            calc = amd.biem(c, centers=_dev(cen)[None], radii=_dev(rad)[None], k=_dev(ks), n_end=7, alpha=a, beta=b, uin=uin, uin_grad=ugr)
            out.append((calc.density.cpu().numpy(), calc.uscat(_dev(x.T)).cpu().numpy()))
        assert np.array_equal(out[0][0], out[1][0])
        if np.isscalar(alpha) and rad[1] != 1.0:
            # the same lattice with a pitch that is no binary fraction: displacements agree to rounding only (classes match them
            # within 32 ulp), so the shared block is that of the representative pair - equal to rounding, not bit for bit
            cen7 = cen * 0.7
            dens = []
            for off in (False, True):
                if off:
                    monkeypatch.setenv("BIEM_FILL_NO_DEDUPE", "1")
                else:
                    monkeypatch.delenv("BIEM_FILL_NO_DEDUPE", raising=False)
                uin, ugr = amd.plane_wave(k=_dev(ks), direction=_dev(dirs))
                dens.append(amd.biem(c, centers=_dev(cen7)[None], radii=_dev(0.8 * rad)[None], k=_dev(ks), n_end=7, alpha=alpha, beta=beta, uin=uin, uin_grad=ugr).density.cpu().numpy())
            assert np.max(np.abs(dens[0] - dens[1])) < 1e-12 * np.max(np.abs(dens[1]))
            monkeypatch.delenv("BIEM_FILL_NO_DEDUPE", raising=False)
        if np.isscalar(alpha) and rad[1] == 1.0:                         # the default threshold: 9 systems per call
            monkeypatch.delenv("BIEM_FILL_DEDUPE_MIN")
            k9 = np.linspace(0.8, 2.1, 9); d9 = np.repeat(dirs[:, :1], 9, axis=1)
            uin9, ugr9 = amd.plane_wave(k=_dev(k9), direction=_dev(d9))
            dens9 = amd.biem(c, centers=_dev(cen)[None], radii=_dev(rad)[None], k=_dev(k9), n_end=7, alpha=alpha, beta=beta, uin=uin9, uin_grad=ugr9).density.cpu().numpy()
            assert np.array_equal(dens9[0], out[0][0][0]) or np.max(np.abs(dens9[0] - out[0][0][0])) < 1e-12 * np.max(np.abs(dens9[0]))
            monkeypatch.setenv("BIEM_FILL_DEDUPE_MIN", "1")
        for i, k in enumerate(ks):
            uo, go = O.plane_wave(k, dirs[:, i])
            res = O.solve_biem("ba", centers=cen, radii=rad, k=k, n_end=7, alpha=np.ravel(alpha) if not np.isscalar(alpha) else alpha,
                               beta=np.ravel(beta) if not np.isscalar(beta) else beta, uin=uo, uin_grad=go)
            ref = O.uscat(res, x)
            assert np.max(np.abs(out[0][1][:, i] - ref)) < 1e-10 * np.max(np.abs(ref))


@pytest.mark.parametrize("force_lu_fallback", [False, True])
def test_batched_geometry_and_points_per_system(amd, force_lu_fallback, monkeypatch):
    """(force_lu_fallback: every system is rejected by the symmetric path and re-solved by the pivoted LU from gathered
    per-system geometry.)  Geometry that differs between systems (centers [K, B, d], radii [K, B]) and evaluation points given per system
    (expand_x=False: x of shape (d, P, K)); every system against its own oracle solve."""
    c = amd.create_from_branching_types("ba")
    rng = np.random.default_rng(5)
    K = 3
    cen = np.array([[[0.0, 1.7, 0.1], [0.2, -1.6, 0.0]], [[0.3, 2.0, -0.2], [0.0, -1.8, 0.4]], [[-0.4, 1.5, 0.0], [0.5, -2.2, 0.1]]])
    rad = np.array([[1.0, 0.7], [0.9, 1.1], [0.6, 0.8]])
    ks = np.array([1.1, 1.9, 2.6])
    dirs = np.zeros((3, K)); dirs[1] = 1.0
    uin, ugr = amd.plane_wave(k=_dev(ks), direction=_dev(dirs))
    if force_lu_fallback:
        monkeypatch.setenv("BIEM_LDLT_PIVOT_REL", "1e30")
    calc = amd.biem(c, centers=_dev(cen), radii=_dev(rad), k=_dev(ks), n_end=8, alpha=0.0, beta=1.0, uin=uin, uin_grad=ugr)
    from biem_helmholtz_sphere_amd import _biem as impl
    assert {q: impl._last_solve_stats[q] for q in ("ldlt_systems", "lu_systems")} == {"ldlt_systems": K, "lu_systems": K if force_lu_fallback else 0}
    xs = 6.0 + rng.normal(size=(K, 4, 3))                               # [K, P, d]
    u = calc.uscat(_dev(np.transpose(xs, (2, 1, 0))), expand_x=False).cpu().numpy()      # (d, P, K) -> [P, K]
    assert u.shape == (4, K)
    for i in range(K):
        uo, go = O.plane_wave(ks[i], [0.0, 1.0, 0.0])
        res = O.solve_biem("ba", centers=cen[i], radii=rad[i], k=ks[i], n_end=8, alpha=0.0, beta=1.0, uin=uo, uin_grad=go)
        ref = O.uscat(res, xs[i])
        assert np.max(np.abs(u[:, i] - ref) / np.abs(ref)) < 1e-10, i


def test_empty_batch(amd):
    """A batch axis of extent 0 yields empty results of the right shape (no kernel is launched on an empty grid)."""
    c = amd.create_from_branching_types("a")
    ks = torch.zeros((0,), dtype=torch.float64, device="cuda")
    uin, _ = amd.plane_wave(k=ks, direction=torch.zeros((2, 0), dtype=torch.float64, device="cuda"))
    calc = amd.biem(c, centers=_dev([[0.0, 2.0], [0.0, -2.0]])[None], radii=_dev([1.0, 1.0])[None], k=ks, n_end=5, uin=uin)
    assert tuple(calc.density.shape) == (0, 2, 9)
    assert tuple(calc.uscat(_dev(np.zeros((2, 3)))).shape) == (3, 0)


@pytest.mark.parametrize("N,nb,nrhs", [(440, 2, 2), (1000, 1, 1)])
def test_lu_solve_without_stored_factors(lib, N, nb, nrhs, monkeypatch):
    """The fused path (biem_solve) keeps L21 only in the panel workspace; the same mode of the stand-alone LU on Gaussian
    matrices (heavy pivoting across the panels of a four-panel group): the solution must not depend on it."""
    monkeypatch.setenv("BIEM_LU_DISCARD_FACTORS", "1")
    test_lu_factor_solve_vs_numpy(lib, N, nb, nrhs)


# ---------------------------------------------------------------------------- complex-symmetric L D L^T path
@pytest.mark.gpu
@pytest.mark.parametrize("N,nb,nrhs", [(64, 2, 1), (17, 3, 2), (40, 2, 8), (64, 300, 8), (150, 2, 2), (256, 1, 1), (300, 3, 1), (576, 2, 3), (1000, 2, 1), (1345, 1, 2), (700, 2, 12)])
@pytest.mark.parametrize("discard", [False, True])
def test_ldlt_factor_solve_vs_numpy(lib, N, nb, nrhs, discard, monkeypatch):
    """biem_ldlt_factor_solve on complex-symmetric (not Hermitian) matrices I + E; only the lower triangle may be read: the
    strict upper triangle outside the diagonal 64 x 64 blocks is filled with garbage."""
    if discard:
        monkeypatch.setenv("BIEM_LU_DISCARD_FACTORS", "1")
    l, L = lib
    rng = np.random.default_rng(N + 7)
    npad = l.biem_lu_npad(N)
    lda = npad + ((nrhs + 7) // 8) * 8
    E = (rng.normal(size=(nb, N, N)) + 1j * rng.normal(size=(nb, N, N))) * (0.12 / np.sqrt(N))
    As = np.eye(N)[None] * (1.0 + 0.2j) + E + np.swapaxes(E, 1, 2)
    Fs = rng.normal(size=(nb, N, nrhs)) + 1j * rng.normal(size=(nb, N, nrhs))
    A = np.zeros((nb, npad, lda), dtype=np.complex128)
    A[:, :N, :N] = As
    for i in range(N, npad):
        A[:, i, i] = 1.0
    blk = np.arange(npad) // 64
    upper_off = (blk[:, None] < blk[None, :])
    A[:, :npad, :npad][:, upper_off] = 1e30                    # must never be read
    A[:, :N, npad:npad + nrhs] = Fs
    dA = _dev(A, torch.complex128)
    ipiv = torch.zeros((nb, npad), dtype=torch.int32, device="cuda")
    info = torch.ones(nb, dtype=torch.int32, device="cuda")
    wb = l.biem_lu_workspace_bytes(nb, npad, nrhs)
    work = torch.empty(wb, dtype=torch.uint8, device="cuda")
    L.check(l.biem_ldlt_factor_solve(nb, npad, nrhs, dA.data_ptr(), lda, npad * lda, ipiv.data_ptr(), info.data_ptr(), work.data_ptr(), wb, None))
    torch.cuda.synchronize()
    assert (info.cpu().numpy() == 0).all(), info.cpu().numpy()
    X = dA.cpu().numpy()[:, :N, npad:npad + nrhs]
    for s in range(nb):
        Xo = np.linalg.solve(As[s], Fs[s])
        assert np.abs(X[s] - Xo).max() / np.abs(Xo).max() < 1e-12, (N, s)


@pytest.mark.gpu
def test_ldlt_rejected_pivot_is_reported(lib):
    l, L = lib
    N = 128
    A = np.zeros((2, N, N + 8), dtype=np.complex128)
    A[:, :, :N] = np.eye(N)
    A[1, 70, 70] = 0.001
    A[1, 90, 70] = A[1, 70, 90] = 1.0                           # the diagonal is 0.1 % of its column's maximum (limit: 1 %)
    dA = _dev(A, torch.complex128)
    ipiv = torch.zeros((2, N), dtype=torch.int32, device="cuda")
    info = torch.zeros(2, dtype=torch.int32, device="cuda")
    wb = l.biem_lu_workspace_bytes(2, N, 1)
    work = torch.empty(wb, dtype=torch.uint8, device="cuda")
    L.check(l.biem_ldlt_factor_solve(2, N, 1, dA.data_ptr(), N + 8, N * (N + 8), ipiv.data_ptr(), info.data_ptr(), work.data_ptr(), wb, None))
    torch.cuda.synchronize()
    assert info.cpu().tolist() == [0, -65]          # -(first row of the 64-column panel holding the rejected pivot + 1)


@pytest.mark.gpu
@pytest.mark.parametrize("tree,B,n_end", [("a", 4, 9), ("ba", 3, 7), ("bba", 2, 4), ("caa", 2, 4), ("bpa", 2, 5)])
def test_ldlt_and_lu_paths_agree(amd, tree, B, n_end, monkeypatch):
    """The symmetric path (default) and the pivoted LU (BIEM_SOLVER=lu, the reference's algorithm) give the same density."""
    from biem_helmholtz_sphere_amd import _biem as impl

    rng = np.random.default_rng(5)
    c = amd.create_from_branching_types(tree)
    d = c.c_ndim
    cen, rad = _rand_geometry(rng, B, d)
    ks = _dev(np.array([0.7, 2.9, 1.1 + 0.3j]), torch.complex128)
    direction = rng.normal(size=d)
    dirs = _dev(np.repeat(direction[:, None], 3, axis=1))
    uin, ugr = amd.plane_wave(k=ks, direction=dirs)
    out = {}
    for solver in ("ldlt", "lu"):
        monkeypatch.setenv("BIEM_SOLVER", solver)
        calc = amd.biem(c, centers=_dev(cen)[None], radii=_dev(rad)[None], k=ks, eta=_dev(np.ones(3)), n_end=n_end, alpha=1.0 + 0.2j,
                        beta=0.4, uin=uin, uin_grad=ugr)
        out[solver] = calc.density.cpu().numpy()
        st = dict(impl._last_solve_stats)
        assert st == ({"ldlt_systems": 3, "lu_systems": 0} if solver == "ldlt" else {"ldlt_systems": 0, "lu_systems": 3}), st
    scale = np.abs(out["lu"]) + 1e-12 * np.abs(out["lu"]).max()
    assert np.max(np.abs(out["ldlt"] - out["lu"]) / scale) < 1e-9


@pytest.mark.gpu
def test_ldlt_near_a_resonance_and_forced_fallback(amd, monkeypatch):
    """k rho = pi: j_0(k rho) = 0 to rounding and the symmetric scaling 1/sqrt(gj gh) is ~1e8 for degree 0 of that ball - the
    symmetric matrix stays benign (that row of M is a unit row) and the result matches the oracle.  Then the fallback: with
    an absurd acceptance threshold (every multiplier must be below 1e-30) every system is rejected and all systems are re-solved with the pivoted LU."""
    from biem_helmholtz_sphere_amd import _biem as impl

    cen = np.array([[0.0, 1.7, 0.1], [0.2, -1.6, 0.0]])
    rad = np.array([1.0, 0.7])
    ks = np.array([1.3, np.pi])
    x = np.array([[4.0, 0.5, 0.2], [-3.0, 2.0, 1.0]])
    c = amd.create_from_branching_types("ba")
    dirs = np.zeros((3, 2)); dirs[0] = 1.0
    uin, _ = amd.plane_wave(k=_dev(ks), direction=_dev(dirs))
    uo = [O.uscat(O.solve_biem("ba", centers=cen, radii=rad, k=float(k), n_end=7, uin=O.plane_wave(float(k), [1.0, 0, 0])[0]), x) for k in ks]
    for rel, stats in ((None, {"ldlt_systems": 2, "lu_systems": 0}), ("1e30", {"ldlt_systems": 2, "lu_systems": 2})):
        if rel is not None:
            monkeypatch.setenv("BIEM_LDLT_PIVOT_REL", rel)
        calc = amd.biem(c, centers=_dev(cen)[None], radii=_dev(rad)[None], k=_dev(ks), eta=_dev(np.ones(2)), n_end=7, uin=uin)
        assert {q: impl._last_solve_stats[q] for q in ("ldlt_systems", "lu_systems")} == stats
        u = calc.uscat(_dev(x.T.copy())).cpu().numpy()
        for i in range(2):
            assert np.abs(u[:, i] - uo[i]).max() / np.abs(uo[i]).max() < 1e-10


@pytest.mark.gpu
def test_close_spheres_stay_on_the_symmetric_path(amd, monkeypatch):
    """Two unit spheres 0.04 apart, Robin rows: at low wavenumbers the first pivot of the second sphere (its monopole after the first
    sphere's elimination) is small - multipliers of 11 (k = 0.5) to 76 (k -> 0), growth 8 .. 45.  The factorisation without interchanges
    solves these to rounding (accepted since round 3: multipliers <= 100, growth <= 200; rejected and sent to the pivoted LU with the
    round-2 limit of 10), so no system leaves the symmetric path, and both paths agree (tools/ldlt_stress.py surveys more)."""
    from biem_helmholtz_sphere_amd import _biem as impl

    c = amd.create_from_branching_types("ba")
    cen, rad = np.array([[0.0, 1.02, 0.0], [0.0, -1.02, 0.0]]), np.array([1.0, 1.0])
    ks = np.array([0.01, 0.5, 10.0])
    dirs = np.zeros((3, 3)); dirs[0] = 1.0
    x = np.array([[6.0, 3.0, 0.1], [-5.0, 2.0, 1.0], [0.2, 7.0, -1.0]]).T
    out = {}
    for solver in ("ldlt", "lu"):
        monkeypatch.setenv("BIEM_SOLVER", solver)
        uin, ugr = amd.plane_wave(k=_dev(ks), direction=_dev(dirs))
        calc = amd.biem(c, centers=_dev(cen)[None], radii=_dev(rad)[None], k=_dev(ks), eta=_dev(np.ones(3)), n_end=14, alpha=1.0, beta=0.3,
                        uin=uin, uin_grad=ugr)
        out[solver] = calc.uscat(_dev(x.copy())).cpu().numpy()
        if solver == "ldlt":
            st = dict(impl._last_solve_stats)
            assert (st["ldlt_systems"], st["lu_systems"]) == (3, 0), st
    assert np.max(np.abs(out["ldlt"] - out["lu"]) / np.abs(out["lu"])) < 1e-12
    # the round-2 limit (multipliers <= 10) hands the k = 0.01 and k = 0.5 systems to the pivoted LU: the fallback route still works
    monkeypatch.setenv("BIEM_SOLVER", "ldlt")
    monkeypatch.setenv("BIEM_LDLT_PIVOT_REL", "0.1")
    uin, ugr = amd.plane_wave(k=_dev(ks), direction=_dev(dirs))
    calc = amd.biem(c, centers=_dev(cen)[None], radii=_dev(rad)[None], k=_dev(ks), eta=_dev(np.ones(3)), n_end=14, alpha=1.0, beta=0.3, uin=uin, uin_grad=ugr)
    st = dict(impl._last_solve_stats)
    assert st["lu_systems"] == 2 and st["rejected_info"] == [-193, -193], st      # the panel of the second sphere's first unknowns
    assert np.max(np.abs(calc.uscat(_dev(x.copy())).cpu().numpy() - out["lu"]) / np.abs(out["lu"])) < 1e-12


# ---------------------------------------------------------------------------- round 2: growth check, factor / solve split, inner kind
@pytest.mark.gpu
def test_ldlt_growth_check_measures_max_u_over_max_a(lib, monkeypatch):
    """The a-posteriori growth check of the symmetric path: max |U| / max |A| (|.| = |re| + |im|, A = the part read) computed on
    the device equals the NumPy L D L^T without interchanges - a limit 1 % below it marks the system (info = -(Npad + 1)), 1 %
    above does not; a NaN in the matrix marks the system at the default limit."""
    l, L = lib
    N, npad = 200, 256
    rng = np.random.default_rng(5)
    E = (rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))) * (0.12 / np.sqrt(N))
    As = np.eye(N) * (1.0 + 0.2j) + E + E.T
    M = As.copy()                                       # reference factorisation: U ends up in the upper triangle
    for c in range(N):
        M[c + 1:, c] /= M[c, c]
        M[c + 1:, c + 1:] -= np.outer(M[c + 1:, c], M[c, c + 1:])
    cabs1 = lambda z: np.abs(z.real) + np.abs(z.imag)
    ratio = cabs1(np.triu(M)).max() / cabs1(As).max()

    def run(limit, poison=False):
        if limit is None:
            monkeypatch.delenv("BIEM_LDLT_GROWTH_MAX", raising=False)
        else:
            monkeypatch.setenv("BIEM_LDLT_GROWTH_MAX", repr(float(limit)))
        A = np.zeros((2, npad, npad + 8), dtype=np.complex128)
        A[:, :N, :N] = As
        A[:, np.arange(N, npad), np.arange(N, npad)] = 1.0
        A[:, :N, npad] = 1.0
        if poison:
            A[1, 150, 20] = np.nan
        dA = _dev(A, torch.complex128)
        ipiv = torch.zeros((2, npad), dtype=torch.int32, device="cuda")
        info = torch.zeros(2, dtype=torch.int32, device="cuda")
        wb = l.biem_lu_workspace_bytes(2, npad, 1)
        work = torch.empty(wb, dtype=torch.uint8, device="cuda")
        L.check(l.biem_ldlt_factor_solve(2, npad, 1, dA.data_ptr(), npad + 8, npad * (npad + 8), ipiv.data_ptr(), info.data_ptr(), work.data_ptr(), wb, None))
        torch.cuda.synchronize()
        return info.cpu().tolist()

    assert run(0.99 * ratio) == [-(npad + 1)] * 2
    assert run(1.01 * ratio) == [0, 0]
    bad = run(None, poison=True)
    assert bad[0] == 0 and bad[1] < 0


@pytest.mark.gpu
@pytest.mark.parametrize("N,nb,nrhs,sym", [(64, 2, 1, False), (300, 2, 3, False), (576, 1, 2, False), (1000, 2, 1, False),
                                           (150, 2, 2, True), (700, 1, 5, True)])
def test_factor_now_solve_later(lib, N, nb, nrhs, sym):
    """biem_lu_factor / biem_ldlt_factor keep the factors (and interchanges); biem_lu_solve serves right-hand sides that arrive
    later, any number of times, without a workspace - against numpy.linalg.solve on Gaussian (heavily pivoting) and
    complex-symmetric matrices."""
    l, L = lib
    rng = np.random.default_rng(N + 31)
    npad = l.biem_lu_npad(N)
    if sym:
        E = (rng.normal(size=(nb, N, N)) + 1j * rng.normal(size=(nb, N, N))) * (0.12 / np.sqrt(N))
        As = np.eye(N)[None] * (1.0 + 0.2j) + E + np.swapaxes(E, 1, 2)
    else:
        As = rng.normal(size=(nb, N, N)) + 1j * rng.normal(size=(nb, N, N))
    A = np.zeros((nb, npad, npad), dtype=np.complex128)
    A[:, :N, :N] = As
    A[:, np.arange(N, npad), np.arange(N, npad)] = 1.0
    dA = _dev(A, torch.complex128)
    ipiv = torch.zeros((nb, npad), dtype=torch.int32, device="cuda")
    info = torch.ones(nb, dtype=torch.int32, device="cuda")
    wb = l.biem_lu_workspace_bytes(nb, npad, 0)
    work = torch.empty(wb, dtype=torch.uint8, device="cuda")
    factor = l.biem_ldlt_factor if sym else l.biem_lu_factor
    L.check(factor(nb, npad, dA.data_ptr(), npad, npad * npad, ipiv.data_ptr(), info.data_ptr(), work.data_ptr(), wb, None))
    torch.cuda.synchronize()
    assert info.cpu().tolist() == [0] * nb
    del work                                          # the solve needs none
    if sym:
        assert (ipiv.cpu().numpy() == np.arange(npad)[None]).all()
    for trial in range(2):
        ldb = nrhs + trial                            # a leading dimension larger than nrhs is allowed
        Fs = rng.normal(size=(nb, N, nrhs)) + 1j * rng.normal(size=(nb, N, nrhs))
        Bm = np.zeros((nb, npad, ldb), dtype=np.complex128)
        Bm[:, :N, :nrhs] = Fs
        dB = _dev(Bm, torch.complex128)
        L.check(l.biem_lu_solve(nb, npad, nrhs, dA.data_ptr(), npad, npad * npad, ipiv.data_ptr(), dB.data_ptr(), ldb, npad * ldb, None))
        torch.cuda.synchronize()
        X = dB.cpu().numpy()
        assert np.abs(X[:, N:, :nrhs]).max(initial=0.0) == 0
        for s in range(nb):
            Xo = np.linalg.solve(As[s], Fs[s])
            assert np.abs(X[s, :N, :nrhs] - Xo).max() / np.abs(Xo).max() < (1e-12 if sym else 1e-9), (N, s, trial)


@pytest.mark.gpu
@pytest.mark.parametrize("tree", ["a", "ba", "bba", "caa"])
def test_kind_inner_interior_expansion(amd, tree):
    """kind="inner" (reference :971-976 keeps the points with r <= rho): there the layer potentials expand in the REGULAR
    functions.  No reference fixture covers it (parity unpinned), so besides the oracle the test checks what defines the
    interior potential: the jump relation across the sphere, u(rho+) - u(rho-) = sum_h density_h Y_h (double layer jumps by the
    density, single layer is continuous), a finite value at the centre, the Helmholtz equation inside, and the NaN mask."""
    tr = O.tree(tree)
    d = tr.d
    k, n_end, eta = 2.0, 7, 1.3
    cen, rad = np.zeros((1, d)), np.array([1.0])
    e = np.zeros(d); e[0] = 1.0; e[1] = 0.3
    c = amd.create_from_branching_types(tree)
    uin, ugr = amd.plane_wave(k=_dev(k), direction=_dev(e))
    kw = dict(centers=_dev(cen), radii=_dev(rad), k=_dev(k), eta=_dev(eta), n_end=n_end, alpha=1.0, beta=0.4, uin=uin, uin_grad=ugr)
    cin, cout = amd.biem(c, kind="inner", **kw), amd.biem(c, kind="outer", **kw)
    assert torch.equal(cin.density, cout.density)           # the reference's solve does not depend on `kind`
    uo_in, ug_in = O.plane_wave(k, e)
    res = O.solve_biem(tree, centers=cen, radii=rad, k=k, n_end=n_end, eta=eta, alpha=1.0, beta=0.4, uin=uo_in, uin_grad=ug_in, kind="inner")
    rng = np.random.default_rng(3)
    pts = rng.normal(size=(12, d))
    pts *= (rng.uniform(0.05, 0.95, size=12) / np.linalg.norm(pts, axis=1))[:, None]
    pts[0] = 0.0                                            # the centre of the ball
    u = cin.uscat(_dev(pts.T)).cpu().numpy()
    ref = O.uscat(res, pts)
    assert np.isfinite(u).all() and np.max(np.abs(u - ref) / np.abs(ref)) < 1e-10
    out_pts = pts[1:4] / np.linalg.norm(pts[1:4], axis=1)[:, None] * 1.7
    assert np.isnan(cin.uscat(_dev(out_pts.T)).cpu().numpy()).all() and np.isnan(cout.uscat(_dev(pts[1:4].T)).cpu().numpy()).all()
    # jump relation on the sphere (r == rho exactly: valid for both kinds, masks are strict)
    xs = np.zeros((3, d)); xs[0, 0] = 1.0; xs[1, 1] = -1.0; xs[2, d - 1] = 1.0
    jump = cout.uscat(_dev(xs.T)).cpu().numpy() - cin.uscat(_dev(xs.T)).cpu().numpy()
    Y = tr.harmonics(xs, n_end)                             # [H, P]
    want = cin.density.cpu().numpy().reshape(-1) @ Y
    assert np.max(np.abs(jump - want)) < 1e-11 * np.max(np.abs(want))
    # Helmholtz equation inside: 2d-point stencil, h = 5e-3 (truncation ~ h^2 k^4 / 12)
    h, x0 = 5e-3, pts[5]
    st = np.concatenate([[x0]] + [[x0 + h * np.eye(d)[i], x0 - h * np.eye(d)[i]] for i in range(d)])
    us = cin.uscat(_dev(st.T)).cpu().numpy()
    lap = (us[1:].sum() - 2 * d * us[0]) / h**2
    assert abs(lap + k * k * us[0]) < 2e-4 * k * k * abs(us[0])


@pytest.mark.gpu
def test_accuracy_sweep_driver_rows_vs_reference_csv(amd, golden_dir, tmp_path):
    """The accuracy-style sweep (reference cli.py:188-271: grids of balls, operator k swept with the incident wave kept at
    k = 1) writes the reference's schema; its rows equal the committed accuracy_n_balls_a.csv / accuracy_k_ba.csv rows."""
    from biem_helmholtz_sphere_amd import sweep

    out = tmp_path / "acc_a.csv"
    sweep.main(["accuracy", "--out", str(out), "--types", "a", "--n-balls", "4,16", "--k", "1", "--n-end", "1,3,6,13,32"])
    with open(out) as f:
        assert f.readline() == sweep.ACCURACY_HEADER
    with open(out) as f:
        mine = {(r["branching_types"], int(r["n_end"]), int(r["n_balls"])): complex(r["uscat"]) for r in csv.DictReader(f)}
    assert len(mine) == 10
    n = 0
    with open(os.path.join(golden_dir, "accuracy_n_balls_a.csv")) as f:
        hdr = f.readline()
        assert hdr == sweep.ACCURACY_HEADER
        f.seek(0)
        for r in csv.DictReader(f):
            key = (r["branching_types"], int(r["n_end"]), int(r["n_balls"]))
            if key in mine:
                assert abs(mine[key] - complex(r["uscat"])) < 1e-11, key
                n += 1
    assert n == 10
    out = tmp_path / "acc_ba.csv"
    kvals = [1.0, 2.0 ** 0.5, 4.0, 2.0 ** 3.5]
    sweep.main(["accuracy", "--out", str(out), "--types", "ba", "--n-balls", "2", "--k", ",".join(repr(v) for v in kvals), "--n-end", "2,6,12,20"])
    with open(out) as f:
        mine = {(r["branching_types"], int(r["n_end"]), round(float(r["k"]), 9)): complex(r["uscat"]) for r in csv.DictReader(f)}
    n = 0
    with open(os.path.join(golden_dir, "accuracy_k_ba.csv")) as f:
        for r in csv.DictReader(f):
            key = (r["branching_types"], int(r["n_end"]), round(float(r["k"]), 9))
            if key in mine:
                assert abs(mine[key] - complex(r["uscat"])) < 1e-11 * max(1.0, abs(complex(r["uscat"]))), key
                n += 1
    assert n >= 12


@pytest.mark.gpu
def test_sharded_solve_on_rccl_world_size_one(amd):
    """_dist.biem_sharded with the real biem() on the `nccl` (= RCCL) backend at world size 1: geometry broadcast, shard, solve,
    all-gather of the densities - equal to a plain biem() call."""
    import socket

    import torch.distributed as dist

    from biem_helmholtz_sphere_amd import _dist

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        c = amd.create_from_branching_types("ba")
        cen = _dev(O.grid_centers(1, 3))
        rad = _dev(np.ones(4))
        ks = _dev(np.linspace(0.7, 2.0, 5))
        dirs = np.zeros((3, 5)); dirs[0] = 1.0

        def incident(k_loc, sl):
            return amd.plane_wave(k=k_loc, direction=_dev(dirs[:, sl]))

        res, full = _dist.biem_sharded(c, centers=cen, radii=rad, k=ks, n_end=5, incident=incident, device=torch.device("cuda", 0))
        uin, _ = amd.plane_wave(k=ks, direction=_dev(dirs))
        plain = amd.biem(c, centers=cen[None], radii=rad[None], k=ks, n_end=5, uin=uin)
        assert full.shape == plain.density.shape and torch.equal(full, plain.density)
        assert torch.equal(res.density, plain.density)
        # BASELINE config 5's shape in small: a (k, eta) batch of 4 x 3 systems, d = 4 'bba', 8 balls on the 2 x 4 grid, per-ball Robin
        # coefficients shared by the batch - the flattened batch is what the rank solves, the gathered density has the 2-D batch shape
        c4 = amd.create_from_branching_types("bba")
        x0, x1 = np.meshgrid(np.arange(2) * 4.0 - 2.0, np.arange(4) * 4.0 - 6.0, indexing="ij")
        cen4 = np.zeros((8, 4)); cen4[:, 0], cen4[:, 1] = x0.ravel(), x1.ravel()
        k2, e2 = _dev(np.linspace(0.5, 4.0, 4))[:, None], _dev(np.linspace(0.25, 4.0, 3))[None, :]
        d4 = np.zeros((4, 12)); d4[0] = 1.0
        al = _dev(np.linspace(1.0, 1.7, 8), torch.complex128)[None, None, :]

        def incident4(k_loc, sl):
            return amd.plane_wave(k=k_loc, direction=_dev(d4[:, sl]))

        res, full = _dist.biem_sharded(c4, centers=_dev(cen4), radii=_dev(np.ones(8)), k=k2, eta=e2, alpha=al, beta=0.25, n_end=4, incident=incident4,
                                       device=torch.device("cuda", 0))
        kf, ef = k2.expand(4, 3).reshape(12), e2.expand(4, 3).reshape(12)
        uin, ugr = amd.plane_wave(k=kf, direction=_dev(d4))
        plain = amd.biem(c4, centers=_dev(cen4)[None], radii=_dev(np.ones(8))[None], k=kf, eta=ef, alpha=al[0], beta=0.25, n_end=4, uin=uin, uin_grad=ugr)
        assert full.shape == (4, 3, 8, 30) and torch.equal(full.reshape(12, 8, 30), plain.density)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["default", "red", "entry", "sys"])
@pytest.mark.parametrize("tree,n_end,B,robin", [("a", 9, 3, False), ("a", 40, 4, True), ("ba", 6, 3, True), ("ba", 12, 2, False), ("bba", 4, 3, True), ("caa", 4, 2, False)])
def test_symmetric_fill_vs_transformed_general_fill(amd, lib, tree, n_end, B, robin, form, monkeypatch):
    """BIEM_FILL_SYMMETRIC (what the L D L^T path factors, written once by the fused kernel) against R W^H M W R^-1 formed in
    NumPy from the general BIEM_FILL_EQUILIBRATED matrix, on everything the factorisation reads (upper triangle + diagonal
    64 x 64 tiles); the result is complex symmetric and has a unit diagonal.  Both forms of the kernel: one unit pair per lane
    (small batches) and one system per lane (batches of >= 32 systems; forced here on a batch of 2)."""
    from biem_helmholtz_sphere_amd import _biem as impl

    # "default": what the product runs (reduced-table kernel; 2-D: the list-free Toeplitz kernels, for the general fill M as well);
    # the named forms force the list kernels (2-D included), so for tree a the two families check each other
    if form != "default":
        monkeypatch.setenv("BIEM_FILL_FORM", form)
    else:
        monkeypatch.delenv("BIEM_FILL_FORM", raising=False)
    l, L = lib
    d = O.tree(tree).d
    rng = np.random.default_rng(n_end + B)
    cen = rng.normal(size=(B, d)) * 0.3 + np.arange(B)[:, None] * np.eye(d)[0] * 2.6
    rad = rng.uniform(0.6, 1.0, size=B)
    ks = np.array([0.9, 2.3 + (0.2j if robin else 0.0)])
    nb = len(ks)
    plan = impl._plan(tree, n_end, torch.device("cuda", 0))
    H = plan.H
    N = B * H
    npad = l.biem_lu_npad(N)
    al = _dev(np.full((1, B), 1.0 + 0.0j), torch.complex128)
    be = _dev(np.full((1, B), (0.4 - 0.1j) if robin else 0.0), torch.complex128)
    k_t, eta_t = _dev(ks, torch.complex128), _dev(np.ones(nb))
    cen_t, rad_t = _dev(cen[None]), _dev(rad[None])
    tab = torch.empty((nb, B, 3, n_end), dtype=torch.complex128, device="cuda")
    L.check(l.biem_ball_tables(plan.handle, nb, B, k_t.data_ptr(), eta_t.data_ptr(), rad_t.data_ptr(), 0, al.data_ptr(), be.data_ptr(), 0, tab.data_ptr(), None))
    wb = l.biem_fill_workspace_bytes(plan.handle, nb, B)        # (with BIEM_FILL_FORM=sys it covers the batch padded to 64 systems)
    work = torch.empty(max(wb, 16), dtype=torch.uint8, device="cuda")
    M = torch.zeros((nb, npad, npad), dtype=torch.complex128, device="cuda")
    L.check(l.biem_fill(plan.handle, nb, B, k_t.data_ptr(), cen_t.data_ptr(), 0, tab.data_ptr(), L.FILL_EQUILIBRATED, M.data_ptr(), npad, npad * npad, npad, work.data_ptr(), wb, None))
    S = torch.full((nb, npad, npad), float("nan"), dtype=torch.complex128, device="cuda")
    L.check(l.biem_fill(plan.handle, nb, B, k_t.data_ptr(), cen_t.data_ptr(), 0, tab.data_ptr(), L.FILL_SYMMETRIC, S.data_ptr(), npad, npad * npad, npad, work.data_ptr(), wb, None))
    torch.cuda.synchronize()
    partner = np.zeros(H, dtype=np.int32)
    slot = np.zeros(H, dtype=np.int32)
    L.check(l.biem_plan_symmetric_order(plan.handle, partner.ctypes.data, slot.ctypes.data))
    assert sorted(slot) == list(range(H)) and (partner[partner] == np.arange(H)).all()
    V = np.zeros((H, H), dtype=np.complex128)
    sdeg = np.zeros(H, dtype=int)
    for h in range(H):
        p = partner[h]
        if p == h:
            V[h, slot[h]] = 1.0
        elif h < p:
            V[h, slot[h]] = V[p, slot[h]] = 1 / np.sqrt(2)
            V[p, slot[p]], V[h, slot[p]] = 1j / np.sqrt(2), -1j / np.sqrt(2)
        sdeg[slot[h]] = plan.degrees[h]
    Mh, Sh, tabh = M.cpu().numpy(), S.cpu().numpy(), tab.cpu().numpy()
    blk = np.arange(npad) // 64
    region = (np.arange(npad)[:, None] <= np.arange(npad)[None, :]) | (blk[:, None] == blk[None, :])      # upper triangle + diagonal tiles
    for s in range(nb):
        W = np.eye(npad, dtype=np.complex128)
        r = np.ones(npad, dtype=np.complex128)
        for b in range(B):
            W[b * H:(b + 1) * H, b * H:(b + 1) * H] = V
            r[b * H:(b + 1) * H] = 1.0 / np.sqrt(tabh[s, b, 0, sdeg] * tabh[s, b, 1, sdeg])
        want = (r[:, None] * (W.conj().T @ Mh[s] @ W)) / r[None, :]
        assert np.abs(want - want.T).max() < 1e-13 * np.abs(want).max()           # the structure the path relies on
        got = Sh[s]
        assert np.isfinite(got[region]).all()
        assert np.abs(got[region] - want[region]).max() < 1e-13 * np.abs(want).max(), (tree, s)
        assert np.abs(np.diag(got) - 1.0).max() < 1e-15


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["reference", "equilibrated"])
def test_2d_toeplitz_general_fill_vs_list_kernel(amd, lib, scaling, monkeypatch):
    """The list-free 2-D general fill (k_fill2d: one Graf term per entry evaluated directly) against the generic list kernel k_fill
    on the same inputs, both scalings, element-wise (identical arithmetic up to the order of two multiplications)."""
    from biem_helmholtz_sphere_amd import _biem as impl

    l, L = lib
    n_end, B, nb = 23, 4, 2
    rng = np.random.default_rng(3)
    cen = rng.normal(size=(B, 2)) * 0.3 + np.arange(B)[:, None] * np.array([2.4, 0.3])
    rad = rng.uniform(0.6, 1.0, size=B)
    ks = np.array([0.9, 2.3 + 0.2j])
    plan = impl._plan("a", n_end, torch.device("cuda", 0))
    N = B * plan.H
    npad = l.biem_lu_npad(N)
    al = _dev(np.full((1, B), 1.0 + 0.1j), torch.complex128)
    be = _dev(np.full((1, B), 0.4 - 0.1j), torch.complex128)
    k_t, eta_t, cen_t, rad_t = _dev(ks, torch.complex128), _dev(np.array([1.0, 0.7])), _dev(cen[None]), _dev(rad[None])
    tab = torch.empty((nb, B, 3, n_end), dtype=torch.complex128, device="cuda")
    L.check(l.biem_ball_tables(plan.handle, nb, B, k_t.data_ptr(), eta_t.data_ptr(), rad_t.data_ptr(), 0, al.data_ptr(), be.data_ptr(), 0, tab.data_ptr(), None))
    wb = l.biem_fill_workspace_bytes(plan.handle, nb, B)
    work = torch.empty(max(wb, 16), dtype=torch.uint8, device="cuda")
    sc = L.FILL_REFERENCE if scaling == "reference" else L.FILL_EQUILIBRATED
    out = {}
    for name in ("direct", "lists"):
        if name == "lists":
            monkeypatch.setenv("BIEM_FILL_FORM", "entry")
        else:
            monkeypatch.delenv("BIEM_FILL_FORM", raising=False)
        M = torch.full((nb, npad, npad), float("nan"), dtype=torch.complex128, device="cuda")
        L.check(l.biem_fill(plan.handle, nb, B, k_t.data_ptr(), cen_t.data_ptr(), 0, tab.data_ptr(), sc, M.data_ptr(), npad, npad * npad, npad, work.data_ptr(), wb, None))
        out[name] = M.cpu().numpy()
    a, b = out["direct"], out["lists"]
    assert np.isfinite(a).all() and np.isfinite(b).all()
    nz = np.abs(b) > 0
    assert np.max(np.abs(a - b)[nz] / np.abs(b)[nz]) < 1e-13 and np.all(a[~nz] == 0)


# ---------------------------------------------------------------------------- the row-form symmetric factorisation (default path)
@pytest.mark.gpu
@pytest.mark.parametrize("N,nb,nrhs", [(64, 2, 1), (17, 3, 2), (40, 2, 8), (64, 300, 8), (150, 2, 2), (256, 1, 1), (300, 3, 1), (576, 2, 3), (1000, 2, 1), (1345, 1, 2), (700, 2, 12), (130, 1, 70), (200, 2, 300)])
def test_sym_factor_solve_vs_numpy(lib, N, nb, nrhs):
    """biem_sym_factor_solve (A = U^T U in row form, what biem_solve_ldlt runs) on complex-symmetric matrices I + E against
    numpy.linalg.solve; only the UPPER triangle and the diagonal tiles may be read: the strict lower triangle outside the
    diagonal 64 x 64 tiles is poisoned.  On return the upper triangle holds U with U^T U = A."""
    l, L = lib
    rng = np.random.default_rng(N + 11)
    npad = l.biem_lu_npad(N)
    lda = npad + ((nrhs + 7) // 8) * 8
    E = (rng.normal(size=(nb, N, N)) + 1j * rng.normal(size=(nb, N, N))) * (0.12 / np.sqrt(N))
    As = np.eye(N)[None] * (1.0 + 0.2j) + E + np.swapaxes(E, 1, 2)
    Fs = rng.normal(size=(nb, N, nrhs)) + 1j * rng.normal(size=(nb, N, nrhs))
    A = np.zeros((nb, npad, lda), dtype=np.complex128)
    A[:, :N, :N] = As
    for i in range(N, npad):
        A[:, i, i] = 1.0
    blk = np.arange(npad) // 64
    lower_off = (blk[:, None] > blk[None, :])
    A[:, :npad, :npad][:, lower_off] = 1e30                    # must never be read
    A[:, :N, npad:npad + nrhs] = Fs
    dA = _dev(A, torch.complex128)
    info = torch.ones(nb, dtype=torch.int32, device="cuda")
    wb = l.biem_lu_workspace_bytes(nb, npad, nrhs)
    work = torch.empty(wb, dtype=torch.uint8, device="cuda")
    L.check(l.biem_sym_factor_solve(nb, npad, nrhs, dA.data_ptr(), lda, npad * lda, info.data_ptr(), work.data_ptr(), wb, None))
    torch.cuda.synchronize()
    assert (info.cpu().numpy() == 0).all(), info.cpu().numpy()
    out = dA.cpu().numpy()
    X = out[:, :N, npad:npad + nrhs]
    for s in range(nb):
        Xo = np.linalg.solve(As[s], Fs[s])
        assert np.abs(X[s] - Xo).max() / np.abs(Xo).max() < 1e-12, (N, s)
        U = np.triu(out[s, :N, :N])
        assert np.abs(U.T @ U - As[s]).max() < 1e-12, (N, s)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"BIEM_DIAG_FORM": "step"}, {"BIEM_BACK_FORM": "col"}, {"BIEM_BACK_FORM": "step"}, {"BIEM_BACK_FORM": "row"},
                                 {"BIEM_DIAG_FORM": "step", "BIEM_BACK_FORM": "row"}, {"BIEM_RHS_SIDE_STREAM": "0"}, {"BIEM_RHS_SIDE_STREAM": "1", "BIEM_BACK_FORM": "row"}])
@pytest.mark.parametrize("N,nb,nrhs", [(150, 2, 2), (576, 1, 1), (700, 3, 12), (1345, 1, 2), (300, 12, 1), (200, 2, 190)])
def test_sym_factor_solve_forms(lib, monkeypatch, env, N, nb, nrhs):
    """The forms the row-form factorisation chooses between by the number of systems, forced through the environment: the diagonal
    block with one pivot per barrier (default: four), and the back substitution by rows / by column blocks with a triangular solve
    per diagonal block / by column blocks with the stored inverses of the diagonal blocks (one launch per block, default for up to
    8 systems); the right-hand sides' update of a group on the caller's stream instead of beside the K = 256 update on a second one.
    Same checks as test_sym_factor_solve_vs_numpy."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    test_sym_factor_solve_vs_numpy(lib, N, nb, nrhs)


@pytest.mark.gpu
def test_sym_factor_rejections_and_growth(lib, monkeypatch):
    """Acceptance tests of the row form: a diagonal below a hundredth of an entry of its row (a multiplier above 100) marks the system with
    the panel's first row; max |u_ii u_ic| / max |a_ij| (moduli) equals the NumPy value - limits 1 % below / above it mark / pass
    the system (info = -(Npad + 1)); a NaN marks it at the default limit."""
    l, L = lib

    def run(A, nrhs=1):
        nb, npad = A.shape[0], A.shape[1]
        dA = _dev(A, torch.complex128)
        info = torch.zeros(nb, dtype=torch.int32, device="cuda")
        wb = l.biem_lu_workspace_bytes(nb, npad, nrhs)
        work = torch.empty(wb, dtype=torch.uint8, device="cuda")
        L.check(l.biem_sym_factor_solve(nb, npad, nrhs, dA.data_ptr(), A.shape[2], npad * A.shape[2], info.data_ptr(), work.data_ptr(), wb, None))
        torch.cuda.synchronize()
        return info.cpu().tolist()

    N = 128
    A = np.zeros((3, N, N + 8), dtype=np.complex128)
    A[:, :, :N] = np.eye(N)
    A[1, 70, 70] = 0.001
    A[1, 90, 70] = A[1, 70, 90] = 1.0                           # inside the second diagonal block
    A[2, 10, 10] = 0.001
    A[2, 10, 100] = A[2, 100, 10] = 1.0                         # multiplier in the strip right of the first block
    assert run(A) == [0, -65, -1]
    N, npad = 200, 256
    rng = np.random.default_rng(5)
    E = (rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))) * (0.12 / np.sqrt(N))
    As = np.eye(N) * (1.0 + 0.2j) + E + E.T
    M = As.copy()
    for c in range(N):
        M[c + 1:, c] /= M[c, c]
        M[c + 1:, c + 1:] -= np.outer(M[c + 1:, c], M[c, c + 1:])
    ratio = np.abs(np.triu(M)).max() / np.abs(As).max()
    B = np.zeros((2, npad, npad + 8), dtype=np.complex128)
    B[:, :N, :N] = As
    B[:, np.arange(N, npad), np.arange(N, npad)] = 1.0
    B[:, :N, npad] = 1.0
    monkeypatch.setenv("BIEM_LDLT_GROWTH_MAX", repr(float(0.99 * ratio)))
    assert run(B) == [-(npad + 1)] * 2
    monkeypatch.setenv("BIEM_LDLT_GROWTH_MAX", repr(float(1.01 * ratio)))
    assert run(B) == [0, 0]
    monkeypatch.delenv("BIEM_LDLT_GROWTH_MAX")
    B[1, 20, 150] = np.nan
    bad = run(B)
    assert bad[0] == 0 and bad[1] < 0
    # the same acceptance tests in the one-launch path of small systems (at most 128 active rows, the whole system in LDS)
    N = 64
    A = np.zeros((4, N, N + 8), dtype=np.complex128)
    A[:, :, :N] = np.eye(N)
    A[:, :, N] = 1.0
    A[1, 10, 10] = 0.001
    A[1, 10, 50] = A[1, 50, 10] = 1.0
    A[2, 63, 63] = 0.0
    A[3, 5, 40] = A[3, 40, 5] = 0.4                             # multiplier 0.4: accepted
    assert run(A) == [0, -1, -1, 0]
    As = (np.eye(N) * (1.0 + 0.2j) + (E + E.T)[:N, :N])
    M = As.copy()
    for c in range(N):
        M[c + 1:, c] /= M[c, c]
        M[c + 1:, c + 1:] -= np.outer(M[c + 1:, c], M[c, c + 1:])
    ratio = np.abs(np.triu(M)).max() / np.abs(As).max()
    B = np.zeros((2, N, N + 8), dtype=np.complex128)
    B[:, :, :N] = As
    B[:, :, N] = 1.0
    monkeypatch.setenv("BIEM_LDLT_GROWTH_MAX", repr(float(0.99 * ratio)))
    assert run(B) == [-(N + 1)] * 2
    monkeypatch.setenv("BIEM_LDLT_GROWTH_MAX", repr(float(1.01 * ratio)))
    assert run(B) == [0, 0]
    monkeypatch.delenv("BIEM_LDLT_GROWTH_MAX")
    B[1, 20, 50] = np.nan
    bad = run(B)
    assert bad[0] == 0 and bad[1] < 0
