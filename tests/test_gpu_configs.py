"""BASELINE.json configurations at (or near) full size on the GPU: parity with the oracle where it finishes in seconds,
size-independent properties (eta-independence of u_scat, batch == one-by-one, residual of the solved system) elsewhere."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import biem_oracle as O  # noqa: E402  (checker)


@pytest.fixture(scope="module")
def amd():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import biem_helmholtz_sphere_amd as amd

    return amd


def _dev(a, dtype=torch.float64):
    return torch.as_tensor(np.array(a), device="cuda").to(dtype).contiguous()


def _probes(d, radius):
    ang = 2 * np.pi * np.arange(15) / 15
    p = np.zeros((16, d))
    p[1:, 0] = radius * np.cos(ang)
    p[1:, 1] = radius * np.sin(ang)
    return p


def test_cfg2_3d_4balls_nend12(amd):
    """configs[1]: d=3, 2x2 grid pitch 4, n_end=12, sound-soft (N = 576) vs the oracle."""
    cen = O.grid_centers(1, 3)
    k = 1.0
    uin_o, _ = O.plane_wave(k, [1.0, 0, 0])
    res = O.solve_biem("ba", centers=cen, radii=np.ones(4), k=k, n_end=12, uin=uin_o)
    x = _probes(3, 1.5 * 3.0)
    uo = O.uscat(res, x)
    c = amd.create_from_branching_types("ba")
    uin, _ = amd.plane_wave(k=_dev(k), direction=_dev([1.0, 0, 0]))
    calc = amd.biem(c, centers=_dev(cen), radii=_dev(np.ones(4)), k=_dev(k), n_end=12, uin=uin)
    u = calc.uscat(_dev(x.T)).cpu().numpy()
    assert np.max(np.abs(u - uo) / np.abs(uo)) < 1e-10
    dg, do = calc.density.cpu().numpy(), res.density
    assert np.max(np.abs(dg - do)) < 1e-10 * np.abs(do).max()                 # entries that vanish by symmetry are rounding noise
    big = np.abs(do) > 1e-6 * np.abs(do).max()
    assert np.max(np.abs(dg - do)[big] / np.abs(do)[big]) < 1e-8


def test_cfg4_2d_32balls_nend64_robin(amd):
    """configs[3]: d=2, 32 balls (4 x 8 grid, pitch 4), n_end=64, Robin alpha=beta=1 (N = 4064, entries span 1e-107..1e+104
    in the reference scaling) vs the oracle; also the reference golden for the 16-ball n_end=64 sound-soft grid."""
    ax0, ax1 = np.arange(4) * 4.0 - 6.0, np.arange(8) * 4.0 - 14.0
    x0, x1 = np.meshgrid(ax0, ax1, indexing="ij")
    cen = np.stack([x0.ravel(), x1.ravel()], -1)
    k = 1.0
    uin_o, ugr_o = O.plane_wave(k, [1.0, 0.0])
    res = O.solve_biem("a", centers=cen, radii=np.ones(32), k=k, n_end=64, alpha=1.0, beta=1.0, uin=uin_o, uin_grad=ugr_o)
    x = _probes(2, 1.5 * 15.0)
    uo = O.uscat(res, x)
    c = amd.create_from_branching_types("a")
    uin, ugr = amd.plane_wave(k=_dev(k), direction=_dev([1.0, 0.0]))
    calc = amd.biem(c, centers=_dev(cen), radii=_dev(np.ones(32)), k=_dev(k), n_end=64, alpha=1.0, beta=1.0, uin=uin, uin_grad=ugr)
    u = calc.uscat(_dev(x.T)).cpu().numpy()
    assert np.max(np.abs(u - uo) / np.abs(uo)) < 1e-10
    # golden accuracy_n_balls_a.csv:41 (16 balls, n_end = 64, sound-soft, probe at the origin)
    cen16 = O.grid_centers(2, 2)
    uin1, _ = amd.plane_wave(k=_dev(1.0), direction=_dev([1.0, 0.0]))
    g = amd.biem(c, centers=_dev(cen16), radii=_dev(np.ones(16)), k=_dev(1.0), eta=_dev(1.0), n_end=64, uin=uin1)
    assert abs(complex(g.uscat(_dev(np.zeros(2))).cpu().numpy()) - (-1.0480631533178735 - 0.27121926513493827j)) < 1e-11


def _cfg5_geometry():
    ax0, ax1 = np.arange(2) * 4.0 - 2.0, np.arange(4) * 4.0 - 6.0
    x0, x1 = np.meshgrid(ax0, ax1, indexing="ij")
    cen = np.zeros((8, 4))
    cen[:, 0], cen[:, 1] = x0.ravel(), x1.ravel()
    return cen


def test_cfg5_4d_8balls_nend10_full_batch_of_512_pairs(amd):
    """configs[4] at its real batch: d=4 'bba', 8 balls (2 x 4 grid in the x0-x1 plane), n_end=10 (N = 3080),
    512 (k, eta) pairs = 32 k's in linspace(0.5, 4) x 16 eta's in linspace(0.25, 4), passed as a (32, 16) batch (81 GB of
    matrices: exercises the chunking).  u_scat does not depend on eta (it only rescales the density): checked across the 16
    eta's of every k; the oracle at two pairs; the density itself scales with eta as blc does, checked through the oracle."""
    cen = _cfg5_geometry()
    ks, etas = np.linspace(0.5, 4.0, 32), np.linspace(0.25, 4.0, 16)
    kk, ee = np.meshgrid(ks, etas, indexing="ij")                      # (32, 16)
    dirs = np.zeros((4, 32, 16)); dirs[0] = 1.0
    c = amd.create_from_branching_types("bba")
    uin, _ = amd.plane_wave(k=_dev(kk), direction=_dev(dirs))
    calc = amd.biem(c, centers=_dev(cen)[None, None], radii=_dev(np.ones(8))[None, None], k=_dev(kk), eta=_dev(ee), n_end=10, uin=uin)
    assert tuple(calc.density.shape) == (32, 16, 8, 385)
    from biem_helmholtz_sphere_amd import _biem as impl

    st = dict(impl._last_solve_stats)
    assert st["ldlt_systems"] + st["lu_systems"] >= 512 and st["ldlt_systems"] == 512, st
    x = _probes(4, 1.5 * 7.0)
    u = calc.uscat(_dev(x.T)).cpu().numpy()        # (16 probes, 32, 16)
    assert u.shape == (16, 32, 16) and np.isfinite(u).all()
    spread = np.max(np.abs(u - u[:, :, :1]), axis=(0, 2)) / np.max(np.abs(u[:, :, 0]), axis=0)     # per k, over probes and eta
    assert spread.max() < 1e-10, spread
    for ik, ie in ((3, 0), (31, 15)):
        uin_o, _ = O.plane_wave(float(ks[ik]), [1.0, 0, 0, 0])
        res = O.solve_biem("bba", centers=cen, radii=np.ones(8), k=float(ks[ik]), n_end=10, eta=float(etas[ie]), uin=uin_o)
        uo = O.uscat(res, x)
        assert np.max(np.abs(u[:, ik, ie] - uo) / np.abs(uo)) < 1e-10, (ik, ie)
        dg, do = calc.density[ik, ie].cpu().numpy(), res.density
        assert np.max(np.abs(dg - do)) < 1e-9 * np.abs(do).max(), (ik, ie)


def _cfg3_inputs(ks):
    cen = O.grid_centers(2, 3)
    dirs = np.zeros((3, len(ks))); dirs[0] = 1.0
    return cen, dirs


def test_cfg3_full_size_vs_oracle(amd):
    """configs[2] at full size (N = 6400): u_scat at 16 probes against the oracle's dense solve for k = 0.5, 3.7, 8 (<= 1e-10),
    densities entry-wise, and the batch against a single-system call."""
    ks = np.array([0.5, 3.7, 8.0])
    cen, dirs = _cfg3_inputs(ks)
    c = amd.create_from_branching_types("ba")
    uin, _ = amd.plane_wave(k=_dev(ks), direction=_dev(dirs))
    calc = amd.biem(c, centers=_dev(cen)[None], radii=_dev(np.ones(16))[None], k=_dev(ks), n_end=20, uin=uin)
    x = _probes(3, 10.5)
    u = calc.uscat(_dev(x.T)).cpu().numpy()
    for i, k in enumerate(ks):
        uo_in, _ = O.plane_wave(float(k), [1.0, 0, 0])
        res = O.solve_biem("ba", centers=cen, radii=np.ones(16), k=float(k), n_end=20, uin=uo_in)
        uo = O.uscat(res, x)
        assert np.max(np.abs(u[:, i] - uo) / np.abs(uo)) < 1e-10, k
        dg, do = calc.density[i].cpu().numpy(), res.density
        assert np.max(np.abs(dg - do)) < 1e-10 * np.abs(do).max(), k
    uin1, _ = amd.plane_wave(k=_dev(ks[1]), direction=_dev([1.0, 0, 0]))
    one = amd.biem(c, centers=_dev(cen), radii=_dev(np.ones(16)), k=_dev(ks[1]), n_end=20, uin=uin1)
    assert torch.max(torch.abs(one.density - calc.density[1])) / torch.max(torch.abs(calc.density[1])) < 1e-12


def test_cfg3_all_256_wavenumbers_ldlt_vs_pivoted_lu(amd, monkeypatch):
    """The headline batch itself: all 256 wavenumbers of linspace(0.5, 8) (interior Dirichlet resonances of the unit spheres lie
    inside the range) solved by the default complex-symmetric L D L^T path and by the pivoted LU (the reference's algorithm,
    BIEM_SOLVER=lu): every system's density agrees to 1e-10 of its largest entry, no system needed the fallback, and u_scat at
    the probes is finite and agrees."""
    from biem_helmholtz_sphere_amd import _biem as impl

    ks = np.linspace(0.5, 8.0, 256)
    cen, dirs = _cfg3_inputs(ks)
    c = amd.create_from_branching_types("ba")
    uin, _ = amd.plane_wave(k=_dev(ks), direction=_dev(dirs))
    kw = dict(centers=_dev(cen)[None], radii=_dev(np.ones(16))[None], k=_dev(ks), n_end=20, uin=uin)
    monkeypatch.delenv("BIEM_SOLVER", raising=False)
    a = amd.biem(c, **kw)
    st = dict(impl._last_solve_stats)
    assert st == {"ldlt_systems": 256, "lu_systems": 0}, st
    da = a.density.clone()
    x = _dev(_probes(3, 10.5).T)
    ua = a.uscat(x)
    del a
    torch.cuda.empty_cache()
    monkeypatch.setenv("BIEM_SOLVER", "lu")
    b = amd.biem(c, **kw)
    assert dict(impl._last_solve_stats) == {"ldlt_systems": 0, "lu_systems": 256}
    db = b.density
    err = torch.amax(torch.abs(da - db), dim=(1, 2)) / torch.amax(torch.abs(db), dim=(1, 2))       # per system
    assert float(err.max()) < 1e-10, (int(err.argmax()), float(err.max()))
    ub = b.uscat(x)
    assert bool(torch.isfinite(ua.real).all()) and float((torch.abs(ua - ub) / torch.abs(ub)).max()) < 1e-10


def test_beyond_the_configs_n12544_vs_oracle(amd):
    """N = 12544 (16 balls, n_end 28: twice the headline N, 196 tile rows, 49 four-panel groups) against the oracle at 8 probe
    points; the oracle's dense solve takes ~30 s on the GPU box's host cores."""
    n_end = 28
    ax = np.arange(-2, 2) * 4.0 + 2.0
    x0, x1 = np.meshgrid(ax, ax, indexing="ij")
    cen = np.stack([x0.ravel(), x1.ravel(), np.zeros(16)], -1)
    ks = np.array([3.0, 7.5])
    dirs = np.zeros((3, 2)); dirs[0] = 1
    uin, _ = amd.plane_wave(k=_dev(ks), direction=_dev(dirs))
    c = amd.create_from_branching_types("ba")
    calc = amd.biem(c, centers=_dev(cen)[None], radii=_dev(np.ones(16))[None], k=_dev(ks), n_end=n_end, uin=uin)
    ang = 2 * np.pi * np.arange(7) / 7
    pts = np.concatenate([np.zeros((1, 3)), np.stack([10.5 * np.cos(ang), 10.5 * np.sin(ang), np.zeros(7)], -1)])
    ug = calc.uscat(_dev(pts.T.copy())).cpu().numpy()
    uo, _ = O.plane_wave(float(ks[1]), [1.0, 0, 0])
    ref = O.uscat(O.solve_biem("ba", centers=cen, radii=np.ones(16), k=float(ks[1]), n_end=n_end, uin=uo), pts)
    assert np.max(np.abs(ug[:, 1] - ref) / np.abs(ref)) < 1e-10


def test_order_ceilings_of_the_fill(amd, golden_dir):
    """Large orders.  (1) 3-D n_end = 39, the highest order the reference's 3-D goldens reach (accuracy_k_ba.csv:382-391; pair table
    95 KB of LDS, N = 3042): golden rows at k = 1 and 8.  (2) n_end = 43: the pair table no longer fits the one-pair-per-lane form
    (H2 = 85^2 > 7168) - the one-system-per-lane form takes over (no ceiling on the order); the solution has converged long
    before, so it must equal the n_end = 39 golden.  (3) 2-D n_end = 152 (radial tables up to order 320; the reference's 2-D
    goldens go on to n_end = 3444, beyond what is built here): high-wavenumber golden rows, k|t| up to 16384.
    (4) 2-D beyond the former ceiling of n_end = 160, up to the reference's largest order 3444."""
    import csv
    import os

    from biem_helmholtz_sphere_amd import _lib as L

    def run(tree, n_end, k_op):
        c = amd.create_from_branching_types(tree)
        d = c.c_ndim
        e0 = np.zeros(d); e0[0] = 1.0
        cen = O.grid_centers(0, d)
        uin, _ = amd.plane_wave(k=_dev(1.0), direction=_dev(e0))      # incident k = 1 quirk of the golden driver (cli.py:238-244)
        calc = amd.biem(c, centers=_dev(cen), radii=_dev(np.ones(2)), k=_dev(k_op), eta=_dev(1.0), n_end=n_end, uin=uin)
        return complex(calc.uscat(_dev(np.zeros(d))).cpu().numpy())

    rows = {}
    with open(os.path.join(golden_dir, "accuracy_k_ba.csv")) as f:
        for r in csv.DictReader(f):
            rows[(r["branching_types"], int(r["n_end"]), round(float(r["k"]), 6))] = complex(r["uscat"])
    with open(os.path.join(golden_dir, "accuracy_k_a.csv")) as f:
        for r in csv.DictReader(f):
            rows[("a2", int(r["n_end"]), round(float(r["k"]), 6))] = complex(r["uscat"])
    for k in (1.0, 8.0):
        assert abs(run("ba", 39, k) - rows[("ba", 39, k)]) < 1e-11, k
    assert abs(run("ba", 43, 1.0) - rows[("ba", 39, 1.0)]) < 1e-11
    for k in (8.0, 64.0, 1024.0, 4096.0):
        got, want = run("a", 152, k), rows[("a2", 152, k)]
        assert abs(got - want) < 1e-9 * max(1.0, abs(want)), (k, got, want)
    # (4) 2-D beyond n_end = 160 (the list-free Toeplitz fill, radial recurrences in global memory): rows of the reference's
    # accuracy sweep up to its largest order, n_end = 3444 at k = 2896.3 (N = 13774; cli.py:223)
    for k, n_end in ((2.0 ** 3.5, 181), (32.0, 256), (128.0, 512), (512.0, 1024), (1024.0, 1722), (4096.0, 1722), (2.0 ** 11.5, 3444)):
        got, want = run("a", n_end, k), rows[("a2", n_end, round(k, 6))]
        assert abs(got - want) < 1e-9 * max(1.0, abs(want)), (k, n_end, got, want)
    # a 2-D plan of such an order holds no term lists (they would be H^2 one-term entries): the accessor says so
    import ctypes as C

    plan = C.c_void_p()
    lib = L.load()
    L.check(lib.biem_plan_create_host(L.TREE_IDS["a"], 200, C.byref(plan)))
    assert lib.biem_plan_terms(plan, None, None, None) == 3 and b"no term lists" in lib.biem_last_error()
    lib.biem_plan_destroy(plan)


@pytest.mark.parametrize("tree,n_end,ks", [("ba", 43, (2.0, 10.0)), ("bba", 15, (2.0, 6.0))])
def test_close_spheres_beyond_the_lds_ceiling_fall_back_to_pivoted_lu(amd, monkeypatch, tree, n_end, ks):
    """The reference hands EVERY system to a pivoted dense solve (_biem.py:797).  Two unit spheres 0.04 apart at orders whose general
    pair table does not fit LDS (3-D n_end 43, 4-D n_end 15): (1) the default path, whatever it decides per system; (2) every system
    rejected by the symmetric factorisation (BIEM_LDLT_PIVOT_REL = 1e30), i.e. re-filled in the general form - pair table read
    from global memory - and solved by the pivoted LU.  Both against the oracle (numpy.linalg.solve of the reference-scaled matrix)."""
    from biem_helmholtz_sphere_amd import _biem as impl

    tr = O.tree(tree)
    d = tr.d
    cen = np.zeros((2, d))
    cen[0, 1], cen[1, 1] = 1.02, -1.02
    e0 = np.zeros(d)
    e0[0] = 1.0
    # probes on the circle of radius 4 (the origin of _probes lies INSIDE the 0.04 gap, 0.02 from either surface: there the series
    # sum_n c_n h_n(1.02 k) Y_n is a sum of terms ~1e3 x its value at these orders and amplifies the rounding of the tiny high-order
    # coefficients - 8e-9 between any two solvers - which says nothing about the solve)
    x = _probes(d, 4.0)[1:]
    c = amd.create_from_branching_types(tree)
    ks = np.asarray(ks)
    dirs = np.zeros((d, len(ks)))
    dirs[0] = 1.0
    uo = []
    for k in ks:
        uin_o, _ = O.plane_wave(float(k), e0)
        uo.append(O.uscat(O.solve_biem(tree, centers=cen, radii=np.ones(2), k=float(k), n_end=n_end, uin=uin_o), x))
    uo = np.stack(uo, -1)

    def run():
        uin, _ = amd.plane_wave(k=_dev(ks), direction=_dev(dirs))
        calc = amd.biem(c, centers=_dev(cen)[None], radii=_dev(np.ones(2))[None], k=_dev(ks), eta=_dev(np.ones(len(ks))), n_end=n_end, uin=uin)
        return calc.uscat(_dev(x.T)).cpu().numpy(), dict(impl._last_solve_stats)

    u, st = run()
    assert st["ldlt_systems"] == len(ks)
    assert np.max(np.abs(u - uo) / np.abs(uo)) < 1e-10, st
    monkeypatch.setenv("BIEM_LDLT_PIVOT_REL", "1e30")
    u, st = run()
    assert st["lu_systems"] == len(ks) > 0
    assert np.max(np.abs(u - uo) / np.abs(uo)) < 1e-10


def test_matrix_attribute_beyond_the_lds_ceiling(amd):
    """`matrix` (reference scaling, _biem.py:792,818) at 3-D n_end = 43 - the general pair table (H2 = 85^2 entries) no longer fits
    LDS and is read from global memory - against the oracle's assembly: norm-wise per block (1e-12), element-wise (1e-9 relative)
    where the degrees satisfy n + n' <= 20, and exact zeros where the oracle has them.  Beyond that an entry (S|R)_{h'->h} is an
    alternating sum over n'' of terms far above its value and rests on triple-integral coefficients below the 1e-16 absolute accuracy of
    a double-precision quadrature (on either side): entries of relative size 1e-20 of their block differ by O(1) between any two
    evaluations (tools/dbg_matrix43.py) - they carry no information in double precision, in the reference's assembly either."""
    tree, n_end, k, eta = "ba", 43, 1.3, 0.7
    tr = O.tree(tree)
    cen = np.array([[0.0, 1.3, 0.2], [0.3, -1.4, -0.1]])
    rad = np.array([1.0, 0.8])
    alpha, beta = 1.0 + 0.25j, 0.4 - 0.1j
    c = amd.create_from_branching_types(tree)
    calc = amd.biem(c, centers=_dev(cen), radii=_dev(rad), k=_dev(k), eta=_dev(eta), n_end=n_end, alpha=alpha, beta=beta)
    M = calc.matrix.cpu().numpy()
    H = tr.n_harm(n_end)
    assert M.shape == (2, H, 2, H)
    A, _ = O.assemble(tr, n_end, k, eta, cen, rad, np.full(2, alpha), np.full(2, beta))
    deg = tr.degrees(n_end)
    low = (deg[:, None] + deg[None, :]) <= 20
    for b in range(2):
        for bp in range(2):
            Mb, Ab = M[b, :, bp, :], A[b, :, bp, :]
            assert np.abs(Mb - Ab).max() < 1e-12 * np.abs(Ab).max(), (b, bp)
            nz = np.abs(Ab) > 1e-200
            assert np.max(np.abs(Mb - Ab)[nz & low] / np.abs(Ab)[nz & low]) < 1e-9, (b, bp)
            if b == bp:
                assert np.all(Mb[~nz] == 0)                     # a ball's own block: exactly diagonal
            else:
                assert np.all(np.abs(Mb[~nz]) < 1e-30 * np.abs(Ab).max())   # (entries the oracle's products underflow on: 1e-37 here)
