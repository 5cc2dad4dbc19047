"""Randomised end-to-end parity: biem() + uscat() of the HIP path against the CPU oracle on seeded random configurations.

Each case draws a tree, an order, a ball count, a non-overlapping geometry, wavenumbers (real and complex), eta, Robin
coefficients (scalar or per ball, real or complex, either may vanish) and an incident field (plane wave or point source); every
fourth case puts equal spheres on a lattice (ball pairs that share a displacement share their block of the fill).  u_scat at
exterior points - near field, per ball and far field - is compared with the oracle to the north-star tolerance (1e-10 relative).
The sizes cover the one-launch path of small systems (N <= 128), the blocked row form (one to four 64-row panels), the
single-ball shortcut and both field-evaluation kernels.  Seeds are fixed: a failure names its case.  BIEM_FUZZ_SEEDS=n runs n
cases (default 96; 1200 pass in 26 s).
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import biem_oracle as O  # noqa: E402  (test infrastructure: the checker)

TREES = {"a": (2, 2, 20), "ba": (3, 2, 9), "bpa": (3, 2, 7), "bba": (4, 2, 5), "bpbpa": (4, 2, 5), "caa": (4, 2, 4)}   # d, n_end range


@pytest.fixture(scope="module")
def amd():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import biem_helmholtz_sphere_amd as amd

    return amd


def _dev(a, dtype=torch.float64):
    return torch.as_tensor(np.array(a), device="cuda").to(dtype).contiguous()


def _geometry(rng, B, d):
    cen, rad = [], []
    while len(cen) < B:
        c = rng.uniform(-3.5, 3.5, size=d)
        r = rng.uniform(0.4, 1.0)
        if all(np.linalg.norm(c - c2) > 1.2 * (r + r2) for c2, r2 in zip(cen, rad)):
            cen.append(c)
            rad.append(r)
    return np.array(cen), np.array(rad)


N_SEEDS = int(os.environ.get("BIEM_FUZZ_SEEDS", "96"))       # (more for a one-off soak)


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_configuration_vs_oracle(amd, seed, monkeypatch):
    monkeypatch.setenv("BIEM_FILL_DEDUPE_MIN", "1")               # pair classes of the fill for these 3-system calls too
    rng = np.random.default_rng(1000 + seed)
    name = list(TREES)[seed % len(TREES)]
    d, lo, hi = TREES[name]
    n_end = int(rng.integers(lo, hi + 1))
    B = int(rng.integers(1, 6 if d < 4 else 4))
    cen, rad = _geometry(rng, B, d)
    lattice = seed % 4 == 3 and B > 2        # equal spheres on a lattice: ball pairs share displacements (pair classes of the fill)
    if lattice:
        pitch, nx = float(rng.uniform(2.3, 3.4)), int(rng.integers(2, 4))
        cen = np.zeros((B, d)); cen[:, 0] = pitch * (np.arange(B) % nx); cen[:, 1] = pitch * (np.arange(B) // nx)
        rad = np.full(B, float(rng.uniform(0.5, 1.0)))
    K = 3
    ks = rng.uniform(0.4, 3.5, size=K).astype(np.complex128)
    if seed % 3 == 1:
        ks = ks + 1j * rng.uniform(0.0, 0.4, size=K)
    eta = float(rng.uniform(0.5, 2.0))
    mode = seed % 4
    alpha = complex(rng.normal(), rng.normal() if mode == 3 else 0.0) if mode != 1 else 0.0
    beta = complex(rng.normal(), rng.normal() if mode == 3 else 0.0) if mode != 0 else 0.0
    if mode == 1:
        beta = 1.0
    if mode == 0:
        alpha = 1.0
    per_ball = seed % 5 == 2 and B > 1 and not lattice
    if per_ball:
        alpha_v = alpha * (1.0 + 0.3 * rng.random(B)); beta_v = beta * (1.0 + 0.3 * rng.random(B))
    source = seed % 6 == 5
    src = 9.0 * np.eye(d)[0] + rng.normal(size=d)
    direction = rng.normal(size=d)
    x = 7.0 * rng.normal(size=(12, d))
    x = x[[all(np.linalg.norm(p - c) > 1.05 * r for c, r in zip(cen, rad)) for p in x]][:6]

    c = amd.create_from_branching_types(name)
    kt = _dev(ks, torch.complex128) if np.iscomplexobj(ks) and np.any(ks.imag != 0) else _dev(ks.real)
    if source:
        uin, ugr = amd.point_source(k=kt, source=_dev(np.repeat(src[:, None], K, axis=1)), n=seed % 2)
    else:
        uin, ugr = amd.plane_wave(k=kt, direction=_dev(np.repeat(direction[:, None], K, axis=1)))
    kw = dict(centers=_dev(cen)[None], radii=_dev(rad)[None], k=kt, eta=_dev(np.full(K, eta)), n_end=n_end, uin=uin, uin_grad=ugr)
    if per_ball:
        kw["alpha"] = _dev(np.broadcast_to(alpha_v, (K, B)).copy(), torch.complex128)
        kw["beta"] = _dev(np.broadcast_to(beta_v, (K, B)).copy(), torch.complex128)
    else:
        kw["alpha"], kw["beta"] = alpha, beta
    calc = amd.biem(c, **kw)
    u = calc.uscat(_dev(x.T)).cpu().numpy()
    ub = calc.uscat(_dev(x.T), per_ball=True).cpu().numpy()
    uf = calc.uscat(_dev(x.T), far_field=True).cpu().numpy()
    for i in range(K):
        kk = complex(ks[i]) if np.any(ks.imag != 0) else float(ks[i].real)
        uo, go = O.point_source(kk, src, seed % 2) if source else O.plane_wave(kk, direction)
        res = O.solve_biem(name, centers=cen, radii=rad, k=kk, n_end=n_end, eta=eta, alpha=alpha_v if per_ball else alpha,
                           beta=beta_v if per_ball else beta, uin=uo, uin_grad=go)
        case = (seed, name, n_end, B, kk)
        ref = O.uscat(res, x)
        assert np.max(np.abs(u[:, i] - ref)) < 1e-10 * np.max(np.abs(ref)), case
        refb = O.uscat(res, x, per_ball=True)
        assert np.max(np.abs(ub[:, i] - refb)) < 1e-10 * np.max(np.abs(refb)), case
        reff = O.uscat(res, x, far_field=True)
        assert np.max(np.abs(uf[:, i] - reff)) < 1e-10 * np.max(np.abs(reff)), case
