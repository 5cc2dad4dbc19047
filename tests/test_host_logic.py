"""CPU tests of the host logic: ABI surface, plan tables, input validation, memory guard, coordinates, sharding."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import biem_helmholtz_sphere_amd as amd
from biem_helmholtz_sphere_amd import _biem, _coords, _dist, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "biem_mi355.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(biem_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/biem_mi355.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.biem_version() >= 100


def test_plan_host_tables_without_gpu():
    lib = _lib.load()
    for tree, n_end, H, Q in (("a", 6, 11, 12), ("ba", 6, 36, 72), ("bba", 4, 30, 128), ("caa", 3, 14, 108)):
        plan = C.c_void_p()
        _lib.check(lib.biem_plan_create_host(_lib.TREE_IDS[tree], n_end, C.byref(plan)))
        d, h, q, h2, nt = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_longlong()
        _lib.check(lib.biem_plan_info(plan, C.byref(d), C.byref(h), C.byref(q), C.byref(h2), C.byref(nt)))
        assert (h.value, q.value) == (H, Q)
        assert h.value == _coords.harm_count(tree, n_end) and h2.value == _coords.harm_count(tree, 2 * n_end - 1)
        w = np.zeros(q.value)
        y = np.zeros((q.value, d.value))
        _lib.check(lib.biem_plan_quadrature(plan, y.ctypes.data, w.ctypes.data))
        area = {2: 2 * np.pi, 3: 4 * np.pi, 4: 2 * np.pi**2}[d.value]
        assert abs(w.sum() - area) < 1e-12 and np.allclose(np.linalg.norm(y, axis=1), 1.0)
        # the reduced-table form of the symmetric fill: every term list of one kind and one phase (checked entry by entry at build time)
        e, npz, rok, rows, gok = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _lib.check(lib.biem_plan_fill_info(plan, C.byref(e), C.byref(npz), C.byref(rok), C.byref(rows), C.byref(gok)))
        assert rok.value == 1 and gok.value == 1 and 0 < npz.value <= e.value < h2.value and rows.value > 0, (tree, e.value, npz.value, rok.value, rows.value)
        lib.biem_plan_destroy(plan)
    plan = C.c_void_p()
    assert lib.biem_plan_create_host(7, 3, C.byref(plan)) == 3          # BIEM_ERR_UNSUPPORTED
    assert b"unsupported" in lib.biem_last_error()


def test_product_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    c = amd.create_from_branching_types("ba")
    with pytest.raises(_lib.BiemLibraryError, match="no CPU fallback"):
        amd.biem(c, centers=np.zeros((2, 3)), radii=np.ones(2), k=np.asarray(1.0), n_end=3)


def test_input_validation_messages():
    c = amd.create_from_branching_types("ba")
    v = _biem._validate_biem_inputs
    assert v(c, np.zeros((5, 2, 3)), np.ones((5, 2)), np.ones(5), None, 1.0, 0.0) == (5,)
    assert v(c, np.zeros((1, 2, 3)), np.ones((1, 2)), np.ones(7), np.ones(7), 1.0, 0.0) == (7,)
    with pytest.raises(ValueError, match="decoupling parameter must be real"):
        v(c, np.zeros((2, 3)), np.ones(2), np.asarray(1.0), np.asarray(1.0 + 1j), 1.0, 0.0)
    with pytest.raises(ValueError, match="are not the same"):
        v(c, np.zeros((2, 3)), np.ones(2), np.ones(4), None, 1.0, 0.0)
    with pytest.raises(ValueError, match="are not broadcastable"):
        v(c, np.zeros((3, 2, 3)), np.ones((3, 2)), np.ones(4), None, 1.0, 0.0)
    with pytest.raises(ValueError, match="centers.shape\\[:-1\\] and radii.shape"):
        v(c, np.zeros((4, 3)), np.ones(5), np.asarray(1.0), None, 1.0, 0.0)
    with pytest.raises(ValueError, match="last dimension of centers must be c.c_ndim=3"):
        v(c, np.zeros((2, 2)), np.ones(2), np.asarray(1.0), None, 1.0, 0.0)
    with pytest.raises(TypeError):
        v(c, [[0, 0, 0]], np.ones(1), np.asarray(1.0), None, 1.0, 0.0)
    # complex wavenumbers are accepted (reference gui.py:296-301); a complex decoupling parameter is not (:267-268)
    assert v(c, np.zeros((2, 3)), np.ones(2), np.asarray(1.0 + 0.1j), None, 1.0, 0.0) == ()
    with pytest.raises(ValueError, match="decoupling parameter must be real"):
        v(c, np.zeros((2, 3)), np.ones(2), np.asarray(1.0), np.asarray(1.0 + 1.0j), 1.0, 0.0)


def test_user_warnings_of_the_input_check():
    """Reference _biem.py:269-285: eta == 0 and (Im k < 0 or eta Re k < 0) warn (texts are API, run-together words included)."""
    import warnings

    w = _biem._warn_biem_inputs
    with pytest.warns(UserWarning, match="The solution may be incorrectif k is an eigenvalue for laplacianon the interior region withNeumann boundary condition."):
        w(np.asarray([1.0, 2.0]), np.asarray([1.0, 0.0]))
    with pytest.warns(UserWarning, match=r"The solution may be incorrectif not \(Im k >= 0 and eta Re k >= 0\)\."):
        w(np.asarray(1.0 - 0.1j), np.asarray(1.0))
    with pytest.warns(UserWarning, match="eta Re k >= 0"):
        w(np.asarray([1.0, 2.0]), np.asarray([1.0, -0.5]))
    with pytest.warns(UserWarning, match="eta Re k >= 0"):
        w(np.asarray(-1.0), None)                       # eta defaults to 1
    import torch

    with pytest.warns(UserWarning, match="eta Re k >= 0"):
        w(torch.tensor([1.0, 2.0]), torch.tensor([-1.0, 1.0]))
    # mixed containers (the reference warns after xp.asarray): list eta, torch k with NumPy eta, NumPy k with torch eta
    with pytest.warns(UserWarning, match="Neumann boundary condition"):
        w(np.asarray([1.0, 2.0]), [0.0, 1.0])
    with pytest.warns(UserWarning, match="Neumann boundary condition"):
        w(torch.tensor([1.0, 2.0]), np.asarray([1.0, 0.0]))
    with pytest.warns(UserWarning, match="eta Re k >= 0"):
        w(torch.tensor([1.0, 2.0]), [1.0, -2.0])
    with pytest.warns(UserWarning, match="eta Re k >= 0"):
        w(np.asarray([1.0, 2.0]), torch.tensor([-1.0, 1.0]))
    with pytest.warns(UserWarning, match="eta Re k >= 0"):
        w(torch.tensor([1.0 - 0.5j, 2.0]), np.asarray([1.0, 1.0]))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        w(torch.tensor([1.0, 2.0]), np.asarray([1.0, 3.0]))
        w(torch.tensor([1.0, 2.0]), [1.0, 3.0])
        w(np.asarray([0.5, 8.0]), None)
        w(np.asarray(1.0 + 0.3j), np.asarray(2.0))
        w(torch.tensor([0.5, 8.0]), torch.tensor([1.0, 1.0]))


def test_isa_check_of_the_built_library():
    """The LDS-DMA ring of k_gemm3m_pipe counts vector-memory instructions by hand: the build pins no scratch, no VGPR spills,
    96 MFMAs in the chunk loop and the reviewed set of vector-memory instructions (runs on the CPU: llvm-objdump on the .so)."""
    from biem_helmholtz_sphere_amd import _build

    _lib.load()
    assert not _build.is_stale() and _build.built_hash() == _build.source_hash()
    assert _lib.load().biem_build_id().decode() == _build.source_hash()
    rep = _build.check_isa()
    assert sorted(rep) == [64, 128, 192, 256]
    for kd, r in rep.items():
        assert r["scratch_bytes"] == 0 and r["vgpr_spills"] == 0 and r["mfma_in_chunk_loop"] == 96, (kd, r)
        assert set(r["vm"]) <= {"global_load_lds_dwordx4", "global_load_dword", "global_store_dwordx4"}, (kd, r)


def test_plane_wave_contract():
    u, g = amd.plane_wave(k=np.asarray(2.0), direction=np.asarray((0.0, 3.0, 0.0)))
    x = np.zeros((3, 4, 2))
    x[1] = 0.5
    assert np.allclose(u(x), np.exp(1j * 2.0 * 0.5)) and u(x).shape == (4, 2)
    assert g(x).shape == (3, 4, 2) and np.allclose(g(x)[1], 2j * np.exp(1j))
    with pytest.raises(ValueError, match="direction.ndim"):
        amd.plane_wave(k=np.ones(3), direction=np.asarray((1.0, 0.0, 0.0)))
    with pytest.raises(ValueError, match="not broadcastable"):
        amd.plane_wave(k=np.ones(3), direction=np.ones((3, 4)))


def test_memory_guard_matches_reference_formulas():
    # reference _biem.py:43-49 (d <= 3 returns an element count)
    assert amd.max_memory(c_ndim=3, n_end=20, n_balls=16) == 16**2 * 400**2
    assert amd.max_memory(c_ndim=2, n_end=64, n_balls=32) == 32**2 * 127**2
    assert amd.max_memory(c_ndim=4, n_end=3, n_balls=2) == 4 * (5 * 27) ** 2 * (11 * 216) * 16
    assert amd.max_n_end(c_ndim=3, memory_limit=16**2 * 400**2, n_balls=16) == 20


def test_coordinates_roundtrip_and_unsupported_tree():
    rng = np.random.default_rng(0)
    for t in ("a", "ba", "bba", "bpa", "bpbpa", "caa"):
        c = amd.create_from_branching_types(t)
        x = rng.normal(size=(c.c_ndim, 9))
        sph = c.from_cartesian(x)
        assert np.allclose(c.to_cartesian(sph, as_array=True), x)
        assert np.allclose(sph["r"], np.linalg.norm(x, axis=0))
    with pytest.raises(NotImplementedError, match="not built"):
        amd.create_from_branching_types("cba")
    assert _coords.n_end_from_harm("bba", 385) == 10 and _coords.n_end_from_harm("a", 127) == 64
    # committed bpa.svg / bpbpa.svg axis conventions
    sph = {"r": np.asarray(2.0), 0: np.asarray(0.3), 1: np.asarray(1.1)}
    x = amd.create_from_branching_types("bpa").to_cartesian(sph, as_array=True)
    assert np.allclose(x, [2 * np.cos(0.3) * np.sin(1.1), 2 * np.cos(0.3) * np.cos(1.1), 2 * np.sin(0.3)])
    sph = {"r": np.asarray(1.5), 0: np.asarray(0.2), 1: np.asarray(-0.4), 2: np.asarray(2.0)}
    x = amd.create_from_branching_types("bpbpa").to_cartesian(sph, as_array=True)
    c0, s0, c1, s1 = np.cos(0.2), np.sin(0.2), np.cos(-0.4), np.sin(-0.4)
    assert np.allclose(x, [1.5 * c0 * c1 * np.sin(2.0), 1.5 * c0 * s1, 1.5 * c0 * c1 * np.cos(2.0), 1.5 * s0])


def test_result_object_is_frozen():
    c = amd.create_from_branching_types("a")
    r = amd.BIEMResultCalculator(c=c, centers=np.zeros((2, 1)), radii=np.ones(1), k=np.asarray(1.0), n_end=3, eta=np.asarray(1.0), kind="outer")
    with pytest.raises(AttributeError):
        r.kind = "inner"
    assert r.matrix is None and r.density is None


def test_shard_bounds_cover_batch():
    for n in (0, 1, 7, 256, 257):
        for world in (1, 2, 3, 8):
            blocks = [_dist.shard_bounds(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
