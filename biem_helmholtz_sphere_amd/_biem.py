"""Host-side mirror of the reference's solver API for the biem() hot path, computing on MI355X.

Same names, argument meaning and error behaviour as reference ``src/biem_helmholtz_sphere/_biem.py``
(``biem`` :453-819, ``BIEMResultCalculator`` :196-237, ``biem_u`` :822-977, ``plane_wave`` :329-388,
``point_source`` :391-450, ``max_memory``/``max_n_end`` :23-74).  All arithmetic of the path runs in
hand-written HIP kernels behind the C ABI of ``include/biem_mi355.h`` (``_lib.py``); PyTorch only owns
device memory and streams.  There is no CPU path: without the HIP library or a GPU every call raises.

Arrays may be torch tensors (any device; results come back on the inputs' device) or NumPy arrays /
Python scalars (moved to the GPU, results returned as NumPy).  The incident field stays an opaque
callable (reference :536-547) and is evaluated in the caller's array namespace.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import warnings
from dataclasses import dataclass
from typing import Any, Callable, Literal, Optional, Protocol, Tuple, TypedDict

import numpy as np
import torch

from . import _lib as L
from ._coords import SphericalCoordinates, canonical_tree, harm_count, n_end_from_harm

try:  # numpy >= 1.25
    from numpy.exceptions import ComplexWarning
except ImportError:  # pragma: no cover
    from numpy import ComplexWarning  # type: ignore

# the reference promotes silent complex->real casts to errors module-wide (_biem.py:18)
warnings.filterwarnings("error", category=ComplexWarning)

Array = Any

__all__ = [
    "BIEMKwargs", "BIEMResultCalculator", "BIEMResultCalculatorProtocol", "UinCallable", "biem", "biem_u",
    "max_memory", "max_n_end", "plane_wave", "point_source",
]


# --------------------------------------------------------------------------------------
# memory guard (reference _biem.py:23-74, including its d<=3 element-count quirk)
# --------------------------------------------------------------------------------------
def _harm_n_ndim_le(n_end: int, c_ndim: int) -> int:
    """Number of harmonics of degree < n_end on S^{c_ndim-1} (``ush.harm_n_ndim_le``)."""
    if n_end <= 0:
        return 0
    if c_ndim == 2:
        return 2 * n_end - 1
    # dim of polynomials of degree <= n_end-1 restricted to harmonics: C(n+d-2, d-1) + C(n+d-3, d-1), n = n_end-1
    n = n_end - 1
    return math.comb(n + c_ndim - 1, c_ndim - 1) + math.comb(n + c_ndim - 2, c_ndim - 1)


_last_solve_stats: dict = {}   # how the last biem() call solved its systems (tests / bench)
_ws_memo: dict = {}            # device -> (shape key of the last solve, its workspace bytes): see biem()


def max_memory(*, c_ndim: int, n_end: int, n_balls: int) -> int:
    """Maximum memory usage in bytes (reference formula, :23-49)."""
    _COMPLEX128_SIZE = 16
    if c_ndim <= 3:
        return n_balls**2 * _harm_n_ndim_le(n_end, c_ndim) ** 2

    def inner(c_ndim: int, n_end: int) -> int:
        return (2 * n_end - 1) * n_end ** (c_ndim - 1)

    return n_balls**2 * inner(c_ndim, n_end) ** 2 * inner(c_ndim, 2 * n_end) * _COMPLEX128_SIZE


def max_n_end(*, c_ndim: int, memory_limit: int, n_balls: int) -> int:
    """Maximum n_end that fits in the given memory limit (:52-74)."""
    for i in range(1000):
        if max_memory(c_ndim=c_ndim, n_end=i, n_balls=n_balls) > memory_limit:
            break
    return i - 1


class BIEMKwargs(TypedDict, total=False):
    """The kwargs for the BIEM (reference :77-101)."""

    centers: Array
    radii: Array
    k: Array
    n_end: int
    eta: Array
    kind: Literal["inner", "outer"]
    force_matrix: bool


class UinCallable(Protocol):
    def __call__(self, x: Array, /, *, expand_x: bool = True) -> Array: ...


class BIEMResultCalculatorProtocol(Protocol):
    c: Any
    uin: Optional[UinCallable]
    centers: Array
    radii: Array
    k: Array
    n_end: int
    eta: Array
    kind: str
    density: Optional[Array]
    matrix: Optional[Array]

    def uscat(self, x: Array, /, far_field: bool = False, per_ball: bool = False, expand_x: bool = True) -> Array: ...


# --------------------------------------------------------------------------------------
# array plumbing
# --------------------------------------------------------------------------------------
@dataclass
class _Origin:
    """Where the caller's arrays live, so results can be handed back the same way."""

    kind: str                      # "numpy" | "torch"
    device: Any                    # torch.device of the inputs (torch) or None
    real_dtype: Any                # torch real dtype of `centers`

    @property
    def complex_dtype(self):
        return torch.complex64 if self.real_dtype == torch.float32 else torch.complex128

    def give(self, t: torch.Tensor, complex_out: bool = True) -> Array:
        if complex_out and t.dtype != self.complex_dtype:
            t = t.to(self.complex_dtype)
        if self.kind == "numpy":
            return t.cpu().numpy()
        return t.to(self.device)

    def user_array(self, t: torch.Tensor) -> Array:
        """A tensor of the compute device handed to a user callable in the caller's namespace."""
        t = t.to(self.real_dtype)
        if self.kind == "numpy":
            return t.cpu().numpy()
        return t.to(self.device)


def _compute_device(pref: Any = None) -> torch.device:
    if not torch.cuda.is_available():
        raise L.BiemLibraryError(
            "no HIP device visible: biem_helmholtz_sphere_amd computes on MI355X only and has no CPU fallback"
        )
    if isinstance(pref, torch.device) and pref.type == "cuda":
        return pref
    return torch.device("cuda", torch.cuda.current_device())


def _origin_of(*arrays: Any) -> Tuple[_Origin, torch.device]:
    tens = [a for a in arrays if isinstance(a, torch.Tensor)]
    if tens:
        dev = tens[0].device
        rd = tens[0].dtype if tens[0].dtype in (torch.float32, torch.float64) else torch.float64
        return _Origin("torch", dev, rd), _compute_device(dev)
    first = arrays[0] if arrays else None
    rd = torch.float32 if isinstance(first, np.ndarray) and first.dtype == np.float32 else torch.float64
    return _Origin("numpy", None, rd), _compute_device()


def _to_dev(a: Any, dev: torch.device, dtype: Any) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        return a.to(device=dev, dtype=dtype)
    if isinstance(a, (int, float, complex)):           # Python scalars: a fill kernel instead of a (synchronous, pageable) host copy
        return torch.full((), a, dtype=dtype, device=dev)
    return torch.as_tensor(np.asarray(a), device=dev).to(dtype)


def _is_complex(a: Any) -> bool:
    if isinstance(a, torch.Tensor):
        return a.is_complex()
    return np.iscomplexobj(np.asarray(a)) if not isinstance(a, (int, float)) else False


def _stream_ptr(dev: torch.device) -> int:
    return int(torch.cuda.current_stream(dev).cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else int(t.data_ptr())


# --------------------------------------------------------------------------------------
# plans (tables per (tree, n_end, device)), cached for the life of the process
# --------------------------------------------------------------------------------------
class _Plan:
    def __init__(self, tree: str, n_end: int, dev: torch.device):
        lib = L.load()
        if tree not in L.TREE_IDS:
            raise NotImplementedError(f"coordinate tree {tree!r} is not built (available: {sorted(L.TREE_IDS)})")
        self.tree, self.n_end, self.dev = tree, n_end, dev
        h = C.c_void_p()
        with torch.cuda.device(dev):
            L.check(lib.biem_plan_create(L.TREE_IDS[tree], n_end, C.byref(h)), "biem_plan_create")
        self.handle = h
        d, H, Q, H2, nt = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_longlong()
        L.check(lib.biem_plan_info(h, C.byref(d), C.byref(H), C.byref(Q), C.byref(H2), C.byref(nt)))
        self.d, self.H, self.Q, self.H2, self.n_terms = d.value, H.value, Q.value, H2.value, nt.value
        y = np.zeros((self.Q, self.d))
        w = np.zeros(self.Q)
        L.check(lib.biem_plan_quadrature(h, y.ctypes.data, w.ctypes.data))
        self.quad_y = torch.as_tensor(y, device=dev)          # [Q, d] unit vectors
        self.y_by_axes: dict = {}                              # the same, [d, ...(quadrature axes)] per axis order of the caller's tree
        lab = np.zeros((self.H, 3), dtype=np.int32)
        deg = np.zeros(self.H, dtype=np.int32)
        L.check(lib.biem_plan_labels(h, lab.ctypes.data, deg.ctypes.data))
        self.labels, self.degrees = lab, deg

    def quad_shape(self) -> Tuple[int, ...]:
        """Tensor-product shape of the rule, one axis per spherical node (the reference's ...(f) axes)."""
        n = self.n_end
        return {"a": (2 * n,), "ba": (n, 2 * n), "bba": (n, n, 2 * n), "caa": (n, 2 * n, 2 * n)}[self.tree]


_PLANS: dict = {}


def _plan(tree: str, n_end: int, dev: torch.device) -> _Plan:
    key = (tree, int(n_end), dev.index if dev.index is not None else torch.cuda.current_device())
    p = _PLANS.get(key)
    if p is None:
        p = _PLANS[key] = _Plan(tree, int(n_end), torch.device("cuda", key[2]))
    return p


# --------------------------------------------------------------------------------------
# incident fields (reference :329-450)
# --------------------------------------------------------------------------------------
def _bshape_ok(a: Tuple[int, ...], b: Tuple[int, ...]) -> bool:
    try:
        np.broadcast_shapes(tuple(a), tuple(b))
        return True
    except ValueError:
        return False


def _like(v: Any, x: Any) -> Any:
    """v (array or scalar of the creator's namespace) as an array usable with x."""
    if isinstance(x, torch.Tensor):
        if isinstance(v, torch.Tensor):
            return v.to(x.device)
        return torch.as_tensor(np.asarray(v), device=x.device)
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy()
    return np.asarray(v)


def plane_wave(*, k: Array, direction: Array) -> Tuple[Callable[[Array], Array], Callable[[Array], Array]]:
    r"""Plane wave :math:`u(x) = e^{i k d\cdot x}`, d = direction/||direction|| (reference :329-388).

    k has shape (...), direction (c_ndim, ...).  Returns (u, grad u); given x of shape (c_ndim, ...(any), ...)
    they return (...(any), ...) and (c_ndim, ...(any), ...).
    """
    k_ = k if isinstance(k, torch.Tensor) else np.asarray(k)
    d_ = direction if isinstance(direction, torch.Tensor) else np.asarray(direction)
    if not isinstance(d_, torch.Tensor) and d_.dtype.kind in "iu":
        d_ = d_.astype(np.float64)
    if not _bshape_ok(tuple(k_.shape), tuple(d_.shape[1:])):
        raise ValueError(
            "Shapes of k and direction[1:] are not broadcastable\n"
            f"tuple(k.shape)={tuple(k_.shape)}\ntuple(direction.shape)={tuple(d_.shape)}"
        )
    if d_.ndim != k_.ndim + 1:
        raise ValueError(f"direction.ndim={d_.ndim} is not k.ndim + 1={k_.ndim + 1}")
    if isinstance(d_, torch.Tensor):
        d_ = d_ / torch.linalg.vector_norm(d_, dim=0, keepdim=True)
    else:
        d_ = d_ / np.linalg.norm(d_, axis=0, keepdims=True)

    def _parts(x):
        dd, kk = _like(d_, x), _like(k_, x)
        dd = dd[(slice(None),) + (None,) * (x.ndim - dd.ndim)]
        if isinstance(x, torch.Tensor):
            dd = dd.to(x.dtype) if not dd.is_complex() else dd
            ip = torch.sum(dd * x, dim=0)
            return dd, kk, ip, torch.exp(1j * kk * ip)
        ip = np.sum(dd * x, axis=0)
        return dd, kk, ip, np.exp(1j * kk * ip)

    def inner(x: Array, /) -> Array:
        return _parts(x)[3]

    def inner_grad(x: Array, /) -> Array:
        dd, kk, _, e = _parts(x)
        return 1j * kk * dd * e[None, ...]

    return inner, inner_grad


def point_source(*, k: Array, source: Array, n: int) -> Tuple[Callable[[Array], Array], Callable[[Array], Array]]:
    r"""Point source :math:`u(x) = h^{(1)}_n(k\|x - source\|)` with the d-dimensional h_n (reference :391-450).

    The radial functions are evaluated by the HIP kernel behind ``biem_radial_complex`` (real or complex k).
    """
    k_ = k if isinstance(k, torch.Tensor) else np.asarray(k)
    s_ = source if isinstance(source, torch.Tensor) else np.asarray(source, dtype=np.float64)
    if not _bshape_ok(tuple(k_.shape), tuple(s_.shape[1:])):
        raise ValueError(f"Shapes of k and source[1:] are not broadcastable\n{tuple(k_.shape)=}\n{tuple(s_.shape)=}")
    if s_.ndim != k_.ndim + 1:
        raise ValueError(f"source.ndim={s_.ndim} is not k.ndim + 1={k_.ndim + 1}")
    n = int(n)

    def _radial(d: int, z: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """h_n(z), h_n'(z) for a (real or complex) tensor z on a cuda device."""
        lib = L.load()
        zz = z.to(torch.complex128).contiguous().reshape(-1)
        out = torch.empty((zz.numel(), 2, n + 2), dtype=torch.complex128, device=zz.device)
        with torch.cuda.device(zz.device):
            L.check(lib.biem_radial_complex(d, n + 1, zz.numel(), _ptr(zz), _ptr(out), _stream_ptr(zz.device)), "biem_radial_complex")
        h, h1 = out[:, 1, n], out[:, 1, n + 1]
        hp = n / zz * h - h1
        return h.reshape(z.shape), hp.reshape(z.shape)

    def _prep(x):
        was_np = not isinstance(x, torch.Tensor)
        dev = _compute_device(None if was_np else x.device)
        xt = _to_dev(x, dev, torch.float64)
        st = _to_dev(s_, dev, torch.float64)
        kt = _to_dev(k_, dev, torch.complex128 if _is_complex(k_) else torch.float64)
        rel = xt - st[(slice(None),) + (None,) * (xt.ndim - st.ndim)]
        r = torch.linalg.vector_norm(rel, dim=0)
        return was_np, (None if was_np else x.device), rel, r, kt

    def _back(t, was_np, dev):
        return t.cpu().numpy() if was_np else t.to(dev)

    def inner(x: Array, /) -> Array:
        was_np, dev, rel, r, kt = _prep(x)
        h, _ = _radial(int(rel.shape[0]), (kt * r).expand(r.shape) if kt.ndim else kt * r)
        return _back(h, was_np, dev)

    def inner_grad(x: Array, /) -> Array:
        was_np, dev, rel, r, kt = _prep(x)
        _, hp = _radial(int(rel.shape[0]), kt * r)
        coeff = kt * hp / r
        return _back(coeff[None, ...] * rel, was_np, dev)

    return inner, inner_grad


# --------------------------------------------------------------------------------------
# result object (reference :196-237)
# --------------------------------------------------------------------------------------
class BIEMResultCalculator:
    """Callable that computes the BIEMResult at the given cartesian coordinates.

    Field-for-field the reference's frozen kw-only record (:196-237).  ``centers`` is stored as ``[d, ..., B]``
    exactly like the reference does (:588,:810).  ``matrix`` (reference scaling, shape
    ``(..., B, harm, B, harm)``, up to 655 MB per system) is assembled on the GPU on first access instead of being
    retained by default.
    """

    __slots__ = ("c", "uin", "centers", "radii", "k", "n_end", "eta", "kind", "density", "_matrix")

    def __init__(self, *, c, centers, radii, k, n_end, eta, kind, uin=None, density=None, matrix=None):
        for name, val in (("c", c), ("uin", uin), ("centers", centers), ("radii", radii), ("k", k), ("n_end", n_end),
                          ("eta", eta), ("kind", kind), ("density", density), ("_matrix", matrix)):
            object.__setattr__(self, name, val)

    def __setattr__(self, name, value):   # frozen, like attrs.frozen
        raise AttributeError(f"BIEMResultCalculator is frozen: cannot assign to field {name!r}")

    def __delattr__(self, name):
        raise AttributeError(f"BIEMResultCalculator is frozen: cannot delete field {name!r}")

    def __repr__(self) -> str:
        return (f"BIEMResultCalculator(c={self.c!r}, n_end={self.n_end}, kind={self.kind!r}, "
                f"density={'None' if self.density is None else tuple(self.density.shape)})")

    @property
    def matrix(self) -> Optional[Array]:
        """The flattened matrix of the BIEM of shape (..., B, harm, B', harm'); None when the shortcut path ran."""
        m = self._matrix
        if callable(m):
            m = m()
            object.__setattr__(self, "_matrix", m)
        return m

    def uscat(self, x: Array, /, far_field: bool = False, per_ball: bool = False, expand_x: bool = True) -> Array:
        return biem_u(self, x, far_field=far_field, per_ball=per_ball, expand_x=expand_x)


# --------------------------------------------------------------------------------------
# input checking (reference :240-326; error and warning texts are part of the API)
# --------------------------------------------------------------------------------------
def _shape(a: Any) -> Tuple[int, ...]:
    return tuple(a.shape) if isinstance(a, (torch.Tensor, np.ndarray)) else tuple(np.shape(a))


def _validate_biem_inputs(c, centers, radii, k, eta, alpha, beta) -> Tuple[int, ...]:
    """Shape / dtype checks of reference :240-326 on metadata only (no device needed); returns the batch shape."""
    for nm, a in (("centers", centers), ("radii", radii), ("k", k)):
        if not isinstance(a, (torch.Tensor, np.ndarray)):
            raise TypeError(f"{nm} must be an array (torch.Tensor or numpy.ndarray), got {type(a).__name__}")
    if eta is not None and _is_complex(eta):
        raise ValueError("The decoupling parameter must be real.")
    ks, cs, rs = _shape(k), _shape(centers), _shape(radii)
    es = (1,) * len(ks) if eta is None else _shape(eta)
    als = _shape(alpha) or (1,) * (len(ks) + 1)
    bes = _shape(beta) or (1,) * (len(ks) + 1)
    if len({len(ks), len(es), len(cs) - 2, len(rs) - 1}) != 1:
        raise ValueError(
            f"k.ndim={len(ks)}, eta.ndim={len(es)}, centers.ndim - 2={len(cs) - 2}, radii.ndim -1={len(rs) - 1}are not the same."
        )
    if len(als) != len(ks) + 1 or len(bes) != len(ks) + 1:
        raise ValueError(f"alpha and beta must be scalars or arrays of shape (..., B) with {len(ks) + 1} axes")
    try:
        batch = np.broadcast_shapes(ks, es, cs[:-2], rs[:-1], als[:-1], bes[:-1])
    except ValueError as e:
        raise ValueError(
            "Shapes of k, eta and "
            "centers.shape[:-2], radii.shape[:-1] "
            "are not broadcastable\n"
            f"tuple(k.shape)={ks}\ntuple(eta.shape)={es}\ntuple(centers.shape)={cs}\ntuple(radii.shape)={rs}\n"
            f"tuple(alpha.shape)={als}\ntuple(beta.shape)={bes}"
        ) from e
    try:
        np.broadcast_shapes(cs[:-1], rs, als, bes)
    except ValueError as e:
        raise ValueError(
            "centers.shape[:-1] and radii.shape "
            "are not broadcastable\n"
            f"tuple(centers.shape)={cs}\ntuple(radii.shape)={rs}\ntuple(alpha.shape)={als}\ntuple(beta.shape)={bes}"
        ) from e
    if cs[-1] != c.c_ndim:
        raise ValueError(f"The last dimension of centers must be c.c_ndim={c.c_ndim}, but got {cs[-1]}")
    return tuple(batch)


def _host_array(a: Any) -> np.ndarray:
    """Any accepted input (torch tensor on any device, NumPy array, list, scalar) as a NumPy array on the host."""
    if isinstance(a, torch.Tensor):
        return a.detach().cpu().numpy()
    return np.asarray(a)


def _warn_biem_inputs(k: Any, eta: Any) -> None:
    """The two UserWarnings of reference :269-285 (texts verbatim, missing blanks included): eta == 0 somewhere; Im k < 0 or
    eta Re k < 0 somewhere.  Like the reference, which warns after ``xp.asarray(eta)``, the test runs on CONVERTED arrays, so
    k and eta may mix torch tensors (any device), NumPy arrays, lists and scalars.  (The reference's own test of the second
    one, :278-280, guards the Im k term with "eta is not castable to float64", which is never true once the complex-eta check
    above it has passed, so there only eta Re k < 0 can fire; here the condition the message states is checked.)"""
    if isinstance(k, torch.Tensor) and isinstance(eta, torch.Tensor) and k.is_cuda and eta.is_cuda and k.device == eta.device:
        # both on one GPU: ONE device -> host copy (each copy is a synchronisation; a call with one small system pays for every one)
        nk = k.numel()
        buf = torch.cat([k.detach().reshape(-1).to(torch.complex128), eta.detach().reshape(-1).to(torch.complex128)]).cpu().numpy()
        k_h = buf[:nk].reshape(tuple(k.shape))
        if not k.is_complex():
            k_h = k_h.real
        eta_h = buf[nk:].real.reshape(tuple(eta.shape))
    else:
        k_h = _host_array(k)
        eta_h = None if eta is None else _host_array(eta)
    if eta_h is not None and bool(np.any(eta_h == 0)):
        warnings.warn(
            "The solution may be incorrect"
            "if k is an eigenvalue for laplacian"
            "on the interior region with"
            "Neumann boundary condition.",
            UserWarning,
            stacklevel=4,
        )
    k_re = k_h.real
    bad = (np.iscomplexobj(k_h) and bool(np.any(k_h.imag < 0))) or bool(np.any((k_re if eta_h is None else eta_h * k_re) < 0))
    if bad:
        warnings.warn("The solution may be incorrectif not (Im k >= 0 and eta Re k >= 0).", UserWarning, stacklevel=4)


def _check_biem_inputs(c, centers, radii, k, eta, alpha, beta):
    batch = _validate_biem_inputs(c, centers, radii, k, eta, alpha, beta)
    _warn_biem_inputs(k, eta)
    origin, dev = _origin_of(centers, radii, k, eta, alpha, beta)
    f64 = torch.float64
    centers_t = _to_dev(centers, dev, f64)
    radii_t = _to_dev(radii, dev, f64)
    k_t = _to_dev(k, dev, torch.complex128)     # the kernels take complex wavenumbers (Im k = 0: real special functions)
    if eta is None:
        eta_t = torch.ones((1,) * k_t.ndim, dtype=f64, device=dev)
    else:
        eta_t = _to_dev(eta, dev, f64)
    alpha_t = _to_dev(alpha, dev, torch.complex128)
    if alpha_t.ndim == 0:
        alpha_t = alpha_t[(None,) * (k_t.ndim + 1)]
    beta_t = _to_dev(beta, dev, torch.complex128)
    if beta_t.ndim == 0:
        beta_t = beta_t[(None,) * (k_t.ndim + 1)]
    return origin, dev, batch, centers_t, radii_t, k_t, eta_t, alpha_t, beta_t


# --------------------------------------------------------------------------------------
# the solver (reference :453-819)
# --------------------------------------------------------------------------------------
@dataclass
class _Flat:
    """Flattened, contiguous device operands of one biem() call."""

    nb: int
    B: int
    k: torch.Tensor          # [nb]
    eta: torch.Tensor        # [nb]
    centers: torch.Tensor    # [nb or 1, B, d]
    radii: torch.Tensor      # [nb or 1, B]
    geom_batched: int
    alpha: torch.Tensor      # [nb or 1, B] complex128
    beta: torch.Tensor
    ab_batched: int


def _flatten(batch, B, centers_t, radii_t, k_t, eta_t, alpha_t, beta_t) -> _Flat:
    nb = int(np.prod(batch)) if len(batch) else 1
    d = centers_t.shape[-1]
    kf = k_t.expand(batch).reshape(nb).contiguous()
    ef = eta_t.expand(batch).reshape(nb).contiguous()
    geom_b = any(s != 1 for s in tuple(centers_t.shape[:-2]) + tuple(radii_t.shape[:-1]))
    if geom_b:
        cf = centers_t.expand(tuple(batch) + (B, d)).reshape(nb, B, d).contiguous()
        rf = radii_t.expand(tuple(batch) + (B,)).reshape(nb, B).contiguous()
    else:
        cf = centers_t.reshape(1, B, d).contiguous()
        rf = radii_t.expand(radii_t.shape[:-1] + (B,)).reshape(1, B).contiguous()
    ab_b = any(s != 1 for s in tuple(alpha_t.shape[:-1]) + tuple(beta_t.shape[:-1]))
    if ab_b:
        af = alpha_t.expand(tuple(batch) + (B,)).reshape(nb, B).contiguous()
        bf = beta_t.expand(tuple(batch) + (B,)).reshape(nb, B).contiguous()
    else:
        af = alpha_t.reshape(-1)[-alpha_t.shape[-1]:].expand(B).reshape(1, B).contiguous()
        bf = beta_t.reshape(-1)[-beta_t.shape[-1]:].expand(B).reshape(1, B).contiguous()
    return _Flat(nb, B, kf, ef, cf, rf, int(geom_b), af, bf, int(ab_b))


def _boundary_samples(plan: _Plan, origin: _Origin, fl: _Flat, batch, uin, uin_grad, perm) -> torch.Tensor:
    """g[nb, B, Q] = (-alpha u_in - beta d_n u_in)(c_b + rho_b y_q): the closure `f` of reference :611-624.

    fl.centers are in the plan's canonical axes; the user's callables see ORIGINAL axes (x_orig[perm[i]] = x_canon[i])."""
    dev, d, Q, B, nb = plan.dev, plan.d, plan.Q, fl.B, fl.nb
    qshape = plan.quad_shape()
    nbt = len(batch)
    inv = [0] * d
    for i, pi_ in enumerate(perm):
        inv[pi_] = i
    ident = inv == list(range(d))                                      # (no primed nodes: no gather kernels for the axis order)
    ykey = tuple(inv)
    y = plan.y_by_axes.get(ykey)
    if y is None:
        y = plan.y_by_axes[ykey] = (plan.quad_y.T if ident else plan.quad_y.T[inv]).reshape((d,) + qshape).contiguous()   # (d, ...(f)), original axes
    x_rel = y[(...,) + (None,) * (nbt + 1)]                            # (d, ...(f), 1.., 1)
    cen = (fl.centers if ident else fl.centers[..., inv]).reshape(((nb,) if fl.geom_batched else (1,)) + (B, d))
    rad = fl.radii.reshape(((nb,) if fl.geom_batched else (1,)) + (B,))
    if fl.geom_batched:
        cen = cen.reshape(tuple(batch) + (B, d))
        rad = rad.reshape(tuple(batch) + (B,))
    else:
        cen = cen.reshape((1,) * nbt + (B, d))
        rad = rad.reshape((1,) * nbt + (B,))
    # (d, ...(f), ...batch, B)
    cen_m = torch.movedim(cen, -1, 0)[(slice(None),) + (None,) * len(qshape)]
    x = rad[(None,) * (1 + len(qshape))] * x_rel + cen_m
    # (d, ...(f), B, ...batch): the layout the reference hands to uin (:620-621)
    x = torch.movedim(x, -1, 1 + len(qshape))
    x = x.expand((d,) + qshape + (B,) + tuple(batch))
    xu = origin.user_array(x)
    # alpha/beta along (B, ...batch)
    def ab(t):
        t = t.reshape(((nb,) if fl.ab_batched else (1,)) + (B,))
        t = t.reshape((tuple(batch) if fl.ab_batched else (1,) * nbt) + (B,))
        return torch.movedim(t, -1, 0)[(None,) * len(qshape)]       # (1.., B, ...batch)
    g = None
    if uin is not None:
        u = _to_dev(uin(xu), dev, torch.complex128)
        g = (ab(fl.alpha) * u).neg_()                                  # (a fresh tensor: negated in place)
    if uin_grad is not None:
        gu = _to_dev(uin_grad(xu), dev, torch.complex128)
        t = ab(fl.beta) * torch.sum(gu * x_rel.to(torch.complex128), dim=0)
        g = t.neg_() if g is None else g - t
    lead = qshape + (B,) + tuple(batch)
    if g is None:
        g = torch.zeros(lead, dtype=torch.complex128, device=dev)
    else:
        tgt = tuple(torch.broadcast_shapes(tuple(g.shape), lead))       # (a callable that returned fewer / shorter axes than it was given)
        if tuple(g.shape) != tgt:
            g = g.expand(tgt)
    # The incident field may vary along batch axes on which the operator (k, eta, geometry, alpha, beta) has size 1
    # (e.g. many incidence directions for one wavenumber): those axes become right-hand sides of ONE factorisation.
    # The reference broadcasts the matrix over them in btensorsolve (_biem.py:797), i.e. factors it again per incidence.
    nq = len(qshape)
    full = tuple(g.shape[nq + 1:])
    if len(full) != nbt:
        raise ValueError(f"uin/uin_grad returned an array with batch shape {full}, expected {nbt} batch axes like k")
    op_axes = [i for i in range(nbt) if batch[i] == full[i]]
    rhs_axes = [i for i in range(nbt) if batch[i] != full[i]]
    nrhs = int(np.prod([full[i] for i in rhs_axes])) if rhs_axes else 1
    g = g.reshape((Q, B) + full)
    g = g.permute([2 + i for i in op_axes] + [2 + i for i in rhs_axes] + [1, 0])      # (*op, *rhs, B, Q)
    return g.reshape(nb, nrhs, B, Q).contiguous(), full, op_axes, rhs_axes


def _restore_batch(t: torch.Tensor, full, op_axes, rhs_axes) -> torch.Tensor:
    """[nb, nrhs, ...] in (op axes, rhs axes) order -> (*full, ...) in the caller's axis order."""
    tail = tuple(t.shape[2:])
    t = t.reshape(tuple(full[i] for i in op_axes) + tuple(full[i] for i in rhs_axes) + tail)
    order = op_axes + rhs_axes
    inv = [order.index(i) for i in range(len(full))]
    return t.permute(inv + [len(full) + j for j in range(len(tail))])


def biem(
    c: Any,
    /,
    *,
    centers: Array,
    radii: Array,
    k: Array,
    n_end: int,
    alpha: Array | complex = 1.0,
    beta: Array | complex = 0.0,
    uin: Callable[[Array], Array] | None = None,
    uin_grad: Callable[[Array], Array] | None = None,
    eta: Array | None = None,
    kind: Literal["inner", "outer"] = "outer",
    force_matrix: bool = False,
    translational_coefficients_method: Literal["gumerov", "plane_wave", "triplet"] | None = None,
    chunk: int = 0,
) -> BIEMResultCalculator:
    r"""Boundary Integral Equation Method (BIEM) for the Helmholtz equation on MI355X.

    Same contract as the reference ``biem`` (``_biem.py:453-581``): solves, per leading batch element,

        A_{b,n,p,b',n',p'} = blc_{n'}(rho_{b'}, eta) * { delta (alpha h_n + beta k h_n')(k rho_b)            b = b'
                                                        (S|R)_{n'p',np}(c_b - c_b') (alpha j_n + beta k j_n')(k rho_b)  b != b'
        sum A phi = f,   f_{b,n,p} = sum_q w_q (-alpha u_in - beta d_n u_in)(c_b + rho_b y_q) conj(Y_{n,p}(y_q))

    ``translational_coefficients_method`` is accepted for signature compatibility; this build always uses the
    exact closed form of SURVEY A.5 (the reference's "triplet" implementation is itself inexact).
    ``chunk`` (extension) bounds how many system matrices are resident at once (0 = choose).

    Solver: the reference passes every system to a general dense solve (``_biem.py:797``).  Here the system is first brought
    to its complex-symmetric form (real harmonics, symmetric scaling) and factored as U^T U (Cholesky-type, no conjugation)
    without interchanges - half the flops; a system in which a multiplier would exceed 100, or whose factor grew by more than
    200, is solved by the pivoted LU instead (``BIEM_SOLVER=lu`` in the environment: pivoted LU for all).  Both give the
    reference's ``density`` to rounding.
    """
    if translational_coefficients_method not in (None, "gumerov", "plane_wave", "triplet"):
        raise ValueError(f"Invalid translational_coefficients_method: {translational_coefficients_method}")
    origin, dev, batch, centers_t, radii_t, k_t, eta_t, alpha_t, beta_t = _check_biem_inputs(c, centers, radii, k, eta, alpha, beta)
    # trees with primed nodes run as their canonical tree in permuted axes (canonical component i = original perm[i])
    tree, perm = canonical_tree(c.branching_types_expression_str)
    lib = L.load()
    B = int(radii_t.shape[-1])
    ndim_first = k_t.ndim
    plan = _plan(tree, n_end, dev)
    H, Q = plan.H, plan.Q
    fl = _flatten(batch, B, centers_t if list(perm) == list(range(len(perm))) else centers_t[..., list(perm)], radii_t, k_t, eta_t, alpha_t, beta_t)
    nb = fl.nb
    sp = _stream_ptr(dev)

    has_rhs = not (uin is None and uin_grad is None)
    g = None
    full, op_axes, rhs_axes, nrhs = tuple(batch), list(range(len(batch))), [], 1
    # A repeated call of the same shape takes its workspace FIRST, before the boundary samples: the block the previous call
    # returned to torch's caching allocator is then still whole.  Taken after them, one of their mid-size temporaries may have
    # been carved out of it - and a second block of that size need not exist (cfg 3's whole batch: 164 of 288 GB).
    ws_key = (tree, int(n_end), int(nb), int(B), int(chunk), os.environ.get("BIEM_MAX_RESIDENT_BYTES"))
    work_pre = None
    if has_rhs and nb > 0 and (B > 1 or force_matrix):
        memo = _ws_memo.get(dev)
        if memo is not None and memo[0] == ws_key:
            try:
                with torch.cuda.device(dev):
                    work_pre = torch.empty(memo[1], dtype=torch.uint8, device=dev)
            except torch.OutOfMemoryError:
                work_pre = None
    if has_rhs:
        # (the all-zero tests are only needed when the matching callable is missing; Python scalars are decided on the host)
        def _all_zero(v, v_t):
            return (v == 0) if isinstance(v, (int, float, complex)) else bool(torch.all(v_t == 0))
        if uin is None and not _all_zero(alpha, alpha_t):
            raise ValueError("alpha is not zero, but uin is None. uin must be provided to compute the boundary condition.")
        if uin_grad is None and not _all_zero(beta, beta_t):
            raise ValueError("beta is not zero, but uin_grad is None. uin_grad must be provided to compute the boundary condition.")
        g, full, op_axes, rhs_axes = _boundary_samples(plan, origin, fl, batch, uin, uin_grad, perm)
        nrhs = int(g.shape[1])

    use_matrix = (not has_rhs) or B > 1 or force_matrix          # reference :643-645
    density_t = None
    with torch.cuda.device(dev):
        if nb == 0:
            # an empty batch axis: nothing to solve, results of the right (empty) shape
            density_t = torch.empty((0, nrhs, B, H), dtype=torch.complex128, device=dev) if has_rhs else None
        elif not use_matrix:
            # single ball: density = f / (blc (alpha h + beta k h'))    (reference :648-691)
            tab = torch.empty((nb, B, 3, n_end), dtype=torch.complex128, device=dev)
            L.check(lib.biem_ball_tables(plan.handle, nb, B, _ptr(fl.k), _ptr(fl.eta), _ptr(fl.radii), fl.geom_batched,
                                         _ptr(fl.alpha), _ptr(fl.beta), fl.ab_batched, _ptr(tab), sp), "biem_ball_tables")
            f = torch.empty((nb, nrhs, B * H), dtype=torch.complex128, device=dev)
            L.check(lib.biem_rhs_project(plan.handle, nb, B, nrhs, _ptr(g), _ptr(f), nrhs * B * H, 1, B * H, sp), "biem_rhs_project")
            density_t = torch.empty((nb, nrhs, B, H), dtype=torch.complex128, device=dev)
            L.check(lib.biem_density(plan.handle, nb, B, nrhs, _ptr(f), nrhs * B * H, 1, B * H, _ptr(tab), _ptr(density_t), sp),
                    "biem_density")
        elif has_rhs:
            chunk = int(chunk)
            if chunk <= 0:
                # resident matrices per pass: as many as fit 85 % of the memory this process can still get (free + cached by
                # torch's allocator).  The per-system kernels of the LU (panel strips, diagonal-block inverses, back
                # substitution) are latency-bound on ONE CU per system, so they cost the same for 32 or 256 systems.
                per = max(1, int(lib.biem_solve_workspace_bytes(plan.handle, 1, B, nrhs, 1)))
                if nb * per <= (1 << 30) and not os.environ.get("BIEM_MAX_RESIDENT_BYTES"):
                    chunk = nb                     # small jobs: everything resident, no memory query (it costs more than the solve)
                else:
                    free, _total = torch.cuda.mem_get_info(dev)
                    avail = free + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
                    if work_pre is not None:
                        avail += int(work_pre.numel())     # (the block taken in advance above is this call's own)
                    budget = int(0.85 * avail)
                    cap = os.environ.get("BIEM_MAX_RESIDENT_BYTES")      # a drop-in inside a larger torch program: bound the workspace
                    if cap:
                        budget = min(budget, max(int(float(cap)), per))
                    chunk = max(1, min(nb, 32768, budget // per))      # 32768: grid dimension of the per-system kernels
            wbytes = int(lib.biem_solve_workspace_bytes(plan.handle, nb, B, nrhs, chunk))
            if work_pre is not None and work_pre.numel() == wbytes:
                work = work_pre
            else:
                work_pre = None                    # (another size after all: back to the allocator before the right one is taken)
                try:
                    work = torch.empty(wbytes, dtype=torch.uint8, device=dev)
                except torch.OutOfMemoryError:
                    torch.cuda.empty_cache()       # cached blocks of other sizes (an earlier, different job): released, one more attempt
                    work = torch.empty(wbytes, dtype=torch.uint8, device=dev)
            _ws_memo[dev] = (ws_key, wbytes)
            density_t = torch.empty((nb, nrhs, B, H), dtype=torch.complex128, device=dev)
            info = torch.zeros(nb, dtype=torch.int32, device=dev)
            # The equilibrated system is complex symmetric in a real-harmonic basis (include/biem_mi355.h, biem_solve_ldlt):
            # U^T U factorisation without interchanges, half the flops of the LU.  Systems whose diagonal pivots were rejected
            # (info < 0: close to a resonance of a sphere, or strongly coupled spheres) are solved again with the pivoted LU,
            # which is what the reference's linalg.solve does for every system (_biem.py:797).  BIEM_SOLVER=lu: LU only.
            solver = os.environ.get("BIEM_SOLVER", "ldlt")
            if solver not in ("ldlt", "lu"):
                raise ValueError(f"BIEM_SOLVER must be 'ldlt' or 'lu', got {solver!r}")
            entry = lib.biem_solve_ldlt if solver == "ldlt" else lib.biem_solve
            L.check(entry(plan.handle, nb, B, nrhs, _ptr(fl.k), _ptr(fl.eta), _ptr(fl.centers), _ptr(fl.radii), fl.geom_batched,
                          _ptr(fl.alpha), _ptr(fl.beta), fl.ab_batched, _ptr(g), _ptr(density_t), _ptr(info), chunk,
                          _ptr(work), wbytes, sp), "biem_solve")
            if solver == "ldlt":
                # (the codes come to the host in ONE copy and the rejected systems are picked there: torch.nonzero on the device is five
                # launches and a synchronisation of its own, which one system per call pays in full)
                info_h = info.cpu()
                redo_h = torch.nonzero(info_h < 0).flatten()
                redo = redo_h.to(dev) if redo_h.numel() > 0 else redo_h
                if redo.numel() > 0:
                    nr = int(redo.numel())
                    pick = lambda t, batched: t[redo].contiguous() if batched else t
                    k_r, eta_r, g_r = fl.k[redo].contiguous(), fl.eta[redo].contiguous(), g[redo].contiguous()
                    cen_r, rad_r = pick(fl.centers, fl.geom_batched), pick(fl.radii, fl.geom_batched)
                    al_r, be_r = pick(fl.alpha, fl.ab_batched), pick(fl.beta, fl.ab_batched)
                    dens_r = torch.empty((nr, nrhs, B, H), dtype=torch.complex128, device=dev)
                    info_r = torch.zeros(nr, dtype=torch.int32, device=dev)
                    L.check(lib.biem_solve(plan.handle, nr, B, nrhs, _ptr(k_r), _ptr(eta_r), _ptr(cen_r), _ptr(rad_r), fl.geom_batched,
                                           _ptr(al_r), _ptr(be_r), fl.ab_batched, _ptr(g_r), _ptr(dens_r), _ptr(info_r), min(chunk, nr),
                                           _ptr(work), wbytes, sp), "biem_solve")
                    density_t[redo] = dens_r
            _last_solve_stats["ldlt_systems"] = nb if solver == "ldlt" else 0
            _last_solve_stats["lu_systems"] = (int(redo.numel()) if solver == "ldlt" else nb)
            # why the symmetric path handed systems over (diagnostics): info = -(first row of the rejecting 64-row panel + 1), or
            # -(n_pad + 1) for the growth check
            _last_solve_stats.pop("rejected_info", None)
            if solver == "ldlt" and redo.numel() > 0:
                _last_solve_stats["rejected_info"] = info_h[redo_h].tolist()[:64]
            del work

    def make_matrix():
        with torch.cuda.device(dev):
            N = B * H
            tab = torch.empty((nb, B, 3, n_end), dtype=torch.complex128, device=dev)
            L.check(lib.biem_ball_tables(plan.handle, nb, B, _ptr(fl.k), _ptr(fl.eta), _ptr(fl.radii), fl.geom_batched,
                                         _ptr(fl.alpha), _ptr(fl.beta), fl.ab_batched, _ptr(tab), _stream_ptr(dev)), "biem_ball_tables")
            wb = int(lib.biem_fill_workspace_bytes(plan.handle, nb, B))
            work = torch.empty(max(wb, 16), dtype=torch.uint8, device=dev)
            A = torch.empty((nb, N, N), dtype=torch.complex128, device=dev)
            L.check(lib.biem_fill(plan.handle, nb, B, _ptr(fl.k), _ptr(fl.centers), fl.geom_batched, _ptr(tab), L.FILL_REFERENCE,
                                  _ptr(A), N, N * N, N, _ptr(work), wb, _stream_ptr(dev)), "biem_fill")
            return origin.give(A.reshape(tuple(batch) + (B, H, B, H)))

    matrix = make_matrix if use_matrix else None
    density = None if density_t is None else origin.give(_restore_batch(density_t, full, op_axes, rhs_axes).contiguous())

    if uin is None:
        uin_wrapped = None
    else:
        def uin_wrapped(x: Array, /, *, expand_x: bool = True) -> Array:   # reference :803-806
            if expand_x:
                x = x[(...,) + (None,) * ndim_first]
            return uin(x)

    real_out = lambda t: origin.give(t.to(origin.real_dtype), complex_out=False)
    return BIEMResultCalculator(
        c=c,
        centers=real_out(torch.movedim(centers_t, -1, 0)),       # [..., B, v] -> [v, ..., B]  (reference :588)
        radii=real_out(radii_t),
        k=(origin.give(k_t) if _is_complex(k) else real_out(k_t.real)),
        n_end=n_end,
        eta=real_out(eta_t),
        kind=kind,
        uin=uin_wrapped,
        density=density,
        matrix=matrix,
    )


# --------------------------------------------------------------------------------------
# field evaluation (reference biem_u :822-977)
# --------------------------------------------------------------------------------------
def biem_u(res: Any, x: Array, /, far_field: bool = False, per_ball: bool = False, expand_x: bool = True) -> Array:
    """Scattered field at cartesian x of shape (c_ndim, ...(x)) [expand_x] or (c_ndim, ...(x), ...(first))."""
    if res.density is None:
        raise ValueError("The BIEMResult does not have density.")
    if res.kind not in ("outer", "inner"):
        raise ValueError(f"Invalid kind: {res.kind}")
    c = res.c
    tree, perm = canonical_tree(c.branching_types_expression_str)
    origin, dev = _origin_of(res.centers, res.radii, res.k, res.density, x)
    if isinstance(res.density, torch.Tensor) and res.density.dtype == torch.complex64:
        origin.real_dtype = torch.float32
    elif isinstance(res.density, np.ndarray) and res.density.dtype == np.complex64:
        origin.real_dtype = torch.float32
    lib = L.load()
    f64 = torch.float64
    k_t = _to_dev(res.k, dev, torch.complex128)
    eta_t = _to_dev(res.eta, dev, f64)
    cen_t = _to_dev(res.centers, dev, f64)[list(perm)]   # [d, ...(first), B], canonical axes
    rad_t = _to_dev(res.radii, dev, f64)            # [...(first), B]
    dens_t = _to_dev(res.density, dev, torch.complex128)
    d = c.c_ndim
    B = int(rad_t.shape[-1])
    H = int(dens_t.shape[-1])
    n_end = n_end_from_harm(tree, H)
    ndim_first = k_t.ndim
    batch = tuple(np.broadcast_shapes(tuple(k_t.shape), tuple(eta_t.shape), tuple(cen_t.shape[1:-1]), tuple(rad_t.shape[:-1]),
                                      tuple(dens_t.shape[:-2])))
    nb = int(np.prod(batch)) if batch else 1
    plan = _plan(tree, n_end, dev)

    if isinstance(x, (list, tuple)):
        x = np.stack([np.asarray(v) for v in x], 0) if not isinstance(x[0], torch.Tensor) else torch.stack(list(x), 0)
    x_t = _to_dev(x, dev, f64)
    if x_t.shape[0] != d:
        raise ValueError(f"x must have shape ({d}, ...), got {tuple(x_t.shape)}")
    x_t = x_t[list(perm)]
    if expand_x:
        xshape = tuple(x_t.shape[1:])
        pts = x_t.reshape(d, -1).contiguous()
        flags = 0
    else:
        nx = x_t.ndim - 1 - ndim_first
        if nx < 0:
            raise ValueError("expand_x=False needs x of shape (c_ndim, ...(x), ...(first))")
        xshape = tuple(x_t.shape[1:1 + nx])
        pts = x_t.expand((d,) + xshape + batch).reshape(d, -1, nb).contiguous()
        flags = L.USCAT_POINTS_BATCHED
    P = int(pts.shape[1])
    if far_field:
        flags |= L.USCAT_FAR_FIELD
    if per_ball:
        flags |= L.USCAT_PER_BALL
    if res.kind == "inner":
        flags |= L.USCAT_KIND_INNER

    kf = k_t.expand(batch).reshape(nb).contiguous()
    ef = eta_t.expand(batch).reshape(nb).contiguous()
    geom_b = any(s != 1 for s in tuple(cen_t.shape[1:-1]) + tuple(rad_t.shape[:-1]))
    cen_bd = torch.movedim(cen_t, 0, -1)            # [...(first), B, d]
    if geom_b:
        cf = cen_bd.expand(batch + (B, d)).reshape(nb, B, d).contiguous()
        rf = rad_t.expand(batch + (B,)).reshape(nb, B).contiguous()
    else:
        cf = cen_bd.reshape(1, B, d).contiguous()
        rf = rad_t.reshape(1, B).contiguous()
    df = dens_t.expand(batch + (B, H)).reshape(nb, B, H).contiguous()
    out = torch.empty((P, nb, B) if per_ball else (P, nb), dtype=torch.complex128, device=dev)
    with torch.cuda.device(dev):
        wb = int(lib.biem_uscat_workspace_bytes(plan.handle, nb, B))
        work = torch.empty(max(wb, 16), dtype=torch.uint8, device=dev)
        if P > 0 and nb > 0:
            L.check(lib.biem_uscat(plan.handle, nb, B, P, _ptr(kf), _ptr(ef), _ptr(cf), _ptr(rf), int(geom_b), _ptr(df), _ptr(pts),
                                   flags, _ptr(out), _ptr(work), wb, _stream_ptr(dev)), "biem_uscat")
    out = out.reshape(xshape + batch + ((B,) if per_ball else ()))
    return origin.give(out)
