"""Minimal hyperspherical coordinate systems for the trees this build supports.

Stands in for ``ultrasphere.SphericalCoordinates`` / ``create_from_branching_types`` (un-vendored; the
reference only needs the duck type listed in SURVEY 8(b): ``c_ndim, s_ndim, root, to_cartesian,
from_cartesian, branching_types_expression_str``).  Axis conventions are read off the reference's
committed ``a.svg / ba.svg / bba.svg``:

    a   : x0 = r cos t0,  x1 = r sin t0
    ba  : x0 = r cos t0,  x1 = r sin t0 cos t1,  x2 = r sin t0 sin t1
    bba : x0 = r cos t0,  x1 = r sin t0 cos t1,  x2 = r sin t0 sin t1 cos t2,  x3 = r sin t0 sin t1 sin t2
    bpa, bpbpa : primed polar nodes measure the angle from the equator (t in [-pi/2, pi/2]); see CANONICAL below
    caa : x0 = r cos t0 cos t1,  x1 = r cos t0 sin t1,  x2 = r sin t0 cos t2,  x3 = r sin t0 sin t2   (t0 in [0, pi/2])

Works on torch tensors and NumPy arrays alike (only elementwise maths is used).
"""
from __future__ import annotations

import math
from typing import Any, Mapping

SUPPORTED = {"a": 2, "ba": 3, "bba": 4, "bpa": 3, "bpbpa": 4, "caa": 4}
NEXT = ()
# Trees with primed (b') nodes, in the axis convention of the reference's committed bpa.svg / bpbpa.svg (its jascome driver
# relabels node 0 <-> d-1 first, cli.py:65-69): they are ba / bba in permuted Cartesian axes, canonical y_i = x_{perm[i]}:
#   bpa   : x2 = r sin t0, x1 = r cos t0 cos t1, x0 = r cos t0 sin t1                      -> (y0, y1, y2) = (x2, x1, x0)
#   bpbpa : x3 = r sin t0, x1 = r cos t0 sin t1, x2 = r cos t0 cos t1 cos t2, x0 = .. sin t2 -> (y0..y3) = (x3, x1, x2, x0)
# (pinned by the jascome goldens through the oracle and the GPU path: tests/test_oracle_golden.py, tests/test_gpu_parity.py)
CANONICAL = {"bpa": ("ba", (2, 1, 0)), "bpbpa": ("bba", (3, 1, 2, 0))}


def canonical_tree(branching_types: str):
    """(canonical tree name, perm) with canonical component i = original component perm[i]; identity for a / ba / bba."""
    if branching_types in CANONICAL:
        return CANONICAL[branching_types]
    return branching_types, tuple(range(SUPPORTED[branching_types]))


def _xp(a: Any):
    import numpy as np

    try:
        import torch

        if isinstance(a, torch.Tensor):
            return torch
    except ImportError:  # pragma: no cover
        pass
    return np


class SphericalCoordinates:
    """Coordinate tree described by its branching-types string (root first)."""

    def __init__(self, branching_types: str):
        if branching_types not in SUPPORTED:
            extra = " (planned: %s)" % ", ".join(NEXT) if branching_types in NEXT else ""
            raise NotImplementedError(
                f"coordinate tree {branching_types!r} is not built in this MI355X implementation; "
                f"available: {sorted(SUPPORTED)}{extra}"
            )
        self.branching_types_expression_str = branching_types
        self.c_ndim = SUPPORTED[branching_types]
        self.s_ndim = self.c_ndim - 1
        self.root = 0
        # adjacency of the tree, enough for display purposes (the reference's plot/CLI read c.G)
        self.G = {i: ([i + 1] if i + 1 < self.s_ndim else []) for i in range(self.s_ndim)}

    def __repr__(self) -> str:
        return f"SphericalCoordinates({self.branching_types_expression_str!r})"

    def __eq__(self, other: object) -> bool:
        return isinstance(other, SphericalCoordinates) and other.branching_types_expression_str == self.branching_types_expression_str

    def __hash__(self) -> int:
        return hash(self.branching_types_expression_str)

    # ------------------------------------------------------------------
    def to_cartesian(self, spherical: Mapping[Any, Any], as_array: bool = False):
        """{"r": r (optional, default 1), 0: theta0, 1: theta1, ...} -> cartesian (stacked on axis 0 if as_array)."""
        th = [spherical[i] for i in range(self.s_ndim)]
        xp = _xp(th[0])
        if self.branching_types_expression_str == "caa":
            c0, s0 = xp.cos(th[0]), xp.sin(th[0])
            comps = [c0 * xp.cos(th[1]), c0 * xp.sin(th[1]), s0 * xp.cos(th[2]), s0 * xp.sin(th[2])]
            if "r" in spherical:
                comps = [spherical["r"] * v for v in comps]
            if as_array:
                comps = xp.broadcast_arrays(*comps) if xp.__name__ == "numpy" else xp.broadcast_tensors(*comps)
                return xp.stack(list(comps), 0)
            return {i: v for i, v in enumerate(comps)}
        if self.branching_types_expression_str in CANONICAL:
            # b' angle t (from the equator) = pi/2 - colatitude of the canonical tree, for every polar node
            base, perm = CANONICAL[self.branching_types_expression_str]
            sph = {i: (math.pi / 2 - th[i]) for i in range(self.s_ndim - 1)}
            sph[self.s_ndim - 1] = th[-1]
            if "r" in spherical:
                sph["r"] = spherical["r"]
            y = SphericalCoordinates(base).to_cartesian(sph, as_array=False)
            comps = [None] * self.c_ndim
            for i, pi_ in enumerate(perm):
                comps[pi_] = y[i]
            if as_array:
                comps = xp.broadcast_arrays(*comps) if xp.__name__ == "numpy" else xp.broadcast_tensors(*comps)
                return xp.stack(list(comps), 0)
            return {i: c for i, c in enumerate(comps)}
        r = spherical["r"] if "r" in spherical else None
        comps = []
        sin_prod = None
        for i in range(self.s_ndim):
            c, s = xp.cos(th[i]), xp.sin(th[i])
            comps.append(c if sin_prod is None else sin_prod * c)
            sin_prod = s if sin_prod is None else sin_prod * s
        comps.append(sin_prod)
        if r is not None:
            comps = [r * c for c in comps]
        if as_array:
            comps = xp.broadcast_arrays(*comps) if xp.__name__ == "numpy" else xp.broadcast_tensors(*comps)
            return xp.stack(list(comps), 0)
        return {i: c for i, c in enumerate(comps)}

    def from_cartesian(self, x: Any):
        """cartesian x[d, ...] (array or mapping 0..d-1) -> {"r": r, 0: theta0, ...}; polar angles in [0, pi], last in (-pi, pi]."""
        xs = [x[i] for i in range(self.c_ndim)]
        xp = _xp(xs[0])
        if self.branching_types_expression_str == "caa":
            r01 = xp.sqrt(xs[0] * xs[0] + xs[1] * xs[1])
            r23 = xp.sqrt(xs[2] * xs[2] + xs[3] * xs[3])
            return {"r": xp.sqrt(r01 * r01 + r23 * r23), 0: xp.arctan2(r23, r01), 1: xp.arctan2(xs[1], xs[0]), 2: xp.arctan2(xs[3], xs[2])}
        if self.branching_types_expression_str in CANONICAL:
            base, perm = CANONICAL[self.branching_types_expression_str]
            sph = SphericalCoordinates(base).from_cartesian([xs[pi_] for pi_ in perm])
            out = {"r": sph["r"], self.s_ndim - 1: sph[self.s_ndim - 1]}
            for i in range(self.s_ndim - 1):
                out[i] = math.pi / 2 - sph[i]
            return out
        out = {}
        # tail norms: rho_i = |(x_i, ..., x_{d-1})|
        tail = xs[-1] * xs[-1]
        rho = [None] * self.c_ndim
        rho[self.c_ndim - 1] = xp.abs(xs[-1])
        for i in range(self.c_ndim - 2, -1, -1):
            tail = tail + xs[i] * xs[i]
            rho[i] = xp.sqrt(tail)
        out["r"] = rho[0]
        for i in range(self.s_ndim - 1):
            out[i] = xp.arctan2(rho[i + 1], xs[i])
        out[self.s_ndim - 1] = xp.arctan2(xs[-1], xs[-2])
        return out


def create_from_branching_types(branching_types: str) -> SphericalCoordinates:
    return SphericalCoordinates(branching_types)


def harm_count(branching_types: str, n_end: int) -> int:
    """Number of harmonics of degree < n_end (``ush.harm_n_ndim_le``)."""
    if n_end <= 0:
        return 0
    branching_types = CANONICAL.get(branching_types, (branching_types,))[0]
    if branching_types == "a":
        return 2 * n_end - 1
    if branching_types == "ba":
        return n_end * n_end
    if branching_types in ("bba", "caa"):
        return n_end * (n_end + 1) * (2 * n_end + 1) // 6
    raise NotImplementedError(branching_types)


def n_end_from_harm(branching_types: str, n_harm: int) -> int:
    """Inverse of :func:`harm_count` (``ush.assume_n_end_and_include_negative_m_from_harmonics``, _biem.py:864)."""
    n = 0
    while harm_count(branching_types, n) < n_harm:
        n += 1
    if harm_count(branching_types, n) != n_harm:
        raise ValueError(f"{n_harm} is not a harmonic count of tree {branching_types!r}")
    return n
