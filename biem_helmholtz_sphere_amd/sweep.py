"""Sweep driver writing the reference's CSV schemas, so results diff directly against its committed files.

Mirrors the *configurations* of the reference's golden-generating commands (`jascome` cli.py:36-115 and `accuracy`
cli.py:188-271: 2 balls at (0, +-2, 0..) or square grids of pitch 4, radius 1, eta = 1, sound-soft, plane wave along +x0
with wavenumber 1.0 whatever the operator's k - reference quirk cli.py:238-244), not its CLI framework.

    python -m biem_helmholtz_sphere_amd.sweep jascome  --out jascome_output.csv [--types a,ba,bpa,bba,bpbpa,caa] [--n-end-max 9]
    python -m biem_helmholtz_sphere_amd.sweep accuracy --out accuracy.csv --types ba --n-balls 2 --k 1,2,4 --n-end 1,2,3,4
"""
from __future__ import annotations

import argparse
from typing import List

import numpy as np
import torch

from . import biem, create_from_branching_types, plane_wave

JASCOME_HEADER = "branching_types,n_end,uscat,device,dtype,density_dtype,density_device,uscat_dtype,uscat_device\n"
ACCURACY_HEADER = "branching_types,n_end,k,n_balls,uscat,device,dtype,density_dtype,density_device,uscat_dtype,uscat_device\n"


def grid_centers(half: int, c_ndim: int) -> np.ndarray:
    """Geometry of the reference's `_center` (cli.py:170-185): half = 0 -> two balls at (0, +-2, 0..), else a (2 half)^2 grid."""
    if half == 0:
        c = np.zeros((2, c_ndim))
        c[0, 1], c[1, 1] = 2.0, -2.0
        return c
    ax = np.arange(-half, half) * 4.0 + 2.0
    x0, x1 = np.meshgrid(ax, ax, indexing="ij")
    return np.stack([x0.ravel(), x1.ravel()] + [np.zeros(x0.size)] * (c_ndim - 2), axis=-1)


def _uscat_origin(btype: str, n_end: int, k: float, centers: np.ndarray, device: str):
    c = create_from_branching_types(btype)
    t = lambda a: torch.as_tensor(np.array(a), dtype=torch.float64, device=device)
    e0 = np.zeros(c.c_ndim)
    e0[0] = 1.0
    uin = plane_wave(k=t(1.0), direction=t(e0))[0]                 # incident wavenumber 1.0 always (reference quirk)
    calc = biem(c, uin=uin, k=t(k), n_end=n_end, eta=t(1.0), centers=t(centers), radii=t(np.ones(len(centers))), kind="outer")
    u = calc.uscat(t(np.zeros(c.c_ndim)))
    return complex(u.cpu().numpy()), calc.density, u


def jascome(out: str, types: List[str], n_end_max: int, device: str) -> None:
    with open(out, "w") as f:
        f.write(JASCOME_HEADER)
        for bt in types:
            d = create_from_branching_types(bt).c_ndim
            for n_end in range(1, n_end_max + 1):
                u, dens, ut = _uscat_origin(bt, n_end, 1.0, grid_centers(0, d), device)
                f.write(f"{bt},{n_end},{u},{device},torch.float64,{dens.dtype},{dens.device},{ut.dtype},{ut.device}\n")


def accuracy(out: str, types: List[str], n_balls: List[int], ks: List[float], n_ends: List[int], device: str) -> None:
    half = {2: 0, 4: 1, 16: 2, 64: 4, 256: 8}
    with open(out, "w") as f:
        f.write(ACCURACY_HEADER)
        for bt in types:
            d = create_from_branching_types(bt).c_ndim
            for nb in n_balls:
                for k in ks:
                    for n_end in n_ends:
                        u, dens, ut = _uscat_origin(bt, n_end, k, grid_centers(half[nb], d), device)
                        if not np.isfinite(u.real):
                            break                        # the reference stops a k at the first overflow (cli.py:255-271)
                        f.write(f"{bt},{n_end},{k},{nb},{u},{device},torch.float64,{dens.dtype},{dens.device},{ut.dtype},{ut.device}\n")


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(prog="biem_helmholtz_sphere_amd.sweep")
    sub = ap.add_subparsers(dest="cmd", required=True)
    j = sub.add_parser("jascome")
    j.add_argument("--out", default="jascome_output.csv")
    j.add_argument("--types", default="a,ba,bpa,bba,bpbpa,caa")
    j.add_argument("--n-end-max", type=int, default=9)
    j.add_argument("--device", default="cuda")
    a = sub.add_parser("accuracy")
    a.add_argument("--out", default="accuracy.csv")
    a.add_argument("--types", default="a")
    a.add_argument("--n-balls", default="2")
    a.add_argument("--k", default="1")
    a.add_argument("--n-end", default="1,2,3,4,5,6")
    a.add_argument("--device", default="cuda")
    args = ap.parse_args(argv)
    if args.cmd == "jascome":
        jascome(args.out, args.types.split(","), args.n_end_max, args.device)
    else:
        accuracy(args.out, args.types.split(","), [int(v) for v in args.n_balls.split(",")], [float(v) for v in args.k.split(",")],
                 [int(v) for v in args.n_end.split(",")], args.device)


if __name__ == "__main__":
    main()
