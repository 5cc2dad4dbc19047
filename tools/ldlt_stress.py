"""Where does the symmetric path reject pivots?  Close spheres x wavenumbers; for each case the number of systems that fell
back to the pivoted LU and the largest relative difference of u_scat between the default path and BIEM_SOLVER=lu.
python tools/ldlt_stress.py [n_end]   (needs a GPU; n_end 14 by default - the blocked row form; 4 or 6 put the pair and the
big+small case on the one-launch path of small systems)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biem_helmholtz_sphere_amd as amd
from biem_helmholtz_sphere_amd import _biem as impl

t = lambda a, dt=torch.float64: torch.as_tensor(np.array(a), dtype=dt, device="cuda")
c = amd.create_from_branching_types("ba")
n_end = int(sys.argv[1]) if len(sys.argv) > 1 else 14
ks = np.array([0.5, 2.0, 5.0, 10.0, 3.0 + 0.5j])
dirs = np.zeros((3, len(ks))); dirs[0] = 1.0
x = np.array([[6.0, 3.0, 0.1], [-5.0, 2.0, 1.0], [0.2, 7.0, -1.0]]).T      # outside every sphere of every case
for gap in (1.5, 1.1, 1.02, 1.002):
    for name, cen, rad in (
        ("pair", [[0, gap, 0], [0, -gap, 0]], [1.0, 1.0]),
        ("chain of 4", [[2 * gap * i, 0, 0] for i in range(4)], [1.0] * 4),
        ("big+small", [[0, 0, 0], [(1 + 0.3) * gap, 0, 0], [0, (1 + 0.3) * gap, 0]], [1.0, 0.3, 0.3]),
    ):
        out = {}
        for solver in ("ldlt", "lu"):
            os.environ["BIEM_SOLVER"] = solver
            uin, ugr = amd.plane_wave(k=t(ks, torch.complex128), direction=t(dirs))
            calc = amd.biem(c, centers=t(cen)[None], radii=t(rad)[None], k=t(ks, torch.complex128), eta=t(np.ones(len(ks))), n_end=n_end,
                            alpha=1.0, beta=0.3, uin=uin, uin_grad=ugr)
            out[solver] = calc.uscat(t(x)).cpu().numpy()
            if solver == "ldlt":
                st = dict(impl._last_solve_stats)
        err = np.max(np.abs(out["ldlt"] - out["lu"]) / np.abs(out["lu"]))
        print(f"n_end {n_end:2d} N {len(rad) * n_end * n_end:4d} gap {gap:6.3f} {name:11s} fell back to LU: {st['lu_systems']} of {st['ldlt_systems']}   max rel diff u_scat {err:.2e}  info {st.get('rejected_info')}", flush=True)
