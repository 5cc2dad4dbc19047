import os, sys, csv
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biem_helmholtz_sphere_amd as amd
from oracle import biem_oracle as O
_dev = lambda a, dt=torch.float64: torch.as_tensor(np.array(a), device="cuda").to(dt).contiguous()
rows = {}
for r in csv.DictReader(open("tests/golden/accuracy_k_a.csv")):
    rows[(int(r["n_end"]), round(float(r["k"]), 6))] = complex(r["uscat"])
c = amd.create_from_branching_types("a")
cen = O.grid_centers(0, 2)
def run(n_end, k_op):
    uin, _ = amd.plane_wave(k=_dev(1.0), direction=_dev([1.0, 0.0]))
    calc = amd.biem(c, centers=_dev(cen), radii=_dev(np.ones(2)), k=_dev(k_op), eta=_dev(1.0), n_end=n_end, uin=uin)
    return complex(calc.uscat(_dev(np.zeros(2))).cpu().numpy()), calc
for k, n_end in ((11.313708498984761, 152), (11.313708498984761, 181), (16.0, 181), (16.0, 215), (32.0, 181), (32.0, 256), (64.0, 256), (128.0, 512)):
    got, calc = run(n_end, k)
    want = rows.get((n_end, round(k, 6)))
    uo_in, _ = O.plane_wave(1.0, [1.0, 0.0])
    res = O.solve_biem("a", centers=cen, radii=np.ones(2), k=k, n_end=n_end, uin=uo_in)
    uo = complex(O.uscat(res, np.zeros(2)))
    dg, do = calc.density.cpu().numpy(), res.density
    print(k, n_end, "gpu-golden", None if want is None else abs(got - want), "oracle-golden", None if want is None else abs(uo - want), "gpu-oracle", abs(got - uo),
          "density normwise", np.abs(dg - do).max() / np.abs(do).max(), flush=True)
