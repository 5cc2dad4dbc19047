import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import biem_helmholtz_sphere_amd as amd
from biem_helmholtz_sphere_amd import _biem as M
lib = M.L.load()
class P:
    def __init__(s, l): s.l = l
    def __getattr__(s, n):
        f = getattr(s.l, n)
        if n != "biem_solve_workspace_bytes": return f
        def g(*a):
            r = f(*a); print("workspace_bytes", a[1:], "->", r, flush=True); return r
        return g
M.L.load = lambda: P(lib)
w = bench.workload(3, 256, 0, 256)
t = lambda a, dt=torch.float64: torch.as_tensor(np.array(a), dtype=dt, device="cuda")
dirs = np.zeros((3, 256)); dirs[0] = 1.0
k = t(w["ks"]); eta = t(w["etas"])
uin, ugr = amd.plane_wave(k=k, direction=t(dirs))
c = amd.create_from_branching_types("ba")
for i in range(3):
    print("free/reserved/allocated GiB", torch.cuda.mem_get_info()[0] / 2**30, torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_allocated() / 2**30, flush=True)
    amd.biem(c, centers=t(w["centers"])[None], radii=t(np.ones(16))[None], k=k, eta=eta, n_end=20, uin=uin)
    torch.cuda.synchronize()
print("ok")
