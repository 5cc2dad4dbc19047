"""Host-side phases of ONE small biem() call (cfg 1): time inside the argument checks, the boundary samples, the C entry point and the
device -> host copies (each one a synchronisation).  python tools/phase_single_call.py   (needs a GPU)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biem_helmholtz_sphere_amd as amd
from biem_helmholtz_sphere_amd import _biem as M
t = lambda a: torch.as_tensor(np.array(a), dtype=torch.float64, device="cuda")
c = amd.create_from_branching_types("ba")
k = t(1.0)
uin, ugr = amd.plane_wave(k=k, direction=t([1.0, 0.0, 0.0]))
cen, rad, eta = t([[0.0, 2.0, 0.0], [0.0, -2.0, 0.0]]), t([1.0, 1.0]), t(1.0)
acc = {}
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **kw):
        t0 = time.perf_counter(); r = f(*a, **kw); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
    setattr(mod, name, g)
for n in ("_check_biem_inputs", "_flatten", "_plan", "_boundary_samples", "_restore_batch", "_warn_biem_inputs", "_validate_biem_inputs", "canonical_tree"):
    if hasattr(M, n): wrap(M, n)
lib = M.L.load()
class P:
    def __init__(s, l): s.l = l
    def __getattr__(s, n):
        f = getattr(s.l, n)
        def g(*a):
            t0 = time.perf_counter(); r = f(*a); acc["C:" + n] = acc.get("C:" + n, 0.0) + time.perf_counter() - t0; return r
        return g
M.L.load = lambda: P(lib)
_cpu = torch.Tensor.cpu
def cpu(self, *a, **kw):
    t0 = time.perf_counter(); r = _cpu(self, *a, **kw); acc["cpu()"] = acc.get("cpu()", 0.0) + time.perf_counter() - t0; return r
torch.Tensor.cpu = cpu
def call(): return amd.biem(c, uin=uin, k=k, n_end=6, eta=eta, centers=cen, radii=rad)
for _ in range(20): call()
torch.cuda.synchronize(); acc.clear()
t0 = time.perf_counter()
for _ in range(200): call()
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 200
print("per call %.1f us" % (tot * 1e6))
for n, v in sorted(acc.items(), key=lambda kv: -kv[1]): print("  %-28s %7.1f us" % (n, v / 200 * 1e6))
