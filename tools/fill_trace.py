"""Where does a combination of the symmetric fill spend its time?  Diagnostic build (-DBIEM_FILL_TRACE): thread 0 of every workgroup
of k_fill_red accumulates s_memtime deltas per phase.  BIEM_HIPCC_FLAGS=-DBIEM_FILL_TRACE BIEM_SKIP_ISA_CHECK=1 python tools/fill_trace.py [cfg]"""
import ctypes as C, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import biem_helmholtz_sphere_amd as amd
from biem_helmholtz_sphere_amd import _lib as L
import bench
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
S = bench.DEFAULT_SYSTEMS[cfg]
w = bench.workload(cfg, S, 0, S)
dev = torch.device("cuda", 0)
t = lambda a, dt=torch.float64: torch.as_tensor(np.array(a), device=dev).to(dt).contiguous()
c = amd.create_from_branching_types(w["tree"])
dirs = np.zeros((w["d"], S)); dirs[0] = 1.0
k_t, eta_t = t(w["ks"]), t(w["etas"])
uin, ugr = amd.plane_wave(k=k_t, direction=t(dirs))
kw = dict(centers=t(w["centers"])[None], radii=t(np.ones(w["B"]))[None], n_end=w["n_end"], alpha=w["alpha"], beta=w["beta"])
if w["beta"] != 0: kw["uin_grad"] = ugr
lib = L.load()
fn = lib.biem_debug_fill_trace; fn.restype = C.c_int; fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
amd.biem(c, k=k_t, eta=eta_t, uin=uin, **kw); torch.cuda.synchronize()
out = (C.c_ulonglong * 8)(); fn(out, 1)
amd.biem(c, k=k_t, eta=eta_t, uin=uin, **kw); torch.cuda.synchronize()
fn(out, 1)
v = np.array(list(out), dtype=np.float64)
names = ["loop head (pair_of)", "barrier 1 (prev readers, stores' vmcnt)", "table regs -> LDS (waits the prefetch)", "q factors (global loads + zsqrt)", "barrier 2", "prefetch issue + contraction", "epilogue + stores", "-"]
print(f"cfg {cfg}: thread 0 of every workgroup, s_memtime ticks (100 MHz) summed; shares of the loop")
for n, x in zip(names, v): print(f"  {n:45s} {x:14.0f}  {100 * x / max(v.sum(), 1):5.1f} %")
