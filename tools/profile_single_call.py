#!/usr/bin/env python
"""Host-side profile of ONE small biem() call (cfg 1: N = 72): where the ~0.55 ms of a single-system call go (cProfile, by own time)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biem_helmholtz_sphere_amd as amd

dev = "cuda"
t = lambda a: torch.as_tensor(np.array(a), dtype=torch.float64, device=dev)
c = amd.create_from_branching_types("ba")
k = t(1.0)
uin, ugr = amd.plane_wave(k=k, direction=t([1.0, 0.0, 0.0]))
cen, rad, eta = t([[0.0, 2.0, 0.0], [0.0, -2.0, 0.0]]), t([1.0, 1.0]), t(1.0)


def call():
    return amd.biem(c, uin=uin, k=k, n_end=6, eta=eta, centers=cen, radii=rad)


for _ in range(5):
    call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    call()
torch.cuda.synchronize()
print("ms per call:", (time.perf_counter() - t0) * 5)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    call()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(30)
