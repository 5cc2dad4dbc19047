"""One-off validation at sizes beyond the test suite: python tools/validate_large_n.py [n_end]  (16 balls, N = 16 n_end^2; GPU vs oracle)."""
import numpy as np, torch, time, sys
sys.path.insert(0, "/root/repo")
import biem_helmholtz_sphere_amd as amd
from oracle import biem_oracle as O
n_end = int(sys.argv[1]) if len(sys.argv) > 1 else 28
ax = np.arange(-2, 2) * 4.0 + 2.0
x0, x1 = np.meshgrid(ax, ax, indexing="ij")
cen = np.stack([x0.ravel(), x1.ravel(), np.zeros(16)], -1)
ks = np.array([3.0, 7.5])
t = lambda a: torch.as_tensor(np.array(a), dtype=torch.float64, device="cuda")
dirs = np.zeros((3, 2)); dirs[0] = 1
uin, _ = amd.plane_wave(k=t(ks), direction=t(dirs))
c = amd.create_from_branching_types("ba")
t0 = time.time()
calc = amd.biem(c, centers=t(cen)[None], radii=t(np.ones(16))[None], k=t(ks), n_end=n_end, uin=uin)
torch.cuda.synchronize(); print("gpu N=%d: %.2f s" % (16 * n_end ** 2, time.time() - t0))
ang = 2 * np.pi * np.arange(7) / 7
pts = np.concatenate([np.zeros((1, 3)), np.stack([10.5 * np.cos(ang), 10.5 * np.sin(ang), np.zeros(7)], -1)])
ug = calc.uscat(t(pts.T)).cpu().numpy()
t0 = time.time()
uo, _ = O.plane_wave(ks[1], [1.0, 0, 0])
res = O.solve_biem("ba", centers=cen, radii=np.ones(16), k=ks[1], n_end=n_end, uin=uo)
ref = O.uscat(res, pts)
print("oracle: %.1f s; max rel err %.3e" % (time.time() - t0, np.max(np.abs(ug[:, 1] - ref) / np.abs(ref))))
