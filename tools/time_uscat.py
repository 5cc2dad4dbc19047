"""Times the field evaluation (reference biem_u, its second hot loop: the 100 x 100 plot grid) on cfg 3 densities."""
import numpy as np, torch, time, sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import biem_helmholtz_sphere_amd as amd
nsys = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ax = np.arange(-2, 2) * 4.0 + 2.0
x0, x1 = np.meshgrid(ax, ax, indexing="ij")
cen = np.stack([x0.ravel(), x1.ravel(), np.zeros(16)], -1)
ks = np.linspace(0.5, 8.0, nsys)
t = lambda a: torch.as_tensor(np.array(a), dtype=torch.float64, device="cuda")
dirs = np.zeros((3, nsys)); dirs[0] = 1
uin, _ = amd.plane_wave(k=t(ks), direction=t(dirs))
c = amd.create_from_branching_types("ba")
calc = amd.biem(c, centers=t(cen)[None], radii=t(np.ones(16))[None], k=t(ks), n_end=20, uin=uin)
g = np.linspace(-12, 12, 100)
X, Y = np.meshgrid(g, g, indexing="ij")
pts = t(np.stack([X, Y, 0.3 * np.ones_like(X)]))          # (3, 100, 100)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    u = calc.uscat(pts)
    torch.cuda.synchronize(); dt = time.time() - t0
print("uscat: %d points x %d systems in %.3f s  (%.2e point-systems/s), nan fraction %.2f" % (pts[0].numel(), nsys, dt, pts[0].numel() * nsys / dt, float(torch.isnan(u.real).float().mean())))
