"""Times the field evaluation (reference biem_u, its second hot loop: the 100 x 100 plot grid).
python tools/time_uscat.py [systems] [case]     case: ba (cfg 3 densities; default) | caa (4-D, 8 balls, n_end 8) | inner (one ball, ba, n_end 20,
points inside) | all.  Each case is timed through the point-per-lane kernel and, with BIEM_USCAT_GENERIC=1, the harmonic-by-harmonic one."""
import numpy as np, torch, time, sys, os
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import biem_helmholtz_sphere_amd as amd
nsys = int(sys.argv[1]) if len(sys.argv) > 1 else 8
case = sys.argv[2] if len(sys.argv) > 2 else "ba"
t = lambda a: torch.as_tensor(np.array(a), dtype=torch.float64, device="cuda")
ks = np.linspace(0.5, 8.0, nsys)
g = np.linspace(-12, 12, 100)
X, Y = np.meshgrid(g, g, indexing="ij")


def timed(name, calc, pts):
    for env in ("", "1"):
        if env: os.environ["BIEM_USCAT_GENERIC"] = "1"
        else: os.environ.pop("BIEM_USCAT_GENERIC", None)
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.time()
            u = calc.uscat(pts)
            torch.cuda.synchronize(); dt = time.time() - t0
        print("%-6s %-9s %d points x %d systems in %.4f s  (%.2e point-systems/s), nan fraction %.2f" % (
            name, "generic" if env else "per-lane", pts[0].numel(), nsys, dt, pts[0].numel() * nsys / dt, float(torch.isnan(u.real).float().mean())), flush=True)
    os.environ.pop("BIEM_USCAT_GENERIC", None)


if case in ("ba", "all"):
    ax = np.arange(-2, 2) * 4.0 + 2.0
    x0, x1 = np.meshgrid(ax, ax, indexing="ij")
    cen = np.stack([x0.ravel(), x1.ravel(), np.zeros(16)], -1)
    dirs = np.zeros((3, nsys)); dirs[0] = 1
    uin, _ = amd.plane_wave(k=t(ks), direction=t(dirs))
    calc = amd.biem(amd.create_from_branching_types("ba"), centers=t(cen)[None], radii=t(np.ones(16))[None], k=t(ks), n_end=20, uin=uin)
    timed("ba", calc, t(np.stack([X, Y, 0.3 * np.ones_like(X)])))
if case in ("caa", "all"):
    cen = np.zeros((8, 4)); cen[:, 0] = 3.0 * (np.arange(8) % 4) - 4.5; cen[:, 2] = 3.0 * (np.arange(8) // 4) - 1.5
    dirs = np.zeros((4, nsys)); dirs[0] = 1
    uin, _ = amd.plane_wave(k=t(ks), direction=t(dirs))
    calc = amd.biem(amd.create_from_branching_types("caa"), centers=t(cen)[None], radii=t(np.ones(8))[None], k=t(ks), n_end=8, uin=uin)
    timed("caa", calc, t(np.stack([X, 0.2 * np.ones_like(X), Y, 0.3 * np.ones_like(X)])))
if case in ("inner", "all"):
    dirs = np.zeros((3, nsys)); dirs[0] = 1
    uin, _ = amd.plane_wave(k=t(ks), direction=t(dirs))
    calc = amd.biem(amd.create_from_branching_types("ba"), centers=t(np.zeros((1, 3)))[None], radii=t([12.0 * 1.5])[None], k=t(ks), n_end=20,
                    uin=uin, kind="inner")
    timed("inner", calc, t(np.stack([X, Y, 0.3 * np.ones_like(X)])))
