"""Kernel timeline of ONE system per call (the reference's own drivers call biem() like this).
  run:     rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/single_timeline.py run CFG   (3 warm-up calls + 5 traced calls)
  report:  python tools/single_timeline.py report DIR     per kernel: launches per call, mean duration, share of the call's GPU span; idle gaps"""
import os, sys, glob, csv, re, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WARM, REPS = 3, 5


def run(cfg):
    import numpy as np, torch, time
    import bench
    import biem_helmholtz_sphere_amd as amd
    w = bench.workload(cfg, 1 if cfg != 5 else 16, 0, 1)
    t = lambda a, dt=torch.float64: torch.as_tensor(np.array(a), dtype=dt, device="cuda")
    dirs = np.zeros((w["d"], 1)); dirs[0] = 1.0
    k = t(w["ks"][:1]); eta = t(w["etas"][:1])
    uin, ugr = amd.plane_wave(k=k, direction=t(dirs))
    c = amd.create_from_branching_types(w["tree"])
    kw = dict(centers=t(w["centers"])[None], radii=t(np.ones(w["B"]))[None], n_end=w["n_end"], alpha=w["alpha"], beta=w["beta"], uin=uin)
    if w["beta"] != 0: kw["uin_grad"] = ugr
    for i in range(WARM + REPS):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        amd.biem(c, k=k, eta=eta, **kw)
        torch.cuda.synchronize()
        print("call %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)


def report(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
    per = len(rows) // (WARM + REPS)
    rows = rows[len(rows) - per * REPS:]
    span = gaps = 0
    byk = collections.defaultdict(lambda: [0, 0])
    for r in range(REPS):
        call = rows[r * per:(r + 1) * per]
        span += call[-1][1] - call[0][0]
        end = call[0][0]
        for s, e, n in call:
            if s > end: gaps += s - end
            end = max(end, e)
            n = re.sub(r"\(.*", "", n).replace("biem::", "").replace("void ", "")
            byk[n][0] += 1; byk[n][1] += e - s
    print("%d launches per call, GPU span %.3f ms per call, idle between kernels %.3f ms" % (per, span / REPS / 1e6, gaps / REPS / 1e6))
    for n, (cnt, ns) in sorted(byk.items(), key=lambda kv: -kv[1][1]):
        print("  %-60s %6.1f launches  %7.1f us each  %7.3f ms  %5.1f %%" % (n[:60], cnt / REPS, ns / cnt / 1e3, ns / REPS / 1e6, 100.0 * ns / span))


if __name__ == "__main__":
    run(int(sys.argv[2])) if sys.argv[1] == "run" else report(sys.argv[2])
