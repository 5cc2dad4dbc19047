#!/usr/bin/env python
"""bench.py -- BIEM systems solved per second on the BASELINE.json configurations (default: configs[2], the headline).

One *step* = one pass of the whole hot path (boundary samples -> RHS projection -> fill -> complex-symmetric U^T U factorisation, or the
pivoted LU for systems it rejects / BIEM_SOLVER=lu -> density) over this rank's systems through the public `biem()` API;
inputs are resident in HBM when the clock starts, the densities are resident in HBM when it stops.

Workloads (SURVEY 8(d) inputs; radii 1, plane wave along +x0 with each system's own k, eta = 1 unless stated):
  --config 1  d=3 'ba', 2 balls at (0, +-2, 0), n_end 6 (N = 72), k = 1            [the config is ONE system: the batch is S copies]
  --config 2  d=3, 4 balls (2 x 2 grid, pitch 4), n_end 12 (N = 576), k = 1         [one system: S copies]
  --config 3  d=3, 16 balls (4 x 4 grid, pitch 4), n_end 20 (N = 6400), 256 wavenumbers linspace(0.5, 8)      (default)
  --config 4  d=2 'a', 32 balls (4 x 8 grid), n_end 64 (N = 4064), Robin alpha = beta = 1, k = 1             [one system: S copies]
  --config 5  d=4 'bba', 8 balls (2 x 4 grid), n_end 10 (N = 3080), 512 pairs = 32 k in linspace(0.5, 4) x 16 eta in linspace(0.25, 4)
S = --systems-per-gpu (defaults 4096 / 512 / 256 / 64 / 512).  For the one-system configs the line also carries
`single_system_ms` (one system per call, what the config literally describes; launch-latency bound).

Multi-GPU: one process per GPU.  Under torchrun (WORLD_SIZE set) the ranks are used as given; `python bench.py --gpus N` alone
starts N child processes itself (the parent never touches a GPU).  --scaling weak (default): every GPU owns S systems, the k's
of the whole job are one linspace of S*N values; --scaling strong: the config's S systems are split S/N per GPU.  No collective
sits in the timed data path (independent systems); RCCL carries the barrier / max-reduce of the timing and,
outside the timed region, one all-gather of the densities (`marshalling_ms`).

Prints ONE JSON line on rank 0 (fields per the driver contract, plus `roofline`, `fill` and `cpu_baseline`).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X FP64 matrix, vendor spec (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz); probe: profiles/r01_mfma_f64_*
HBM_PEAK_GBS = 8000.0
CLASSES = ["tables", "fill", "rhs", "panel", "swap", "trsm", "gemm", "back", "gemm_small"]
DEFAULT_SYSTEMS = {1: 4096, 2: 512, 3: 256, 4: 64, 5: 512}


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------
def _grid(ax0, ax1, d):
    x0, x1 = np.meshgrid(ax0, ax1, indexing="ij")
    cen = np.zeros((x0.size, d))
    cen[:, 0], cen[:, 1] = x0.ravel(), x1.ravel()
    return cen


def workload(cfg: int, n_total: int, lo: int, hi: int):
    """Systems lo..hi-1 of a job of n_total systems of configuration `cfg` (NumPy; the caller moves them to the device)."""
    if cfg == 1:
        tree, n_end, cen = "ba", 6, np.array([[0.0, 2.0, 0.0], [0.0, -2.0, 0.0]])
        ks, etas, ab = np.ones(n_total), np.ones(n_total), (1.0, 0.0)
        desc = f"cfg1: d=3 'ba', 2 balls at (0,+-2,0), n_end=6, N=72, k=1, sound-soft; {n_total} copies of the one system of the config"
    elif cfg == 2:
        ax = np.array([-2.0, 2.0])
        tree, n_end, cen = "ba", 12, _grid(ax, ax, 3)
        ks, etas, ab = np.ones(n_total), np.ones(n_total), (1.0, 0.0)
        desc = f"cfg2: d=3 'ba', 4 balls (2x2 grid, pitch 4), n_end=12, N=576, k=1, sound-soft; {n_total} copies of the one system of the config"
    elif cfg == 3:
        ax = np.arange(-2, 2) * 4.0 + 2.0
        tree, n_end, cen = "ba", 20, _grid(ax, ax, 3)
        ks, etas, ab = np.linspace(0.5, 8.0, n_total), np.ones(n_total), (1.0, 0.0)
        desc = f"cfg3: d=3 'ba', n_balls=16 (4x4 grid, pitch 4), n_end=20, N=6400, sound-soft, {n_total} wavenumbers linspace(0.5, 8, {n_total})"
    elif cfg == 4:
        tree, n_end, cen = "a", 64, _grid(np.arange(4) * 4.0 - 6.0, np.arange(8) * 4.0 - 14.0, 2)
        ks, etas, ab = np.ones(n_total), np.ones(n_total), (1.0, 1.0)
        desc = f"cfg4: d=2 'a', 32 balls (4x8 grid, pitch 4), n_end=64, N=4064, Robin alpha=beta=1, k=1; {n_total} copies of the one system of the config"
    elif cfg == 5:
        tree, n_end, cen = "bba", 10, _grid(np.arange(2) * 4.0 - 2.0, np.arange(4) * 4.0 - 6.0, 4)
        nk = max(1, n_total // 16)
        kk, ee = np.meshgrid(np.linspace(0.5, 4.0, nk), np.linspace(0.25, 4.0, 16), indexing="ij")
        ks, etas, ab = kk.ravel()[:n_total], ee.ravel()[:n_total], (1.0, 0.0)
        if len(ks) < n_total:
            raise SystemExit("cfg5: the number of systems must be a multiple of 16 (k's x 16 eta's)")
        desc = f"cfg5: d=4 'bba', 8 balls (2x4 grid, pitch 4), n_end=10, N=3080, sound-soft, {n_total} (k, eta) pairs = {nk} k in linspace(0.5,4) x 16 eta in linspace(0.25,4)"
    else:
        raise SystemExit(f"unknown --config {cfg}")
    return dict(tree=tree, n_end=n_end, centers=cen, ks=ks[lo:hi], etas=etas[lo:hi], alpha=ab[0], beta=ab[1], desc=desc,
                d=cen.shape[1], B=len(cen))


def harm_count(tree: str, n_end: int) -> int:
    return {"a": 2 * n_end - 1, "ba": n_end**2, "bba": n_end * (n_end + 1) * (2 * n_end + 1) // 6}[tree]


# ------------------------------------------------------------------------------------------------
# CPU baseline: the oracle = this repo's NumPy/LAPACK restatement of the reference's CPU path
# ------------------------------------------------------------------------------------------------
def _blas_threads():
    try:
        from threadpoolctl import threadpool_info

        return max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        return os.cpu_count() or 1


def _oracle_systems(w, idx, reps):
    """Per-system wall times (median over `reps` after one warm-up when reps > 1) of oracle fill + numpy.linalg.solve."""
    from oracle import biem_oracle as O   # checker / baseline only

    e0 = np.zeros(w["d"])
    e0[0] = 1.0
    results, times, solves = [], [], []
    warmed = False
    for i in idx:
        k, eta = float(w["ks"][i]), float(w["etas"][i])
        uin, ugr = O.plane_wave(k, e0)

        def one():
            t0 = time.perf_counter()
            r = O.solve_biem(w["tree"], centers=w["centers"], radii=np.ones(w["B"]), k=k, n_end=w["n_end"], eta=eta, alpha=w["alpha"],
                             beta=w["beta"], uin=uin, uin_grad=ugr if w["beta"] != 0 else None)
            return r, time.perf_counter() - t0

        if reps > 1 and not warmed:      # one warm-up in all (caches, allocator); the tables are built before the timing
            one()
            warmed = True
        ts = []
        for _ in range(reps):
            r, t = one()
            ts.append(t)
        results.append(r)
        times.append(statistics.median(ts))
        if r.matrix is not None and r.rhs is not None:          # the LAPACK part alone (the rest is the NumPy assembly)
            n = r.rhs.size
            t0 = time.perf_counter()
            np.linalg.solve(r.matrix.reshape(n, n), r.rhs.reshape(n))
            solves.append(time.perf_counter() - t0)
    return results, times, solves


def cpu_baseline(w, n_sys: int, reps: int, one_thread: bool):
    """Oracle timed on `n_sys` systems spread over the rank's batch (all BLAS threads), and on one system with one thread."""
    from oracle import biem_oracle as O

    if w["tree"] == "ba":
        O._terms3(w["n_end"])             # table build is amortised over a sweep: keep it out of the timing
    else:
        _oracle_systems(w, [0], 1)        # warm the tree's table caches
    nb = len(w["ks"])
    idx = sorted(set(int(round(v)) for v in np.linspace(0, nb - 1, min(n_sys, nb))))
    t_all0 = time.perf_counter()
    res, times, solves = _oracle_systems(w, idx, reps)
    threads = _blas_threads()
    out = {"value": 1.0 / statistics.median(times), "unit": "systems/s", "cores": int(threads), "os_cpu_count": os.cpu_count(),
           "kind": "port",
           "sample": f"{len(idx)} systems of the workload (batch indices {idx}), per system the median of {reps} run(s)"
                     f"{' (one untimed warm-up solve first)' if reps > 1 else ''} of oracle fill + numpy.linalg.solve; value = 1 / median over the systems "
                     f"({', '.join(f'{t:.2f}' for t in times)} s, of which numpy.linalg.solve alone {statistics.median(solves) if solves else float('nan'):.2f} s: "
                     f"the assembly is single-threaded NumPy as in the reference); {time.perf_counter() - t_all0:.1f} s in all"}
    if one_thread:
        try:
            from threadpoolctl import threadpool_limits

            with threadpool_limits(limits=1):
                t0 = time.perf_counter()
                _, t1, s1 = _oracle_systems(w, idx[:1], 1)
            out["one_thread"] = {"value": 1.0 / t1[0], "unit": "systems/s", "cores": 1,
                                 "sample": f"system {idx[0]} once with BLAS limited to 1 thread ({t1[0]:.2f} s, numpy.linalg.solve alone {s1[0] if s1 else float('nan'):.2f} s)"}
        except Exception as e:  # noqa: BLE001
            out["one_thread"] = {"value": None, "error": str(e)}
    return res, idx, out


# ------------------------------------------------------------------------------------------------
def _spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks as child processes.  This parent never initialises a GPU
    (device_count() does not, on this image) and does not exec over itself; it returns rank 0's exit code (first failure wins)."""
    import socket

    import torch

    share = os.environ.get("BIEM_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not share:
        print(f"bench.py: --gpus {args.gpus} but only {have} HIP device(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=[1, 2, 3, 4, 5])
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--systems-per-gpu", type=int, default=0, help="0 = the config's batch (cfg3: 256, cfg5: 512; one-system configs: copies)")
    ap.add_argument("--chunk", type=int, default=0, help="resident matrices per pass (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-systems", type=int, default=4)
    ap.add_argument("--cpu-baseline-reps", type=int, default=5, help="BASELINE.md section 3: median of 5 repetitions after one warm-up")
    ap.add_argument("--sym-vs-lu-systems", type=int, default=16, help="systems of the timed batch re-solved with the pivoted LU for the whole-batch accuracy figure")
    ap.add_argument("--single-system", action="store_true", help="also for cfg 3 / 5 (batches of different systems): latency of the first system alone, one system per call (default only for the one-system configs 1, 2, 4: its small launches would enter a profiler's per-kernel averages of the default run)")
    ap.add_argument("--no-single-system", action="store_true", help="skip the one-system-per-call latency measurement")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return _spawn_ranks(args)

    import torch

    # stdout carries exactly ONE line, the JSON: libraries that print there (RCCL's version banner at communicator set-up) are
    # sent to stderr by pointing fd 1 at fd 2 for the life of the process; the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # one rank per GPU; BIEM_BENCH_SHARE_GPU=1 (rehearsal on a 1-GPU box only) folds the ranks onto the visible devices
    share = os.environ.get("BIEM_BENCH_SHARE_GPU") == "1"
    if not share and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} HIP device(s) visible")
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # BIEM_BENCH_FORCE_DIST=1: build the process group at world size 1 too (exercises the RCCL barrier / max-reduce / gather on a 1-GPU box)
    if world > 1 or os.environ.get("BIEM_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if share:
            dist.init_process_group(backend="gloo")           # RCCL refuses two ranks on one device
        else:
            dist.init_process_group(backend="nccl", device_id=dev)   # "nccl" is RCCL on ROCm

    import biem_helmholtz_sphere_amd as amd
    from biem_helmholtz_sphere_amd import _biem as impl
    from biem_helmholtz_sphere_amd import _dist as D
    from biem_helmholtz_sphere_amd import _lib as L

    lib = L.load()
    cfg = args.config
    S = args.systems_per_gpu or DEFAULT_SYSTEMS[cfg]
    if args.scaling == "weak":
        n_total, (lo, hi) = S * world, (rank * S, (rank + 1) * S)
    else:
        n_total, (lo, hi) = S, D.shard_bounds(S, rank, world)
    w = workload(cfg, n_total, lo, hi)
    nloc = hi - lo
    t = lambda a, dt=torch.float64: torch.as_tensor(np.array(a), device=dev).to(dt).contiguous()
    c = amd.create_from_branching_types(w["tree"])
    dirs = np.zeros((w["d"], nloc))
    dirs[0] = 1.0
    k_t, eta_t = t(w["ks"]), t(w["etas"])
    uin, ugr = amd.plane_wave(k=k_t, direction=t(dirs))
    kw = dict(centers=t(w["centers"])[None], radii=t(np.ones(w["B"]))[None], n_end=w["n_end"], alpha=w["alpha"], beta=w["beta"])
    if w["beta"] != 0:
        kw["uin_grad"] = ugr

    def step():
        return amd.biem(c, k=k_t, eta=eta_t, uin=uin, chunk=args.chunk, **kw)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    calc = None
    for _ in range(args.warmup):
        calc = step()
    barrier()
    L.check(lib.biem_profile_begin())
    t0 = time.perf_counter()
    for _ in range(args.steps):
        calc = step()
    barrier()
    dt = time.perf_counter() - t0
    ms = (C.c_double * 9)()
    work = (C.c_double * 9)()
    launches = (C.c_longlong * 9)()
    L.check(lib.biem_profile_end(ms, work, launches))
    marshalling_ms, rccl_ranks = None, None
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # output marshalling (outside the timed region): all-gather of the densities, as _dist.biem_sharded does
        rccl_ranks = dist.get_world_size()
        dens = calc.density if not share else calc.density.cpu()
        D.gather_batch(dens, n_total if args.scaling == "strong" else S * world)          # warm-up (communicator set-up)
        barrier()
        tg = time.perf_counter()
        full = D.gather_batch(dens, n_total if args.scaling == "strong" else S * world)
        barrier()
        marshalling_ms = (time.perf_counter() - tg) * 1e3
        assert full.shape[0] == n_total
        del full

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return 0

    systems_per_step = n_total
    stats_timed = dict(impl._last_solve_stats)
    value = systems_per_step * args.steps / dt
    ms, work, launches = list(ms), list(work), list(launches)
    gi, fi = CLASSES.index("gemm"), CLASSES.index("fill")
    alg_tflops = work[gi] / (ms[gi] * 1e-3) / 1e12 if ms[gi] > 0 else None     # 8 real flops per complex multiply-add
    fill_gbs = work[fi] / (ms[fi] * 1e-3) / 1e9 if ms[fi] > 0 else None
    H = harm_count(w["tree"], w["n_end"])
    N = w["B"] * H
    solver = os.environ.get("BIEM_SOLVER", "ldlt")

    # latency of a single system per call (what the reference's own drivers do; cfg 3 / 5 - the first system of the batch - with --single-system), rank 0 only
    single_ms = None
    if rank == 0 and not args.no_single_system and (cfg in (1, 2, 4) or args.single_system):
        u1, g1 = amd.plane_wave(k=k_t[:1], direction=t(dirs[:, :1]))
        kw1 = dict(kw)
        if w["beta"] != 0:
            kw1["uin_grad"] = g1
        for _ in range(3):
            amd.biem(c, k=k_t[:1], eta=eta_t[:1], uin=u1, **kw1)
        torch.cuda.synchronize(dev)
        ts = time.perf_counter()
        for _ in range(10):
            amd.biem(c, k=k_t[:1], eta=eta_t[:1], uin=u1, **kw1)
        torch.cuda.synchronize(dev)
        single_ms = (time.perf_counter() - ts) * 100.0

    # whole-batch accuracy figure (cheap): systems spread over the timed batch solved again by the pivoted LU (the reference's
    # algorithm, _biem.py:797) and compared with the timed run's densities, max over systems of max |d_sym - d_lu| / max |d_lu|
    sym_vs_lu = None
    if solver == "ldlt" and args.sym_vs_lu_systems > 0 and rank == 0:
        pick = sorted(set(int(round(v)) for v in np.linspace(0, nloc - 1, min(args.sym_vs_lu_systems, nloc))))
        pt = torch.as_tensor(pick, device=dev)
        u2, g2 = amd.plane_wave(k=k_t[pt], direction=t(dirs[:, pick]))
        kw2 = dict(kw)
        if w["beta"] != 0:
            kw2["uin_grad"] = g2
        os.environ["BIEM_SOLVER"] = "lu"
        try:
            ref = amd.biem(c, k=k_t[pt], eta=eta_t[pt], uin=u2, **kw2).density
        finally:
            del os.environ["BIEM_SOLVER"]
        got = calc.density[pt]
        num = (got - ref).abs().reshape(len(pick), -1).amax(dim=1)
        den = ref.abs().reshape(len(pick), -1).amax(dim=1)
        sym_vs_lu = {"systems": len(pick), "batch_indices": pick, "max_rel_diff_density": float((num / den).max().item()),
                     "what": "timed run's densities (symmetric path) vs the same systems solved by the pivoted LU: max_s max|d_sym - d_lu| / max|d_lu|"}
        impl._last_solve_stats.update(stats_timed)

    # accuracy of this run's densities vs the CPU oracle at probe points (the metric's second half) + the CPU baseline
    cpu, relerr = None, None
    if not args.no_cpu_baseline and world == 1:     # rank 0 at N = 1 only
        from oracle import biem_oracle as O

        res, idx, cpu = cpu_baseline(w, args.cpu_baseline_systems, args.cpu_baseline_reps, one_thread=True)
        half = float(np.max(np.abs(w["centers"]))) + 1.0
        ang = 2 * np.pi * np.arange(63) / 63
        probes = np.zeros((64, w["d"]))
        probes[1:, 0], probes[1:, 1] = 1.5 * half * np.cos(ang), 1.5 * half * np.sin(ang)
        ug = calc.uscat(t(probes.T.copy())).cpu().numpy()
        relerr = 0.0
        for r, i in zip(res, idx):
            uo = O.uscat(r, probes)
            relerr = max(relerr, float(np.max(np.abs(ug[:, i] - uo) / np.abs(uo))))

    # HBM traffic of the dominant kernel: rocprofv3 PMC passes cannot run inside bench.py; bytes per launch per system from
    # the committed summary, scaled to this run's systems per launch
    def profile_json(name):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            with open(p) as f:
                return json.load(f)
        return None

    traffic, traffic_src = None, None
    tname = next((n for n in ("r03_gemm_traffic.json", "r02_gemm_traffic.json", "r01_gemm_traffic.json") if profile_json(n)), None)
    tj = profile_json(tname) if tname else None
    if tj and cfg == 3:
        key = "bytes_per_launch_per_system" if solver == "ldlt" else "bytes_per_launch_per_system_lu"
        per = min(nloc, args.chunk or nloc)
        traffic = tj[key] * per
        same = solver == "ldlt" and tj.get("systems_per_launch") == per
        traffic_src = (f"profiles/{tname}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs of this bench.py; FETCH_SIZE doubled per the gfx950 "
                       f"correction) at {tj.get('systems_per_launch', 8)} systems per launch"
                       + (" = this run's launch shape" if same else f", scaled to {per} systems per launch") + "; counters cannot be read inside the timed run")
    fname = next((n for n in ("r03_fill_traffic.json", "r02_fill_traffic.json") if profile_json(n)), None)
    fj = profile_json(fname) if fname else None
    fill_traffic = fj["bytes_per_system"] * nloc if fj and cfg == 3 else None

    cen_ = np.asarray(w["centers"], dtype=np.float64)
    disp_ = [tuple((cen_[bp] - cen_[b]).tolist()) for bp in range(len(cen_)) for b in range(bp)]
    pair_classes = {"pairs": len(disp_), "classes": len(set(disp_)) if solver == "ldlt" and not os.environ.get("BIEM_FILL_NO_DEDUPE") else len(disp_)}
    issued = 0.75 * alg_tflops if alg_tflops else None     # 3M: 3 real products per complex multiply-add = 6 of the 8 flops
    # what a pure stream of the same MFMA instruction sustains on THIS box (0.3 s, after the timed region): context for `frac`,
    # which stays priced against the nominal 78.6 TFLOP/s
    sustained = None
    if rank == 0:
        tf = C.c_double()
        L.check(lib.biem_bench_mfma_f64_ex(2000000, 1, C.byref(tf), None))
        sustained = tf.value
    out = {
        "metric": "BIEM systems solved/sec + max |u_scat| rel-err vs NumPy ref",
        "value": value,
        "unit": "systems/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "c128",
        "data": "synthetic",
        "config": {"workload": w["desc"], "config_id": cfg, "N": N, "systems_per_step": systems_per_step, "systems_per_gpu": nloc,
                   "parallelism": f"batch-shard x{world}", "solver": solver, "solved_by": stats_timed},
        "max_rel_err_uscat": relerr,
        "sym_vs_lu": sym_vs_lu,
        "roofline": {
            "bound": "mfma",
            "kernel": "k_gemm3m_pipe<256> (zgemm3m K=256 trailing update of a four-panel group, v_mfma_f64_4x4x4_4b_f64; upper-triangle tiles in the symmetric path)",
            # `achieved` = flops the matrix pipe EXECUTES: the 3M form does a complex multiply-add in 3 real products = 6 real flops
            "achieved": issued, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": issued / FP64_MFMA_PEAK_TFLOPS if issued else None,
            # the same launches priced at the textbook 8 real flops per complex multiply-add (what a 4M zgemm would execute)
            "algorithmic_tflops_8flop": alg_tflops,
            "pure_mfma_stream_tflops": sustained, "frac_of_pure_mfma_stream": issued / sustained if issued and sustained else None,
            "traffic": traffic, "traffic_source": traffic_src,
            "avg_launch_ms": ms[gi] / launches[gi] if launches[gi] else None, "launches": launches[gi],
            "share_of_step": ms[gi] / (dt * 1e3) if dt > 0 else None,
        },
        # (pair classes: ball pairs with the same displacement and the same sphere on either side share their block of the matrix,
        # contracted once and stored to every pair of the class - all radii and Robin coefficients are equal in these workloads)
        "fill": {"bound": "hbm", "pair_classes": pair_classes, "achieved": fill_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fill_gbs / HBM_PEAK_GBS if fill_gbs else None,
                 "bytes_counted": "what the factorisation reads: upper triangle + diagonal 64 x 64 tiles of the symmetric form (default path), or 16 N^2 per system (BIEM_SOLVER=lu)",
                 "traffic": fill_traffic, "traffic_source": f"profiles/{fname} (rocprofv3 --pmc WRITE_SIZE + 2 x FETCH_SIZE at 8 systems, pair classes on), scaled to {nloc} systems" if fill_traffic else None},
        "stage_ms_per_step": {n: m / args.steps for n, m in zip(CLASSES, ms)},
        "single_system_ms": single_ms,
        "marshalling_ms": marshalling_ms, "rccl_ranks": rccl_ranks,
        "cpu_baseline": cpu,
    }
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
