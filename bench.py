#!/usr/bin/env python
"""bench.py -- BIEM systems solved per second on the BASELINE.json headline configuration.

Workload (BASELINE.json configs[2], the one the metric is quoted on; it fits one GPU):
    d=3 ('ba'), n_balls=16 on the 4x4 grid {-6,-2,2,6}^2 x {0} (reference cli.py:170-185 `_center(2,3)`), radius 1,
    n_end=20 (N = 6400 unknowns per system), sound-soft, eta=1, plane wave along +x0 with the system's own k,
    batch of wavenumbers k in [0.5, 8].
One *step* = one pass of the whole hot path (boundary samples -> RHS projection -> fill -> symmetrise -> L D L^T (pivoted LU
for systems whose diagonal pivots are rejected; BIEM_SOLVER=lu: LU for all) -> density) over this
rank's shard of the batch through the public `biem()` API; inputs are resident in HBM when the clock starts, the
densities are resident in HBM when it stops.  Weak scaling: every GPU owns `--systems-per-gpu` systems (default 256 =
the whole 256-wavenumber batch of the config, which fits one MI355X: 256 x 656 MB of matrices); the k's of the whole job
are linspace(0.5, 8, 256*N), rank r takes the r-th contiguous block.  No collective sits in the data path (independent systems); RCCL is
used only for the barrier / max-reduce of the timing.

Prints ONE JSON line on rank 0 (fields per the driver contract, plus `roofline` and `cpu_baseline`).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X FP64 matrix, vendor spec (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
CLASSES = ["tables", "fill", "rhs", "panel", "swap", "trsm", "gemm", "back", "gemm_small"]


def workload(n_sys_total: int, rank: int, world: int, per_gpu: int, dev):
    half, d = 2, 3
    ax = np.arange(-half, half) * 4.0 + 2.0
    x0, x1 = np.meshgrid(ax, ax, indexing="ij")
    centers = np.stack([x0.ravel(), x1.ravel(), np.zeros(x0.size)], axis=-1)          # [16, 3]
    ks_all = np.linspace(0.5, 8.0, n_sys_total)
    ks = ks_all[rank * per_gpu:(rank + 1) * per_gpu]
    t = lambda a: torch.as_tensor(np.array(a), dtype=torch.float64, device=dev)
    dirs = np.zeros((d, len(ks)))
    dirs[0] = 1.0
    return dict(centers=t(centers)[None], radii=t(np.ones(len(centers)))[None], k=t(ks), eta=t(np.ones(len(ks))),
                direction=t(dirs), ks=ks, centers_np=centers)


def cpu_baseline(n_end: int, centers: np.ndarray, ks):
    """The oracle (CPU restatement of the reference path: materialise the matrix, numpy.linalg.solve) timed on a few
    systems of the same workload on this host's cores, one after the other as the reference would run them.
    Returns the systems' results (for the accuracy check), the total time and the BLAS thread count."""
    from oracle import biem_oracle as O   # checker / baseline only

    O._terms3(n_end)                      # table build is amortised over a sweep: keep it out of the timing
    results = []
    t0 = time.perf_counter()
    for k in ks:
        uin, _ = O.plane_wave(float(k), [1.0, 0.0, 0.0])
        res = O.solve_biem("ba", centers=centers, radii=np.ones(len(centers)), k=float(k), n_end=n_end, eta=1.0, uin=uin)
        results.append(res)
    dt = time.perf_counter() - t0
    res = results
    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    return res, dt, threads


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--systems-per-gpu", type=int, default=256)
    ap.add_argument("--n-end", type=int, default=20)
    ap.add_argument("--chunk", type=int, default=0, help="resident matrices per pass (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # one rank per GPU; BIEM_BENCH_SHARE_GPU=1 (rehearsal on a 1-GPU box only) folds the ranks onto the visible devices
    share = os.environ.get("BIEM_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # BIEM_BENCH_FORCE_DIST=1: build the process group at world size 1 too (exercises the RCCL barrier / max-reduce on a 1-GPU box)
    if world > 1 or os.environ.get("BIEM_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")           # RCCL refuses two ranks on one device
        else:
            dist.init_process_group(backend="nccl", device_id=dev)   # "nccl" is RCCL on ROCm

    import biem_helmholtz_sphere_amd as amd
    from biem_helmholtz_sphere_amd import _biem as impl
    from biem_helmholtz_sphere_amd import _lib as L

    lib = L.load()
    per_gpu = args.systems_per_gpu
    w = workload(per_gpu * world, rank, world, per_gpu, dev)
    c = amd.create_from_branching_types("ba")
    uin, _ = amd.plane_wave(k=w["k"], direction=w["direction"])

    def step():
        return amd.biem(c, centers=w["centers"], radii=w["radii"], k=w["k"], eta=w["eta"], n_end=args.n_end, uin=uin,
                        chunk=args.chunk)

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
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    n_sys = per_gpu * world * args.steps
    value = n_sys / dt
    ms, work, launches = list(ms), list(work), list(launches)
    gi = CLASSES.index("gemm")
    gemm_tflops = work[gi] / (ms[gi] * 1e-3) / 1e12 if ms[gi] > 0 else None
    fi = CLASSES.index("fill")
    fill_gbs = work[fi] / (ms[fi] * 1e-3) / 1e9 if ms[fi] > 0 else None
    N = 16 * args.n_end ** 2
    lu_flops = (8.0 / 3.0) * N ** 3

    # accuracy of this run's densities vs the CPU oracle at probe points (max rel-err, metric's second half)
    cpu = None
    relerr = None
    if not args.no_cpu_baseline and world == 1:     # the CPU baseline and the accuracy check run at N = 1 only
        cpu_ks = [w["ks"][0], w["ks"][-1]] if len(w["ks"]) > 1 else [w["ks"][0]]
        res, cpu_dt, threads = cpu_baseline(args.n_end, w["centers_np"], cpu_ks)
        ang = 2 * np.pi * np.arange(63) / 63
        probes = np.concatenate([np.zeros((1, 3)), np.stack([10.5 * np.cos(ang), 10.5 * np.sin(ang), np.zeros(63)], -1)])
        from oracle import biem_oracle as O

        ug = calc.uscat(torch.as_tensor(probes.T.copy(), dtype=torch.float64, device=dev)).cpu().numpy()
        relerr = 0.0
        for r, col in zip(res, [0, -1]):
            uo = O.uscat(r, probes)
            relerr = max(relerr, float(np.max(np.abs(ug[:, col] - uo) / np.abs(uo))))
        cpu = {"value": len(cpu_ks) / cpu_dt, "unit": "systems/s", "cores": int(threads), "kind": "port",
               "sample": f"{len(cpu_ks)} systems of the workload (k = " + ", ".join(f"{k:.4g}" for k in cpu_ks)
                         + f"), oracle fill + numpy.linalg.solve one after the other, {cpu_dt:.1f} s"}

    # HBM traffic of the dominant kernel per launch: measured once with rocprofv3 PMC passes (cannot run inside bench.py);
    # bytes per launch per system from the committed summary, scaled to this run's systems per launch
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_gemm_traffic.json")
    if os.path.exists(tpath) and args.n_end == 20:
        with open(tpath) as f:
            tj = json.load(f)
            traffic = tj["bytes_per_launch_per_system" if os.environ.get("BIEM_SOLVER", "ldlt") == "ldlt" else "bytes_per_launch_per_system_lu"] * per_gpu

    out = {
        "metric": "BIEM systems solved/sec + max |u_scat| rel-err vs NumPy ref",
        "value": value,
        "unit": "systems/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "c128",
        "data": "synthetic",
        "config": {"workload": f"cfg3: d=3 'ba', n_balls=16 (4x4 grid, pitch 4), n_end={args.n_end}, N={N}, sound-soft, "
                               f"{per_gpu} wavenumbers per GPU from linspace(0.5, 8, {per_gpu * world})",
                   "systems_per_gpu": per_gpu, "parallelism": f"batch-shard x{world}",
                   "solver": os.environ.get("BIEM_SOLVER", "ldlt"), "solved_by": dict(impl._last_solve_stats)},
        "max_rel_err_uscat": relerr,
        "roofline": {
            "bound": "mfma", "kernel": "k_gemm3m_pipe<256> (zgemm3m K=256 trailing update of a four-panel group, v_mfma_f64_4x4x4_4b_f64; lower-triangle tiles in the L D L^T path)",
            "achieved": gemm_tflops, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": (gemm_tflops / FP64_MFMA_PEAK_TFLOPS) if gemm_tflops else None,
            # `achieved` counts the ALGORITHMIC 8 real flops per complex multiply-add; the 3M form issues 6 of them as MFMAs
            # (3 real products instead of 4), so the matrix pipe itself runs at 3/4 of `achieved`:
            "mfma_issued_tflops": 0.75 * gemm_tflops if gemm_tflops else None,
            "mfma_issued_frac": (0.75 * gemm_tflops / FP64_MFMA_PEAK_TFLOPS) if gemm_tflops else None,
            "traffic": traffic,
            "avg_launch_ms": ms[gi] / launches[gi] if launches[gi] else None,
            "lu_effective_tflops": lu_flops * per_gpu * args.steps / (sum(ms[3:9]) * 1e-3) / 1e12 if sum(ms[3:9]) > 0 else None,
        },
        "fill": {"bound": "hbm", "achieved": fill_gbs, "peak": 8000.0, "unit": "GB/s", "frac": fill_gbs / 8000.0 if fill_gbs else None},
        "stage_ms_per_step": {n: m / args.steps for n, m in zip(CLASSES, ms)},
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
