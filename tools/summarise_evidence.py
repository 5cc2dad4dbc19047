#!/usr/bin/env python
"""Turn the output of tools/collect_evidence.sh (gpurun_out/ev/) into the committed evidence files under profiles/ (prefix r03_).

    python tools/summarise_evidence.py [gpurun_out/ev] [profiles]
"""
import csv
import json
import os
import re
import shutil
import sys

ev = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ev"
out = sys.argv[2] if len(sys.argv) > 2 else "profiles"
P = "r03_"


def line_of(path):
    with open(path) as f:
        rows = [l for l in f if l.startswith("{")]
    return rows[-1] if rows else None


def copy_line(src, dst):
    p = os.path.join(ev, src)
    if os.path.exists(p) and line_of(p):
        with open(os.path.join(out, P + dst), "w") as f:
            f.write(line_of(p))
        return json.loads(line_of(p))
    return None


main = copy_line("bench_cfg3.json", "bench_cfg3_256sys.json")
copy_line("bench_under_rocprof.json", "bench_under_rocprof.json")
for c in (1, 2, 4, 5):
    copy_line(f"bench_cfg{c}.json", f"bench_cfg{c}.json")
copy_line("bench_cfg3_cpu5.json", "bench_cfg3_cpu5.json")
copy_line("bench_cfg3_lu.json", "bench_cfg3_256sys_lu_only.json")
copy_line("bench_2ranks_weak_shared_gpu.json", "bench_2ranks_weak_shared_gpu.json")
copy_line("bench_2ranks_strong_shared_gpu.json", "bench_2ranks_strong_shared_gpu.json")
copy_line("bench_rccl_1rank.json", "bench_rccl_1rank.json")
copy_line("bench_rccl_1rank_cfg5.json", "bench_rccl_1rank_cfg5.json")
copy_line("bench_cfg3_no_pair_classes.json", "bench_cfg3_256sys_no_pair_classes.json")
for c in (4, 5):
    copy_line(f"bench_cfg{c}_no_pair_classes.json", f"bench_cfg{c}_no_pair_classes.json")
rows_ = []
for sz in (32, 64, 128):
    jj = copy_line(f"bench_cfg3_{sz}sys.json", f"bench_cfg3_{sz}sys.json")
    if jj:
        rows_.append((sz, jj["value"], jj["ms_per_step"]))
if rows_ and main:
    with open(os.path.join(out, P + "throughput_vs_systems_per_gpu.txt"), "w") as f_:
        f_.write("cfg 3 on one GPU at the per-rank batch of the strong-scaling split (8 / 4 / 2 / 1 GPUs): systems per step, systems/s, ms per step, share of the full-batch rate\n")
        for sz, v, ms_ in rows_ + [(256, main["value"], main["ms_per_step"])]:
            f_.write(f"{sz:4d} {v:8.1f} {ms_:9.1f} {100 * v / main['value']:6.1f} %\n")

ks = os.path.join(ev, "kt", "p_kernel_stats.csv")
if os.path.exists(ks):
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(out, P + "kernel_stats_cfg3_256sys.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:32]:
            w.writerow([r["Name"].split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

ps = os.path.join(ev, "pmc_summary.txt")
if os.path.exists(ps):
    shutil.copy(ps, os.path.join(out, P + "pmc_summary.txt"))
    txt = open(ps).read()

    def counter(kernel, name):
        # first block whose header line starts with `kernel` and that holds `name`
        for m in re.finditer(r"^(\S.*?) dispatches (\d+)\n((?:   .*\n)+)", txt, flags=re.M):
            if m.group(1).split("(")[0].strip().endswith(kernel) or kernel in m.group(1):
                mm = re.search(r"^\s+" + name + r"\s+(\S+)$", m.group(3), flags=re.M)
                if mm:
                    return float(mm.group(1)), int(m.group(2))
        return None, None

    nsys = 8
    fetch, nl = counter("k_gemm3m_pipe<256>", "FETCH_SIZE")
    write, _ = counter("k_gemm3m_pipe<256>", "WRITE_SIZE")
    keep = os.path.join(out, P + "gemm_traffic.json")
    if os.path.exists(keep) and json.load(open(keep)).get("systems_per_launch") == 256:
        fetch = None          # the traffic file measured at the timed launch shape (256 systems) stays
    if fetch and write:
        tiles = sum(t * (t + 1) // 2 for t in range(96, 0, -4))          # upper-triangle tiles of the 24 K = 256 updates at N = 6400
        alg = tiles * 64 * 64 * 16 * 2 / nl                                 # C read + written once, per launch per system (average)
        j = {"kernel": "k_gemm3m_pipe<256> (upper-triangle tile order of the row-form symmetric path)",
             "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --systems-per-gpu 8 --steps 1 --warmup 0; "
                       f"sums over the {nl} K = 256 launches in profiles/{P}pmc_summary.txt",
             "fetch_bytes_corrected": 2 * fetch * 1024, "write_bytes": write * 1024, "launches": nl, "systems_per_launch": nsys,
             "bytes_per_launch_per_system": (2 * fetch + write) * 1024 / nl / nsys,
             "algorithmic_bytes_per_launch_per_system": alg,
             "note": "FETCH_SIZE (KiB) doubled per the guide's gfx950 correction; averaged over the K = 256 updates of one N = 6400 factorisation"}
        old = os.path.join(out, "r01_gemm_traffic.json")
        if os.path.exists(old):
            j["bytes_per_launch_per_system_lu"] = json.load(open(old)).get("bytes_per_launch_per_system_lu")
            j["note_lu"] = "general (square) tile order of the pivoted LU path, unchanged since round 1 (profiles/r01_v8_gemm3m_pipe_pmc.txt)"
        json.dump(j, open(os.path.join(out, P + "gemm_traffic.json"), "w"), indent=1)
        print("gemm traffic ratio", j["bytes_per_launch_per_system"] / alg)
    fw, _ = counter("k_fill_red<", "WRITE_SIZE")
    ff, _ = counter("k_fill_red<", "FETCH_SIZE")
    dw, _ = counter("k_fill_sym_diag", "WRITE_SIZE")
    pw, _ = counter("k_pair_tables", "WRITE_SIZE")
    if fw:
        j = {"kernels": "k_pair_tables + k_fill_red + k_fill_sym_diag (symmetric fill of the default path)",
             "source": f"rocprofv3 --pmc passes at {nsys} systems, profiles/{P}pmc_summary.txt",
             "write_bytes_per_system": (fw + (dw or 0) + (pw or 0)) * 1024 / nsys, "fetch_bytes_per_system_corrected": 2 * (ff or 0) * 1024 / nsys,
             "bytes_per_system": ((fw + (dw or 0) + (pw or 0)) + 2 * (ff or 0)) * 1024 / nsys,
             "needed_bytes_per_system": 16.0 * 64 * 64 * 100 * 101 / 2}
        json.dump(j, open(os.path.join(out, P + "fill_traffic.json"), "w"), indent=1)
        print("fill traffic", j)
if main:
    print("headline", main["value"], main["ms_per_step"], main["stage_ms_per_step"], main["roofline"]["achieved"], main["roofline"]["frac"])
