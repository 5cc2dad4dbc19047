"""Build libbiem_mi355.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library carries a hash of the sources it was built from (`biem_build_id`); `is_stale()` compares it with the checkout's,
so an edited `csrc/` is never run against an old binary (file times do not survive a copy to another machine, the hash does).
After the link the gfx950 code object is disassembled and the trailing-update kernels are checked (`check_isa`): their LDS-DMA
ring counts vector-memory instructions by hand (`s_waitcnt vmcnt(N)`), which is only right while the compiler adds none.
"""
from __future__ import annotations

import hashlib
import os
import re
import shutil
import subprocess
import tempfile
from collections import Counter

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbiem_mi355.so")
SOURCES = ["abi.cpp", "plan.cpp", "kernels_fill.hip", "kernels_uscat.hip", "kernels_lu.hip"]
HEADERS = ["common.hpp", "plan.hpp", "special.hpp", os.path.join("..", "..", "include", "biem_mi355.h")]
_MARK = b"BIEM_SRC_HASH="


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def _llvm_tool(name: str) -> str | None:
    for cand in (shutil.which(name), os.path.join("/opt/rocm/lib/llvm/bin", name), os.path.join("/opt/rocm/llvm/bin", name)):
        if cand and os.path.exists(cand):
            return cand
    return None


def source_hash() -> str:
    """sha256 over the translation units, the headers and the extra compiler flags (first 16 hex digits)."""
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read() + b"\0")
    h.update(os.environ.get("BIEM_HIPCC_FLAGS", "").encode())
    return h.hexdigest()[:16]


def built_hash(path: str = LIB) -> str | None:
    """The source hash stored in a built library (read from the file, without loading it)."""
    try:
        with open(path, "rb") as fh:
            blob = fh.read()
    except OSError:
        return None
    i = blob.find(_MARK)
    if i < 0:
        return None
    j = blob.find(b";", i)
    return blob[i + len(_MARK):j].decode(errors="replace") if j > 0 else None


def is_stale() -> bool:
    return built_hash() != source_hash()


# ------------------------------------------------------------------------------------------------
# ISA check of the trailing-update kernels (CPU side: llvm-objdump / llvm-readelf on the built library)
# ------------------------------------------------------------------------------------------------
# What the check pins for every k_gemm3m_pipe<KD> (whole kernel; the compiler places cold paths of the chunk loop outside its
# address range, so per-loop counts are not stable):
#   * no scratch, no VGPR spills (either would add vector-memory instructions the vmcnt waits do not count);
#   * exactly 96 MFMAs, all inside one loop (3 real products x 2 k4-steps x 16 accumulators per chunk);
#   * the only vector-memory instructions are global_load_lds_dwordx4 (the ring), global_store_dwordx4 (48 = three epilogue
#     forms x 16 result stores, outside the chunk loop) and at most 2 global_load_dword (the triangular tile map, read when the
#     producer moves to its next tile; hipcc waits vmcnt(0) for it, which only drains the ring early);
#   * the static number of LDS-DMA instructions equals the count of the reviewed build (prologue + fused groups with / without a
#     C unit + the clamped edge-tile group).  A different count means the compiler restructured the groups: re-derive the
#     vmcnt(N) constants of kernels_lu.hip against the new disassembly before changing the numbers here.
EXPECTED_LDS_DMA = {64: 34, 128: 28, 192: 32, 256: 32}
EXPECTED_STORES = 48
MFMA_PER_CHUNK = 96
MAX_TILE_MAP_LOADS = 2


class IsaCheckError(RuntimeError):
    pass


def _kernel_meta(readelf: str, obj: str) -> dict:
    out = subprocess.run([readelf, "--notes", obj], check=True, capture_output=True, text=True).stdout
    metas, cur = {}, {}
    for line in out.splitlines():
        m = re.match(r"\s+-?\s*\.(name|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|vgpr_count):\s+(\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key in cur:           # a new kernel record starts when a key repeats
            if "name" in cur:
                metas[cur["name"]] = cur
            cur = {}
        cur[key] = val
    if "name" in cur:
        metas[cur["name"]] = cur
    return metas


def _disassemble(objdump: str, obj: str) -> dict:
    out = subprocess.run([objdump, "-d", obj], check=True, capture_output=True, text=True).stdout
    kernels, name = {}, None
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            name = m.group(1)
            kernels[name] = []
            continue
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and name is not None:
            kernels[name].append((int(m.group(3), 16), m.group(1), m.group(2)))
    return kernels


def _chunk_loop(ins):
    """Instructions of the smallest backward-branch loop that holds every MFMA of the kernel."""
    n_mfma = sum(1 for _, op, _ in ins if op.startswith("v_mfma"))
    best = None
    for a, op, args in ins:
        if not (op.startswith("s_cbranch") or op == "s_branch"):
            continue
        if not re.fullmatch(r"-?\d+", args.strip()):
            continue
        off = int(args)
        if off >= 32768:
            off -= 65536
        tgt = a + 4 + 4 * off
        if tgt >= a:
            continue
        body = [x for x in ins if tgt <= x[0] <= a]
        if sum(1 for _, o, _ in body if o.startswith("v_mfma")) == n_mfma and (best is None or len(body) < len(best)):
            best = body
    return best, n_mfma


def check_isa(lib_path: str = LIB, verbose: bool = False) -> dict:
    """Disassemble the gfx950 code objects of `lib_path`; raise IsaCheckError unless every k_gemm3m_pipe<*> has no scratch,
    no VGPR spills, 96 MFMAs in its chunk loop and only the expected vector-memory instructions there.  Returns the report."""
    objdump, readelf = _llvm_tool("llvm-objdump"), _llvm_tool("llvm-readelf")
    if not objdump or not readelf:
        raise IsaCheckError("llvm-objdump / llvm-readelf not found: cannot check the code object")
    report = {}
    with tempfile.TemporaryDirectory(prefix="biem_isa_") as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)
        subprocess.run([objdump, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        objs = sorted(f for f in os.listdir(tmp) if "gfx950" in f)
        if not objs:
            raise IsaCheckError("no gfx950 code object found in " + lib_path)
        for o in objs:
            path = os.path.join(tmp, o)
            metas = _kernel_meta(readelf, path)
            gemm = [n for n in metas if "k_gemm3m_pipe" in n]
            if not gemm:
                continue
            dis = _disassemble(objdump, path)
            for name in gemm:
                kd = int(re.search(r"k_gemm3m_pipeILi(\d+)E", name).group(1))
                meta, ins = metas[name], dis[name]
                loop, n_mfma = _chunk_loop(ins)
                if loop is None:
                    raise IsaCheckError(f"k_gemm3m_pipe<{kd}>: no loop holds the kernel's MFMAs")
                ops = Counter(op for _, op, _ in ins)
                lops = Counter(op for _, op, _ in loop)
                vm = {op: c for op, c in ops.items() if op.startswith(("global_", "buffer_", "flat_", "scratch_"))}
                rep = {"scratch_bytes": int(meta.get("private_segment_fixed_size", -1)), "vgpr_spills": int(meta.get("vgpr_spill_count", -1)),
                       "sgpr_spills": int(meta.get("sgpr_spill_count", -1)), "vgprs": int(meta.get("vgpr_count", -1)),
                       "mfma_in_chunk_loop": sum(c for op, c in lops.items() if op.startswith("v_mfma")), "mfma_total": n_mfma,
                       "vm": vm, "lane_spill_ops_in_chunk_loop": lops.get("v_readlane_b32", 0) + lops.get("v_writelane_b32", 0)}
                report[kd] = rep
                bad = []
                if rep["scratch_bytes"] != 0:
                    bad.append(f"scratch {rep['scratch_bytes']} B")
                if rep["vgpr_spills"] != 0:
                    bad.append(f"{rep['vgpr_spills']} VGPR spills")
                if rep["mfma_in_chunk_loop"] != MFMA_PER_CHUNK or n_mfma != MFMA_PER_CHUNK:
                    bad.append(f"{rep['mfma_in_chunk_loop']} MFMAs in the chunk loop, {n_mfma} in the kernel (expected {MFMA_PER_CHUNK})")
                extra = {op: c for op, c in vm.items() if op not in ("global_load_lds_dwordx4", "global_load_dword", "global_store_dwordx4")}
                if extra:
                    bad.append(f"unexpected vector-memory instructions: {extra}")
                if vm.get("global_load_lds_dwordx4", 0) != EXPECTED_LDS_DMA[kd]:
                    bad.append(f"{vm.get('global_load_lds_dwordx4', 0)} LDS-DMA instructions (reviewed build: {EXPECTED_LDS_DMA[kd]})")
                if vm.get("global_store_dwordx4", 0) != EXPECTED_STORES:
                    bad.append(f"{vm.get('global_store_dwordx4', 0)} result stores (reviewed build: {EXPECTED_STORES})")
                if vm.get("global_load_dword", 0) > MAX_TILE_MAP_LOADS:
                    bad.append(f"{vm.get('global_load_dword', 0)} global_load_dword (expected <= {MAX_TILE_MAP_LOADS})")
                if bad:
                    raise IsaCheckError(f"k_gemm3m_pipe<{kd}>: the hand-counted vmcnt scheme is not safe with this code object: " + "; ".join(bad))
    if set(report) != set(EXPECTED_LDS_DMA):
        raise IsaCheckError(f"k_gemm3m_pipe instances found: {sorted(report)}, expected {sorted(EXPECTED_LDS_DMA)}")
    if verbose:
        for kd in sorted(report):
            print(f"isa check k_gemm3m_pipe<{kd}>: {report[kd]}")
    return report


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit into one shared library (written next to it, then renamed over it); returns its path."""
    if not force and not is_stale():
        return LIB
    tmp = LIB + f".tmp{os.getpid()}"
    cmd = [_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", f'-DBIEM_SRC_HASH="{source_hash()}"', "-o", tmp]
    cmd += os.environ.get("BIEM_HIPCC_FLAGS", "").split()      # experiments only
    cmd += [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.run(cmd, check=True, cwd=CSRC)
        if os.environ.get("BIEM_SKIP_ISA_CHECK") != "1":       # (timing-ablation builds change the instruction counts on purpose)
            check_isa(tmp, verbose=verbose)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
