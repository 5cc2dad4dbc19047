#!/bin/bash
# builds the diagnostic tool tools/gemm_trace<suffix> (see tools/gemm_trace.cpp); usage: build_trace.sh [suffix] [extra hipcc flags...]
# flags: -DBIEM_TR_STAMPS (in-kernel timeline; perturbs) and the timing ablations -DBIEM_ABL_{NOMFMA,NODMA,NOCDMA,ONLYCDMA,NOLDS,NOSUMS,NOCADD,
# NOBARRIER,NOEPI,NOSTORE,ONESTORE,CHOT} (results wrong by construction, only the launch time is meaningful)
cd "$(dirname "$0")/.." || exit 1
suf="$1"; shift
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DBIEM_GEMM_TRACE "$@" tools/gemm_trace.cpp \
  biem_helmholtz_sphere_amd/csrc/abi.cpp biem_helmholtz_sphere_amd/csrc/plan.cpp biem_helmholtz_sphere_amd/csrc/kernels_fill.hip \
  biem_helmholtz_sphere_amd/csrc/kernels_uscat.hip biem_helmholtz_sphere_amd/csrc/kernels_lu.hip -o tools/gemm_trace$suf 2>/dev/null
