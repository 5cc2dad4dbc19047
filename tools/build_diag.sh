#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DBIEM_DIAG_TRACE "$@" tools/diag_trace.cpp \
  biem_helmholtz_sphere_amd/csrc/abi.cpp biem_helmholtz_sphere_amd/csrc/plan.cpp biem_helmholtz_sphere_amd/csrc/kernels_fill.hip \
  biem_helmholtz_sphere_amd/csrc/kernels_uscat.hip biem_helmholtz_sphere_amd/csrc/kernels_lu.hip -o tools/diag_trace$SUF 2>&1 | grep -E "error" | head
