#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DBIEM_GEMM_TRACE "$@" tools/gemm_small.cpp \
  biem_helmholtz_sphere_amd/csrc/abi.cpp biem_helmholtz_sphere_amd/csrc/plan.cpp biem_helmholtz_sphere_amd/csrc/kernels_fill.hip \
  biem_helmholtz_sphere_amd/csrc/kernels_uscat.hip biem_helmholtz_sphere_amd/csrc/kernels_lu.hip -o tools/gemm_small 2>&1 | grep -v warning | head
