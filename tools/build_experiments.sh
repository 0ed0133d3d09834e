#!/bin/bash
# Diagnostic build of the library (-DFE_EXPERIMENTS: skip-experiments, per-wave time stamps, the
# in-kernel clock) and a copy of fe_check linked against it.  Never shipped, never loaded by the
# package:  build/libfeinsum_hip_exp.so, build/fe_check_exp
set -e
cd "$(dirname "$0")/.."
mkdir -p build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -Wl,-Bsymbolic -DFE_EXPERIMENTS \
    feinsum_amd/csrc/feinsum_hip.hip -o build/libfeinsum_hip_exp.so
hipcc -O2 -std=c++17 tools/fe_check.cpp -Lbuild -l:libfeinsum_hip_exp.so -Wl,-rpath,'$ORIGIN' -ldl -o build/fe_check_exp
ls -la build/libfeinsum_hip_exp.so build/fe_check_exp
