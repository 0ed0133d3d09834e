#!/bin/bash
# Rebuilds the library on the GPU box with variants of the dynamic tail and times each with tools/tail_ab.py (one process each):
# number of ticket pools, priority toggling in the dynamic walk.   bash tools/tail_variants.sh > gpurun_out/.../variants.txt
set -e
run() { echo "== $1"; python3 tools/tail_ab.py grad 1000000 "-1 12 1000" 2>&1 | grep "split allocator"; }
build() { rm -f feinsum_amd/libfeinsum_hip.so; python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1; }
cp feinsum_amd/csrc/fe_common.h /tmp/fe_common.h.keep; cp feinsum_amd/csrc/fe_grad.h /tmp/fe_grad.h.keep
run "8 pools (as committed)"
sed -i 's/constexpr int kTailPools = 8;/constexpr int kTailPools = 16;/' feinsum_amd/csrc/fe_common.h; build; run "16 pools"
sed -i 's/constexpr int kTailPools = 16;/constexpr int kTailPools = 4;/' feinsum_amd/csrc/fe_common.h; build; run "4 pools"
cp /tmp/fe_common.h.keep feinsum_amd/csrc/fe_common.h
sed -i 's/                balance_priority(younger_half, iteration++);   \/\/ dyn/                iteration++;/' feinsum_amd/csrc/fe_grad.h; build; run "8 pools, no priority toggling"
cp /tmp/fe_grad.h.keep feinsum_amd/csrc/fe_grad.h
