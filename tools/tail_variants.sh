#!/bin/bash
# Rebuilds the library on the GPU box with variants of the dynamic walk and times each with tools/tail_ab.py (one process each):
# number of ticket pools (default: "8 32" against the committed 16).   bash tools/tail_variants.sh "8 32" > gpurun_out/.../variants.txt
set -e
run() { echo "== $1"; for w in grad div; do python3 tools/tail_ab.py $w 1000000 "-1 1000" 2>&1 | grep "allocat"; done; }
build() { rm -f feinsum_amd/libfeinsum_hip.so; python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1; }
cp feinsum_amd/csrc/fe_common.h /tmp/fe_common.h.keep
run "16 pools (as committed)"
for n in ${1:-8 32}; do
  sed -i "s/^constexpr int kTailPools = [0-9]*;/constexpr int kTailPools = $n;/" feinsum_amd/csrc/fe_common.h; build; run "$n pools"
done
cp /tmp/fe_common.h.keep feinsum_amd/csrc/fe_common.h; build
