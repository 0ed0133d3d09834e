#!/bin/bash
# A/B between builds of the library in one process (fe_check abl) over a list of sizes:
#   tools/abl_sizes.sh <family> "<E ...>" libA.so libB.so ...   (output: stdout)
fam=$1; sizes=$2; shift 2
for E in $sizes; do
  n=$(( 40000000 / E )); [ $n -gt 400 ] && n=400; [ $n -lt 20 ] && n=20
  timeout -k 10 120 build/fe_check abl $fam $E 9 $n "$@" || exit 1
done
