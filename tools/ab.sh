#!/bin/bash
# tools/ab.sh LIB_A LIB_B [bench.py args...] -- alternate two builds of librupphash_hip.so on the SAME GPU box (boxes differ by
# several percent, so kernel variants are only comparable within one call).  Build a variant with
#   make -C rupphash_amd/csrc OUT=$PWD/gpurun_in/libA.so BUILD=build_a EXTRA_ALL="-DSOMETHING"
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in $A $B; do
    RPH_LIB_PATH=$L python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-reference-cases "$@" > /tmp/ab.json 2>/dev/null
    echo "$(basename $L): $(python tools/show_bench.py /tmp/ab.json)"
  done
done
