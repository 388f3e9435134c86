#!/bin/bash
# tools/asmstat.sh FILE.hip [extra flags] -- compile one translation unit for gfx950 with the library's flags and print each
# kernel's registers, spills, LDS and scratch (assembly left in /tmp/asm)
R=$(cd "$(dirname "$0")/.." && pwd)
F=$1; shift
mkdir -p /tmp/asm
EXTRA=""
[ "$(basename $F)" = "pdq_fused512.hip" ] && EXTRA="-mllvm -amdgpu-sched-strategy=max-ilp -mllvm -greedy-regclass-priority-trumps-globalness=1"
cd $R/rupphash_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math $EXTRA "$@" -x hip -c $(basename $F) -o /tmp/asm/x.o -save-temps=obj 2>&1 | grep -E "error" | head
S=/tmp/asm/$(basename $F .hip)-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^\s+\.(name|vgpr_count|sgpr_spill_count|private_segment_fixed_size|vgpr_spill_count|group_segment_fixed_size):" $S | paste - - - - - - | sed 's/\s\+/ /g; s/_ZN12_GLOBAL__N_1//'
