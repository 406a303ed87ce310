#!/bin/bash
# Link a variant of the library with ONE source recompiled under extra flags (same-box A/B, tools/ab.sh):
#   bash tools/mkvariant.sh <name> <source.hip> <extra flags...>   -> build_variants/<name>.so
set -e
NAME=$1; SRC=$2; shift 2
mkdir -p build_variants
FLAGS="-O3 -std=c++17 -fno-slp-vectorize -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value"
# the per-file flags of the Makefile (a variant built without them is not comparable: the first 5 : 3 quad A/B of k_attn_m was run that way and had to be repeated)
case "$SRC" in k_attn_m.hip|k_attn_bwd_m.hip) FLAGS="$FLAGS -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans";; k_ffn_xr.hip) FLAGS="$FLAGS -mllvm -amdgpu-sched-strategy=iterative-ilp";; esac
/opt/rocm/bin/hipcc $FLAGS "$@" -c lgteun_amd/csrc/$SRC -o build_variants/$NAME.o
OBJS=$(ls lgteun_amd/csrc/*.o | grep -v "\.ab\.o" | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS build_variants/$NAME.o -o build_variants/$NAME.so
echo build_variants/$NAME.so
