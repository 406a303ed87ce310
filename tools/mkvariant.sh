#!/bin/bash
# Link a variant of the library with ONE source recompiled under extra flags (same-box A/B, tools/ab.sh):
#   bash tools/mkvariant.sh <name> <source.hip> <extra flags...>   -> build_variants/<name>.so
set -e
NAME=$1; SRC=$2; shift 2
mkdir -p build_variants
FLAGS="-O3 -std=c++17 -fno-slp-vectorize -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value"
/opt/rocm/bin/hipcc $FLAGS "$@" -c lgteun_amd/csrc/$SRC -o build_variants/$NAME.o
OBJS=$(ls lgteun_amd/csrc/*.o | grep -v "\.ab\.o" | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS build_variants/$NAME.o -o build_variants/$NAME.so
echo build_variants/$NAME.so
