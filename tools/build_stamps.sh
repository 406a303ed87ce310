#!/bin/bash
# Diagnostic build: the same sources with -DLG_STAMPS (in-kernel s_memtime phase stamps) -> lgteun_amd/_lgteun_hip_stamps.so.
# Use:  LGTEUN_HIP_LIB=$PWD/lgteun_amd/_lgteun_hip_stamps.so python tools/ffn_stamps.py
set -e
cd "$(dirname "$0")/.."
mkdir -p build/stamps
FLAGS="-O3 -std=c++17 -fno-slp-vectorize -fPIC -fvisibility=hidden --offload-arch=gfx950 -DLG_STAMPS -DLG_BUILD_AB=1 $STAMP_FLAGS -Wno-unused-value -Wno-unused-function"
OBJS=""
for f in lgteun_amd/csrc/*.hip; do
  o=build/stamps/$(basename ${f%.hip}).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ lgteun_amd/csrc/split_bf16.h -nt $o ]; then /opt/rocm/bin/hipcc $FLAGS -c $f -o $o & fi
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -o lgteun_amd/_lgteun_hip_stamps.so
