#!/bin/bash
# Same-box A/B of the e = 16 FFN forward alone (tools/ffn_probe.py, kernel time by lg_prof): bash tools/xr_ab.sh lib1.so lib2.so ...  ("xs" = the default library under LG_FFN_FWD=xs)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "xs" ]; then LG_FFN_FWD=xs python $ROOT/tools/ffn_probe.py 4 0 32 128 30 2>/dev/null | grep "kernel alone"
    elif [ "$lib" = "default" ]; then python $ROOT/tools/ffn_probe.py 4 0 32 128 30 2>/dev/null | grep "kernel alone"
    else LGTEUN_HIP_LIB=$ROOT/$lib python $ROOT/tools/ffn_probe.py 4 0 32 128 30 2>/dev/null | grep "kernel alone"; fi
  done
done
