#!/bin/bash
# Register / LDS / spill summary of every kernel in one HIP source (compile only, no GPU): bash tools/kres.sh lgteun_amd/csrc/k_ffn_bwd.hip [filter]
F=$1; PAT=${2:-.}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fno-slp-vectorize -fPIC -fvisibility=hidden --offload-arch=gfx950 $KRES_FLAGS -c "$F" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/^Function Name/{if(n)print n,v,a,s,o,sp,l; n=$3} /^VGPRs:/{v="vgpr="$2} /^AGPRs:/{a="agpr="$2} /^ScratchSize/{s="scratch="$3} /^Occupancy/{o="occ="$3} /^VGPRs Spill/{sp="spill="$3} /^LDS Size/{l="lds="$4} END{print n,v,a,s,o,sp,l}' | c++filt | grep -E "$PAT" || true
