#!/bin/bash
# ON THE GPU BOX: SQ counters of any probe script (two passes of 8 SQ counters), per-launch means of the kernels matching <filter>.
# usage: bash tools/pmc_probe.sh <tag> <filter> <script.py> [args]
TAG=$1; FIL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/p1 -- python $ROOT/$@ > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS --kernel-trace --output-format csv -d $OUT/p2 -- python $ROOT/$@ > $OUT/p2.log 2>&1
for p in p1 p2; do
  F=$(find $OUT/$p -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && python3 $ROOT/tools/pmc_table.py "$F" $FIL
done
rm -rf $OUT/p1/*/ $OUT/p2/*/ 2>/dev/null
