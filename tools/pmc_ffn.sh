#!/bin/bash
# ON THE GPU BOX: SQ counters of the FFN probe (two passes of 8 SQ counters), per-launch means.   usage: bash tools/pmc_ffn.sh <tag> [probe args]
TAG=${1:-run}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/p1 -- python $ROOT/tools/ffn_probe.py "$@" > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS --kernel-trace --output-format csv -d $OUT/p2 -- python $ROOT/tools/ffn_probe.py "$@" > $OUT/p2.log 2>&1
for p in p1 p2; do
  F=$(find $OUT/$p -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && python3 $ROOT/tools/pmc_table.py "$F" ffn
done
tail -1 $OUT/p1.log
