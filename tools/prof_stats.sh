#!/bin/bash
# Runs ON THE GPU BOX: kernel-trace stats of a short bench.py run -> gpurun_out/prof_<tag>/stats ; prints the top kernels.
# usage: bash tools/prof_stats.sh <tag> [extra bench.py args]
TAG=${1:-run}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-live "$@" > $OUT/stats.log 2>&1
F=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
python3 - "$F" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:28]:
    print('%-90s n=%5s avg=%8.1f us  %5.1f%%'%(r['Name'][:90],r['Calls'],float(r['AverageNs'])/1e3,100*float(r['TotalDurationNs'])/tot))
PY
