#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of a short bench.py run; prints, for the LAST traced train step, every launch in order with its
# duration and the gap to the previous kernel's end (what the per-kernel averages hide: which launches of a kernel are the slow ones).
# usage: bash tools/trace_step.sh <tag> [extra bench.py args]   -> gpurun_out/trace_<tag>/step.txt
TAG=${1:-run}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-live --prof-kernel none "$@" > $OUT/run.log 2>&1
F=$(find $OUT/raw -name '*kernel_trace.csv' | head -1)
python3 - "$F" > $OUT/step.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# a train step ends with the Adam kernel: take the launches between the last two of them
adam = [i for i, n in enumerate(names) if 'adam' in n.lower()]
lo, hi = (adam[-2] + 1, adam[-1] + 1) if len(adam) >= 2 else (0, len(rows))
prev_end = None
tot = 0
for r in rows[lo:hi]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = max(e, prev_end or e)
    nm = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '')).replace('void ', '')
    print('%8.1f us  gap %6.1f  %s' % ((e - s) / 1e3, gap, nm[:70]))
    tot += e - s
print('launches', hi - lo, 'kernel time %.3f ms' % (tot / 1e6), 'span %.3f ms' % ((int(rows[hi-1]['End_Timestamp']) - int(rows[lo]['Start_Timestamp'])) / 1e6))
PY
tail -1 $OUT/step.txt
rm -rf $OUT/raw
