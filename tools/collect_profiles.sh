#!/bin/bash
# Runs ON THE GPU BOX (through gpurun) from the repo root: kernel-trace stats + the two HBM PMC passes of bench.py.
# usage: bash tools/collect_profiles.sh <tag> [extra bench.py args, e.g. --config c3]      -> gpurun_out/prof_<tag>/{stats,fetch,write}
# rocprofv3 gets `python ...` directly after `--` and --pmc is never combined with other trace domains (pool rules).
set -e
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live "$@" > $OUT/write.log 2>&1
echo done > $OUT/done.txt
