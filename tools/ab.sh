#!/bin/bash
# Same-box A/B of library builds (boxes differ by several per cent, so only alternating runs inside ONE gpurun call compare):
#   [AB_ROUNDS=n] [AB_ARGS='--config c3'] bash tools/ab.sh [--prof-kernel K] libA.so libB.so [...]      (paths relative to the repo root; 3 alternating rounds)
# prints ms/step and the live-timed kernel's average per run.  Build variants with:  make LIB=build_variants/x.so FLAGS+=-DSOMETHING
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PK=ffn
if [ "$1" = "--prof-kernel" ]; then PK=$2; shift 2; fi
for round in $(seq 1 ${AB_ROUNDS:-3}); do
  for lib in "$@"; do
    LGTEUN_HIP_LIB=$ROOT/$lib python $ROOT/bench.py --no-cpu-baseline --no-live --steps 30 --warmup 5 --prof-kernel $PK $AB_ARGS 2>/dev/null | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'ms/step', d['ms_per_step'], 'kernel_us', d['roofline']['avg_launch_us'])"
  done
done
