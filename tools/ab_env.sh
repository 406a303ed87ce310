#!/bin/bash
# Same-box A/B of ENVIRONMENT switches with the library as built: bash tools/ab_env.sh "LG_FFN_SAVE=5" "LG_FFN_SAVE=3" ...   (3 alternating rounds)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for e in "$@"; do
    env $e python $ROOT/bench.py --no-cpu-baseline --no-live --steps 30 --warmup 5 $AB_ARGS 2>/dev/null | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', 'ms/step', d['ms_per_step'], 'kernel_us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
  done
done
