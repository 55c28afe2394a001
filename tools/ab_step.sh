#!/bin/bash
# same-box A/B of the t2v step: ab_step.sh "<ENV=a> <ENV=b> ..." [reps]  — boxes of the pool differ by +-5 %, so variants
# are only ever compared inside one gpurun call, interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
REPS=${2:-2}
for r in $(seq $REPS); do
  for v in $1; do
    env $v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-vae --no-e2e --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'])"
  done
done
