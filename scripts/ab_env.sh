#!/bin/bash
# same-box A/B of environment settings: bash ab_env.sh "<env A>" "<env B>" ...
cd ${GRAFT_REPO_ROOT}
run() {
    env $1 timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-extras --no-validate --reps 5 $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s ms_per_step %.3f' % ('$1', d['ms_per_step']), flush=True)"
}
for k in 1 2 3; do for e in "$@"; do run "$e"; done; done
