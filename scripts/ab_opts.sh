#!/bin/bash
# Same-box A/B of bench.py argument sets (library options via --opt), three alternating passes.
# usage (GPU box): bash scripts/ab_opts.sh "<args A>" "<args B>" ...
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for k in 1 2 3; do
  for a in "$@"; do
    timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-extras --no-validate --reps 5 $a 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d.get('stage_ms_per_step')
print('%-44s ms_per_step %.4f  bp %.3f np %.3f br %.3f' % ('$a', d['ms_per_step'], s['broadphase'], s['narrowphase'], s['bristle']), flush=True)"
  done
done
