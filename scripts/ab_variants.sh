#!/bin/bash
# Same-box A/B of the product library against variant builds (scripts/mkvar.sh): boxes differ by ~2 % among themselves, so
# only runs of ONE gpurun call are compared; three alternating passes.  Extra bench.py arguments in $ARGS.
# usage (GPU box): [ARGS="--poses 4096"] bash scripts/ab_variants.sh <name> [<name> ...]
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() {
    timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-extras --no-validate --reps 5 $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d.get('stage_ms_per_step')
print('%-10s $ARGS ms_per_step %.3f  bp %.2f np %.2f br %.2f' % ('$1', d['ms_per_step'], s['broadphase'], s['narrowphase'], s['bristle']), flush=True)"
}
for k in 1 2 3; do
  unset PFC_LIB; run product
  for v in "$@"; do export PFC_LIB=$PWD/build/variants/$v.so; run $v; done
done
