#!/bin/bash
# batch-size sweep of the split narrowphase (clip-only kernel + k_integ) against the one-kernel form
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for poses in 128 256 512 1024 2048 4096 8192; do
  for cm in 0 1; do
    python bench.py --poses $poses --clip-min $cm --cpu-seconds 0 --no-extras --no-validate --reps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('poses $poses clip_min $cm: %.3f ms/step  %.3e ops/s  np %.3f' % (d['ms_per_step'], d['value'], d['stage_ms_per_step']['narrowphase']))"
  done
done
