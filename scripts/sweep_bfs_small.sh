#!/bin/bash
# seed-level sweep for small C3 batches (big trees, few items): ms per step against bfs_levels
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for poses in 1 4 16 64 200 256 512; do
  for L in -1 2 3 4 5 6 7 8; do
    python bench.py --poses $poses --steps 20 --warmup 5 --cpu-seconds 0 --no-extras --no-validate --reps 3 --bfs-levels $L 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('poses %4d bfs_levels %2d: %.4f ms/step' % ($poses, $L, d['ms_per_step']))"
  done
done
