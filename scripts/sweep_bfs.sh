#!/bin/bash
# sweep the number of level-synchronous broadphase levels before the depth-first kernel
for L in "$@"; do
  python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --bfs-levels $L 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bfs_levels', $L, 'ms/step %.3f' % d['ms_per_step'], 'broadphase %.3f' % d['stage_ms_per_step']['broadphase'], 'narrow %.3f' % d['stage_ms_per_step']['narrowphase'], 'bristle %.3f' % d['stage_ms_per_step']['bristle'])"
done
