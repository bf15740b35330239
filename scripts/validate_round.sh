#!/bin/bash
# End-of-round parity at scale with the final kernels -> gpurun_out/<tag>_validation.txt   (bash scripts/validate_round.sh [tag])
T=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
O=$R/gpurun_out/${T}_validation.txt
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%s: %.3f ms per step, %.3e ops/s, path: %s; %d oracle-checked items: %s' % ('$1', d['ms_per_step'], d['value'], d['path'], d['validated_items'], d['validation']))"; }
{
timeout -k 10 400 python bench.py --validate 96 --cpu-seconds 0 --no-extras 2>/dev/null | line "C3, 8 192 poses"
timeout -k 10 400 python bench.py --poses 32768 --validate 64 --cpu-seconds 0 --no-extras --steps 4 2>/dev/null | line "C3, 32 768 poses"
timeout -k 10 500 python bench.py --poses 65536 --validate 16 --cpu-seconds 0 --no-extras --steps 3 --reps 2 2>/dev/null | line "C3, 65 536 poses"
timeout -k 10 300 python bench.py --poses 700 --validate 48 --cpu-seconds 0 --no-extras 2>/dev/null | line "C3, 700 poses (one launch sequence, 512-thread broadphase workgroups)"
timeout -k 10 300 python bench.py --config C5 --steps 50 --validate 400 --cpu-seconds 0 --no-extras 2>/dev/null | line "C5 (sparse pile: one launch sequence, 512-thread broadphase workgroups)"
timeout -k 10 300 python bench.py --config C4 --steps 50 --validate 256 --cpu-seconds 0 --no-extras 2>/dev/null | line "C4"
timeout -k 10 900 python scripts/extended_fuzz.py 30 2>&1 | tail -1
} > $O 2>&1
cat $O
