#!/bin/bash
# The soak set of a round's final library -> gpurun_out/<tag>_final_soak.txt   (bash scripts/final_soak.sh [tag]; GPU box, ~3 min)
T=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
O=$R/gpurun_out/${T}_final_soak.txt
{
timeout -k 10 300 python scripts/soak.py 200 2>&1 | tail -1
timeout -k 10 300 python scripts/soak.py 60 big 2>&1 | tail -1
PFC_SOAK_POISON=1 timeout -k 10 300 python scripts/soak.py 100 2>&1 | tail -1
timeout -k 10 300 python scripts/soak.py 150 reg 2>&1 | tail -1
timeout -k 10 300 python scripts/soak_pile.py 120 2>&1 | tail -1
timeout -k 10 300 python scripts/soak_options.py 150 2>&1 | tail -1
timeout -k 10 300 python scripts/soak_threads.py 4 100 2>&1 | tail -1
timeout -k 10 300 python scripts/soak_shapes.py 60 2>&1 | tail -1
for c in c3r pile c2; do timeout -k 10 300 python scripts/soak_multi.py 120 $c 2>&1 | tail -1; done
timeout -k 10 600 python scripts/dirty_tests.py 2>&1 | tail -1
} > $O 2>&1
cat $O
