#!/bin/bash
# PMC passes that do not stop at a group the profiler refuses: bash scripts/pmc_groups.sh <tag> "<group>" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
k=0
for grp in "$@"; do
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_g$k -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-extras --no-validate --reps 1 > $R/gpurun_out/${TAG}_g$k.log 2>&1
  echo "group $k rc=$?"
  k=$((k+1))
done
