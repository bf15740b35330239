#!/bin/bash
# PMC passes that do not stop at a group the profiler refuses: bash scripts/pmc_groups.sh <tag> "<group>" ...
# Every name is checked first (scripts/pmc_lib.sh: TA_* refused, unknown names need PFC_PMC_FORCE=1) and every pass runs
# under `timeout -k`, so one bad group costs one pass.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
. $R/scripts/pmc_lib.sh
TAG=$1; shift
for grp in "$@"; do pmc_check_names "$grp" || exit 2; done
cd /tmp && export TMPDIR=/tmp
k=0
for grp in "$@"; do
  pmc_pass $R/gpurun_out/${TAG}_g$k $R/gpurun_out/${TAG}_g$k.log "$grp" --steps 2 --warmup 1 --cpu-seconds 0 --no-extras --no-validate --reps 1
  echo "group $k rc=$?"
  k=$((k+1))
done
