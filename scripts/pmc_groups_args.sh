#!/bin/bash
# Like pmc_groups.sh, with extra bench.py arguments: bash scripts/pmc_groups_args.sh <tag> "<bench args>" "<group>" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
. $R/scripts/pmc_lib.sh
TAG=$1; shift
BARGS=$1; shift
for grp in "$@"; do pmc_check_names "$grp" || exit 2; done
cd /tmp && export TMPDIR=/tmp
k=0
for grp in "$@"; do
  pmc_pass $R/gpurun_out/${TAG}_g$k $R/gpurun_out/${TAG}_g$k.log "$grp" --steps 2 --warmup 1 --cpu-seconds 0 --no-extras --no-validate --reps 1 $BARGS
  echo "group $k rc=$?"
  k=$((k+1))
done
