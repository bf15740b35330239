#!/bin/bash
# ad-hoc PMC probes of the bench workload: one counter group per pass (never combined with trace domains)
# usage (GPU box): bash scripts/pmc_probe.sh <tag> "<CTR1 CTR2 ...>" ["<group 2>" ...]
# Names are checked first and every pass runs under `timeout -k` (scripts/pmc_lib.sh); stops at the first failing pass.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
. $R/scripts/pmc_lib.sh
TAG=$1; shift
for grp in "$@"; do pmc_check_names "$grp" || exit 2; done
cd /tmp && export TMPDIR=/tmp
k=0
for grp in "$@"; do
  pmc_pass $R/gpurun_out/${TAG}_g$k $R/gpurun_out/${TAG}_g$k.log "$grp" --steps 2 --warmup 1 --cpu-seconds 0 --no-extras --no-validate --reps 1
  k=$((k+1))
done
echo done
