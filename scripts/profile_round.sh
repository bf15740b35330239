#!/bin/bash
# rocprofv3 evidence for one round: kernel-trace stats, then FETCH_SIZE / WRITE_SIZE in their own passes.
# usage (on the GPU box): bash scripts/profile_round.sh <tag>      writes gpurun_out/<tag>_*
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r03}
. $R/scripts/pmc_lib.sh
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-extras --no-validate --reps 1 > $R/gpurun_out/${TAG}_stats.log 2>&1
pmc_pass $R/gpurun_out/${TAG}_fetch $R/gpurun_out/${TAG}_fetch.log FETCH_SIZE --steps 3 --warmup 2 --cpu-seconds 0 --no-extras --no-validate --reps 1
pmc_pass $R/gpurun_out/${TAG}_write $R/gpurun_out/${TAG}_write.log WRITE_SIZE --steps 3 --warmup 2 --cpu-seconds 0 --no-extras --no-validate --reps 1
echo done
