#!/bin/bash
# rocprofv3 evidence of the Dual path: per-kernel durations (kernel trace) and SQ counters (own passes) of device-resident
# Dual(6) evaluations -- a first chunk + 4 further chunks each -- of C5 (2 016 pile instructions) and of a 2 048-pose C3 batch.
# usage (GPU box): bash scripts/profile_dual.sh <tag>      writes gpurun_out/<tag>_dual_{c5,c3b}_{stats,pmc}; then
#                  python scripts/pmc_dual.py <tag> <round>  (here) -> profiles/<round>_dual_kernel_stats_*.csv, profiles/pmc_dual.json
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
. $R/scripts/pmc_lib.sh
cd /tmp && export TMPDIR=/tmp
for cfg in c5 c3b; do
  n=30; [ $cfg = c3b ] && n=6
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_dual_${cfg}_stats -- python3 $R/scripts/dual_trace.py $cfg $n 6 4 > $R/gpurun_out/${TAG}_dual_${cfg}_stats.log 2>&1
  pmc_check_names "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" || exit 2
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/${TAG}_dual_${cfg}_pmc -- python3 $R/scripts/dual_trace.py $cfg $n 6 4 > $R/gpurun_out/${TAG}_dual_${cfg}_pmc.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/${TAG}_dual_${cfg}_pmc2 -- python3 $R/scripts/dual_trace.py $cfg $n 6 4 > $R/gpurun_out/${TAG}_dual_${cfg}_pmc2.log 2>&1
done
echo done
