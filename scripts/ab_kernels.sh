#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of bench.py under several option sets, on one box.
# usage (GPU box): bash scripts/ab_kernels.sh <tag> "<bench args A>" "<bench args B>" ...   -> gpurun_out/<tag>_<k>_stats.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
k=0
for args in "$@"; do
  rm -rf /tmp/abk_$k
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk_$k -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-extras --no-validate --reps 1 $args > $R/gpurun_out/${TAG}_$k.log 2>&1
  f=$(find /tmp/abk_$k -name "*kernel_stats.csv" | head -1)
  echo "== $args" > $R/gpurun_out/${TAG}_${k}_stats.txt
  python3 - "$f" >> $R/gpurun_out/${TAG}_${k}_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']}%")
PY
  cat $R/gpurun_out/${TAG}_${k}_stats.txt
  k=$((k+1))
done
