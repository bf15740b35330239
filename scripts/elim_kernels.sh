#!/bin/bash
# rocprofv3 per-kernel durations of elimination / variant builds (build/variants/<name>.so), one box: 
# usage (GPU box): bash scripts/elim_kernels.sh "<bench args>" <name> ...   ("product" = the product library)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
BARGS=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = product ]; then unset PFC_LIB PFC_ALLOW_DIAGNOSTIC; else export PFC_LIB=$R/build/variants/$v.so PFC_ALLOW_DIAGNOSTIC=1; fi
  rm -rf /tmp/ek_$v
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ek_$v -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-seconds 0 --no-extras --no-validate --reps 1 $BARGS > $R/gpurun_out/ek_$v.log 2>&1
  f=$(find /tmp/ek_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v ($BARGS)"
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:5]:
    print(f"   {r['Name'][:60]:60s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
done
