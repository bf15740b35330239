#!/bin/bash
# Kernel timeline of one steady-state step of bench.py (rocprofv3 --kernel-trace): start/end of every kernel relative to the
# step's first kernel, per queue -- what the two concurrent halves actually overlap.
# usage (GPU box): bash scripts/timeline.sh <tag> [bench args]     -> gpurun_out/<tag>_timeline.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$TAG
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -- python3 $R/bench.py --steps 4 --warmup 3 --cpu-seconds 0 --no-extras --no-validate --reps 1 "$@" > $R/gpurun_out/${TAG}_timeline.log 2>&1
f=$(find /tmp/tl_$TAG -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $R/gpurun_out/${TAG}_timeline.txt <<'PY'
import csv, sys, re
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "pfc::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps: split at k_setup_items launches that follow a k_final by the same queue ... simpler: a step starts at a k_setup_items whose
# previous pfc kernel (in start order) is a k_final
steps, cur = [], []
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("pfc::", "")
    r["_n"] = name
    if name == "k_setup_items" and cur and sum(1 for x in cur if x["_n"] == "k_final") >= max(1, len({x["Queue_Id"] for x in cur})):
        steps.append(cur); cur = []
    cur.append(r)
if cur: steps.append(cur)
st = steps[-2] if len(steps) >= 2 else steps[-1]
t0 = min(int(r["Start_Timestamp"]) for r in st)
qs = sorted({r["Queue_Id"] for r in st})
print(f"step with {len(st)} kernels on {len(qs)} queues; times in us relative to the first kernel start")
for r in st:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"  q{qs.index(r['Queue_Id'])} {r['_n']:28s} {a:9.1f} -> {b:9.1f}  ({b - a:8.1f})  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))} lds {r.get('LDS_Block_Size', '?')} vgpr {r.get('VGPR_Count', '?')}")
print(f"step length {(max(int(r['End_Timestamp']) for r in st) - t0) / 1e3:.1f} us")
PY
cat $R/gpurun_out/${TAG}_timeline.txt
