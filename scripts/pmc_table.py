"""Per-kernel averages of the counters collected by scripts/pmc_probe.sh.  usage: python scripts/pmc_table.py <tag>"""
import csv, glob, os, re, sys, collections


def newest_per_dir(pattern):
    """gpurun merges every call's output into the same directories: keep only the newest file of each directory, so
    that counters of an earlier build never mix with the current one's"""
    best = {}
    for f in glob.glob(pattern, recursive=True):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in newest_per_dir(os.path.join(root, "gpurun_out", sys.argv[1] + "_g*", "**", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        vals[re.sub(r"\(.*", "", row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(vals):
    if not name.startswith(("pfc::", "void pfc::")):
        continue
    print(name)
    for c, v in sorted(vals[name].items()):
        # steady state: launches of the first step see work lists that are still growing (truncated work)
        keep = [x for x in v if x >= 0.5 * max(v)]
        print(f"    {c:36s} {sum(keep) / len(keep):16.1f}  per steady-state launch ({len(keep)} of {len(v)} launches)")
