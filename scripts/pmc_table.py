"""Per-kernel averages of the counters collected by scripts/pmc_probe.sh.  usage: python scripts/pmc_table.py <tag>"""
import csv, glob, os, re, sys, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "gpurun_out", sys.argv[1] + "_g*", "**", "*counter_collection.csv"), recursive=True):
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
