"""Per-kernel averages of the counters collected by scripts/pmc_probe.sh.  usage: python scripts/pmc_table.py <tag>"""
import csv, glob, os, re, sys, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(root, "gpurun_out", sys.argv[1] + "_g*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", row["Kernel_Name"])
        a = acc[name][row["Counter_Name"]]
        a[0] += 1; a[1] += float(row["Counter_Value"])
for name in sorted(acc):
    if not name.startswith(("pfc::", "void pfc::")):
        continue
    print(name)
    for c, (n, v) in sorted(acc[name].items()):
        print(f"    {c:36s} {v / n:16.1f}  per launch ({n} launches)")
