"""profiles/<round>_dual_kernel_stats_{c5,c3b}.csv and profiles/pmc_dual.json from the passes of scripts/profile_dual.sh.
usage: python scripts/pmc_dual.py <tag> <round>
Per Dual kernel and configuration: launches, mean duration (kernel trace), wave-level VALU instructions, Float64 share, active
lanes, wait share per launch -- the numbers behind bench.py's roofline_dual (instructions per (contributing pair, direction),
fraction of the vector-issue bound)."""
import collections, csv, glob, json, os, re, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]


def newest(pattern):
    best = {}
    for f in glob.glob(pattern, recursive=True):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


def kname(full):
    return re.sub(r"^void ", "", re.sub(r"\(.*", "", full))


out = {"source": "rocprofv3 kernel trace + --pmc passes of scripts/dual_trace.py <cfg> n 6 4 (device-resident Dual(6): a first chunk and "
                 "four further chunks per evaluation, dense seeds), scripts/profile_dual.sh", "measured": rnd}
for cfg in ("c5", "c3b"):
    dur = collections.defaultdict(list)
    for f in newest(os.path.join(root, "gpurun_out", f"{tag}_dual_{cfg}_stats", "**", "*kernel_trace.csv")):
        for row in csv.DictReader(open(f)):
            dur[kname(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
    for f in newest(os.path.join(root, "gpurun_out", f"{tag}_dual_{cfg}_stats", "**", "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(root, "profiles", f"{rnd}_dual_kernel_stats_{cfg}.csv"))
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("pmc", "pmc2"):
        for f in newest(os.path.join(root, "gpurun_out", f"{tag}_dual_{cfg}_{sub}", "**", "*counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                ctr[kname(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res = {}
    for k in sorted(dur):
        if "dual" not in k:
            continue
        v = dur[k]
        big = [x for x in v if x >= 0.5 * max(v)]      # steady launches (the first evaluation sizes its buffers)
        e = {"launches": len(v), "mean_us": sum(big) / len(big)}
        c = ctr.get(k, {})
        def mean(name):
            x = c.get(name)
            if not x:
                return None
            keep = [y for y in x if y >= 0.5 * max(x)]
            return sum(keep) / len(keep)
        iv = mean("SQ_INSTS_VALU")
        if iv:
            e["valu_insts_per_launch"] = iv
            f64 = sum((mean(n) or 0.0) for n in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
            e["f64_share"] = f64 / iv if f64 else None
            tc = mean("SQ_THREAD_CYCLES_VALU")
            e["active_lanes"] = tc / iv if tc else None
            wc, wa = mean("SQ_WAVE_CYCLES"), mean("SQ_WAIT_ANY")
            e["wait_any_share"] = wa / wc if (wc and wa) else None
            cyc = 4.2 * (e["f64_share"] or 0.7) + 3.4 * (1.0 - (e["f64_share"] or 0.7))      # profiles/r02_valu_rate.txt
            e["issue_bound_us"] = iv * cyc / (1024 * 2.4e9) * 1e6
            e["frac_of_issue_bound"] = e["issue_bound_us"] / e["mean_us"]
        res[k] = e
    out[cfg] = res
json.dump(out, open(os.path.join(root, "profiles", "pmc_dual.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
