"""Summarise the rocprofv3 --pmc passes of scripts/profile_round.sh into profiles/: per kernel the FETCH_SIZE and
WRITE_SIZE totals (KB as reported) and the per-launch traffic bench.py reports as roofline.traffic.
usage: python scripts/pmc_summary.py <tag> [round]      reads gpurun_out/<tag>_{fetch,write,stats}"""
import csv, glob, json, os, re, shutil, sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = {}
for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    acc = {}
    for f in glob.glob(os.path.join(root, "gpurun_out", f"{tag}_{sub}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != ctr:
                continue
            name = re.sub(r"\(.*", "", row["Kernel_Name"])
            a = acc.setdefault(name, {"calls": 0, "sum_KB": 0.0})
            a["calls"] += 1; a["sum_KB"] += float(row["Counter_Value"])
    for a in acc.values():
        a["per_call_KB"] = a["sum_KB"] / a["calls"]
    raw[ctr] = acc
json.dump(raw, open(os.path.join(root, "profiles", f"{rnd}_pmc_raw.json"), "w"), indent=1)

def per_launch(name):
    return 1024.0 * (raw["FETCH_SIZE"].get(name, {}).get("per_call_KB", 0.0) + raw["WRITE_SIZE"].get(name, {}).get("per_call_KB", 0.0))

out = {"k_bp_dfs32_bytes_per_launch": per_launch("pfc::k_bp_dfs32") + per_launch("pfc::k_bp_dfs") + per_launch("pfc::k_bp_expand"),
       "k_narrow0_bytes_per_launch": per_launch("void pfc::k_narrow<false>"),
       "k_fric_bytes_per_launch": per_launch("pfc::k_fric"),
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KB x 1024, per launch, bench.py default "
               "workload (2048 poses); FETCH_SIZE is NOT doubled: MI355X_MICROARCH.md says it under-reports wide coalesced "
               "streaming reads by 2x and is uncalibrated for other widths; these kernels gather 16-byte pieces of "
               "64..256-byte records"}
json.dump(out, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
for f in glob.glob(os.path.join(root, "gpurun_out", f"{tag}_stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(root, "profiles", f"{rnd}_kernel_stats.csv"))
print(json.dumps(out, indent=1))
