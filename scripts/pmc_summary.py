"""Summarise the rocprofv3 --pmc passes of scripts/profile_round.sh into profiles/: per kernel the FETCH_SIZE and
WRITE_SIZE totals (KB as reported) and the per-launch traffic bench.py reports as roofline.traffic.
usage: python scripts/pmc_summary.py <tag> [round]      reads gpurun_out/<tag>_{fetch,write,stats}"""
import csv, glob, json, os, re, shutil, sys


def newest_per_dir(pattern):
    """gpurun merges every call's output into the same directories: keep only the newest file of each directory, so
    that counters of an earlier build never mix with the current one's"""
    best = {}
    for f in glob.glob(pattern, recursive=True):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())



def kname(full):
    """kernel name without its argument list; the instantiations of k_bp_dfs32<BLK> under one name"""
    n = re.sub(r"^void (pfc::k_bp_dfs32)<\d+>$", r"\1", re.sub(r"\(.*", "", full))
    # the clip-only forms (ring by lane / by survivor rank; round 3: survivors queued in the ring, k_clip_queue) under one name
    return n.replace("k_narrow<false, 3>", "k_narrow<false, 2>").replace("pfc::k_clip_queue", "void pfc::k_narrow<false, 2>")


tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = {}
for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    acc = {}
    for f in newest_per_dir(os.path.join(root, "gpurun_out", f"{tag}_{sub}", "**", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != ctr:
                continue
            name = kname(row["Kernel_Name"])
            a = acc.setdefault(name, {"calls": 0, "sum_KB": 0.0, "values_KB": []})
            a["calls"] += 1; a["sum_KB"] += float(row["Counter_Value"]); a["values_KB"].append(float(row["Counter_Value"]))
    for a in acc.values():
        # steady-state launches only: while the work lists are still growing (first step of a run) an overflowing
        # candidate list truncates what the later kernels process; launches below half of the maximum are dropped
        keep = [x for x in a["values_KB"] if x >= 0.5 * max(a["values_KB"])]
        a["per_call_KB"] = sum(keep) / len(keep); a["steady_calls"] = len(keep)
        del a["values_KB"]
    raw[ctr] = acc
json.dump(raw, open(os.path.join(root, "profiles", f"{rnd}_pmc_raw.json"), "w"), indent=1)

def per_launch(name):
    # MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B while the L2 fills 128-byte lines, i.e. it
    # reports half of the bytes read -> doubled; WRITE_SIZE is exact.  Both are KB.
    return 1024.0 * (2.0 * raw["FETCH_SIZE"].get(name, {}).get("per_call_KB", 0.0) +
                     raw["WRITE_SIZE"].get(name, {}).get("per_call_KB", 0.0))

out = {"k_bp_dfs32_bytes_per_launch": per_launch("pfc::k_bp_dfs32") + per_launch("pfc::k_bp_dfs") + per_launch("pfc::k_bp_expand"),
       "k_narrow0_bytes_per_launch": per_launch("void pfc::k_narrow<false, 2>") + per_launch("pfc::k_integ"),
       "k_narrow_clip_bytes_per_launch": per_launch("void pfc::k_narrow<false, 2>"),
       "k_integ_bytes_per_launch": per_launch("pfc::k_integ"),
       "k_fric_bytes_per_launch": per_launch("pfc::k_fric"),
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), per launch of bench.py's default workload "
               "(8192 poses as two concurrent 4096-pose halves: a launch covers one half); bytes = 1024 x (2 x FETCH_SIZE "
               "+ WRITE_SIZE): the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE tallies 128-byte line fills at "
               "64 B); Infinity-Cache hits are counted, so this is memory-side traffic of the L2, an upper bound of HBM bytes"}
json.dump(out, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
for f in newest_per_dir(os.path.join(root, "gpurun_out", f"{tag}_stats", "**", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(root, "profiles", f"{rnd}_kernel_stats.csv"))
# rocprofv3's own summary averages over ALL launches, including the few of the first step that ran on truncated work
# lists (buffers still growing); the steady-state averages below are the ones bench.py's HIP-event times agree with
steady = {}
for f in newest_per_dir(os.path.join(root, "gpurun_out", f"{tag}_stats", "**", "*kernel_trace.csv")):
    dur = {}
    for row in csv.DictReader(open(f)):
        dur.setdefault(kname(row["Kernel_Name"]), []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for name, v in dur.items():
        if name.startswith(("pfc::", "void pfc::")):
            keep = [x for x in v if x >= 0.5 * max(v)]
            steady[name] = {"launches": len(v), "steady_launches": len(keep), "all_avg_ns": sum(v) / len(v),
                            "steady_avg_ns": sum(keep) / len(keep)}
steady["_names"] = ("the key 'void pfc::k_narrow<false, 2>' is the clip-only narrowphase kernel of the run, whichever it was "
                    "(pfc::k_clip_queue since round 3; k_narrow<.., 2 / 3> before and for tet-tet scenarios): one key across rounds")
json.dump(steady, open(os.path.join(root, "profiles", f"{rnd}_kernel_steady.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
