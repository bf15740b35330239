"""profiles/pmc_valu.json from the counter groups of scripts/pmc_probe.sh (SQ_INSTS_VALU, wait and L2 counters) and the
unit counts of a bench line.  usage: python scripts/pmc_valu.py <tag>[,<tag2>...] <bench.json>
Values are PER STEP: parts x the per-launch averages (a split step launches every kernel once per half)."""
import csv, glob, json, os, re, sys, collections


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


root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, bench = sys.argv[1], json.load(open(sys.argv[2]))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for tg in tag.split(","):
    seen = set()       # a counter that several groups of one tag collected is taken from the first group only
    for f in newest_per_dir(os.path.join(root, "gpurun_out", tg + "_g*", "**", "*counter_collection.csv")):
        mine = set()
        for row in csv.DictReader(open(f)):
            if (tg, row["Counter_Name"]) in seen:
                continue
            mine.add((tg, row["Counter_Name"]))
            vals[kname(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        seen |= mine


def steady(v):
    """Mean over the steady-state launches: the first launches of a run see work lists that are still growing (an
    overflowing candidate list truncates what the later kernels process), so launches below half of the maximum are
    dropped."""
    if not v:
        return None
    keep = [x for x in v if x >= 0.5 * max(v)]
    return sum(keep) / len(keep)


parts = int(bench.get("concurrent_parts", 1))
out = {"source": "rocprofv3 --pmc (scripts/pmc_probe.sh, one counter group per pass), bench.py default workload "
                 f"({int(bench['config'].get('items_per_step', bench['config'].get('poses_per_gpu', 0)))} poses as {parts} concurrent part(s)); PER STEP = {parts} x the per-launch averages",
       "measured": "round 3",
       "node_tests": int(bench["config"]["node_tests_per_step"]), "candidates": int(bench["config"]["ops_per_step"])}
for kern, name in (("pfc::k_bp_dfs32", "k_bp_dfs32"), ("void pfc::k_narrow<false, 2>", "k_narrow"), ("pfc::k_integ", "k_integ"),
                   ("pfc::k_fric", "k_fric")):
    for ctr, key in (("SQ_INSTS_VALU", "valu_insts"), ("SQ_WAVE_CYCLES", "wave_cycles"), ("SQ_WAIT_ANY", "wait_any_cycles"),
                     ("SQ_WAIT_INST_ANY", "wait_inst_any_cycles"), ("SQ_ACTIVE_INST_ANY", "active_inst_any_cycles"),
                     ("SQ_INSTS_SALU", "salu_insts"), ("SQ_INSTS_LDS", "lds_insts"), ("TCC_HIT_sum", "tcc_hit"), ("TCC_MISS_sum", "tcc_miss"),
                     # instruction mix (scripts/pmc_groups.sh r02x): the issue cost of a wave64 instruction depends on its type
                     ("SQ_INSTS_VALU_ADD_F32", "valu_add_f32"), ("SQ_INSTS_VALU_MUL_F32", "valu_mul_f32"), ("SQ_INSTS_VALU_FMA_F32", "valu_fma_f32"),
                     ("SQ_INSTS_VALU_ADD_F64", "valu_add_f64"), ("SQ_INSTS_VALU_MUL_F64", "valu_mul_f64"), ("SQ_INSTS_VALU_FMA_F64", "valu_fma_f64"),
                     ("SQ_INSTS_VALU_TRANS_F64", "valu_trans_f64"), ("SQ_INSTS_VALU_IOPS", "valu_int"), ("SQ_INSTS_VALU_CVT", "valu_cvt"),
                     ("SQ_THREAD_CYCLES_VALU", "valu_thread_cycles"), ("SQ_INSTS_BRANCH", "branch_insts"), ("SQ_INSTS_VMEM", "vmem_insts")):
        m = steady(vals[kern][ctr])
        if m is not None:
            out[f"{name}_{key}"] = round(parts * m)
json.dump(out, open(os.path.join(root, "profiles", "pmc_valu.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
