"""Diagnostic: per-phase cycle shares of k_narrow (needs a -DPFC_STAMPS build: scripts/build_stamps.sh)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = sys.argv[2] if len(sys.argv) > 2 else "c3"          # c3 (n poses) | c5 (the pile) ; argv[3]: bfs_levels
w = pfc.configs.c5_pile() if cfg == "c5" else pfc.configs.c3_blob_tool(n)
m = pfc.configs.build_scenario(w)
if len(sys.argv) > 3:
    m.set_option("bfs_levels", int(sys.argv[3]))
if os.environ.get("PFC_SPLIT_MIN"):
    m.set_option("split_min", int(os.environ["PFC_SPLIT_MIN"]))
for _ in range(3):
    m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
out = (C.c_longlong * 16)()
pfc._lib.lib().pfc_debug_stamps(m._h, out)
v = [int(x) for x in out]
names = ["gather", "clip", "reserve", "integrate", "reduce"]
if v[5] == 0:      # the clip-only narrowphase of a big batch carries no stamps; slots 0..2 then hold the broadphase barrier waits
    names = []
tot = sum(v[:5]) if names else 0
if names:
    print("rounds", v[5], "cycles/round", tot / max(v[5], 1))
for k, nm in enumerate(names):
    print(f"  {nm:10s} {v[k] / max(v[5], 1):10.0f} cycles/round  {100.0 * v[k] / max(tot, 1):5.1f} %")

if names:
  print("  reduce sub-phases (cycles/round): slot atomic + polygon store %.0f, ten sums %.0f, moments + record %.0f, counters %.0f" % (
    v[6] / max(v[5], 1), v[7] / max(v[5], 1), v[14] / max(v[5], 1), (v[4] - v[6] - v[7] - v[14]) / max(v[5], 1)))
it = max(v[11], 1)
print("broadphase workgroup iterations", v[11], "pairs/iteration %.1f" % (v[12] / it))
tb = v[8] + v[9] + v[10] + v[13]
for k, nm in ((8, "pop + node loads"), (9, "single-precision test"), (10, "ballots + barrier"), (13, "prefix + push + barrier")):
    print(f"  {nm:24s} {v[k] / it:10.0f} cycles/iteration  {100.0 * v[k] / max(tb, 1):5.1f} %")
if v[2] and not names:
    print("  inside the barriers (mean over the waves of a workgroup): first %.0f, second %.0f cycles/iteration" % (v[0] / v[2], v[1] / v[2]))
