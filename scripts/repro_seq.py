"""Diagnostic: the first three evaluations of scripts/soak.py on the long-lived handle only (argv[1] = 1: with the
fresh comparison handles alive as in the soak)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
rng = np.random.default_rng(2026)
w = pfc.configs.c3_blob_tool(700, seed=9, n_div_blob=6, n_div_tool=4)
w.s[:] = rng.standard_normal((w.n_items, 6)) * 1e-3
nd = 6
d_pose = rng.standard_normal((w.n_items, nd, 24)) * 1e-2
d_twist = rng.standard_normal((w.n_items, nd, 6)) * 0.1
d_s = rng.standard_normal((w.n_items, nd, 6)) * 1e-3
with_fresh = len(sys.argv) > 1 and sys.argv[1] == "1"
m = pfc.configs.build_scenario(w)
for size, lo, dual in ((64, 242, True), (86, 349, True), (512, 59, False)):
    sl = slice(lo, lo + size)
    f = pfc.configs.build_scenario(w) if with_fresh else None
    print("m:", size, lo, "dual" if dual else "value", flush=True)
    if dual:
        m.force_all_elastic_intersections_dual(w.pose[sl], w.twist[sl], w.s[sl], d_pose[sl], d_twist[sl], d_s[sl], w.ins_ids[sl])
    else:
        m.force_all_elastic_intersections(w.pose[sl], w.twist[sl], w.s[sl], w.ins_ids[sl])
    print("  m done", m.stats(), flush=True)
    if f is not None:
        if dual:
            f.force_all_elastic_intersections_dual(w.pose[sl], w.twist[sl], w.s[sl], d_pose[sl], d_twist[sl], d_s[sl], w.ins_ids[sl])
        else:
            f.force_all_elastic_intersections(w.pose[sl], w.twist[sl], w.s[sl], w.ins_ids[sl])
        print("  f done", flush=True)
        f.close()
m.close()
print("all ok")
