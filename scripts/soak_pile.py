"""Soak run on a multi-instruction scene (C5-like pile: many bodies, all pairs as bristle instructions over different
meshes): random subsets of the instructions in random order, value and Dual, one long-lived handle against fresh ones.
usage: python scripts/soak_pile.py [n_evals]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
n_evals = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(5)
w = pfc.configs.c5_pile(n_side=3, n_divs=(1, 2, 3))
n = w.n_items
nd = 3
d_pose = rng.standard_normal((n, nd, 24)) * 1e-2
d_twist = rng.standard_normal((n, nd, 6)) * 0.1
d_s = rng.standard_normal((n, nd, 6)) * 1e-3
m = pfc.configs.build_scenario(w)
bad = 0
for it in range(n_evals):
    k = int(rng.choice([1, 3, 20, 100, n // 2, n]))
    sel = rng.permutation(n)[:k]                      # arbitrary instruction order
    dual = rng.random() < 0.4
    f = pfc.configs.build_scenario(w)
    args = (w.pose[sel], w.twist[sel], w.s[sel])
    if dual:
        got = m.force_all_elastic_intersections_dual(*args, d_pose[sel], d_twist[sel], d_s[sel], w.ins_ids[sel])
        ref = f.force_all_elastic_intersections_dual(*args, d_pose[sel], d_twist[sel], d_s[sel], w.ins_ids[sel])
        pairs = ((got[0], ref[0], 1e-10), (got[2], ref[2], 1e-8))
        cg, cr = got[4], ref[4]
    else:
        got = m.force_all_elastic_intersections(*args, w.ins_ids[sel])
        ref = f.force_all_elastic_intersections(*args, w.ins_ids[sel])
        pairs = ((got[0], ref[0], 1e-10),)
        cg, cr = got[2], ref[2]
    f.close()
    ok = np.array_equal(cg, cr)
    for a, b, tol in pairs:
        ok = ok and np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-300)
    if not ok:
        bad += 1
        print("MISMATCH", it, k, "dual" if dual else "value", flush=True)
    if it % 40 == 0:
        print("eval", it, "...", flush=True)
m.close()
print(f"{n_evals} evaluations on {n} instructions, mismatches: {bad}")
sys.exit(1 if bad else 0)
