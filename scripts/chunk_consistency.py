import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
w = pfc.configs.c5_pile()
n = w.n_items
rng = np.random.default_rng(5)
seeds = (rng.standard_normal((n, 6, 24)) * 1e-2, rng.standard_normal((n, 6, 6)) * 0.1, rng.standard_normal((n, 6, 6)) * 1e-3)
worst = []; cnt6 = []; cnt12 = []
for rep in range(6):
    m = pfc.configs.build_scenario(w)
    a = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *seeds, w.ins_ids)
    b = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *seeds, w.ins_ids)      # the same directions as a further chunk
    assert m.last_dual_reused()
    d = np.maximum(np.abs(a[2] - b[2]).reshape(n, -1).max(1) / np.maximum(np.abs(b[2]).reshape(n, -1).max(1), 1e-300),
                   np.abs(a[3] - b[3]).reshape(n, -1).max(1) / np.maximum(np.abs(b[3]).reshape(n, -1).max(1), 1e-300))
    worst.append(d.max()); cnt6.append(int((d > 1e-6).sum())); cnt12.append(int((d > 1e-12).sum()))
    m.close()
print("PFC_DUAL_VALUE_K =", os.environ.get("PFC_DUAL_VALUE_K"), "| same directions, first chunk vs further chunk of one value pass: worst", ["%.1e" % x for x in worst], "items > 1e-6:", cnt6, "> 1e-12:", cnt12)
