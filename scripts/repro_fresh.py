import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
rng = np.random.default_rng(2026)
w = pfc.configs.c3_blob_tool(700, seed=9, n_div_blob=6, n_div_tool=4)
w.s[:] = rng.standard_normal((w.n_items, 6)) * 1e-3
size = int(sys.argv[1]); lo = int(sys.argv[2])
sl = slice(lo, lo + size)
f = pfc.configs.build_scenario(w)
print("evaluating", size, lo, flush=True)
out = f.force_all_elastic_intersections(w.pose[sl], w.twist[sl], w.s[sl], w.ins_ids[sl])
print("ok", out[2][:, 1].sum(), f.stats(), flush=True)
f.close()
