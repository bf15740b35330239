import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
L = int(sys.argv[1]); which = sys.argv[2] if len(sys.argv) > 2 else "c1"
opts = dict(kv.split("=") for kv in sys.argv[3:])
w = pfc.configs.c1_boxes() if which == "c1" else pfc.configs.c3_blob_tool(4, n_div_blob=6, n_div_tool=5)
m = pfc.configs.build_scenario(w)
m.set_option("bfs_levels", L)
for k, v in opts.items():
    m.set_option(k, int(v))
print("eval", which, "bfs_levels", L, opts, flush=True)
wrench, sdot, counts = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
print(counts.tolist(), m.stats(), flush=True)
