import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
C = pfc.configs
w = C.c5_pile()
m = C.build_scenario(w)
m.set_option("fixed_order", 1)
for k in range(3):
    try:
        out = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        print("ok", int((out[-1][:, 3] > 0).sum()), m.stats() if hasattr(m, "stats") else "")
    except Exception as e:
        print("ERR", e)
