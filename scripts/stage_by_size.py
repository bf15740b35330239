import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
for n in (16, 128, 400, 700, 1000):
    w = pfc.configs.c3_blob_tool(n)
    m = pfc.configs.build_scenario(w)
    m.set_option("fused", 0); m.set_option("team", 0)
    for _ in range(4): m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    m.set_option("profile", 1)
    acc = {}
    for _ in range(10):
        m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        for k, v in m.stage_ms().items(): acc[k] = acc.get(k, 0) + v / 10
    print("c3 full x %4d: stages (us) %s" % (n, {k: round(v * 1e3, 1) for k, v in acc.items()}), flush=True)
    m.close()
