import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
for n in (520, 600, 700, 800, 900, 1000):
    w = pfc.configs.c3_blob_tool(n)
    res = []
    for cq in (1, 2):
        m = pfc.configs.build_scenario(w)
        m.set_option("clip_queue", cq)
        for _ in range(4): m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        b = m.bind(w.pose, w.twist, w.s, w.ins_ids)
        for _ in range(4): b()
        ts = []
        for _ in range(9):
            t0 = time.perf_counter()
            for _ in range(10): b()
            ts.append((time.perf_counter() - t0) / 10)
        res.append(np.median(ts) * 1e6)
        m.close()
    print("c3 full x %4d: clip_queue for launches >= 1024 items (default) %.0f us | for every clip-only launch %.0f us" % (n, res[0], res[1]), flush=True)
