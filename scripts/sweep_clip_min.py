"""One-kernel narrowphase against clip-only kernel + k_integ by batch size (option clip_min).  usage: python scripts/sweep_clip_min.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
C = pfc.configs
shapes = [(f"c3 full x {n}", (lambda n=n: C.c3_blob_tool(n))) for n in (16, 64, 128, 256, 400, 511)]
shapes += [("c3 reduced (8/6) x 300 [batched]", lambda: C.c3_blob_tool(300, n_div_blob=8, n_div_tool=6)), ("c2 x 400", lambda: C.c2_box_on_plane(400, montecarlo=True))]
for name, mk in shapes:
    w = mk()
    res = []
    for cm in (0, 1):
        m = C.build_scenario(w)
        m.set_option("clip_min", cm); m.set_option("fused", 0); m.set_option("team", 0)
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
    print("%-34s one-kernel narrowphase %.0f us | clip-only kernel + k_integ %.0f us" % (name, res[0], res[1]), flush=True)
