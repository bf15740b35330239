"""Small scenes under option fixed_order, and whether the one-launch kernel is bit-reproducible by itself (two fresh default handles).
usage: python scripts/fixed_small.py"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
C = pfc.configs
for name, w in (("c1", C.c1_boxes()), ("c2", C.c2_box_on_plane(1)), ("pencil-like", C.c3_blob_tool(8, seed=3, n_div_blob=4, n_div_tool=2, distance=0.195)),
                ("c3 single", C.c3_blob_tool(1)), ("c3 x 8", C.c3_blob_tool(8))):
    n = w.n_items
    rng = np.random.default_rng(3)
    seeds = (rng.standard_normal((n, 6, 24)) * 1e-2, rng.standard_normal((n, 6, 6)) * 0.1, rng.standard_normal((n, 6, 6)) * 1e-3)
    row = {}
    outs = {}
    for fixed in (0, 1):
        res = []
        for rep in range(2):
            m = C.build_scenario(w)
            m.set_option("fixed_order", fixed)
            v = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
            d = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *seeds, w.ins_ids)
            res.append(tuple(v) + tuple(d))
            if rep == 1:
                tv, td = [], []
                for _ in range(30):
                    t0 = time.perf_counter(); m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids); tv.append(time.perf_counter() - t0)
                for k in range(20):
                    w.s[0, 0] += 1e-12
                    t0 = time.perf_counter(); m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *seeds, w.ins_ids); td.append(time.perf_counter() - t0)
                row[fixed] = (statistics.median(tv[5:]) * 1e6, statistics.median(td[5:]) * 1e6, m.last_parts())
            m.close()
        outs[fixed] = [bool(np.array_equal(x, y)) for x, y in zip(res[0], res[1])]
    print("%-12s value %6.0f -> %6.0f us | Dual(6) %6.0f -> %6.0f us | two fresh handles bit-equal (wrench sdot counts | wrench sdot d_wrench d_sdot counts): default %s, fixed_order %s" %
          (name, row[0][0], row[1][0], row[0][1], row[1][1], outs[0], outs[1]), flush=True)
