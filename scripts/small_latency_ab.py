"""Small-scene latency through the bound host path (value) and the Dual chunk path, medians of many short runs: for A/B of
host-side changes (PFC_LIB selects the library).  usage: python scripts/small_latency_ab.py [label]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
C = pfc.configs
out = []
for name, w in (("c1", C.c1_boxes()), ("c2", C.c2_box_on_plane(1)), ("c2x48", C.c2_box_on_plane(48, montecarlo=True))):
    m = C.build_scenario(w); n = w.n_items
    b = m.bind(w.pose, w.twist, w.s, w.ins_ids)
    for _ in range(200): b()
    tv = []
    for _ in range(15):
        t0 = time.perf_counter()
        for _ in range(400): b()
        tv.append((time.perf_counter() - t0) / 400)
    rng = np.random.default_rng(1)
    dp = rng.standard_normal((n, 6, 24)) * 1e-3; dt = rng.standard_normal((n, 6, 6)) * 1e-2; ds = rng.standard_normal((n, 6, 6)) * 1e-4
    for k in range(50): m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, dp * (1 + 0.01 * k), dt, ds, w.ins_ids)
    td = []
    for _ in range(15):
        t0 = time.perf_counter()
        for k in range(100): m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, dp * (1 + 0.01 * k), dt, ds, w.ins_ids)
        td.append((time.perf_counter() - t0) / 100)
    out.append("%s value %.2f dual-chunk %.2f" % (name, np.median(tv) * 1e6, np.median(td) * 1e6))
    m.close()
print("%-10s" % (sys.argv[1] if len(sys.argv) > 1 else "product"), " | ".join(out), flush=True)
