"""C5 (or argv[1] = c3s: one full-size C3 pose on the batched path): per-stage times (HIP events, option profile) of the unsplit
evaluation by seed level.  usage: [PFC_LIB=...] python scripts/c5_stage.py [c5|c3s] [levels ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
levels = [int(x) for x in sys.argv[2:]] or [-1]
w = pfc.configs.c5_pile() if cfg == "c5" else pfc.configs.c3_blob_tool(1)
m = pfc.configs.build_scenario(w)
m.set_option("split_min", 0); m.set_option("fused", 0); m.set_option("team", 0)
for L in levels:
    m.set_option("bfs_levels", L)
    m.set_option("profile", 0)
    for _ in range(4): m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    b = m.bind(w.pose, w.twist, w.s, w.ins_ids)
    for _ in range(5): b()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        for _ in range(20): b()
        ts.append((time.perf_counter() - t0) / 20)
    m.set_option("profile", 1)
    acc = {}
    for _ in range(20):
        m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        for k, v in m.stage_ms().items(): acc[k] = acc.get(k, 0) + v / 20
    print("%-12s %s levels %2d: %6.1f us per evaluation (host buffers) | stages (us) %s" % (os.environ.get("PFC_LIB", "product").split("/")[-1], cfg, L, np.median(ts) * 1e6, {k: round(v * 1e3, 1) for k, v in acc.items()}), flush=True)
