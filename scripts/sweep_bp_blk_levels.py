import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
C = pfc.configs
shapes = [("c5 pile", lambda: C.c5_pile(), (0, 1, 2))]
for n, Ls in ((1, (3, 4, 5, 6, 7)), (4, (2, 3, 4, 5, 6)), (16, (1, 2, 3, 4, 5)), (64, (0, 1, 2, 3, 4)), (128, (0, 1, 2, 3)), (256, (0, 1, 2)), (600, (0, 1)), (900, (0, 1))):
    shapes.append((f"c3 full x {n}", lambda n=n: C.c3_blob_tool(n), Ls))
for n, Ls in ((64, (0, 1, 2, 3)), (512, (0, 1, 2))):
    shapes.append((f"c3 reduced (8/6) x {n}", lambda n=n: C.c3_blob_tool(n, n_div_blob=8, n_div_tool=6), Ls))
blk = os.environ.get("PFC_BP_BLK", "256")
for name, mk, Ls in shapes:
    w = mk()
    m = C.build_scenario(w)
    m.set_option("fused", 0); m.set_option("team", 0)
    res = []
    for L in (-1,) + tuple(Ls):
        m.set_option("bfs_levels", L)
        for _ in range(4): m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        b = m.bind(w.pose, w.twist, w.s, w.ins_ids)
        for _ in range(4): b()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            for _ in range(10): b()
            ts.append((time.perf_counter() - t0) / 10)
        res.append("L%d %.0f" % (L, np.median(ts) * 1e6))
    print("blk %s %-26s %s" % (blk, name, "  ".join(res)), flush=True)
    m.close()
