import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
C = pfc.configs
shapes = [("c5 pile", lambda: C.c5_pile()), ("c3 reduced (8/6) x 2000", lambda: C.c3_blob_tool(2000, n_div_blob=8, n_div_tool=6)),
          ("c3 reduced (8/6) x 1200", lambda: C.c3_blob_tool(1200, n_div_blob=8, n_div_tool=6)),
          ("c3 reduced (12/10) x 1500", lambda: C.c3_blob_tool(1500, n_div_blob=12, n_div_tool=10)),
          ("c2 x 2500", lambda: C.c2_box_on_plane(2500, montecarlo=True)), ("c3 full x 1100", lambda: C.c3_blob_tool(1100)), ("c3 full x 1500", lambda: C.c3_blob_tool(1500))]
blk = os.environ.get("PFC_BP_BLK", "256")
for name, mk in shapes:
    w = mk()
    m = C.build_scenario(w)
    m.set_option("fused", 0); m.set_option("team", 0)
    res = []
    for split in (1025, 0):
        for L in (-1, 0, 1):
            m.set_option("bfs_levels", L); m.set_option("split_min", split)
            for _ in range(4): m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
            b = m.bind(w.pose, w.twist, w.s, w.ins_ids)
            for _ in range(4): b()
            ts = []
            for _ in range(7):
                t0 = time.perf_counter()
                for _ in range(8): b()
                ts.append((time.perf_counter() - t0) / 8)
            res.append("%s L%d %.0f" % ("split" if split else "whole", L, np.median(ts) * 1e6))
    print("blk %s %-26s leaves %5d | %s" % (blk, name, max(ms.tree.n_leaf for ms in w.meshes), "  ".join(res)), flush=True)
    m.close()
