import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
for nb, nt, ns in ((15, 12, (8, 16, 24, 32, 48)), (18, 14, (8, 16, 24, 32)), (22, 16, (8, 12, 16, 20, 24, 32))):
    for n in ns:
        w = pfc.configs.c3_blob_tool(n, seed=31, n_div_blob=nb, n_div_tool=nt)
        res = []
        for team in (-1, 0):
            m = pfc.configs.build_scenario(w)
            if team == 0: m.set_option("team", 0)
            for _ in range(4): out = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
            b = m.bind(w.pose, w.twist, w.s, w.ins_ids)
            for _ in range(4): b()
            ts = []
            for _ in range(7):
                t0 = time.perf_counter()
                for _ in range(10): b()
                ts.append((time.perf_counter() - t0) / 10)
            res.append((np.median(ts) * 1e6, m.last_team(), m.last_parts(), out[2].copy()))
            m.close()
        assert np.array_equal(res[0][3], res[1][3])
        leaves = sum(ms.tree.n_leaf for ms in w.meshes)
        print("blob %2d / tool %2d (%5d leaves) x %3d: default %.0f us (team %d, parts %d) | batched %.0f us" % (nb, nt, leaves, n, res[0][0], res[0][1], res[0][2], res[1][0]), flush=True)
