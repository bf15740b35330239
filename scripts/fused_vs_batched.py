"""Mid-sized pairs (up to kFusedMaxLeaves leaves), up to 256 items: the one-launch kernel (default: a workgroup or a small team per
item) against the batched path (option fused = 0).  usage: python scripts/fused_vs_batched.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
for nb, nt in ((8, 6), (12, 10), (15, 12)):
    for n in (4, 16, 64, 128, 256):
        w = pfc.configs.c3_blob_tool(n, seed=31, n_div_blob=nb, n_div_tool=nt)
        res = []
        for fused in (1, 0):
            m = pfc.configs.build_scenario(w)
            m.set_option("fused", fused)
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
        print("blob %2d / tool %2d (%5d leaves) x %3d: default %.0f us (parts %d, team %d) | batched %.0f us" % (nb, nt, leaves, n, res[0][0], res[0][2], res[0][1], res[1][0]), flush=True)
