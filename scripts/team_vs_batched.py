"""A few full-size C3 poses: the one-launch kernel with teams of workgroups (default) against the batched path (option team = 0).
usage: python scripts/team_vs_batched.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
for n in [int(x) for x in sys.argv[1:]] or (1, 2, 4, 8, 16, 24, 32, 48, 64, 80):
    w = pfc.configs.c3_blob_tool(n, seed=31)
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
    print("c3 full x %3d: default %.0f us (team of %d, parts %d) | batched %.0f us" % (n, res[0][0], res[0][1], res[0][2], res[1][0]), flush=True)
