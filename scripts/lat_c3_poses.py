"""Single full-size C3 poses (8 different ones, each evaluated alone) by team size: mean / max latency over the poses."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
w = pfc.configs.c3_blob_tool(8)
teams = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "64,48,32,24,16,0").split(",")]
for team in teams:
    m = pfc.configs.build_scenario(w)
    m.set_option("team", team)
    res = []
    for k in range(8):
        a = (w.pose[k:k + 1], w.twist[k:k + 1], w.s[k:k + 1], w.ins_ids[k:k + 1])
        for _ in range(10): m.force_all_elastic_intersections(*a)
        blocks = []
        for _ in range(6):
            t0 = time.perf_counter()
            for _ in range(25): m.force_all_elastic_intersections(*a)
            blocks.append((time.perf_counter() - t0) / 25)
        res.append(np.median(blocks) * 1e6)
    print(f"team {team:3d}: mean {np.mean(res):6.1f} us, max {np.max(res):6.1f}, min {np.min(res):6.1f}  (path {m.last_parts()}, team {m.last_team()})")
    m.close()
