"""A single full-size C3 pose (BASELINE config 3 as written: one 9 680-tet x 5 120-triangle pair) through host buffers, by
team size (option "team": workgroups per item of the one-launch kernel; 0 = the batched launch sequence)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
w = pfc.configs.c3_blob_tool(1)
for team in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "64,48,32,16,8,0".split(","))]:
    m = pfc.configs.build_scenario(w)
    m.set_option("team", team)
    for _ in range(20): m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    blocks = []
    for _ in range(12):
        t0 = time.perf_counter()
        for _ in range(25): out = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        blocks.append((time.perf_counter() - t0) / 25)
    print(f"team {team}: {np.median(blocks)*1e6:.1f} us/eval path {m.last_parts()} counts {out[2][0]}")
    m.close()
