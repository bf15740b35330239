import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
C = pfc.configs
shapes = [("c2", lambda: C.c2_box_on_plane(1)), ("c2 x4", lambda: C.c2_box_on_plane(4, montecarlo=True))]
for nb, nt in ((8, 6), (12, 10)):
    for n in (1, 4, 16):
        shapes.append((f"blob {nb}/tool {nt} x{n}", lambda nb=nb, nt=nt, n=n: C.c3_blob_tool(n, seed=31, n_div_blob=nb, n_div_tool=nt)))
out = []
for name, mk in shapes:
    w = mk(); m = C.build_scenario(w)
    for _ in range(4): m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    b = m.bind(w.pose, w.twist, w.s, w.ins_ids)
    for _ in range(4): b()
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        for _ in range(20): b()
        ts.append((time.perf_counter() - t0) / 20)
    out.append("%s: %.0f us (team %d)" % (name, np.median(ts) * 1e6, m.last_team()))
    m.close()
print("cap %s per %s | " % (os.environ.get("PFC_TEAM_CAP", "8"), os.environ.get("PFC_TEAM_PER", "256")) + " | ".join(out), flush=True)
