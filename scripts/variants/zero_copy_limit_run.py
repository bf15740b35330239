import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
C = pfc.configs
out = []
for name, mk in (("c5", C.c5_pile), ("c2 x600", lambda: C.c2_box_on_plane(600, montecarlo=True)), ("c2 x1000", lambda: C.c2_box_on_plane(1000, montecarlo=True)), ("c3 x600", lambda: C.c3_blob_tool(600)), ("c3 x1024", lambda: C.c3_blob_tool(1024)), ("c3 x2048", lambda: C.c3_blob_tool(2048))):
    w = mk(); m = C.build_scenario(w)
    for _ in range(4): m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    b = m.bind(w.pose, w.twist, w.s, w.ins_ids)
    for _ in range(4): b()
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        for _ in range(10): b()
        ts.append((time.perf_counter() - t0) / 10)
    out.append("%s %.0f" % (name, np.median(ts) * 1e6)); m.close()
print("in place up to %s items |" % os.environ.get("PFC_ZC", "512"), " | ".join(out), flush=True)
