import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
C = pfc.configs
out = []
for name, mk in (("c1", C.c1_boxes), ("c2 x16", lambda: C.c2_box_on_plane(16, montecarlo=True)), ("c2 x64", lambda: C.c2_box_on_plane(64, montecarlo=True)), ("c4 (256)", lambda: C.c2_box_on_plane(256, montecarlo=True)),
                 ("c3 x8", lambda: C.c3_blob_tool(8, seed=31)), ("c3 x16", lambda: C.c3_blob_tool(16, seed=31)), ("12/10 x16", lambda: C.c3_blob_tool(16, seed=31, n_div_blob=12, n_div_tool=10)), ("c3 x128 [batched]", lambda: C.c3_blob_tool(128, seed=31))):
    w = mk(); m = C.build_scenario(w)
    for _ in range(4): m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    b = m.bind(w.pose, w.twist, w.s, w.ins_ids)
    for _ in range(4): b()
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        for _ in range(20): b()
        ts.append((time.perf_counter() - t0) / 20)
    out.append("%s %.1f" % (name, np.median(ts) * 1e6)); m.close()
print("bar items <= %s |" % os.environ.get("PFC_BAR_ITEMS", "16"), " | ".join(out), flush=True)
