import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
out = []
for name, mk in (("c3 x1", lambda: pfc.configs.c3_blob_tool(1)), ("c3 x1 b", lambda: pfc.configs.c3_blob_tool(1, seed=77)), ("c3 x4", lambda: pfc.configs.c3_blob_tool(4, seed=31)), ("c3 x8", lambda: pfc.configs.c3_blob_tool(8, seed=31)), ("c3 x12", lambda: pfc.configs.c3_blob_tool(12, seed=31)),
                 ("15/12 x16", lambda: pfc.configs.c3_blob_tool(16, seed=31, n_div_blob=15, n_div_tool=12)), ("12/10 x4", lambda: pfc.configs.c3_blob_tool(4, seed=31, n_div_blob=12, n_div_tool=10)), ("c2", lambda: pfc.configs.c2_box_on_plane(1))):
    w = mk(); m = pfc.configs.build_scenario(w)
    for _ in range(4): m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    b = m.bind(w.pose, w.twist, w.s, w.ins_ids)
    for _ in range(4): b()
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        for _ in range(20): b()
        ts.append((time.perf_counter() - t0) / 20)
    out.append("%s %.0f" % (name, np.median(ts) * 1e6)); m.close()
print(os.environ.get("PFC_LIB", "product(16)").split("/")[-1], "|", " | ".join(out), flush=True)
