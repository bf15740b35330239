import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
C = pfc.configs
rng = np.random.default_rng(3)
out = []
for name, mk in (("c1", C.c1_boxes), ("c2 x8", lambda: C.c2_box_on_plane(8, montecarlo=True)), ("c2 x16", lambda: C.c2_box_on_plane(16, montecarlo=True)), ("c2 x48", lambda: C.c2_box_on_plane(48, montecarlo=True)), ("c2 x80", lambda: C.c2_box_on_plane(80, montecarlo=True)),
                 ("c3r(8/6) x16", lambda: C.c3_blob_tool(16, seed=31, n_div_blob=8, n_div_tool=6))):
    w = mk(); m = C.build_scenario(w); n = w.n_items; nd = 6
    dp = rng.standard_normal((n, nd, 24)) * 1e-3; dt = rng.standard_normal((n, nd, 6)) * 1e-2; ds = rng.standard_normal((n, nd, 6)) * 1e-4
    dp2 = dp * 1.5
    def first(): return m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, dp, dt, ds, w.ins_ids)
    def further(): return m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, dp2, dt, ds, w.ins_ids)
    for _ in range(4): first(); further()
    tf, tm = [], []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(10): first()      # alternating seeds at the same point: every call after the first is a "further chunk"
        tf.append((time.perf_counter() - t0) / 10)
    out.append("%s %.1f" % (name, np.median(tf) * 1e6)); m.close()
print("bar keys <= %s | Dual(6) chunk, us:" % os.environ.get("PFC_BAR_KEYS", "64"), " | ".join(out), flush=True)
