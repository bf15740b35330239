import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
C = pfc.configs
rng = np.random.default_rng(3)
out = []
for name, mk in (("c2 x80", lambda: C.c2_box_on_plane(80, montecarlo=True)), ("c2 x128", lambda: C.c2_box_on_plane(128, montecarlo=True)), ("c4 (256)", lambda: C.c2_box_on_plane(256, montecarlo=True)), ("c3r(8/6) x100", lambda: C.c3_blob_tool(100, seed=31, n_div_blob=8, n_div_tool=6)), ("c3 x64", lambda: C.c3_blob_tool(64, seed=31))):
    w = mk(); m = C.build_scenario(w); n = w.n_items; nd = 6
    dp = rng.standard_normal((n, nd, 24)) * 1e-3; dt = rng.standard_normal((n, nd, 6)) * 1e-2; ds = rng.standard_normal((n, nd, 6)) * 1e-4
    w2 = w.pose.copy()
    def chunk(k): return m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, dp * (1.0 + 0.1 * k), dt, ds, w.ins_ids)
    def firstc(k):
        p = w.pose.copy(); p[:, 21] += 1e-9 * (k + 1)      # a new point: the value pass runs again
        return m.force_all_elastic_intersections_dual(p, w.twist, w.s, dp, dt, ds, w.ins_ids)
    for k in range(4): chunk(k)
    tf, tc = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        for k in range(8): firstc(k)
        tf.append((time.perf_counter() - t0) / 8)
        chunk(0)
        t0 = time.perf_counter()
        for k in range(8): chunk(k + 1)
        tc.append((time.perf_counter() - t0) / 8)
    out.append("%s first %.0f further %.0f" % (name, np.median(tf) * 1e6, np.median(tc) * 1e6)); m.close()
print("seeds in place up to %s keys |" % os.environ.get("PFC_ZK", "512"), " | ".join(out), flush=True)
