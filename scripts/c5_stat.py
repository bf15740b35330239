import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
w = pfc.configs.c5_pile()
m = pfc.configs.build_scenario(w)
wr, sd, ct = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
nt = ct[:, 0]; nc = ct[:, 1]
print("items", len(nt), "with >1 node test", int((nt > 1).sum()), "with candidates", int((nc > 0).sum()))
o = np.argsort(-nt)[:12]
print("top node tests", nt[o].tolist()); print("their candidates", nc[o].tolist())
print("total tests", int(nt.sum()), "total cands", int(nc.sum()))
for L in (-1, 0, 1, 2, 3, 4):
    m.set_option("bfs_levels", L)
    for _ in range(5): m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    b = m.bind(w.pose, w.twist, w.s, w.ins_ids)
    for _ in range(5): b()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        for _ in range(20): b()
        ts.append((time.perf_counter() - t0) / 20)
    m.set_option("profile", 1); b(); sm = m.stage_ms(); m.set_option("profile", 0)
    print("bfs_levels %2d: %.1f us per eval (host buffers, bound)  stages %s parts %d" % (L, np.median(ts) * 1e6, {k: round(v * 1e3, 1) for k, v in sm.items()}, m.last_parts()), flush=True)
