"""Evaluation time over a range of batch shapes on the batched path (fused / teams off): the data behind the workgroup size
of the depth-first broadphase for small and mid launches.  usage: [PFC_LIB=...] python scripts/sweep_shapes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
C = pfc.configs
shapes = [("c5 pile", lambda: C.c5_pile())]
for n in (300, 1000, 2500):
    shapes.append((f"c2 x {n}", lambda n=n: C.c2_box_on_plane(n, montecarlo=True)))
for n in (64, 512, 2000):
    shapes.append((f"c3 reduced (8/6) x {n}", lambda n=n: C.c3_blob_tool(n, n_div_blob=8, n_div_tool=6)))
for n in (1, 16, 128, 600, 1024, 2048):
    shapes.append((f"c3 full x {n}", lambda n=n: C.c3_blob_tool(n)))
tag = os.environ.get("PFC_LIB", "product").split("/")[-1]
for name, mk in shapes:
    w = mk()
    m = C.build_scenario(w)
    m.set_option("fused", 0); m.set_option("team", 0)
    for _ in range(4): m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    b = m.bind(w.pose, w.twist, w.s, w.ins_ids)
    for _ in range(4): b()
    ts = []
    reps = 10 if w.n_items <= 1024 else 4
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(reps): b()
        ts.append((time.perf_counter() - t0) / reps)
    print("%-12s %-28s %9.1f us" % (tag, name, np.median(ts) * 1e6), flush=True)
    m.close()
