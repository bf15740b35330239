"""Latency of bound host-buffer evaluations of the small scenes with the package found under <root> (default: this repository) --
for a same-box comparison with an older tree (git archive <commit> | tar -x -C build/<name>; build its library there).
usage: python scripts/lat_bound.py [root] [label]"""
import os, sys, time
root = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
C = pfc.configs
out = []
for name, w in (("c1", C.c1_boxes()), ("c2", C.c2_box_on_plane(1)), ("c4", C.c2_box_on_plane(256, montecarlo=True)), ("c3 single", C.c3_blob_tool(1)), ("c3 x8", C.c3_blob_tool(8))):
    m = C.build_scenario(w)
    b = m.bind(w.pose, w.twist, w.s, w.ins_ids)
    for _ in range(100): b()
    ts = []
    for _ in range(15):
        t0 = time.perf_counter()
        for _ in range(100): b()
        ts.append((time.perf_counter() - t0) / 100)
    out.append("%s %.1f" % (name, np.median(ts) * 1e6))
    m.close()
# eight different single full-size poses on one handle: mean and worst of their medians (one pose alone is one descent shape)
ws = [C.c3_blob_tool(1, seed=100 + k) for k in range(8)]
m = C.build_scenario(ws[0])
bs = [m.bind(w.pose, w.twist, w.s, w.ins_ids) for w in ws]
med = []
for b in bs:
    for _ in range(40): b()
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        for _ in range(60): b()
        ts.append((time.perf_counter() - t0) / 60)
    med.append(np.median(ts) * 1e6)
out.append("c3 8 poses mean %.1f worst %.1f" % (np.mean(med), np.max(med)))
m.close()
print("%-10s" % (sys.argv[2] if len(sys.argv) > 2 else "here"), " | ".join(out), flush=True)
