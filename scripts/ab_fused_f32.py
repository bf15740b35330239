"""A/B of the single-precision SAT filter in the one-launch kernel (option fused_f32): bound host-buffer evaluations, medians of
blocks, alternating the option on ONE handle per scene (same box, same buffers).  usage: python scripts/ab_fused_f32.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
C = pfc.configs
cases = (("c1", C.c1_boxes()), ("c2", C.c2_box_on_plane(1)), ("c2 tilted x1", C.c2_box_on_plane(1, montecarlo=True)), ("c4 x256", C.c2_box_on_plane(256, montecarlo=True)),
         ("pads x32", C.spoon_pencil_pads(32)), ("c3r x4", C.c3_blob_tool(4, n_div_blob=8, n_div_tool=6)), ("c3 single", C.c3_blob_tool(1)),
         ("c3 single b", C.c3_blob_tool(1, seed=5)), ("c3 x8", C.c3_blob_tool(8)))
for name, w in cases:
    m = C.build_scenario(w)
    b = m.bind(w.pose, w.twist, w.s, w.ins_ids)
    res = {}
    for opt in (1, 0, 1, 0):
        m.set_option("fused_f32", opt)
        for _ in range(60): b()
        ts = []
        for _ in range(12):
            t0 = time.perf_counter()
            for _ in range(100): b()
            ts.append((time.perf_counter() - t0) / 100)
        res.setdefault(opt, []).append(np.median(ts) * 1e6)
        c = b.counts.copy()
        res.setdefault("c", []).append(c)
    assert all(np.array_equal(res["c"][0], x) for x in res["c"])
    print("%-14s f32 on %6.1f %6.1f us | off %6.1f %6.1f us | path %d team %d | node tests %d" %
          (name, res[1][0], res[1][1], res[0][0], res[0][1], m.last_parts(), m.last_team(), int(b.counts[:, 0].sum())), flush=True)
    m.close()
