"""Share of the broadphase's node tests that the single-precision filter leaves to the exact Float64 test, per config
(batched path).  usage: [PFC_LIB=...] python scripts/undecided.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
L = pfc._lib.lib()
cfgs = (("C2 x 64 (axis-aligned: every cross axis of a resting box is degenerate)", pfc.configs.c2_box_on_plane(64, montecarlo=True)),
        ("C3, 2 048 poses", pfc.configs.c3_blob_tool(2048)),
        ("C5 pile", pfc.configs.c5_pile()))
for name, w in cfgs:
    m = pfc.configs.build_scenario(w)
    m.set_option("fused", 0)
    for _ in range(3):
        m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    out = (C.c_longlong * 16)()
    assert L.pfc_debug_stamps(m._h, out) == 0
    st = m.stats()
    print("%-80s node tests %10d  settled exactly %8d  (%.2e)" % (name, st["node_tests"], out[7], out[7] / max(st["node_tests"], 1)), flush=True)
    m.close()
