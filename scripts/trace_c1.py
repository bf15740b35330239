"""Diagnostic: a few C1 evaluations for a rocprofv3 --kernel-trace timeline (where do the ~100 us of a small scene go?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
w = pfc.configs.c1_boxes() if len(sys.argv) < 2 or sys.argv[1] == "c1" else pfc.configs.c3_blob_tool(1)
m = pfc.configs.build_scenario(w)
for _ in range(30):
    m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
m.close()
