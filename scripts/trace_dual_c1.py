"""Diagnostic: a few C1 Dual evaluations for a rocprofv3 --kernel-trace timeline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
w = pfc.configs.c1_boxes()
m = pfc.configs.build_scenario(w)
nd = 6
rng = np.random.default_rng(7)      # dense seeds: keys with all-zero seeds are skipped by the Dual passes
dz = (rng.standard_normal((w.n_items, nd, 24)) * 1e-3, rng.standard_normal((w.n_items, nd, 6)) * 1e-2, rng.standard_normal((w.n_items, nd, 6)) * 1e-4)
for _ in range(30):
    m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *dz, w.ins_ids)
m.close()
