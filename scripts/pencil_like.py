"""Latency of a pencil.jl-sized bristle scene (test/pencil.jl: sphere pads of 320 tets against a < 100-triangle pencil surface,
bristle friction): 8 instructions, value and Dual(6) evaluations through host buffers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
w = pfc.configs.c3_blob_tool(8, seed=3, n_div_blob=4, n_div_tool=2, distance=0.195)
for fused in (1, 0):
    m = pfc.configs.build_scenario(w)
    m.set_option("fused", fused)
    nd = 6
    rng = np.random.default_rng(7)      # dense seeds: keys with all-zero seeds are skipped by the Dual passes
    dz = (rng.standard_normal((w.n_items, nd, 24)) * 1e-3, rng.standard_normal((w.n_items, nd, 6)) * 1e-2, rng.standard_normal((w.n_items, nd, 6)) * 1e-4)
    for _ in range(10):
        m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *dz, w.ins_ids)
    def med(f, blocks=8, per=25):        # median over blocks: a process sees one ~36 ms runtime stall early on
        ts = []
        for _ in range(blocks):
            t0 = time.perf_counter()
            for _ in range(per):
                r = f()
            ts.append((time.perf_counter() - t0) / per)
        return float(np.median(ts)), r
    tv, out = med(lambda: m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids))
    pv = m.last_parts()
    # the first chunk of a Jacobian (a value evaluation at another point precedes it, as in Radau), then its further chunks
    def first():
        m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        return m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *dz, w.ins_ids)
    t1, _ = med(first, blocks=6, per=10)
    td, _ = med(lambda: m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *dz, w.ins_ids))
    print(f"fused={fused}: value {tv*1e6:.1f} us (path {pv}), Dual(6) first chunk {(t1 - tv)*1e6:.1f} us, further chunks {td*1e6:.1f} us "
          f"(value pass reused: {m.last_dual_reused()}); counts {out[2].sum(axis=0)}")
    m.close()
