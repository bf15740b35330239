"""Per-evaluation cost of the CPU restatement (oracle, one thread = the reference's execution model) next to the
library's host-buffer latency for the five configurations: where routing a scene through the GPU starts to pay.
usage: python scripts/crossover.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
for name, w in (("C1 boxes (4 instructions)", pfc.configs.c1_boxes()),
                ("C2 box on plane", pfc.configs.c2_box_on_plane(1)),
                ("C3 single pose (bristle)", pfc.configs.c3_blob_tool(1)),
                ("C4 256 scenes", pfc.configs.c2_box_on_plane(256, montecarlo=True)),
                ("C5 pile 2016 instructions", pfc.configs.c5_pile())):
    m = pfc.configs.build_scenario(w)
    for _ in range(5):
        m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    ts = []
    for _ in range(8):            # median over blocks (one-off runtime stalls, see scripts/latency.py)
        t0 = time.perf_counter()
        for _ in range(25):
            m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        ts.append((time.perf_counter() - t0) / 25)
    t_gpu = float(np.median(ts))
    m.close()
    # the CPU side as bench.py times it: one C call evaluating all items serially (pfo_eval_batch)
    from oracle import oracle as O
    om = [O.OracleMesh(ms.mesh, ms.tree, ms.Ebar or 0.0) for ms in w.meshes]
    oi, m1, m2 = [], [], []
    for c in w.instructions:
        mu_s, mu_d = pfc.scenario.determine_mu_s_mu_d(c.mu_s, c.mu_d)
        if c.model == "regularized":
            oi.append(O.make_ins(c.chi, c.n_quad_rule, O.REGULARIZED, mu_s, mu_d, v_c=c.v_tol))
        else:
            oi.append(O.make_ins(c.chi, c.n_quad_rule, O.BRISTLE, mu_s, mu_d, tau=c.tau, k_bar=c.k_bar, magic=c.magic))
        m1.append(c.id_1); m2.append(c.id_2)
    O.evaluate_batch(om, oi, m1, m2, w.ins_ids, w.pose, w.twist, w.s, 1)
    reps = 3 if w.n_items > 100 else 50
    t0 = time.perf_counter()
    for _ in range(reps):
        O.evaluate_batch(om, oi, m1, m2, w.ins_ids, w.pose, w.twist, w.s, 1)
    t_cpu = (time.perf_counter() - t0) / reps
    print(f"{name:28s} GPU {t_gpu * 1e6:9.1f} us   CPU oracle (1 thread) {t_cpu * 1e6:11.1f} us   ratio {t_cpu / t_gpu:7.1f}")
