"""Latency of one forceAllElasticIntersections! evaluation for the small reference-sized scenes (C1, C2):
this is what a Radau stage evaluation pays.  Host-buffer path (pfc_eval: H2D copy, kernels, D2H copy, sync)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
for name, w in (("C1 boxes (4 instructions)", pfc.configs.c1_boxes()),
                ("C2 box on plane (1 instruction, 972 tets)", pfc.configs.c2_box_on_plane(1)),
                ("C3 single pose (bristle)", pfc.configs.c3_blob_tool(1)),
                ("C4 256 scenes", pfc.configs.c2_box_on_plane(256, montecarlo=True)),
                ("C5 pile 2016 instructions", pfc.configs.c5_pile())):
    m = pfc.configs.build_scenario(w)
    if os.environ.get("PFC_BFS"):
        m.set_option("bfs_levels", int(os.environ["PFC_BFS"]))
    for _ in range(5):
        m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    n = 100

    def med(f, blocks=8, per=25):
        """Median over blocks of calls: a process sees ONE ~36 ms stall some 100-150 launches in (runtime / power-state
        housekeeping, with polling and with hipStreamSynchronize alike); a plain mean over a few hundred calls smears it."""
        ts = []
        for _ in range(blocks):
            t0 = time.perf_counter()
            for _ in range(per):
                r = f()
            ts.append((time.perf_counter() - t0) / per)
        return float(np.median(ts)), r

    dt, (wr, sd, ct) = med(lambda: m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids))
    st = m.stats()
    print(f"{name:45s} {dt*1e6:9.1f} us/eval   ops {st['candidates']:8d}  node tests {st['node_tests']:9d}  "
          f"-> {st['candidates']/dt:.3g} ops/s, {w.n_items/dt:.3g} contact pairs/s")
    bound = m.bind(w.pose, w.twist, w.s, w.ins_ids)
    for _ in range(5):
        bound()
    dtb, _ = med(bound)
    print(f"{'':45s} {dtb*1e6:9.1f} us/eval on bound buffers (MechanismScenario.bind: one foreign call, no array checks / allocations)")
    # the Dual evaluation of the same scene (6 partials, DENSE seeds: every (item, direction) carries non-zero partials --
    # keys whose seeds are all zero are skipped by the Dual passes, see the sparse figures below), alternating with value
    # evaluations as Radau does
    nd = 6
    rng = np.random.default_rng(7)
    dz = (rng.standard_normal((w.n_items, nd, 24)) * 1e-3, rng.standard_normal((w.n_items, nd, 6)) * 1e-2,
          rng.standard_normal((w.n_items, nd, 6)) * 1e-4)
    for _ in range(3):
        m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *dz, w.ins_ids)
    def pair():
        m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *dz, w.ins_ids)
        return m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    dd = med(pair, blocks=6, per=10)[0] - dt
    print(f"{'':45s} {dd*1e6:9.1f} us/Dual eval (6 partials)")
    # the further chunks of a Jacobian: same values, other partials (src/radau/radau_functions.jl:2-14)
    def again():
        return m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *dz, w.ins_ids)
    again()
    da = med(again, blocks=6, per=10)[0]
    print(f"{'':45s} {da*1e6:9.1f} us/further Dual eval at the same point (value pass reused: {m.last_dual_reused()})")
    # device-resident Dual evaluation (pfc_eval_dual_device + pfc_check): what the host-buffer figure above pays on top is
    # 288 B per (item, direction) of seeds over PCIe and the staging copies
    try:
        import torch
        dev = torch.device("cuda", 0)
        T = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
        ni = w.n_items
        t = [T(w.ins_ids, torch.int32), T(w.pose), T(w.twist), T(w.s), T(dz[0]), T(dz[1]), T(dz[2])]
        o = [torch.zeros((ni, 6), dtype=torch.float64, device=dev), torch.zeros((ni, 6), dtype=torch.float64, device=dev),
             torch.zeros((ni, nd, 6), dtype=torch.float64, device=dev), torch.zeros((ni, nd, 6), dtype=torch.float64, device=dev),
             torch.zeros((ni, 4), dtype=torch.int32, device=dev)]
        st = torch.cuda.current_stream().cuda_stream
        def dual_dev():
            for _ in range(40):
                m.eval_dual_device(ni, nd, *[x.data_ptr() for x in t], *[x.data_ptr() for x in o], st)
                if m.check() == 0:
                    return
        for _ in range(5):
            dual_dev()
        print(f"{'':45s} {med(dual_dev, blocks=6, per=10)[0] * 1e6:9.1f} us/Dual eval (6 partials), device-resident buffers")
        def more_dev():
            m.eval_dual_device_more(nd, t[4].data_ptr(), t[5].data_ptr(), t[6].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st)
            m.check()
        dual_dev()
        print(f"{'':45s} {med(more_dev, blocks=6, per=10)[0] * 1e6:9.1f} us/further Dual eval, device-resident (pfc_eval_dual_device_more)")
        # What a chunk of a Radau Jacobian looks like for a scene of many bodies: N_chunk = 6 state variables of ONE body
        # are seeded, so only the instructions that body takes part in carry non-zero partials (C5: 63 of 2 016; C4: the
        # one scene of 256 the body belongs to); all other keys are zero and skipped.
        act = None
        if "n_body" in w.meta:
            nb, k, pairs = w.meta["n_body"], 0, []
            for i in range(nb):
                for j in range(i + 1, nb):
                    pairs.append((i, j))
            act = np.array([5 in pr for pr in pairs])
        elif ni == 256:
            act = np.zeros(ni, dtype=bool); act[17] = True
        if act is not None:
            ds_ = [x * act[:, None, None] for x in dz]
            ts = [T(x) for x in ds_]
            def more_sparse():
                m.eval_dual_device_more(nd, ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st)
                m.check()
            dual_dev()
            print(f"{'':45s} {med(more_sparse, blocks=6, per=10)[0] * 1e6:9.1f} us/further Dual eval, device-resident, seeds of one body's states "
                  f"({int(act.sum())} of {ni} instructions carry partials)")
            def again_sparse():
                return m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *ds_, w.ins_ids)
            again_sparse()
            print(f"{'':45s} {med(again_sparse, blocks=6, per=10)[0] * 1e6:9.1f} us/further Dual eval, host buffers, the same seeds")
    except ImportError:
        pass
    m.close()
