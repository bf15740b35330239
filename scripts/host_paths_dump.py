"""Results of the host-buffer entry points over the sizes where their staging changes (in place up to 512 items / pairs from
pinned memory, up to 4 096 with BAR-resident inputs, staged copies above), written to an .npz -- run once as is and once with
PFC_NO_BAR_INPUTS=1 and compare (tests/test_gpu_scale.py::test_host_paths_without_bar_resident_inputs).
usage: python scripts/host_paths_dump.py out.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
C = pfc.configs
out = {}
cases = (("c1", C.c1_boxes()), ("c2x256", C.c2_box_on_plane(256, montecarlo=True)), ("c2x1000", C.c2_box_on_plane(1000, montecarlo=True)),
         ("c2x1500", C.c2_box_on_plane(1500, montecarlo=True)), ("c2x3000", C.c2_box_on_plane(3000, montecarlo=True)),
         ("c3r x 40", C.c3_blob_tool(40, seed=5, n_div_blob=8, n_div_tool=6)), ("c2x5000", C.c2_box_on_plane(5000, montecarlo=True)))
for name, w in cases:
    m = C.build_scenario(w)
    for rep in range(2):      # the second call runs on settled buffers (and replays graphs where the path records them)
        wr, sd, cn = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    out[name + "/wrench"], out[name + "/sdot"], out[name + "/counts"] = wr.copy(), sd.copy(), cn.copy()
    # OTHER inputs at the same size on the same handle (identical repetitions would not notice a stale line of
    # the BAR-resident input block in the GPU's L2, nor a write-combined store that arrives behind the doorbell): the items in
    # reverse order, then every pose / twist / state a little off
    n = w.n_items
    rng = np.random.default_rng(5)
    for var in range(2):
        if var == 0:
            ids, po, tw, ss = w.ins_ids[::-1].copy(), w.pose[::-1].copy(), w.twist[::-1].copy(), w.s[::-1].copy()
        else:
            ids, po, tw, ss = w.ins_ids, w.pose.copy(), w.twist * 1.25 + 0.01, w.s * 0.5 + 1e-3
            po[:, 9:12] += rng.uniform(-2e-4, 2e-4, (n, 3))                   # t21 shifts; x_r1_r2 follows: t12 = -R12 t21
            for k in range(n):
                po[k, 21:24] = -(po[k, 12:21].reshape(3, 3, order="F") @ po[k, 9:12])
        wr, sd, cn = m.force_all_elastic_intersections(po, tw, ss, ids)
        out["%s/var%d/wrench" % (name, var)], out["%s/var%d/sdot" % (name, var)], out["%s/var%d/counts" % (name, var)] = wr.copy(), sd.copy(), cn.copy()
        if var == 0:      # the reversed batch is the first one, item for item
            assert np.array_equal(cn[::-1], out[name + "/counts"]), name
    if w.n_items <= 1000:
        rng = np.random.default_rng(11)
        for nd in (2, 6):
            n = w.n_items
            dp = rng.standard_normal((n, nd, 24)) * 1e-3; dt = rng.standard_normal((n, nd, 6)) * 1e-2; ds = rng.standard_normal((n, nd, 6)) * 1e-4
            for rep in range(2):
                r = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, dp * (1 + rep), dt, ds, w.ins_ids)
            out["%s/dual%d/dwrench" % (name, nd)], out["%s/dual%d/dsdot" % (name, nd)] = np.array(r[2]), np.array(r[3])
    m.close()
np.savez(sys.argv[1], **out)
print("wrote", len(out), "arrays")
