"""Option "fixed_order" (sorted candidate list, run records added in chunk order, the Dual passes' eigen-decomposition on the value
pass's K): agreement with the default path, and run-to-run repeatability of values, K and partials on fresh handles.
usage: python scripts/fixed_order_probe.py [c5|c5ps|c3x64|c4] [--time]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
C = pfc.configs


def workload(name):
    if name == "c5":
        return C.c5_pile()
    if name == "c5ps":
        return C.c5_pile(pencil_spoon=True)
    if name == "c3x64":
        return C.c3_blob_tool(64)
    if name == "c4":
        return C.c2_box_on_plane(256, montecarlo=True)
    raise KeyError(name)


def run(w, fixed, dual, seeds):
    m = C.build_scenario(w)
    if fixed:
        m.set_option("fixed_order", 1)
    t0 = time.perf_counter()
    if dual:
        out = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *seeds, w.ins_ids)
    else:
        out = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    t1 = time.perf_counter()
    if dual:
        out = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *seeds, w.ins_ids)
    else:
        out = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    t2 = time.perf_counter()
    K = []
    counts = out[-1]
    for k in range(w.n_items):
        if counts[k, 3] > 0 and w.instructions[int(w.ins_ids[k])].model != "regularized" and len(K) < 400:
            st = m.debug_stiffness(k)
            if st is not None:
                K.append((k, st[0].copy(), st[1].copy()))
    m.close()
    return out, K, (t1 - t0, t2 - t1)


def per_item(a, b):
    n = a.shape[0]
    d = np.abs(a - b).reshape(n, -1).max(1)
    s = np.maximum(np.abs(b).reshape(n, -1).max(1), 1e-300)
    return d / s


for name in ([a for a in sys.argv[1:] if not a.startswith("--")] or ([] if "--time" in sys.argv else ["c5", "c3x64", "c4"])):
    w = workload(name)
    n = w.n_items
    rng = np.random.default_rng(11)
    nd = 2
    seeds = (rng.standard_normal((n, nd, 24)) * 1e-2, rng.standard_normal((n, nd, 6)) * 0.1, rng.standard_normal((n, nd, 6)) * 1e-3)
    for dual in (False, True):
        base, Kb, tb = run(w, False, dual, seeds)
        f1, K1, t1 = run(w, True, dual, seeds)
        f2, K2, t2 = run(w, True, dual, seeds)
        b2, Kb2, _ = run(w, False, dual, seeds)
        assert np.array_equal(base[-1], f1[-1]), "counters differ"
        incontact = int((base[-1][:, 3] > 0).sum())
        print("%s %s: %d items, %d in contact | second call default %.0f us, fixed_order %.0f us" %
              (name, "dual" if dual else "value", n, incontact, tb[1] * 1e6, t1[1] * 1e6), flush=True)
        print("   fixed vs default: wrench %.2e  sdot %.2e (worst item, relative)" %
              (per_item(f1[0], base[0]).max(), per_item(f1[1], base[1]).max()))
        print("   two fresh fixed handles: wrench %.2e  sdot %.2e | bit-equal K: %d of %d items, K^-1/2: %d" %
              (per_item(f1[0], f2[0]).max(), per_item(f1[1], f2[1]).max(),
               sum(np.array_equal(a[1], b[1]) for a, b in zip(K1, K2)), len(K1), sum(np.array_equal(a[2], b[2]) for a, b in zip(K1, K2))))
        print("   two fresh default handles: wrench %.2e  sdot %.2e | bit-equal K: %d of %d items" %
              (per_item(base[0], b2[0]).max(), per_item(base[1], b2[1]).max(),
               sum(np.array_equal(a[1], b[1]) for a, b in zip(Kb, Kb2)), len(Kb)))
        if dual:
            pf = np.maximum(per_item(f1[2], f2[2]), per_item(f1[3], f2[3]))
            pb = np.maximum(per_item(base[2], b2[2]), per_item(base[3], b2[3]))
            px = np.maximum(per_item(f1[2], base[2]), per_item(f1[3], base[3]))
            print("   partials, two fresh handles: fixed worst %.2e (items > 1e-12: %d, > 1e-6: %d) | default worst %.2e (items > 1e-12: %d, > 1e-6: %d)" %
                  (pf.max(), int((pf > 1e-12).sum()), int((pf > 1e-6).sum()), pb.max(), int((pb > 1e-12).sum()), int((pb > 1e-6).sum())))
            print("   partials, fixed vs default: worst %.2e (items > 1e-6: %d)" % (px.max(), int((px > 1e-6).sum())))


def timing():
    """Cost of the option: warm evaluations through host buffers, median of 15."""
    import statistics
    for name, w in (("c5 (2 016 pairs)", C.c5_pile()), ("c3 x 64", C.c3_blob_tool(64)), ("c4 (256 scenes)", C.c2_box_on_plane(256, montecarlo=True)),
                    ("c3 x 2048", C.c3_blob_tool(2048))):
        n = w.n_items
        rng = np.random.default_rng(3)
        nd = 6
        seeds = (rng.standard_normal((n, nd, 24)) * 1e-2, rng.standard_normal((n, nd, 6)) * 0.1, rng.standard_normal((n, nd, 6)) * 1e-3)
        row = []
        for fixed in (0, 1):
            m = C.build_scenario(w)
            m.set_option("fixed_order", fixed)
            tv, td = [], []
            for k in range(18):
                t0 = time.perf_counter()
                m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
                tv.append(time.perf_counter() - t0)
            for k in range(8 if n > 1000 else 18):
                t0 = time.perf_counter()
                m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *seeds, w.ins_ids)
                td.append(time.perf_counter() - t0)
            m.close()
            row.append((statistics.median(tv[3:]) * 1e6, statistics.median(td[3:]) * 1e6))
        print("%-18s value %8.0f -> %8.0f us | Dual(6) %8.0f -> %8.0f us   (default -> fixed_order, host buffers)" %
              (name, row[0][0], row[1][0], row[0][1], row[1][1]), flush=True)


if "--time" in sys.argv:
    timing()
