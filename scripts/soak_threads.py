"""Soak run of the threading contract (include/pfc.h: one handle = one stream set; different handles may be used from
different host threads): N threads, each with its own handle, evaluate random slices concurrently (ctypes releases the
GIL during the calls) and compare with references computed beforehand on a single thread.
usage: python scripts/soak_threads.py [n_threads] [n_evals_per_thread]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
n_thr = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_ev = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(7)
w = pfc.configs.c3_blob_tool(700, seed=9, n_div_blob=6, n_div_tool=4)
w.s[:] = rng.standard_normal((w.n_items, 6)) * 1e-3
nd = 6
d_pose = rng.standard_normal((w.n_items, nd, 24)) * 1e-2
d_twist = rng.standard_normal((w.n_items, nd, 6)) * 0.1
d_s = rng.standard_normal((w.n_items, nd, 6)) * 1e-3
sizes = [1, 5, 64, 86, 200, 512, 513, 700]
cases = [(s, int(rng.integers(0, w.n_items - s + 1)), bool(rng.random() < 0.4)) for s in sizes for _ in range(3)]
ref = {}
f = pfc.configs.build_scenario(w)
f.set_option("fused", 0)
for c in cases:
    s, lo, dual = c
    sl = slice(lo, lo + s)
    if dual:
        ref[c] = f.force_all_elastic_intersections_dual(w.pose[sl], w.twist[sl], w.s[sl], d_pose[sl], d_twist[sl], d_s[sl], w.ins_ids[sl])
    else:
        ref[c] = f.force_all_elastic_intersections(w.pose[sl], w.twist[sl], w.s[sl], w.ins_ids[sl])
f.close()
bad = [0] * n_thr


errors = []


def worker(t):
    try:
        work(t)
    except Exception as e:      # a thread's exception must fail the run, not only print
        errors.append((t, repr(e)))
        raise


def work(t):
    r = np.random.default_rng(100 + t)
    m = pfc.configs.build_scenario(w)      # (finalized here, while other threads may be recording graphs)
    for _ in range(n_ev):
        c = cases[int(r.integers(0, len(cases)))]
        s, lo, dual = c
        sl = slice(lo, lo + s)
        if dual:
            got = m.force_all_elastic_intersections_dual(w.pose[sl], w.twist[sl], w.s[sl], d_pose[sl], d_twist[sl], d_s[sl], w.ins_ids[sl])
            tol = (1e-10, 1e-6, 1e-9, 1e-6)
            ok = np.array_equal(got[4], ref[c][4])
        else:
            got = m.force_all_elastic_intersections(w.pose[sl], w.twist[sl], w.s[sl], w.ins_ids[sl])
            tol = (1e-10, 1e-6)
            ok = np.array_equal(got[2], ref[c][2])
        for k, tl in enumerate(tol):
            ok = ok and np.abs(got[k] - ref[c][k]).max() <= tl * max(np.abs(ref[c][k]).max(), 1e-300)
        if not ok:
            bad[t] += 1
    m.close()


threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_thr)]
for th in threads:
    th.start()
for th in threads:
    th.join()
print(f"{n_thr} threads x {n_ev} evaluations, mismatches per thread: {bad}")
if errors:
    print("thread errors:", errors)
sys.exit(1 if (any(bad) or errors) else 0)
