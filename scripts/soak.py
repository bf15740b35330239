"""Soak run of the host paths: one long-lived handle, hundreds of evaluations of random sizes (1 .. 700 items), value
and Dual mixed at random -- zero-copy and staged paths, one-graph and two-stage Dual paths, graph re-captures, list
growth, the fused small-scene kernel with its polled completion and its in-kernel Dual passes -- each compared with a fresh
handle that only ever sees that one call and always takes the batched path (option fused = 0).  A third of the Dual
evaluations repeat the point of the previous Dual evaluation with other seeds and, sometimes, another number of
directions (the chunks of a Jacobian: the value pass is reused on every path that can).
usage: python scripts/soak.py [n_evals] [big | reg]      reg: all-regularized box-on-plane scenes (small-scene Dual passes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
n_evals = int(sys.argv[1]) if len(sys.argv) > 1 else 300
big = len(sys.argv) > 2 and sys.argv[2] == "big"      # batches around the two-halves threshold (value evaluations split)
rng = np.random.default_rng(2026)
reg = len(sys.argv) > 2 and sys.argv[2] == "reg"
w = (pfc.configs.c2_box_on_plane(700, montecarlo=True, n_div=2) if reg else
     pfc.configs.c3_blob_tool(3000 if big else 700, seed=9, n_div_blob=6, n_div_tool=4))
w.s[:] = rng.standard_normal((w.n_items, 6)) * 1e-3
nd = 6
d_pose = rng.standard_normal((w.n_items, nd, 24)) * 1e-2
d_twist = rng.standard_normal((w.n_items, nd, 6)) * 0.1
d_s = rng.standard_normal((w.n_items, nd, 6)) * 1e-3
m = pfc.configs.build_scenario(w)
if os.environ.get("PFC_SOAK_POISON"):      # the long-lived handle starts every evaluation from poisoned work lists
    m.set_option("poison", 1)
bad = 0
kinds = {"value": 0, "dual": 0, "dual, value pass reused": 0}
last_dual_sl = None
for it in range(n_evals):
    size = int(rng.choice([5, 600, 1023, 1024, 1025, 1500, 2048, 3000] if big else [1, 2, 5, 17, 64, 85, 86, 200, 511, 512, 513, 700]))
    lo = int(rng.integers(0, w.n_items - size + 1))
    sl = slice(lo, lo + size)
    dual = rng.random() < 0.5
    ndk = nd
    if dual and last_dual_sl is not None and rng.random() < 0.5:      # the next chunk of the same Jacobian
        sl = last_dual_sl
        size = sl.stop - sl.start
        ndk = int(rng.choice([nd, nd, 3]))
        d_pose[sl] = rng.standard_normal((size, nd, 24)) * 1e-2
        d_twist[sl] = rng.standard_normal((size, nd, 6)) * 0.1
    last_dual_sl = sl if dual else None
    if it % 50 == 0:
        print(f"eval {it} ...", flush=True)
    f = pfc.configs.build_scenario(w)
    f.set_option("fused", 0)
    if dual:
        sd = (np.ascontiguousarray(d_pose[sl][:, :ndk]), np.ascontiguousarray(d_twist[sl][:, :ndk]), np.ascontiguousarray(d_s[sl][:, :ndk]))
        got = m.force_all_elastic_intersections_dual(w.pose[sl], w.twist[sl], w.s[sl], *sd, w.ins_ids[sl])
        if m.last_dual_reused():
            kinds["dual, value pass reused"] += 1
        f.set_option("dual_reuse", 0)
        ref = f.force_all_elastic_intersections_dual(w.pose[sl], w.twist[sl], w.s[sl], *sd, w.ins_ids[sl])
        pairs = ((got[0], ref[0], 1e-10), (got[1], ref[1], 1e-6), (got[2], ref[2], 1e-9), (got[3], ref[3], 1e-6))
        cg, cr = got[4], ref[4]
    else:
        got = m.force_all_elastic_intersections(w.pose[sl], w.twist[sl], w.s[sl], w.ins_ids[sl])
        ref = f.force_all_elastic_intersections(w.pose[sl], w.twist[sl], w.s[sl], w.ins_ids[sl])
        pairs = ((got[0], ref[0], 1e-10), (got[1], ref[1], 1e-6))
        cg, cr = got[2], ref[2]
    f.close()
    kinds["dual" if dual else "value"] += 1
    ok = np.array_equal(cg, cr)
    for a, b, tol in pairs:
        ok = ok and np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-300)
    if not ok:
        bad += 1
        print("MISMATCH at evaluation", it, "size", size, "dual" if dual else "value", flush=True)
m.close()
print(f"{n_evals} evaluations ({kinds}), mismatches: {bad}")
sys.exit(1 if bad else 0)
