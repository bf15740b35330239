"""C5: the Dual(6) chunks of ONE body (its 63 instructions passed as a call of their own through ins_ids), host buffers:
what a shim that knows which instructions a chunk of the Jacobian touches pays.  usage: python scripts/c5_subset.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
w = pfc.configs.c5_pile()
nb = w.meta["n_body"]
pairs = [(i, j) for i in range(nb) for j in range(i + 1, nb)]
rng = np.random.default_rng(3)
m = pfc.configs.build_scenario(w)
nd = 6
def med(f, blocks=6, per=10):
    ts = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        for _ in range(per):
            f()
        ts.append((time.perf_counter() - t0) / per)
    return float(np.median(ts)) * 1e6
subs = []
for b in (5, 21, 42):
    idx = np.array([k for k, pr in enumerate(pairs) if b in pr], dtype=np.int64)
    sd = (rng.standard_normal((len(idx), nd, 24)) * 1e-3, rng.standard_normal((len(idx), nd, 6)) * 1e-2, rng.standard_normal((len(idx), nd, 6)) * 1e-4)
    subs.append((idx, sd, np.ascontiguousarray(w.pose[idx]), np.ascontiguousarray(w.twist[idx]), np.ascontiguousarray(w.s[idx]), np.ascontiguousarray(w.ins_ids[idx])))
state = {"k": 0}
def first():      # another body's instructions each time: the value pass of the subset is new
    idx, sd, p, t, s, ids = subs[state["k"] % 3]; state["k"] += 1
    m.force_all_elastic_intersections_dual(p, t, s, *sd, ids)
def again():
    idx, sd, p, t, s, ids = subs[0]
    m.force_all_elastic_intersections_dual(p, t, s, *sd, ids)
for _ in range(6): first()
print(f"C5, the 63 instructions of one body as a call of their own (ins_ids), Dual(6), host buffers: first chunk {med(first):.1f} us", flush=True)
again(); again()
print(f"   further chunks of the same body {med(again):.1f} us (value pass reused: {m.last_dual_reused()})")
