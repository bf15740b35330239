"""Experiment: the Dual evaluation of a 2048-pose C3 batch as two concurrent 1024-pose halves (two handles, two host
threads; ctypes releases the GIL during the call) against one call."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, pfc_pkg
from test_oracle_dual import tangents
pfc = pfc_pkg.load()
n, nd = 2048, 6
w = pfc.configs.c3_blob_tool(n, seed=20260103)
rng = np.random.default_rng(0)
w.s[:] = rng.standard_normal((n, 6)) * 1e-3
dq = rng.standard_normal((n, nd, 6)) * np.array([1, 1, 1, 0.05, 0.05, 0.05])
d_pose = np.stack([tangents(w.pose[k][:9].reshape(3, 3, order="F"), w.pose[k][9:12], dq[k]) for k in range(n)])
d_twist = rng.standard_normal((n, nd, 6)); d_s = rng.standard_normal((n, nd, 6)) * 1e-3
hs = [pfc.configs.build_scenario(w) for _ in range(2)]
def call(h, a, b):
    return h.force_all_elastic_intersections_dual(w.pose[a:b], w.twist[a:b], w.s[a:b], d_pose[a:b], d_twist[a:b], d_s[a:b], w.ins_ids[a:b])
for _ in range(2):
    call(hs[0], 0, n); call(hs[0], 0, n // 2); call(hs[1], n // 2, n)
t = time.perf_counter()
for _ in range(5): call(hs[0], 0, n)
t1 = (time.perf_counter() - t) / 5
t = time.perf_counter()
for _ in range(5):
    th = [threading.Thread(target=call, args=(hs[j], j * n // 2, (j + 1) * n // 2)) for j in range(2)]
    for x in th: x.start()
    for x in th: x.join()
t2 = (time.perf_counter() - t) / 5
print(f"one call {t1*1e3:.2f} ms; two concurrent halves {t2*1e3:.2f} ms")
