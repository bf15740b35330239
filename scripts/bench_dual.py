"""Timing of pfc_eval_dual (value + n_dir partials, host buffers) against pfc_eval on the C3 batch, and the Dual
oracle on a few poses.  Usage: python scripts/bench_dual.py [poses] [n_dir] [reps]"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pfc_pkg
pfc = pfc_pkg.load()
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_oracle_dual import tangents

poses = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_dir = int(sys.argv[2]) if len(sys.argv) > 2 else 6
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
w = pfc.configs.c3_blob_tool(poses, seed=20260103)
rng = np.random.default_rng(0)
n = w.n_items
w.s[:] = rng.standard_normal((n, 6)) * 1e-3
dq = rng.standard_normal((n, n_dir, 6)) * np.array([1, 1, 1, 0.05, 0.05, 0.05])
d_pose = np.zeros((n, n_dir, 24))
for k in range(n):
    d_pose[k] = tangents(w.pose[k][:9].reshape(3, 3, order="F"), w.pose[k][9:12], dq[k])
d_twist = rng.standard_normal((n, n_dir, 6)); d_s = rng.standard_normal((n, n_dir, 6)) * 1e-3
m = pfc.configs.build_scenario(w)
for _ in range(2):
    m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, d_s, w.ins_ids)
t = time.perf_counter()
for _ in range(reps):
    _, _, counts = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
t_val = (time.perf_counter() - t) / reps
t = time.perf_counter()
for _ in range(reps):
    m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, d_s, w.ins_ids)
t_dual = (time.perf_counter() - t) / reps
ops = float(counts[:, 1].sum() + counts[:, 3].sum())
out = {"poses": poses, "n_dir": n_dir, "value_ms": t_val * 1e3, "dual_ms": t_dual * 1e3,
       "dual_over_value": t_dual / t_val, "pairs_plus_points": ops}
if "--cpu" in sys.argv:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_ins, oracle_meshes
    from oracle import oracle as O
    om = oracle_meshes(w); c = w.instructions[0]; ins = oracle_ins(pfc, c)
    k_n = min(8, n)
    t = time.perf_counter()
    for k in range(k_n):
        O.evaluate(om[c.id_1], om[c.id_2], ins, w.pose[k], w.twist[k], w.s[k], debug=False)
    t_cv = (time.perf_counter() - t) / k_n
    t = time.perf_counter()
    for k in range(k_n):
        O.evaluate_dual(om[c.id_1], om[c.id_2], ins, w.pose[k], w.twist[k], w.s[k], d_pose[k], d_twist[k], d_s[k])
    t_cd = (time.perf_counter() - t) / k_n
    out.update({"cpu_value_ms_per_pose": t_cv * 1e3, "cpu_dual_ms_per_pose": t_cd * 1e3,
                "gpu_dual_ms_per_pose": t_dual * 1e3 / poses})
print(json.dumps(out))
