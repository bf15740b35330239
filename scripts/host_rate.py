"""PCIe-inclusive rate of the host-buffer entry point (pfc_eval) on the bench workload: what a caller that hands over
host arrays every step gets, next to bench.py's device-resident figure.  usage: python scripts/host_rate.py [poses]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
w = pfc.configs.c3_blob_tool(n, seed=20260103)
m = pfc.configs.build_scenario(w)
for _ in range(4):
    m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
reps = 10
t0 = time.perf_counter()
for _ in range(reps):
    m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
dt = (time.perf_counter() - t0) / reps
st = m.stats()
print(f"{n} poses through host buffers: {dt * 1e3:.3f} ms per step, {st['candidates'] / dt:.3e} ops/s "
      f"({n * 288 / 1e6:.2f} MB up, {n * 112 / 1e6:.2f} MB down per step)")
m.close()
