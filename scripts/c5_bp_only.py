import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pfc_pkg
pfc = pfc_pkg.load()
w = pfc.configs.c5_pile()
dev = torch.device("cuda:0")
T = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
m = pfc.configs.build_scenario(w)
m.set_option("split_min", 0)
n = w.n_items
b = [T(w.ins_ids, torch.int32), T(w.pose, torch.float64), T(w.twist, torch.float64), T(w.s, torch.float64),
     torch.zeros((n, 6), dtype=torch.float64, device=dev), torch.zeros((n, 6), dtype=torch.float64, device=dev), torch.zeros((n, 4), dtype=torch.int32, device=dev)]
def run():
    m.eval_device(n, *[x.data_ptr() for x in b]); return m.check()
for _ in range(5): run()
def timed(K=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): run()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e6
print("C5 unsplit full evaluation: %.1f us" % timed())
for L in (1, 2):
    m.set_option("bfs_levels", L)
    m.set_option("phase", 0); run(); run()
    full = timed()
    m.set_option("phase", 1); run(); run()
    print("bfs_levels %d: full %.1f us, setup + broadphase only, back to back: %.1f us" % (L, full, timed()))
