"""Experiment: what would the C5 pile gain if the broadphase took its heavy items first?  The caller permutes the items
(descending node tests of a previous evaluation; ascending; random) -- the library is unchanged -- and the device-resident
evaluation is timed per order, with the broadphase stage time beside it.
usage: python scripts/c5_order_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pfc_pkg
pfc = pfc_pkg.load()
w = pfc.configs.c5_pile()
m = pfc.configs.build_scenario(w)
dev = torch.device("cuda:0")
wr, sd, cn = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
cost = cn[:, 0].astype(np.int64)
print("items %d, in contact %d, node tests: total %d, max %d, items with > 1 test %d" % (len(cost), int((cn[:, 3] > 0).sum()), cost.sum(), cost.max(), int((cost > 1).sum())))
rng = np.random.default_rng(0)
orders = {"as given": np.arange(len(cost)), "heavy first": np.argsort(-cost, kind="stable"), "heavy last": np.argsort(cost, kind="stable"),
          "random": rng.permutation(len(cost))}
def T(a, dt): return torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
for rep in range(2):
    for name, p in orders.items():
        n = len(p)
        b = [T(w.ins_ids[p], torch.int32), T(w.pose[p], torch.float64), T(w.twist[p], torch.float64), T(w.s[p], torch.float64),
             torch.zeros((n, 6), dtype=torch.float64, device=dev), torch.zeros((n, 6), dtype=torch.float64, device=dev),
             torch.zeros((n, 4), dtype=torch.int32, device=dev)]
        args = [x.data_ptr() for x in b]
        for _ in range(8):      # (the first calls may grow the work lists: re-issue)
            m.eval_device(n, *args); m.check()
        m.eval_device(n, *args); assert m.check() == 0
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            for _ in range(50):
                m.eval_device(n, *args); assert m.check() == 0
            ts.append((time.perf_counter() - t0) / 50)
        assert np.array_equal(b[6].cpu().numpy(), cn[p])
        m.set_option("profile", 1)
        for _ in range(3):
            m.eval_device(n, *args); assert m.check() == 0
        st = m.stage_ms() if hasattr(m, "stage_ms") else None
        m.set_option("profile", 0)
        print("%-12s %.1f us per evaluation   stages %s" % (name, np.median(ts) * 1e6, st), flush=True)
m.close()
