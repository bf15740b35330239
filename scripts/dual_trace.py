"""Diagnostic: N device-resident Dual evaluations (pfc_eval_dual_device) of one config, for a rocprofv3 kernel trace.
usage: dual_trace.py <c1|c2|c4|c5|c3|c3b> [n_evals] [n_dir] [more]      (more: that many further chunks (pfc_eval_dual_device_more) after each evaluation)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pfc_pkg
pfc = pfc_pkg.load()
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
nd = int(sys.argv[3]) if len(sys.argv) > 3 else 6
more = int(sys.argv[4]) if len(sys.argv) > 4 else 0
w = {"c1": pfc.configs.c1_boxes, "c2": lambda: pfc.configs.c2_box_on_plane(1), "c3": lambda: pfc.configs.c3_blob_tool(1),
     "c4": lambda: pfc.configs.c2_box_on_plane(256, montecarlo=True), "c5": pfc.configs.c5_pile,
     "c3b": lambda: pfc.configs.c3_blob_tool(2048)}[cfg]()
m = pfc.configs.build_scenario(w)
dev = torch.device("cuda", 0)
T = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
ni = w.n_items
t = [T(w.ins_ids, torch.int32), T(w.pose), T(w.twist), T(w.s), T(np.random.default_rng(7).standard_normal((ni, nd, 24)) * 1e-3), T(np.random.default_rng(8).standard_normal((ni, nd, 6)) * 1e-2), T(np.random.default_rng(9).standard_normal((ni, nd, 6)) * 1e-4)]      # dense seeds (zero keys are skipped)
o = [torch.zeros((ni, 6), dtype=torch.float64, device=dev), torch.zeros((ni, 6), dtype=torch.float64, device=dev),
     torch.zeros((ni, nd, 6), dtype=torch.float64, device=dev), torch.zeros((ni, nd, 6), dtype=torch.float64, device=dev),
     torch.zeros((ni, 4), dtype=torch.int32, device=dev)]
st = torch.cuda.current_stream().cuda_stream
def run():
    for _ in range(40):
        m.eval_dual_device(ni, nd, *[x.data_ptr() for x in t], *[x.data_ptr() for x in o], st)
        if m.check() == 0:
            break
    for _ in range(more):
        m.eval_dual_device_more(nd, t[4].data_ptr(), t[5].data_ptr(), t[6].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st)
        assert m.check() == 0
for _ in range(5):
    run()
t0 = time.perf_counter()
for _ in range(n):
    run()
print(f"{cfg}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per device-resident Dual({nd}) evaluation + {more} further chunks; stats {m.stats()}")
m.close()
