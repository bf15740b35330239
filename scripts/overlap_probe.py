"""Experiment: one 2048-pose C3 batch as two 1024-pose halves on two handles / two streams (the broadphase of one half
can share the CUs with the narrowphase of the other), against the single-handle evaluation."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, pfc_pkg
pfc = pfc_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
w = pfc.configs.c3_blob_tool(n, seed=20260103)
dev = torch.device("cuda:0")
def T(a, dt): return torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
pose, twist, s, ids = T(w.pose, torch.float64), T(w.twist, torch.float64), T(w.s, torch.float64), T(w.ins_ids, torch.int32)
wr = torch.zeros((n, 6), dtype=torch.float64, device=dev); sd = torch.zeros_like(wr)
ct = torch.zeros((n, 4), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
def run(hs, reps):
    k = len(hs); m = n // k
    t0 = time.perf_counter()
    for _ in range(reps):
        for attempt in range(40):
            for j, h in enumerate(hs):
                a, b = j * m, (j + 1) * m
                h.eval_device(m, ids[a:b].data_ptr(), pose[a:b].data_ptr(), twist[a:b].data_ptr(), s[a:b].data_ptr(),
                              wr[a:b].data_ptr(), sd[a:b].data_ptr(), ct[a:b].data_ptr())
            if all([h.check() == 0 for h in hs]):   # overflow: capacities were grown, re-issue
                break
    return (time.perf_counter() - t0) / reps
one = [pfc.configs.build_scenario(w)]
for h in one: h.set_option("profile", 0)
run(one, 3); t1 = run(one, 20)
ref = wr.clone()
many = [pfc.configs.build_scenario(w) for _ in range(parts)]
run(many, 3); t2 = run(many, 20)
print(f"single handle {t1*1e3:.3f} ms/step; {parts} handles x {n//parts} poses {t2*1e3:.3f} ms/step; max |dw| {float((wr-ref).abs().max()):.3e}")
