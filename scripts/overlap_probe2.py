"""Experiment: software pipelining of a batch over two handles/streams: k chunks issued alternately (chunk j on handle
j % 2), so that at any time two chunks are in flight and naturally out of phase."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, pfc_pkg
pfc = pfc_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
w = pfc.configs.c3_blob_tool(n, seed=20260103)
dev = torch.device("cuda:0")
def T(a, dt): return torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
pose, twist, s, ids = T(w.pose, torch.float64), T(w.twist, torch.float64), T(w.s, torch.float64), T(w.ins_ids, torch.int32)
wr = torch.zeros((n, 6), dtype=torch.float64, device=dev); sd = torch.zeros_like(wr)
ct = torch.zeros((n, 4), dtype=torch.int32, device=dev)
hs = [pfc.configs.build_scenario(w) for _ in range(2)]
for h in hs: h.set_option("split_min", 0)
def run(chunks, reps):
    bounds = [round(j * n / chunks) for j in range(chunks + 1)]
    t0 = time.perf_counter()
    for _ in range(reps):
        for attempt in range(40):
            for j in range(chunks):
                a, b = bounds[j], bounds[j + 1]
                hs[j % 2].eval_device(b - a, ids[a:b].data_ptr(), pose[a:b].data_ptr(), twist[a:b].data_ptr(), s[a:b].data_ptr(),
                                      wr[a:b].data_ptr(), sd[a:b].data_ptr(), ct[a:b].data_ptr())
            if all([h.check() == 0 for h in hs]):
                break
    return (time.perf_counter() - t0) / reps
for chunks in (2, 3, 4, 6, 8):
    run(chunks, 4)
    print(f"{chunks} chunks on 2 streams: {run(chunks, 20)*1e3:.3f} ms/step")
