"""Experiment: a batch cut into k chunks evaluated on s concurrent streams (one handle per stream, chunk j on handle
j % s): does more than two-way concurrency pay with the round-2 kernel set?  usage: overlap_probe3.py [poses]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, pfc_pkg
pfc = pfc_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
w = pfc.configs.c3_blob_tool(n, seed=20260103)
dev = torch.device("cuda:0")
def T(a, dt): return torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
pose, twist, s, ids = T(w.pose, torch.float64), T(w.twist, torch.float64), T(w.s, torch.float64), T(w.ins_ids, torch.int32)
wr = torch.zeros((n, 6), dtype=torch.float64, device=dev); sd = torch.zeros_like(wr)
ct = torch.zeros((n, 4), dtype=torch.int32, device=dev)
hs = [pfc.configs.build_scenario(w) for _ in range(4)]
for h in hs: h.set_option("split_min", 0)
def run(streams, chunks, reps):
    bounds = [round(j * n / chunks) for j in range(chunks + 1)]
    t0 = time.perf_counter()
    for _ in range(reps):
        for attempt in range(40):
            for j in range(chunks):
                a, b = bounds[j], bounds[j + 1]
                hs[j % streams].eval_device(b - a, ids[a:b].data_ptr(), pose[a:b].data_ptr(), twist[a:b].data_ptr(), s[a:b].data_ptr(),
                                            wr[a:b].data_ptr(), sd[a:b].data_ptr(), ct[a:b].data_ptr())
            if all([h.check() == 0 for h in hs[:streams]]):
                break
    return (time.perf_counter() - t0) / reps
for streams, chunks in ((1, 1), (2, 2), (3, 3), (4, 4), (2, 4), (3, 6), (4, 8)):
    run(streams, chunks, 4)
    print(f"{chunks} chunks on {streams} streams: {run(streams, chunks, 12)*1e3:.3f} ms/step", flush=True)
