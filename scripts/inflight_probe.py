"""Experiment: two independent 8 192-pose batches (two handles, two caller streams) kept in flight, against one batch at a
time.  usage: inflight_probe.py [poses] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pfc_pkg
pfc = pfc_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
OWN = len(sys.argv) > 3 and sys.argv[3] == "own"     # "own": each handle runs on its own internal stream (stream argument 0)
dev = torch.device("cuda:0")
w = pfc.configs.c3_blob_tool(n)
H = []
for k in range(2):
    m = pfc.configs.build_scenario(w)
    st = torch.cuda.Stream()
    bufs = dict(ins=torch.from_numpy(w.ins_ids.astype(np.int32)).to(dev), pose=torch.from_numpy(w.pose).to(dev),
                twist=torch.from_numpy(w.twist).to(dev), s=torch.from_numpy(w.s).to(dev),
                wrench=torch.zeros((n, 6), dtype=torch.float64, device=dev), sdot=torch.zeros((n, 6), dtype=torch.float64, device=dev),
                counts=torch.zeros((n, 4), dtype=torch.int32, device=dev))
    H.append((m, st, bufs))

def enqueue(h):
    m, st, b = h
    m.eval_device(n, b["ins"].data_ptr(), b["pose"].data_ptr(), b["twist"].data_ptr(), b["s"].data_ptr(),
                  b["wrench"].data_ptr(), b["sdot"].data_ptr(), b["counts"].data_ptr(), 0 if OWN else st.cuda_stream)

def check(h):
    rc = h[0].check()
    if rc != 0:
        enqueue(h); rc = h[0].check()
    assert rc == 0, rc

for h in H:
    for _ in range(4):
        enqueue(h); check(h)
torch.cuda.synchronize()
# one at a time
t0 = time.perf_counter()
for i in range(K):
    enqueue(H[0]); check(H[0])
torch.cuda.synchronize()
serial = (time.perf_counter() - t0) / K * 1e3
# two in flight
enqueue(H[0]); enqueue(H[1])
torch.cuda.synchronize()
t0 = time.perf_counter()
enqueue(H[0]); enqueue(H[1])
for i in range(K - 2):
    check(H[i & 1]); enqueue(H[i & 1])
check(H[K & 1]); check(H[(K + 1) & 1])
torch.cuda.synchronize()
pipe = (time.perf_counter() - t0) / K * 1e3
print("own streams" if OWN else "caller streams", end=": ")
print("poses %d: one batch at a time %.3f ms per step, two in flight %.3f ms per step" % (n, serial, pipe))
a = H[0][2]["wrench"].cpu().numpy(); b = H[1][2]["wrench"].cpu().numpy()
print("max |wrench_A - wrench_B| = %.3e" % float(np.max(np.abs(a - b))))
