"""Does an initialised RCCL communicator change what the 8 192-pose step costs?  (The two halves of a step only overlap if
their streams sit on different hardware queues; the runtime deals its streams out to a few queues in order of creation.)
usage: python scripts/rccl_queue_probe.py <mode>     modes: none | before | after | before+xchg | after+xchg; "nocoll" in the mode: no collective inside the
initialisation; "lazy": init_process_group without device_id"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
def init():
    if "lazy" in mode: dist.init_process_group("nccl", rank=0, world_size=1)      # communicator created at the first collective
    else: dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)    # eager (what bench.py does)
    if "nocoll" not in mode:
        t = torch.zeros(8, device=dev); g = torch.zeros(8, device=dev); dist.all_gather_into_tensor(g, t); torch.cuda.synchronize()
if mode.startswith("before"): init()
import pfc_pkg
pfc = pfc_pkg.load()
w = pfc.configs.c3_blob_tool(n)
m = pfc.configs.build_scenario(w)
if mode.startswith("after"): init()
T = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
b = [T(w.ins_ids, torch.int32), T(w.pose, torch.float64), T(w.twist, torch.float64), T(w.s, torch.float64),
     torch.zeros((n, 6), dtype=torch.float64, device=dev), torch.zeros((n, 6), dtype=torch.float64, device=dev), torch.zeros((n, 4), dtype=torch.int32, device=dev)]
d_out = torch.zeros((n, 12), dtype=torch.float64, device=dev); gathered = torch.zeros((n, 12), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
def step():
    for _ in range(40):
        m.eval_device(n, *[x.data_ptr() for x in b], st)
        if m.check() == 0: break
    if mode.endswith("xchg"):
        d_out[:, :6] = b[4]; d_out[:, 6:] = b[5]
        dist.all_gather_into_tensor(gathered, d_out)
for _ in range(4): step()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10 * 1e3)
import ctypes as C
out = (C.c_longlong * 16)(); pfc._lib.lib().pfc_debug_stamps(m._h, out)
print("mode %-22s poses %d: %.3f ms per step (parts %d; stream pair: %s)" % (mode, n, float(np.median(ts)), m.last_parts(),
      {0: "side by side as created", 1: "twin re-created with another priority", 2: "serial"}[int(out[6])]), flush=True)
if dist.is_initialized(): dist.destroy_process_group()
