"""Experiment (variant library build/variants/phase.so, scripts/variants/phase_probe.py): what would the 8 192-pose step cost if
the narrowphase + bristle passes of PARTS of the batch ran beside ONE broadphase launch over the whole batch (a device-side
hand-over per part instead of the two-half scheme)?  The probe runs the broadphase of the whole batch (handle A, phase 1) on
one stream and the narrowphase + bristle passes of k parts (handles B, C: phase 2 on candidate lists of earlier normal
evaluations) on one or two other streams, all enqueued at once -- the upper bound of what such a hand-over could gain
(nothing waits for its producer).
usage: PFC_LIB=.../build/variants/phase.so PFC_ALLOW_DIAGNOSTIC=1 python scripts/dataflow_probe.py [poses] [parts]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, pfc_pkg
pfc = pfc_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K = 10
dev = torch.device("cuda:0")
def T(a, dt): return torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)

class Job:
    def __init__(self, poses, seed, split):
        self.n = poses
        self.w = w = pfc.configs.c3_blob_tool(poses, seed=seed)
        self.m = pfc.configs.build_scenario(w)
        if not split: self.m.set_option("split_min", 0)
        self.b = [T(w.ins_ids, torch.int32), T(w.pose, torch.float64), T(w.twist, torch.float64), T(w.s, torch.float64),
                  torch.zeros((poses, 6), dtype=torch.float64, device=dev), torch.zeros((poses, 6), dtype=torch.float64, device=dev),
                  torch.zeros((poses, 4), dtype=torch.int32, device=dev)]
    def enqueue(self):
        self.m.eval_device(self.n, *[x.data_ptr() for x in self.b])
    def check(self):
        return self.m.check()
    def settle(self):
        for _ in range(6):
            self.enqueue()
            if self.check() == 0: return
        raise RuntimeError("evaluation does not settle")

ref = Job(n, 20260103, True)            # the product scheme (two halves)
A = Job(n, 20260103, False)             # whole batch, unsplit
x = torch.cuda.Stream()                 # shifts the hardware queue of the next handles' streams
B = Job(n // parts, 20260104, False)
C = Job(n // parts, 20260105, False)
for j in (ref, A, B, C):
    j.settle(); j.settle()
torch.cuda.synchronize()

def timed(fn, reps=K):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def step_ref():
    ref.enqueue(); assert ref.check() == 0
print("poses %d, parts %d" % (n, parts))
print("product, two halves:                         %.3f ms" % timed(step_ref), flush=True)
def step_unsplit():
    A.enqueue(); assert A.check() == 0
print("one launch sequence (unsplit):               %.3f ms" % timed(step_unsplit), flush=True)
A.m.set_option("phase", 1); B.m.set_option("phase", 2); C.m.set_option("phase", 2)
def bp_only():
    A.enqueue(); A.check()
print("broadphase of the whole batch alone:         %.3f ms" % timed(bp_only), flush=True)
def np_one():
    for k in range(parts):
        B.enqueue()
    B.check()
print("narrowphase + bristle of %d parts, 1 stream:  %.3f ms" % (parts, timed(np_one)), flush=True)
def np_two():
    for k in range(parts // 2):
        B.enqueue(); C.enqueue()
    B.check(); C.check()
print("narrowphase + bristle of %d parts, 2 streams: %.3f ms" % (parts, timed(np_two)), flush=True)
def both_one():
    A.enqueue()
    for k in range(parts):
        B.enqueue()
    A.check(); B.check()
print("broadphase || parts on 1 stream:             %.3f ms" % timed(both_one), flush=True)
def both_two():
    A.enqueue()
    for k in range(parts // 2):
        B.enqueue(); C.enqueue()
    A.check(); B.check(); C.check()
print("broadphase || parts on 2 streams:            %.3f ms" % timed(both_two), flush=True)
def both_two_late():
    # the parts' passes enqueued first would be unrealistic; this order lets the broadphase take the chip first
    A.enqueue()
    for k in range(parts // 2):
        C.enqueue(); B.enqueue()
    A.check(); B.check(); C.check()
print("same, other stream first:                    %.3f ms" % timed(both_two_late), flush=True)
