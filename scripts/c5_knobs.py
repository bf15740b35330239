"""C5 (2 016 bristle instructions) device-resident: value and Dual(6) latency against the seed-level and narrowphase-form
knobs.  usage: python scripts/c5_knobs.py"""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pfc_pkg
pfc = pfc_pkg.load()
w = pfc.configs.c5_pile()
dev = torch.device("cuda", 0)
T = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
ni, nd = w.n_items, 6
t = [T(w.ins_ids, torch.int32), T(w.pose), T(w.twist), T(w.s), T(np.random.default_rng(7).standard_normal((ni, nd, 24)) * 1e-3), T(np.random.default_rng(8).standard_normal((ni, nd, 6)) * 1e-2), T(np.random.default_rng(9).standard_normal((ni, nd, 6)) * 1e-4)]      # dense seeds (zero keys are skipped)
o = [torch.zeros((ni, 6), dtype=torch.float64, device=dev), torch.zeros((ni, 6), dtype=torch.float64, device=dev),
     torch.zeros((ni, nd, 6), dtype=torch.float64, device=dev), torch.zeros((ni, nd, 6), dtype=torch.float64, device=dev),
     torch.zeros((ni, 4), dtype=torch.int32, device=dev)]
st = torch.cuda.current_stream().cuda_stream


def med(f, blocks=6, per=20):
    ts = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        for _ in range(per):
            f()
        ts.append((time.perf_counter() - t0) / per)
    return float(np.median(ts)) * 1e6


for bfs, clip in itertools.product((-1, 0, 1, 2), (1024, 0)):
    m = pfc.configs.build_scenario(w)
    m.set_option("bfs_levels", bfs)
    m.set_option("clip_min", clip)

    def val():
        m.eval_device(ni, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[4].data_ptr(), st)
        m.check()

    def dual():
        for _ in range(40):
            m.eval_dual_device(ni, nd, *[x.data_ptr() for x in t], *[x.data_ptr() for x in o], st)
            if m.check() == 0:
                return
    for _ in range(5):
        val(); dual()
    print(f"bfs_levels {bfs:2d} clip_min {clip:4d}: value {med(val):7.1f} us   Dual(6) {med(dual):7.1f} us", flush=True)
    m.close()
