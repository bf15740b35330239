"""Soak run of the host paths: one long-lived handle, hundreds of evaluations of random sizes (1 .. 700 items), value
and Dual mixed at random -- zero-copy and staged paths, one-graph and two-stage Dual paths, graph re-captures, list
growth, the fused small-scene kernel with its polled completion and its in-kernel Dual passes -- each compared with a fresh
handle that only ever sees that one call and always takes the batched path (option fused = 0).  A third of the Dual
evaluations repeat the point of the previous Dual evaluation with other seeds and, sometimes, another number of
directions (the chunks of a Jacobian: the value pass is reused on every path that can).  A third of the Dual evaluations
go through the device-resident entry points on the same handle (pfc_eval_dual_device + pfc_check with its re-issue
protocol, and pfc_eval_dual_device_more for a repeated point), interleaved with the host-buffer calls.
usage: python scripts/soak.py [n_evals] [big | reg]      reg: all-regularized box-on-plane scenes (small-scene Dual passes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pfc_pkg
pfc = pfc_pkg.load()
dev = torch.device("cuda", 0)
T = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
n_evals = int(sys.argv[1]) if len(sys.argv) > 1 else 300
big = len(sys.argv) > 2 and sys.argv[2] == "big"      # batches around the two-halves threshold (value evaluations split)
rng = np.random.default_rng(2026)
reg = len(sys.argv) > 2 and sys.argv[2] == "reg"
w = (pfc.configs.c2_box_on_plane(700, montecarlo=True, n_div=2) if reg else
     pfc.configs.c3_blob_tool(3000 if big else 700, seed=9, n_div_blob=6, n_div_tool=4))
w.s[:] = rng.standard_normal((w.n_items, 6)) * 1e-3
nd = 6
d_pose = rng.standard_normal((w.n_items, nd, 24)) * 1e-2
d_twist = rng.standard_normal((w.n_items, nd, 6)) * 0.1
d_s = rng.standard_normal((w.n_items, nd, 6)) * 1e-3
m = pfc.configs.build_scenario(w)
if os.environ.get("PFC_SOAK_POISON"):      # the long-lived handle starts every evaluation from poisoned work lists
    m.set_option("poison", 1)
bad = 0
kinds = {"value": 0, "dual": 0, "dual, value pass reused": 0, "dual, device-resident": 0, "re-issues": 0}
last_dual_sl = None
last_dual_dev = False      # the previous Dual evaluation went through pfc_eval_dual_device (and was checked)


def dual_device(sl, sd, more):
    """the Dual evaluation of items sl through the device entry points; more: only further directions"""
    n, k = sl.stop - sl.start, sd[0].shape[1]
    t = [T(w.ins_ids[sl], torch.int32), T(w.pose[sl]), T(w.twist[sl]), T(w.s[sl]), T(sd[0]), T(sd[1]), T(sd[2])]
    o = [torch.zeros((n, 6), dtype=torch.float64, device=dev), torch.zeros((n, 6), dtype=torch.float64, device=dev),
         torch.zeros((n, k, 6), dtype=torch.float64, device=dev), torch.zeros((n, k, 6), dtype=torch.float64, device=dev),
         torch.zeros((n, 4), dtype=torch.int32, device=dev)]
    st = torch.cuda.current_stream().cuda_stream
    if more:
        m.eval_dual_device_more(k, t[4].data_ptr(), t[5].data_ptr(), t[6].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st)
        assert m.check() == 0
        return None, None, o[2].cpu().numpy(), o[3].cpu().numpy(), None
    for attempt in range(40):
        m.eval_dual_device(n, k, *[x.data_ptr() for x in t], *[x.data_ptr() for x in o], st)
        if m.check() == 0:
            break
        kinds["re-issues"] += 1
    else:
        raise RuntimeError("no success in 40 issues")
    return tuple(x.cpu().numpy() for x in o)
for it in range(n_evals):
    size = int(rng.choice([5, 600, 1023, 1024, 1025, 1500, 2048, 3000] if big else [1, 2, 5, 17, 64, 85, 86, 200, 511, 512, 513, 700]))
    lo = int(rng.integers(0, w.n_items - size + 1))
    sl = slice(lo, lo + size)
    dual = rng.random() < 0.5
    ndk = nd
    if dual and last_dual_sl is not None and rng.random() < 0.5:      # the next chunk of the same Jacobian
        sl = last_dual_sl
        size = sl.stop - sl.start
        ndk = int(rng.choice([nd, nd, 3]))
        d_pose[sl] = rng.standard_normal((size, nd, 24)) * 1e-2
        d_twist[sl] = rng.standard_normal((size, nd, 6)) * 0.1
    same_point = dual and last_dual_sl is not None and sl is last_dual_sl
    last_dual_sl = sl if dual else None
    if it % 50 == 0:
        print(f"eval {it} ...", flush=True)
    f = pfc.configs.build_scenario(w)
    f.set_option("fused", 0)
    if dual:
        sd = (np.ascontiguousarray(d_pose[sl][:, :ndk]), np.ascontiguousarray(d_twist[sl][:, :ndk]), np.ascontiguousarray(d_s[sl][:, :ndk]))
        on_device = rng.random() < 1.0 / 3.0
        more = on_device and same_point and last_dual_dev
        if on_device:
            got = dual_device(sl, sd, more)
            kinds["dual, device-resident"] += 1
        else:
            got = m.force_all_elastic_intersections_dual(w.pose[sl], w.twist[sl], w.s[sl], *sd, w.ins_ids[sl])
        last_dual_dev = on_device
        if m.last_dual_reused():
            kinds["dual, value pass reused"] += 1
        f.set_option("dual_reuse", 0)
        ref = f.force_all_elastic_intersections_dual(w.pose[sl], w.twist[sl], w.s[sl], *sd, w.ins_ids[sl])
        if more:
            pairs = ((got[2], ref[2], 1e-9), (got[3], ref[3], 1e-6))
            cg = cr = ref[4]
        else:
            pairs = ((got[0], ref[0], 1e-10), (got[1], ref[1], 1e-6), (got[2], ref[2], 1e-9), (got[3], ref[3], 1e-6))
            cg, cr = got[4], ref[4]
    else:
        got = m.force_all_elastic_intersections(w.pose[sl], w.twist[sl], w.s[sl], w.ins_ids[sl])
        ref = f.force_all_elastic_intersections(w.pose[sl], w.twist[sl], w.s[sl], w.ins_ids[sl])
        pairs = ((got[0], ref[0], 1e-10), (got[1], ref[1], 1e-6))
        cg, cr = got[2], ref[2]
        last_dual_dev = False
    f.close()
    kinds["dual" if dual else "value"] += 1
    ok = np.array_equal(cg, cr)
    for a, b, tol in pairs:
        ok = ok and np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-300)
    if not ok:
        bad += 1
        print("MISMATCH at evaluation", it, "size", size, "dual" if dual else "value", flush=True)
m.close()
print(f"{n_evals} evaluations ({kinds}), mismatches: {bad}")
sys.exit(1 if bad else 0)
