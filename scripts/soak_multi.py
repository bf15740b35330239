"""Soak run of a multi-device handle (pfc_create_multi over PFC_SOAK_DEVICES, default "0,0"): one long-lived handle, a few hundred
evaluations of random sizes -- value and Dual (with and without a broadphase pose, repeated points = chunks of a Jacobian), host
buffers and device buffers at random -- each compared with a fresh single-device handle that only ever sees that one call.
usage: python scripts/soak_multi.py [n_evals] [c3r | pile | c2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pfc_pkg
pfc = pfc_pkg.load()
C = pfc.configs
dev = torch.device("cuda", 0)
T = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
n_evals = int(sys.argv[1]) if len(sys.argv) > 1 else 200
kind = sys.argv[2] if len(sys.argv) > 2 else "c3r"
devs = [int(x) for x in os.environ.get("PFC_SOAK_DEVICES", "0,0").split(",")]
rng = np.random.default_rng(404)
w = {"c3r": lambda: C.c3_blob_tool(1500, seed=9, n_div_blob=6, n_div_tool=4), "pile": lambda: C.c5_pile(),
     "c2": lambda: C.c2_box_on_plane(1500, montecarlo=True, n_div=3)}[kind]()
w.s[:] = rng.standard_normal((w.n_items, 6)) * 1e-3
nd = 6
d_pose = rng.standard_normal((w.n_items, nd, 24)) * 1e-2
d_twist = rng.standard_normal((w.n_items, nd, 6)) * 0.1
d_s = rng.standard_normal((w.n_items, nd, 6)) * 1e-3
bp_all = w.pose.copy()
bp_all[:, 21:24] += rng.standard_normal((w.n_items, 3)) * 2e-3      # another x_r1_r2 translation for the broadphase
m = C.build_scenario(w, devices=devs)
bad = 0
kinds = {"value host": 0, "value device": 0, "dual host": 0, "dual host, bp pose": 0, "dual device": 0, "dual reused": 0, "re-issues": 0, "shards>1": 0}
sizes = [1, 5, 17, 64, 200, 511, 513, 700, 1025, 1500] if kind != "pile" else [8, 63, 500, 1024, 2016]
last = None


def close(a, b, tol):
    a, b = np.asarray(a), np.asarray(b)
    s = np.abs(b).max(axis=tuple(range(1, b.ndim)), keepdims=True) + 1e-300
    return bool((np.abs(a - b) <= tol * s).mean() > 0.97)      # (flat patches: a few items' sdot / partials are rounding noise)


for it in range(n_evals):
    size = int(rng.choice(sizes)); size = min(size, w.n_items)
    lo = int(rng.integers(0, w.n_items - size + 1))
    sl = slice(lo, lo + size)
    mode = rng.choice(["vh", "vd", "dh", "dhb", "dd"])
    repeat = last is not None and rng.random() < 0.3 and mode in ("dh", "dhb") and last[0] == mode
    if repeat:
        sl = last[1]; size = sl.stop - sl.start
    ids = w.ins_ids[sl]; po, tw, ss = w.pose[sl], w.twist[sl], w.s[sl]
    sc = 1.0 + 0.1 * it
    sd = (d_pose[sl] * sc, d_twist[sl], d_s[sl])
    ref = C.build_scenario(w)
    # option fixed_order on the long-lived handle (every shard), switched every ~8 evaluations (any pfc_set_option drops a kept
    # value pass): results as before up to the order of the sums
    if it % 8 == 0:
        fixed = bool(rng.random() < 0.4)
        m.set_option("fixed_order", int(fixed))
    kinds["fixed_order"] = kinds.get("fixed_order", 0) + int(fixed)
    if mode == "vh":
        got = m.force_all_elastic_intersections(po, tw, ss, ids); want = ref.force_all_elastic_intersections(po, tw, ss, ids)
        ok = np.array_equal(got[2], want[2]) and close(got[0], want[0], 1e-8)
        kinds["value host"] += 1
    elif mode == "vd":
        t = [T(ids, torch.int32), T(po), T(tw), T(ss)]
        o = [torch.zeros((size, 6), dtype=torch.float64, device=dev), torch.zeros((size, 6), dtype=torch.float64, device=dev), torch.zeros((size, 4), dtype=torch.int32, device=dev)]
        st = torch.cuda.current_stream().cuda_stream
        for attempt in range(40):
            m.eval_device(size, *[x.data_ptr() for x in t], *[x.data_ptr() for x in o], st)
            if m.check() == 0:
                break
            kinds["re-issues"] += 1
        want = ref.force_all_elastic_intersections(po, tw, ss, ids)
        ok = np.array_equal(o[2].cpu().numpy(), want[2]) and close(o[0].cpu().numpy(), want[0], 1e-8)
        kinds["value device"] += 1
    elif mode in ("dh", "dhb"):
        bp = bp_all[sl] if mode == "dhb" else None
        got = m.force_all_elastic_intersections_dual(po, tw, ss, *sd, ids, bp_pose=bp)
        if m.last_dual_reused():
            kinds["dual reused"] += 1
        want = ref.force_all_elastic_intersections_dual(po, tw, ss, *sd, ids, bp_pose=bp)
        ok = np.array_equal(got[4], want[4]) and close(got[0], want[0], 1e-8) and close(got[2], want[2], 1e-6)
        kinds["dual host" + (", bp pose" if bp is not None else "")] += 1
        last = (mode, sl)
    else:
        t = [T(ids, torch.int32), T(po), T(tw), T(ss), T(sd[0]), T(sd[1]), T(sd[2])]
        o = [torch.zeros((size, 6), dtype=torch.float64, device=dev), torch.zeros((size, 6), dtype=torch.float64, device=dev),
             torch.zeros((size, nd, 6), dtype=torch.float64, device=dev), torch.zeros((size, nd, 6), dtype=torch.float64, device=dev),
             torch.zeros((size, 4), dtype=torch.int32, device=dev)]
        st = torch.cuda.current_stream().cuda_stream
        for attempt in range(40):
            m.eval_dual_device(size, nd, *[x.data_ptr() for x in t], *[x.data_ptr() for x in o], st)
            if m.check() == 0:
                break
            kinds["re-issues"] += 1
        want = ref.force_all_elastic_intersections_dual(po, tw, ss, *sd, ids)
        ok = np.array_equal(o[4].cpu().numpy(), want[4]) and close(o[0].cpu().numpy(), want[0], 1e-8) and close(o[2].cpu().numpy(), want[2], 1e-6)
        kinds["dual device"] += 1
    if mode not in ("dh", "dhb"):
        last = None
    kinds["shards>1"] += m.last_shards() > 1
    ref.close()
    if not ok:
        bad += 1
        print("MISMATCH at evaluation", it, mode, "size", size, "lo", lo, flush=True)
print(f"soak_multi {kind} over devices {devs}: {n_evals} evaluations, {bad} mismatches; {kinds}")
m.close()
sys.exit(1 if bad else 0)
