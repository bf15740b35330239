"""Multi-device handles (pfc_create_multi, SURVEY 8(b) "pfc_create(device_mask)") and the broadphase pose of a Dual
evaluation (pfc_eval_dual_bp; calcTriTetIntersections!, src/contact_algorithms_non_friction.jl:94-101).

The GPU box of the test run has ONE device, so the device list is {0, 0}: two shard contexts -- two handles, two stream
sets, two host threads, staging buffers and copies of the device-pointer path -- on the one visible GPU; when more devices
are visible the same tests also run over all of them."""
import ctypes
import threading
import time

import numpy as np
import pytest

import helpers as H
from helpers import oracle_ins, oracle_meshes
from test_oracle_dual import tangents

pytestmark = pytest.mark.gpu


def _device_lists():
    import torch
    n = torch.cuda.device_count()
    out = [[0, 0]]
    if n >= 2:
        out.append(list(range(n)))
    return out


def _workload(pfc, cfg):
    C = pfc.configs
    if cfg == "c4":
        return C.c2_box_on_plane(256, montecarlo=True)
    if cfg == "c5":
        return C.c5_pile()
    if cfg == "c3x64":
        return C.c3_blob_tool(64)
    raise KeyError(cfg)


def _close(a, b, tol):
    for k in range(a.shape[0]):
        nb = np.linalg.norm(b[k])
        if nb == 0.0:
            assert np.linalg.norm(a[k]) == 0.0, k
        else:
            assert np.linalg.norm(a[k] - b[k]) <= tol * nb, (k, a[k], b[k])


@pytest.mark.parametrize("cfg", ["c4", "c5", "c3x64"])
def test_multi_handle_host_buffers(pfc, cfg):
    """pfc_eval on a handle over {0, 0} (and over every visible device): counters bit-equal and wrench 1e-9 against the
    single-device handle and against the oracle (C3: 8 sampled full-size items); the ranges are rebalanced from the measured
    costs and the results do not move."""
    w = _workload(pfc, cfg)
    m1 = pfc.configs.build_scenario(w)
    ref = m1.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    m1.close()
    for devs in _device_lists():
        m = pfc.configs.build_scenario(w, devices=devs)
        for rep in range(3):       # leaf-product costs, then measured costs (a new partition), then the settled partition
            wrench, sdot, counts = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
            assert m.last_shards() == len(devs)
            assert np.array_equal(counts, ref[2]), (cfg, devs, rep)
            # (the pile has grazing contacts whose wrench is a nearly cancelling sum: reproducible to the north_star tolerance between
            # two launches, tests/test_gpu_scale.py::test_c5_pile_all_pairs)
            _close(wrench, ref[0], 1e-6 if cfg == "c5" else 1e-9)
            st = m.stats()
            assert st["candidates"] == int(counts[:, 1].sum()) and st["node_tests"] == int(counts[:, 0].sum()) and st["n_items"] == w.n_items
        if cfg == "c3x64":
            items = [0, 7, 21, 31, 32, 40, 55, 63]
            for k, r in zip(items, H.oracle_run(pfc, w, items=items, debug=False)):
                assert np.array_equal(counts[k], r.counts), k
                assert H.rel_err(wrench[k], r.wrench) < 1e-9 and H.rel_err(sdot[k], r.sdot) < 1e-6, k
        else:
            from test_gpu_scale import _check_vs_oracle
            _check_vs_oracle(pfc, w, m, wrench, sdot, counts, tol=1e-6 if cfg == "c5" else 1e-9, oracle_debug=cfg == "c5")
        # fewer items than multi_min per device: one shard takes the evaluation
        m.set_option("multi_min", w.n_items)
        a = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        assert m.last_shards() == 1 and np.array_equal(a[2], ref[2])
        # a subset with ins_ids == None where allowed (item i = instruction i): ids are synthesised per range
        if cfg == "c5":
            m.set_option("multi_min", 8)
            b = m.force_all_elastic_intersections(w.pose[:500], w.twist[:500], w.s[:500], None)
            assert m.last_shards() == len(devs) and np.array_equal(b[2], ref[2][:500])
        m.close()


def test_multi_handle_debug_views_and_errors(pfc):
    """pfc_debug_* of a multi-device handle go to the shard that evaluated the item; argument errors come back as on the
    single-device handle."""
    w = pfc.configs.c2_box_on_plane(40, montecarlo=True)
    m = pfc.configs.build_scenario(w, devices=[0, 0], debug=True)
    wrench, sdot, counts = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert m.last_shards() == 2
    ref = H.oracle_run(pfc, w, items=[3, 37])
    for k, r in zip([3, 37], ref):
        gp, gc = H.sorted_pairs(*m.debug_pairs(k))
        rp, rc = H.sorted_pairs(r.pairs, r.clip_n)
        assert np.array_equal(gp, rp) and np.array_equal(gc, rc), k
        assert m.debug_tractions(k).shape[0] == r.counts[3]
    with pytest.raises(pfc._lib.PFCError) as e:
        m.force_all_elastic_intersections(w.pose, w.twist, w.s, np.full(40, 7, dtype=np.int32))      # bad instruction id
    assert e.value.status == pfc._lib.ERR_BAD_ARG
    bad = w.pose.copy(); bad[33, 2] = np.nan
    with pytest.raises(pfc._lib.PFCError) as e:
        m.force_all_elastic_intersections(bad, w.twist, w.s, w.ins_ids)
    assert e.value.status == pfc._lib.ERR_NONFINITE
    again = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert np.array_equal(again[2], counts)
    m.close()


@pytest.mark.parametrize("cfg", ["c4", "c5"])
def test_multi_handle_device_buffers(pfc, cfg):
    """pfc_eval_device + pfc_check on a multi-device handle: buffers on the first device, the other shards' ranges staged by
    (peer) copies, results back in the caller's arrays; d_ins_ids given and NULL; scatter on the first device."""
    import torch
    w = _workload(pfc, cfg)
    m1 = pfc.configs.build_scenario(w)
    ref = m1.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    m1.close()
    dev = torch.device("cuda:0")
    n = w.n_items
    t = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    d_ids, d_pose, d_twist, d_s = t(w.ins_ids, torch.int32), t(w.pose), t(w.twist), t(w.s)
    for devs in _device_lists():
        m = pfc.configs.build_scenario(w, devices=devs)
        for use_ids in (True, False) if cfg == "c5" else (True,):
            for rep in range(3):
                d_w = torch.full((n, 6), np.nan, dtype=torch.float64, device=dev); d_sd = torch.full_like(d_w, np.nan)
                d_c = torch.full((n, 4), -1, dtype=torch.int32, device=dev)
                stream = torch.cuda.current_stream().cuda_stream
                for attempt in range(40):
                    m.eval_device(n, d_ids.data_ptr() if use_ids else 0, d_pose.data_ptr(), d_twist.data_ptr(), d_s.data_ptr(),
                                  d_w.data_ptr(), d_sd.data_ptr(), d_c.data_ptr() if rep != 1 else 0, stream)
                    if m.check() == pfc._lib.OK:
                        break
                assert m.last_shards() == len(devs)
                if rep != 1:
                    assert np.array_equal(d_c.cpu().numpy(), ref[2]), (cfg, devs, use_ids, rep)
                _close(d_w.cpu().numpy(), ref[0], 1e-6 if cfg == "c5" else 1e-9)
                if cfg != "c5":      # (the pile's sdot of sliver patches: 1e-3 rule of tests/test_gpu_scale.py)
                    _close(d_sd.cpu().numpy(), ref[1], 1e-6)
                assert m.stats()["candidates"] == int(ref[2][:, 1].sum())
        m.close()


def _dual_inputs(w, n_dir, seed):
    rng = np.random.default_rng(seed)
    n = w.n_items
    dq = rng.standard_normal((n, n_dir, 6)) * np.array([1, 1, 1, 0.05, 0.05, 0.05])
    d_twist = rng.standard_normal((n, n_dir, 6)) * np.array([1, 1, 1, 0.1, 0.1, 0.1])
    d_s = rng.standard_normal((n, n_dir, 6)) * 1e-3
    d_pose = np.zeros((n, n_dir, 24))
    for k in range(n):
        R0 = w.pose[k][:9].reshape(3, 3, order="F"); t0 = w.pose[k][9:12]
        d_pose[k] = tangents(R0, t0, dq[k])
    return d_pose, d_twist, d_s


def _nearby_pose(w, seed, scale=2e-3):
    """A pose a little off w.pose -- what m.float holds when Radau evaluates its Jacobian at another point."""
    rng = np.random.default_rng(seed)
    S = None
    out = np.zeros_like(w.pose)
    for k in range(w.n_items):
        R21 = w.pose[k][:9].reshape(3, 3, order="F"); t21 = w.pose[k][9:12]
        a = rng.standard_normal(3) * scale
        th = np.linalg.norm(a); ax = a / th
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0.0]])
        dR = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
        R = dR @ R21; t = t21 + rng.standard_normal(3) * scale * 0.1
        R12 = R.T; t12 = -(R12 @ t)
        out[k] = np.concatenate([R.reshape(-1, order="F"), t, R12.reshape(-1, order="F"), t12])
    return out


@pytest.mark.parametrize("cfg", ["c1", "c2x24", "blob12", "blob600", "volvol", "pile"])
def test_dual_with_a_broadphase_pose_of_its_own(pfc, O, cfg):
    """pfc_eval_dual_bp: the candidates come from bp_pose (m.float's transforms, non_friction.jl:94-101), everything else from
    pose.  Against the Dual oracle with the same two poses, through every path a Dual evaluation can take: the all-in-one
    small-scene kernel (c1, c2), the hybrid path (blob12: bristle), the batched path (blob600, pile), tet-tet (k_repose puts
    x_r1_r2 back for the plane); counters with bp_pose differ from those without (the test would not notice a bp_pose that is
    ignored otherwise); then chunks at the same point, the same point with another bp_pose, and bp_pose = None again."""
    C = pfc.configs
    w = {"c1": lambda: C.c1_boxes(), "c2x24": lambda: C.c2_box_on_plane(24, montecarlo=True),
         "blob12": lambda: C.c3_blob_tool(12, seed=3, n_div_blob=6, n_div_tool=4),
         "blob600": lambda: C.c3_blob_tool(600, seed=4, n_div_blob=6, n_div_tool=4),
         "volvol": lambda: C.vol_vol(6, model="bristle"), "pile": lambda: C.c5_pile(n_side=3)}[cfg]()
    n_dir = 6
    d_pose, d_twist, d_s = _dual_inputs(w, n_dir, 11)
    bp = _nearby_pose(w, 5, scale=8e-3 if cfg in ("c1", "c2x24") else 2e-2)
    om = oracle_meshes(w)
    items = range(w.n_items) if w.n_items <= 64 else list(range(0, w.n_items, max(1, w.n_items // 40)))

    def oracle(bp_pose):
        out = {}
        for k in items:
            c = w.instructions[int(w.ins_ids[k])]
            ins = oracle_ins(pfc, c)
            v = O.evaluate(om[c.id_1], om[c.id_2], ins, w.pose[k], w.twist[k], w.s[k], debug=False, bp_pose=None if bp_pose is None else bp_pose[k])
            st, rw, rs, rdw, rdsd = O.evaluate_dual(om[c.id_1], om[c.id_2], ins, w.pose[k], w.twist[k], w.s[k], d_pose[k], d_twist[k], d_s[k],
                                                    bp_pose=None if bp_pose is None else bp_pose[k])
            assert st == 0 and v.status == 0
            out[k] = (v.counts, v.wrench, v.sdot, rdw, rdsd)
        return out

    def compare(got, want):
        wr, sd, dw, dsd, counts = got
        for k in items:
            rc, rw, rs, rdw, rdsd = want[k]
            assert np.array_equal(counts[k], rc), (cfg, k, counts[k], rc)
            if np.linalg.norm(rw) > 0:
                assert H.rel_err(wr[k], rw) < 1e-9, k
            sw = max(np.abs(rdw).max(), 1e-300); ss = max(np.abs(rdsd).max(), 1e-300)
            assert np.abs(dw[k] - rdw).max() <= 1e-6 * sw, (cfg, k)
            assert np.abs(dsd[k] - rdsd).max() <= 1e-5 * ss, (cfg, k)

    ref_bp, ref_plain = oracle(bp), oracle(None)
    assert any(not np.array_equal(ref_bp[k][0][:2], ref_plain[k][0][:2]) for k in items), "bp_pose does not change the candidate sets of this scene"
    for devs in (None, [0, 0]):
        if devs is not None and cfg not in ("blob600", "pile", "c2x24"):
            continue
        m = C.build_scenario(w, devices=devs)
        compare(m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, d_s, w.ins_ids, bp_pose=bp), ref_bp)
        # the next chunk of the same Jacobian (same values, same bp_pose): reuses the value pass where the path keeps one
        compare(m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, d_s, w.ins_ids, bp_pose=bp), ref_bp)
        # same values, NO bp_pose: the lists of the previous evaluation must not be reused
        compare(m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, d_s, w.ins_ids), ref_plain)
        assert not (devs is None and m.last_dual_reused())
        compare(m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, d_s, w.ins_ids, bp_pose=bp), ref_bp)
        # a value evaluation in between never sees the broadphase pose
        v = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        for k in items:
            assert np.array_equal(v[2][k], ref_plain[k][0]), k
        m.close()


def test_dual_device_entry_points_with_bp_pose_and_multi(pfc, O):
    """pfc_eval_dual_device_bp + pfc_check + pfc_eval_dual_device_more on a single-device and on a {0, 0} handle."""
    import torch
    C = pfc.configs
    w = C.c3_blob_tool(96, seed=9, n_div_blob=6, n_div_tool=4)
    n, n_dir = w.n_items, 6
    d_pose, d_twist, d_s = _dual_inputs(w, n_dir, 21)
    d_pose2, d_twist2, d_s2 = _dual_inputs(w, 3, 22)
    bp = _nearby_pose(w, 6, scale=2e-2)
    dev = torch.device("cuda:0")
    t = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    ref = C.build_scenario(w)
    want = ref.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, d_s, w.ins_ids, bp_pose=bp)
    want2 = ref.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose2, d_twist2, d_s2, w.ins_ids, bp_pose=bp)
    ref.close()
    T = dict(ids=t(w.ins_ids, torch.int32), pose=t(w.pose), bp=t(bp), twist=t(w.twist), s=t(w.s), dp=t(d_pose), dt=t(d_twist), ds=t(d_s),
             dp2=t(d_pose2), dt2=t(d_twist2), ds2=t(d_s2))
    for devs in (None, [0, 0]):
        m = C.build_scenario(w, devices=devs)
        o = dict(w=torch.zeros((n, 6), dtype=torch.float64, device=dev), sd=torch.zeros((n, 6), dtype=torch.float64, device=dev),
                 dw=torch.zeros((n, n_dir, 6), dtype=torch.float64, device=dev), dsd=torch.zeros((n, n_dir, 6), dtype=torch.float64, device=dev),
                 c=torch.zeros((n, 4), dtype=torch.int32, device=dev), dw2=torch.zeros((n, 3, 6), dtype=torch.float64, device=dev),
                 dsd2=torch.zeros((n, 3, 6), dtype=torch.float64, device=dev))
        st = torch.cuda.current_stream().cuda_stream
        for attempt in range(40):
            m.eval_dual_device(n, n_dir, T["ids"].data_ptr(), T["pose"].data_ptr(), T["twist"].data_ptr(), T["s"].data_ptr(), T["dp"].data_ptr(),
                               T["dt"].data_ptr(), T["ds"].data_ptr(), o["w"].data_ptr(), o["sd"].data_ptr(), o["dw"].data_ptr(), o["dsd"].data_ptr(),
                               o["c"].data_ptr(), st, d_bp_pose=T["bp"].data_ptr())
            if m.check() == pfc._lib.OK:
                break
        assert np.array_equal(o["c"].cpu().numpy(), want[4])
        _close(o["w"].cpu().numpy(), want[0], 1e-9)
        sw = np.abs(want[2]).max(axis=(1, 2), keepdims=True) + 1e-300
        assert (np.abs(o["dw"].cpu().numpy() - want[2]) <= 1e-7 * sw).all()
        ss = np.abs(want[3]).max(axis=(1, 2), keepdims=True) + 1e-300
        assert (np.abs(o["dsd"].cpu().numpy() - want[3]) <= 1e-5 * ss).all()
        # further directions at the same point
        m.eval_dual_device_more(3, T["dp2"].data_ptr(), T["dt2"].data_ptr(), T["ds2"].data_ptr(), o["dw2"].data_ptr(), o["dsd2"].data_ptr(), st)
        assert m.check() == pfc._lib.OK and m.last_dual_reused()
        sw2 = np.abs(want2[2]).max(axis=(1, 2), keepdims=True) + 1e-300
        assert (np.abs(o["dw2"].cpu().numpy() - want2[2]) <= 1e-7 * sw2).all()
        if devs is not None:
            assert m.last_shards() == 2
        m.close()


def test_multi_handle_dual_host_buffers(pfc):
    """pfc_eval_dual on {0, 0}: every shard runs the ordinary Dual entry point on its range; chunks at the same point reuse
    each shard's value pass once the ranges have settled."""
    C = pfc.configs
    w = C.c5_pile(n_side=3)
    n_dir = 6
    d_pose, d_twist, d_s = _dual_inputs(w, n_dir, 31)
    d_pose2, d_twist2, d_s2 = _dual_inputs(w, n_dir, 32)
    ref = C.build_scenario(w)
    want = ref.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, d_s, w.ins_ids)
    want2 = ref.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose2, d_twist2, d_s2, w.ins_ids)
    ref.close()
    m = C.build_scenario(w, devices=[0, 0])
    reused = []
    for rep in range(4):
        dp, dt, dsd, wnt = (d_pose, d_twist, d_s, want) if rep % 2 == 0 else (d_pose2, d_twist2, d_s2, want2)
        got = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, dp, dt, dsd, w.ins_ids)
        assert m.last_shards() == 2 and np.array_equal(got[4], wnt[4])
        _close(got[0], wnt[0], 1e-6)
        contact = wnt[4][:, 3] > 0
        sw = np.abs(wnt[2]).max(axis=(1, 2), keepdims=True) + 1e-300
        # (flat face-to-face patches: the partials of K̄^{-1/2} along its clamped null direction are rounding noise)
        ok = (np.abs(got[2] - wnt[2]) <= 1e-6 * sw).all(axis=(1, 2))
        assert ok[~contact].all() and ok[contact].mean() > 0.95
        reused.append(m.last_dual_reused())
    assert reused[-1], reused       # by the fourth chunk the partition has settled and every shard reuses its value pass
    m.close()


def test_team_mate_that_times_out_is_reported_not_summed(pfc):
    """Round 3's advisor finding: a rank of a team whose wait after pass 0 times out (while rank 0 sees every granule) went on
    with stale totals and a status word only it held; rank 0 summed its friction partials and reported success.  The status
    word now travels with the friction sums.  Diagnostic option team_fault makes rank 1 of every team behave that way: the
    evaluation must come back from the batched path (re-issued), correct."""
    w = pfc.configs.c3_blob_tool(1)
    m = pfc.configs.build_scenario(w)
    good = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert m.last_parts() == 0 and m.last_team() > 1
    r = H.oracle_run(pfc, w, debug=False)[0]
    assert np.array_equal(good[2][0], r.counts) and H.rel_err(good[0][0], r.wrench) < 1e-9
    m.set_option("team_fault", 1)
    bad = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert m.last_parts() == 1, "the faulty team evaluation was accepted"
    assert np.array_equal(bad[2][0], r.counts) and H.rel_err(bad[0][0], r.wrench) < 1e-9 and H.rel_err(bad[1][0], r.sdot) < 1e-6
    m.close()


def test_two_handles_evaluate_big_pairs_at_the_same_moment(pfc):
    """Two handles on two host threads, each evaluating single full-size C3 poses back to back: teams wait for their team-mates
    inside the launch, so before the per-device team slot two such launches could each be partly resident and sit out the
    bounded spin (65 ms) -- a 700x latency cliff.  Both must be correct and no call may take longer than 1 ms once warm."""
    ws = [pfc.configs.c3_blob_tool(1, seed=s) for s in (1, 2)]
    ms = [pfc.configs.build_scenario(w) for w in ws]
    refs = [H.oracle_run(pfc, w, debug=False)[0] for w in ws]
    for m, w in zip(ms, ws):      # warm: allocations, first launches
        for _ in range(3):
            m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    worst = [0.0, 0.0]
    teams = [0, 0]
    errs = []
    start = threading.Barrier(2)

    def work(i):
        try:
            m, w, r = ms[i], ws[i], refs[i]
            start.wait()
            for _ in range(200):
                t0 = time.perf_counter()
                wr, sd, c = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
                worst[i] = max(worst[i], time.perf_counter() - t0)
                teams[i] += m.last_team() > 1
                assert np.array_equal(c[0], r.counts) and H.rel_err(wr[0], r.wrench) < 1e-9 and H.rel_err(sd[0], r.sdot) < 1e-6
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    assert max(worst) < 1e-3, worst
    assert sum(teams) > 0, "no evaluation ran as a team"
    for m in ms:
        m.close()


def test_bound_evaluation_after_close_raises(pfc):
    """a BoundEvaluation kept across MechanismScenario.close() handed a freed handle to the library."""
    w = pfc.configs.c1_boxes()
    m = pfc.configs.build_scenario(w)
    b = m.bind(w.pose, w.twist, w.s, w.ins_ids)
    b()
    m.close()
    with pytest.raises(RuntimeError):
        b()
