"""pfc_eval_dual: the HIP Dual-number path against the Dual oracle (oracle/pfc_oracle_dual.cpp, itself
pinned by central differences in tests/test_oracle_dual.py).  Tolerances: 1e-6 relative on the wrench partials
(north_star's Float64 tolerance); the partials of sdot go through K̄^{-1/2} and inherit its conditioning, so they are
compared at 1e-5 of the largest partial of the item (values of sdot themselves are only reproducible to ~1e-4 on
sliver patches, tests/test_gpu_scale.py)."""
import numpy as np
import pytest

from helpers import oracle_ins, oracle_meshes
from test_oracle_dual import tangents

pytestmark = pytest.mark.gpu


def run_case(pfc, O, w, n_dir, seed, wr_tol=1e-6, sd_tol=1e-5, zero_s=False, options=None):
    rng = np.random.default_rng(seed)
    n = w.n_items
    if not zero_s:
        w.s[:] = rng.standard_normal((n, 6)) * 1e-3
    dq = rng.standard_normal((n, n_dir, 6)) * np.array([1, 1, 1, 0.05, 0.05, 0.05])
    d_twist = rng.standard_normal((n, n_dir, 6)) * np.array([1, 1, 1, 0.1, 0.1, 0.1])
    d_s = rng.standard_normal((n, n_dir, 6)) * 1e-3
    d_pose = np.zeros((n, n_dir, 24))
    for k in range(n):
        R0 = w.pose[k][:9].reshape(3, 3, order="F"); t0 = w.pose[k][9:12]
        d_pose[k] = tangents(R0, t0, dq[k])
    m = pfc.configs.build_scenario(w)
    for name, value in (options or {}).items():
        m.set_option(name, value)
    wr, sd, dw, dsd, counts = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, d_s, w.ins_ids)
    wr0, sd0, counts0 = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    # same value path (atomic summation order differs between two launches: rounding-level differences only)
    np.testing.assert_allclose(wr, wr0, rtol=1e-11, atol=1e-11 * np.abs(wr0).max())
    np.testing.assert_allclose(sd, sd0, rtol=1e-7, atol=1e-7 * max(np.abs(sd0).max(), 1e-300))
    assert np.array_equal(counts, counts0)
    m.close()
    om = oracle_meshes(w)
    n_contact = 0
    for k in range(n):
        c = w.instructions[int(w.ins_ids[k])]
        st, rw, rs, rdw, rdsd = O.evaluate_dual(om[c.id_1], om[c.id_2], oracle_ins(pfc, c), w.pose[k], w.twist[k],
                                                w.s[k], d_pose[k], d_twist[k], d_s[k])
        assert st == 0
        n_contact += counts[k, 3] > 0
        sw = max(np.abs(rdw).max(), 1e-300)
        assert np.abs(dw[k] - rdw).max() <= wr_tol * sw, (k, np.abs(dw[k] - rdw).max() / sw)
        ss = max(np.abs(rdsd).max(), 1e-300)
        assert np.abs(dsd[k] - rdsd).max() <= sd_tol * ss, (k, np.abs(dsd[k] - rdsd).max() / ss)
    assert n_contact > 0
    return n_contact


def test_dual_regularized_c4(pfc, O):
    w = pfc.configs.c2_box_on_plane(24, montecarlo=True)
    assert run_case(pfc, O, w, 6, 1) == 24


def test_dual_bristle_blob_tool(pfc, O):
    w = pfc.configs.c3_blob_tool(12, seed=3, n_div_blob=6, n_div_tool=4)
    run_case(pfc, O, w, 6, 2)


@pytest.mark.parametrize("n_dir", [1, 3, 7, 12, 16])
def test_dual_direction_counts(pfc, O, n_dir):
    w = pfc.configs.c3_blob_tool(5, seed=4, n_div_blob=5, n_div_tool=3)
    run_case(pfc, O, w, n_dir, 10 + n_dir)


def test_dual_tet_tet_and_mixed(pfc, O):
    for model in ("regularized", "bristle"):
        w = pfc.configs.vol_vol(6, n_div=3, model=model)
        run_case(pfc, O, w, 6, 5)


def test_dual_c1_boxes(pfc, O):
    w = pfc.configs.c1_boxes()
    run_case(pfc, O, w, 6, 6)


def test_dual_linear_in_seed(pfc):
    """Partials are linear in the seed: a Dual evaluation with seeds (a, b, a + 2b) returns d3 = d1 + 2 d2."""
    w = pfc.configs.c3_blob_tool(4, seed=8, n_div_blob=6, n_div_tool=4)
    rng = np.random.default_rng(9)
    n = w.n_items
    w.s[:] = rng.standard_normal((n, 6)) * 1e-3
    dq = rng.standard_normal((n, 2, 6)) * 0.1
    d_pose = np.zeros((n, 3, 24)); d_twist = np.zeros((n, 3, 6)); d_s = np.zeros((n, 3, 6))
    for k in range(n):
        R0 = w.pose[k][:9].reshape(3, 3, order="F"); t0 = w.pose[k][9:12]
        d_pose[k, :2] = tangents(R0, t0, dq[k])
    d_twist[:, :2] = rng.standard_normal((n, 2, 6)); d_s[:, :2] = rng.standard_normal((n, 2, 6)) * 1e-3
    for a in (d_pose, d_twist, d_s):
        a[:, 2] = a[:, 0] + 2 * a[:, 1]
    m = pfc.configs.build_scenario(w)
    _, _, dw, dsd, _ = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, d_s, w.ins_ids)
    m.close()
    np.testing.assert_allclose(dw[:, 2], dw[:, 0] + 2 * dw[:, 1], rtol=1e-9, atol=1e-9 * np.abs(dw).max())
    np.testing.assert_allclose(dsd[:, 2], dsd[:, 0] + 2 * dsd[:, 1], rtol=1e-7, atol=1e-7 * np.abs(dsd).max())


def test_dual_bad_arguments(pfc):
    w = pfc.configs.c1_boxes()
    m = pfc.configs.build_scenario(w)
    n = w.n_items
    with pytest.raises(pfc._lib.PFCError):
        m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, np.zeros((n, 17, 24)), np.zeros((n, 17, 6)), None,
                                               w.ins_ids)
    # the wrapper validates buffer sizes before the library copies from them
    with pytest.raises(ValueError):
        m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s[:2], np.zeros((n, 2, 24)), np.zeros((n, 2, 6)), None,
                                               w.ins_ids)
    with pytest.raises(ValueError):
        m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, np.zeros((n, 2, 24)), np.zeros((n, 2, 6)), None,
                                               w.ins_ids[:2])
    m.close()


def test_dual_argument_checks_do_not_depend_on_the_path(pfc):
    """Bristle instructions without the state buffer s: PFC_ERR_BAD_ARG on the two-stage path (first Dual evaluation of
    a handle) AND on the one-graph small-scene path (every later one), which used to evaluate silently with s = 0."""
    L = pfc._lib
    w = pfc.configs.c3_blob_tool(3, n_div_blob=5, n_div_tool=4)
    m = pfc.configs.build_scenario(w)
    n = w.n_items
    dp, dt = np.zeros((n, 2, 24)), np.zeros((n, 2, 6))
    with pytest.raises(L.PFCError) as ei:
        m.force_all_elastic_intersections_dual(w.pose, w.twist, None, dp, dt, None, w.ins_ids)
    assert ei.value.status == L.ERR_BAD_ARG
    m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, dp, dt, None, w.ins_ids)    # sets the small-scene hint
    m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, dp, dt, None, w.ins_ids)    # one-graph path
    with pytest.raises(L.PFCError) as ei:
        m.force_all_elastic_intersections_dual(w.pose, w.twist, None, dp, dt, None, w.ins_ids)
    assert ei.value.status == L.ERR_BAD_ARG
    m.close()


def test_dual_small_scene_path_and_its_fallback(pfc):
    """Scenes of <= 512 (item, direction) pairs take a one-synchronisation path from the second Dual evaluation of a
    handle on (value and Dual passes back to back, kept Dual polygons sized from the previous evaluation's pair count).
    It must reproduce the two-stage path (what a fresh handle runs), also when the pair count jumps past the
    speculative size (barely touching -> deep contact), and when it drops again."""
    rng = np.random.default_rng(11)
    w = pfc.configs.c3_blob_tool(8, seed=5, n_div_blob=6, n_div_tool=4)
    n, nd = w.n_items, 6
    w.s[:] = rng.standard_normal((n, 6)) * 1e-3
    d_pose = rng.standard_normal((n, nd, 24)) * 1e-2
    d_twist = rng.standard_normal((n, nd, 6)) * 0.1
    d_s = rng.standard_normal((n, nd, 6)) * 1e-3

    def shifted(scale):
        # move the tool along the centre line: scale < 1 deepens the contact, > 1 leaves only a grazing one
        p = w.pose.copy()
        R21 = p[:, :9].reshape(n, 3, 3).transpose(0, 2, 1)        # column-major -> R
        p[:, 9:12] *= scale
        t12 = -np.einsum("nji,nj->ni", R21, p[:, 9:12])             # x_r1_r2 = inverse of x_r2_r1
        p[:, 21:24] = t12
        return p

    def fresh(pose):
        f = pfc.configs.build_scenario(w)
        out = f.force_all_elastic_intersections_dual(pose, w.twist, w.s, d_pose, d_twist, d_s, w.ins_ids)
        f.close()
        return out

    m = pfc.configs.build_scenario(w)
    pairs = []
    for scale in (1.045, 1.045, 0.97, 1.0, 1.045, 0.97):   # graze (two-stage), graze (fast), deep (miss -> fallback), ...
        pose = shifted(scale)
        got = m.force_all_elastic_intersections_dual(pose, w.twist, w.s, d_pose, d_twist, d_s, w.ins_ids)
        ref = fresh(pose)
        assert np.array_equal(got[4], ref[4]), scale
        pairs.append(int(got[4][:, 2].sum()))
        for a, b, tol in ((got[0], ref[0], 1e-11), (got[1], ref[1], 1e-7), (got[2], ref[2], 1e-9), (got[3], ref[3], 1e-6)):
            np.testing.assert_allclose(a, b, rtol=tol, atol=tol * max(np.abs(b).max(), 1e-300), err_msg=str(scale))
    m.close()
    assert pairs[2] > 2 * pairs[1] + 64 > 64, pairs      # the deep contact really overshoots the speculative size


def test_dual_one_graph_path_with_copied_seeds(pfc):
    """Between 513 and 4096 (item, direction) pairs the one-graph Dual path copies seeds and results with memcpy nodes
    instead of reading them in place: 120 scenes x 6 directions, second and third evaluation of a handle against a
    fresh handle (which runs the two-stage path)."""
    rng = np.random.default_rng(21)
    w = pfc.configs.c2_box_on_plane(120, montecarlo=True)
    n, nd = w.n_items, 6
    d_pose = rng.standard_normal((n, nd, 24)) * 1e-2
    d_twist = rng.standard_normal((n, nd, 6)) * 0.1
    d_s = np.zeros((n, nd, 6))
    f = pfc.configs.build_scenario(w)
    ref = f.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, d_s, w.ins_ids)
    f.close()
    m = pfc.configs.build_scenario(w)
    for rep in range(3):
        got = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose * (1.0 if rep < 2 else 2.0), d_twist, d_s, w.ins_ids)
        assert np.array_equal(got[4], ref[4])
        np.testing.assert_allclose(got[0], ref[0], rtol=1e-11, atol=1e-11 * np.abs(ref[0]).max())
        if rep < 2:
            np.testing.assert_allclose(got[2], ref[2], rtol=1e-9, atol=1e-9 * np.abs(ref[2]).max())
        else:   # linear in the seeds: doubling d_pose (with the same d_twist) changes the partials, not the values
            assert np.abs(got[2] - ref[2]).max() > 1e-6 * np.abs(ref[2]).max()
    m.close()


def test_dual_device_resident_entry_point(pfc):
    """pfc_eval_dual_device + pfc_check (buffers in HBM, caller's stream, one synchronisation) against the host-buffer
    call; the first call of a handle has no pair count to size the kept Dual polygons from and may ask for a re-issue."""
    import torch
    rng = np.random.default_rng(23)
    w = pfc.configs.c3_blob_tool(40, n_div_blob=8, n_div_tool=6)
    n, nd = w.n_items, 6
    d_pose = rng.standard_normal((n, nd, 24)) * 1e-2
    d_twist = rng.standard_normal((n, nd, 6)) * 0.1
    d_s = rng.standard_normal((n, nd, 6)) * 1e-3
    f = pfc.configs.build_scenario(w)
    ref = f.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, d_s, w.ins_ids)
    f.close()
    dev = torch.device("cuda", 0)
    T = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    t_ins, t_pose, t_tw, t_s = T(w.ins_ids, torch.int32), T(w.pose), T(w.twist), T(w.s)
    t_dp, t_dt, t_ds = T(d_pose), T(d_twist), T(d_s)
    o_w, o_sd = torch.zeros((n, 6), dtype=torch.float64, device=dev), torch.zeros((n, 6), dtype=torch.float64, device=dev)
    o_dw, o_dsd = torch.zeros((n, nd, 6), dtype=torch.float64, device=dev), torch.zeros((n, nd, 6), dtype=torch.float64, device=dev)
    o_ct = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    m = pfc.configs.build_scenario(w)
    stream = torch.cuda.current_stream().cuda_stream
    attempts = []
    for rep in range(3):
        for k in range(40):
            m.eval_dual_device(n, nd, t_ins.data_ptr(), t_pose.data_ptr(), t_tw.data_ptr(), t_s.data_ptr(), t_dp.data_ptr(),
                               t_dt.data_ptr(), t_ds.data_ptr(), o_w.data_ptr(), o_sd.data_ptr(), o_dw.data_ptr(), o_dsd.data_ptr(),
                               o_ct.data_ptr(), stream)
            if m.check() == 0:
                break
        attempts.append(k + 1)
        assert np.array_equal(o_ct.cpu().numpy(), ref[4])
        for got, want, tol in ((o_w, ref[0], 1e-11), (o_sd, ref[1], 1e-7), (o_dw, ref[2], 1e-9), (o_dsd, ref[3], 1e-6)):
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=tol, atol=tol * max(np.abs(want).max(), 1e-300))
    assert attempts[-1] == 1, attempts          # settled: no re-issue once the pair count is known
    # d_ds = NULL means zeros
    m.eval_dual_device(n, nd, t_ins.data_ptr(), t_pose.data_ptr(), t_tw.data_ptr(), t_s.data_ptr(), t_dp.data_ptr(),
                       t_dt.data_ptr(), 0, o_w.data_ptr(), o_sd.data_ptr(), o_dw.data_ptr(), o_dsd.data_ptr(), 0, stream)
    assert m.check() == 0
    f = pfc.configs.build_scenario(w)
    ref0 = f.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, None, w.ins_ids)
    f.close()
    np.testing.assert_allclose(o_dsd.cpu().numpy(), ref0[3], rtol=1e-6, atol=1e-6 * np.abs(ref0[3]).max())
    m.close()


def test_dual_large_batch_one_sync_equals_two_stage(pfc):
    """Above the small-scene limits pfc_eval_dual goes through pfc_eval_dual_device (one synchronisation per attempt);
    the former two-stage path (PFC_DUAL_TWO_STAGE=1: value pass, host reads the pair count, Dual passes) is the reference."""
    import os
    rng = np.random.default_rng(29)
    w = pfc.configs.c3_blob_tool(700, n_div_blob=6, n_div_tool=5)
    n, nd = w.n_items, 6
    d_pose = rng.standard_normal((n, nd, 24)) * 1e-2
    d_twist = rng.standard_normal((n, nd, 6)) * 0.1
    d_s = rng.standard_normal((n, nd, 6)) * 1e-3
    outs = []
    for two_stage in (True, False):
        if two_stage:
            os.environ["PFC_DUAL_TWO_STAGE"] = "1"
        else:
            os.environ.pop("PFC_DUAL_TWO_STAGE", None)
        m = pfc.configs.build_scenario(w)
        for _ in range(2):
            out = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, d_s, w.ins_ids)
        outs.append(out)
        m.close()
    os.environ.pop("PFC_DUAL_TWO_STAGE", None)
    a, b = outs
    assert np.array_equal(a[4], b[4])
    for x, y, tol in ((a[0], b[0], 1e-11), (a[1], b[1], 1e-7), (a[2], b[2], 1e-9), (a[3], b[3], 1e-6)):
        np.testing.assert_allclose(y, x, rtol=tol, atol=tol * max(np.abs(x).max(), 1e-300))


def test_jacobian_chunks_reuse_the_value_pass(pfc):
    """The chunks of one Jacobian (src/radau/radau_functions.jl:2-14): Dual evaluations with the same values and different
    partials.  pfc_eval_dual recognises bitwise-equal value inputs and runs only the Dual passes on the previous call's
    value pass (option dual_reuse); every call must equal a fresh handle's full evaluation, a changed value input must
    not be mistaken for a repeat, and a different number of directions may follow."""
    rng = np.random.default_rng(31)
    w = pfc.configs.c3_blob_tool(700, n_div_blob=6, n_div_tool=5)
    n = w.n_items

    def seeds(nd):
        return (rng.standard_normal((n, nd, 24)) * 1e-2, rng.standard_normal((n, nd, 6)) * 0.1,
                rng.standard_normal((n, nd, 6)) * 1e-3)

    def fresh(pose, twist, s, sd):
        f = pfc.configs.build_scenario(w)
        f.set_option("dual_reuse", 0)
        out = f.force_all_elastic_intersections_dual(pose, twist, s, *sd, w.ins_ids)
        f.close()
        return out

    def same(a, b):
        assert np.array_equal(a[4], b[4])
        for x, y, tol in ((a[0], b[0], 1e-11), (a[1], b[1], 1e-7), (a[2], b[2], 1e-9), (a[3], b[3], 1e-6)):
            np.testing.assert_allclose(y, x, rtol=tol, atol=tol * max(np.abs(x).max(), 1e-300))

    m = pfc.configs.build_scenario(w)
    stats = []
    pose2 = w.pose.copy()
    pose2[5, 21] += 1e-9            # one translation component of one item, far below any tolerance of the comparison
    for pose, nd in ((w.pose, 6), (w.pose, 6), (w.pose, 3), (pose2, 6), (pose2, 2), (w.pose, 6)):
        sd = seeds(nd)
        got = m.force_all_elastic_intersections_dual(pose, w.twist, w.s, *sd, w.ins_ids)
        stats.append(m.last_dual_reused())
        same(fresh(pose, w.twist, w.s, sd), got)
    assert stats == [False, True, True, False, True, False], stats
    # the repeat of a point is cheaper than its first evaluation (only the Dual passes run): timing is not asserted, the
    # path is: a value evaluation in between must end the reuse (the handle's device state is overwritten)
    sd = seeds(6)
    a = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *sd, w.ins_ids)
    m.force_all_elastic_intersections(pose2, w.twist, w.s, w.ins_ids)
    b = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *sd, w.ins_ids)
    assert not m.last_dual_reused()
    same(a, b)
    m.close()


def test_dual_device_more_directions(pfc):
    """pfc_eval_dual_device_more: further seed directions on the value pass of the previous pfc_eval_dual_device
    evaluation; refused (PFC_ERR_STATE) without one or after another evaluation."""
    import torch
    rng = np.random.default_rng(37)
    w = pfc.configs.c3_blob_tool(40, n_div_blob=8, n_div_tool=6)
    n = w.n_items
    dev = torch.device("cuda", 0)
    T = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    t_ins, t_pose, t_tw, t_s = T(w.ins_ids, torch.int32), T(w.pose), T(w.twist), T(w.s)
    o_w, o_sd = torch.zeros((n, 6), dtype=torch.float64, device=dev), torch.zeros((n, 6), dtype=torch.float64, device=dev)
    o_ct = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    m = pfc.configs.build_scenario(w)

    def seeds(nd):
        return (rng.standard_normal((n, nd, 24)) * 1e-2, rng.standard_normal((n, nd, 6)) * 0.1,
                rng.standard_normal((n, nd, 6)) * 1e-3)

    def ref(sd):
        f = pfc.configs.build_scenario(w)
        f.set_option("dual_reuse", 0)
        out = f.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *sd, w.ins_ids)
        f.close()
        return out

    sd0 = seeds(6)
    t0 = [T(x) for x in sd0]
    o_dw = torch.zeros((n, 6, 6), dtype=torch.float64, device=dev); o_dsd = torch.zeros_like(o_dw)
    with pytest.raises(Exception):           # nothing to extend yet
        m.eval_dual_device_more(6, t0[0].data_ptr(), t0[1].data_ptr(), t0[2].data_ptr(), o_dw.data_ptr(), o_dsd.data_ptr(), stream)
    for k in range(40):
        m.eval_dual_device(n, 6, t_ins.data_ptr(), t_pose.data_ptr(), t_tw.data_ptr(), t_s.data_ptr(), t0[0].data_ptr(),
                           t0[1].data_ptr(), t0[2].data_ptr(), o_w.data_ptr(), o_sd.data_ptr(), o_dw.data_ptr(), o_dsd.data_ptr(),
                           o_ct.data_ptr(), stream)
        if m.check() == 0:
            break
    for nd in (6, 4, 9):
        sd = seeds(nd)
        t = [T(x) for x in sd]
        dw = torch.zeros((n, nd, 6), dtype=torch.float64, device=dev); dsd = torch.zeros_like(dw)
        m.eval_dual_device_more(nd, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), dw.data_ptr(), dsd.data_ptr(), stream)
        assert m.check() == 0
        want = ref(sd)
        np.testing.assert_allclose(dw.cpu().numpy(), want[2], rtol=1e-9, atol=1e-9 * np.abs(want[2]).max())
        np.testing.assert_allclose(dsd.cpu().numpy(), want[3], rtol=1e-6, atol=1e-6 * np.abs(want[3]).max())
    # a value evaluation ends it
    m.eval_device(n, t_ins.data_ptr(), t_pose.data_ptr(), t_tw.data_ptr(), t_s.data_ptr(), o_w.data_ptr(), o_sd.data_ptr(),
                  o_ct.data_ptr(), stream)
    assert m.check() == 0
    with pytest.raises(Exception):
        m.eval_dual_device_more(6, t0[0].data_ptr(), t0[1].data_ptr(), t0[2].data_ptr(), o_dw.data_ptr(), o_dsd.data_ptr(), stream)
    m.close()


def test_one_graph_path_at_the_selection_threshold(pfc):
    """512 items: the largest scene of the one-graph Dual path and the smallest that selects the pairs its seeds touch.
    A fresh handle records that graph with the selection lists still unallocated (an allocation inside a stream capture
    fails: found by scripts/soak_threads.py) -- the captured sequence must do without the selection; later chunks at the
    same point, launched eagerly, use it.  All equal a handle that never reuses."""
    rng = np.random.default_rng(53)
    w = pfc.configs.c3_blob_tool(512, seed=6, n_div_blob=5, n_div_tool=4)
    n, nd = w.n_items, 6
    w.s[:] = rng.standard_normal((n, 6)) * 1e-3

    def seeds():
        return (rng.standard_normal((n, nd, 24)) * 1e-2, rng.standard_normal((n, nd, 6)) * 0.1, rng.standard_normal((n, nd, 6)) * 1e-3)

    m = pfc.configs.build_scenario(w)
    f = pfc.configs.build_scenario(w)
    f.set_option("dual_reuse", 0)
    # a Dual evaluation of a few items first: the pair count it leaves opens the one-graph path for the 512 that follow,
    # and nothing so far has allocated the selection lists
    sd0 = seeds()
    m.force_all_elastic_intersections_dual(w.pose[:64], w.twist[:64], w.s[:64], sd0[0][:64], sd0[1][:64], sd0[2][:64], w.ins_ids[:64])
    for k in range(5):
        pose = w.pose.copy()
        if k in (1, 2):
            pose[:, 21:24] += 1e-5 * k       # new points: the graph is recorded / replayed; k = 3, 4 repeat the point of k = 2
        elif k > 2:
            pose[:, 21:24] += 2e-5
        sd = seeds()
        got = m.force_all_elastic_intersections_dual(pose, w.twist, w.s, *sd, w.ins_ids)
        want = f.force_all_elastic_intersections_dual(pose, w.twist, w.s, *sd, w.ins_ids)
        assert np.array_equal(got[4], want[4])
        for x, y, tol in ((want[0], got[0], 1e-10), (want[1], got[1], 1e-6), (want[2], got[2], 1e-8), (want[3], got[3], 1e-5)):
            np.testing.assert_allclose(y, x, rtol=tol, atol=tol * max(np.abs(x).max(), 1e-300))
    assert m.last_dual_reused()
    m.close(); f.close()


@pytest.mark.parametrize("cfg", ["pile", "c3batch", "c4", "volvol"])
def test_zero_seeds_are_skipped_and_give_zero_partials(pfc, cfg):
    """Every partial of an (item, direction) is linear in that key's 36 seed components, so the Dual passes skip keys
    whose seeds are all zero (what nearly all keys of a Radau chunk are in a scene of many bodies: a contact instruction
    depends on the states of its two bodies only).  Checked: a sparsely seeded evaluation returns exact zeros for the
    zero keys and, for the others, what the densely seeded evaluation returns for the same seeds (the keys do not mix);
    values and counters do not depend on the seeds; a key seeded in its bristle state alone is evaluated; the same on the
    reused value pass and through the device entry points."""
    import torch
    rng = np.random.default_rng(47)
    if cfg == "pile":
        w = pfc.configs.c5_pile(n_side=3, n_divs=(1, 2, 3))        # 351 bristle instructions over different meshes
    elif cfg == "c3batch":
        w = pfc.configs.c3_blob_tool(300, seed=5, n_div_blob=6, n_div_tool=4)     # beyond the small-scene limits
    elif cfg == "c4":
        w = pfc.configs.c2_box_on_plane(40, montecarlo=True)       # regularized
    else:
        w = pfc.configs.vol_vol(6, model="bristle")                # tet-tet
    n, nd = w.n_items, 6
    w.s[:] = rng.standard_normal((n, 6)) * 1e-3
    dense = (rng.standard_normal((n, nd, 24)) * 1e-2, rng.standard_normal((n, nd, 6)) * 0.1, rng.standard_normal((n, nd, 6)) * 1e-3)
    mask = rng.random((n, nd)) < 0.15          # keys that keep their seeds
    mask[rng.integers(0, n, 5)] = False        # some items without any seeded direction
    sparse = [x * mask[:, :, None] for x in dense]
    only_s = np.zeros((n, nd), dtype=bool)     # keys seeded in the bristle state alone
    only_s[rng.integers(0, n, 7), rng.integers(0, nd, 7)] = True
    only_s &= ~mask
    sparse[2] = sparse[2] + dense[2] * only_s[:, :, None]
    live = mask | only_s

    def fresh(sd):
        f = pfc.configs.build_scenario(w)
        f.set_option("dual_reuse", 0)
        out = f.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *sd, w.ins_ids)
        f.close()
        return out

    want_dense = fresh(dense)
    # the s-only keys against a dense evaluation of exactly those seeds
    s_only_seeds = (dense[0] * mask[:, :, None], dense[1] * mask[:, :, None], sparse[2])
    m = pfc.configs.build_scenario(w)
    got = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *sparse, w.ins_ids)
    again = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *sparse, w.ins_ids)       # the reused value pass
    for g in (got, again):
        assert np.array_equal(g[4], want_dense[4])
        np.testing.assert_allclose(g[0], want_dense[0], rtol=1e-12, atol=1e-12 * np.abs(want_dense[0]).max())
        np.testing.assert_allclose(g[1], want_dense[1], rtol=1e-9, atol=1e-9 * max(np.abs(want_dense[1]).max(), 1e-300))
        assert np.all(g[2][~live] == 0.0) and np.all(g[3][~live] == 0.0)
        # seeded in pose / twist / state exactly as in the dense evaluation: the same partials
        same = mask & (np.abs(dense[2]).max(axis=2) >= 0)      # (all of mask: d_s is kept with the other seeds)
        np.testing.assert_allclose(g[2][same], want_dense[2][same], rtol=1e-10, atol=1e-10 * np.abs(want_dense[2]).max())
        np.testing.assert_allclose(g[3][same], want_dense[3][same], rtol=1e-7, atol=1e-7 * max(np.abs(want_dense[3]).max(), 1e-300))
    # linearity for the s-only keys: d sdot = -(K^-1/2 ... + I) ds / tau is what a dense evaluation with the other seeds
    # removed gives
    zp = (np.zeros_like(dense[0]), np.zeros_like(dense[1]), dense[2] * only_s[:, :, None])
    want_s = fresh(zp)
    np.testing.assert_allclose(got[3][only_s], want_s[3][only_s], rtol=1e-9, atol=1e-9 * max(np.abs(want_s[3]).max(), 1e-300))
    np.testing.assert_allclose(got[2][only_s], want_s[2][only_s], rtol=1e-9, atol=1e-9 * max(np.abs(want_dense[2]).max(), 1e-300))
    # device entry points
    dev = torch.device("cuda", 0)
    T = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    t = [T(w.ins_ids, torch.int32), T(w.pose), T(w.twist), T(w.s)] + [T(x) for x in sparse]
    o = [torch.zeros((n, 6), dtype=torch.float64, device=dev), torch.zeros((n, 6), dtype=torch.float64, device=dev),
         torch.full((n, nd, 6), 7.0, dtype=torch.float64, device=dev), torch.full((n, nd, 6), 7.0, dtype=torch.float64, device=dev),
         torch.zeros((n, 4), dtype=torch.int32, device=dev)]
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(40):
        m.eval_dual_device(n, nd, *[x.data_ptr() for x in t], *[x.data_ptr() for x in o], st)
        if m.check() == 0:
            break
    np.testing.assert_allclose(o[2].cpu().numpy(), got[2], rtol=1e-10, atol=1e-10 * np.abs(want_dense[2]).max())
    np.testing.assert_allclose(o[3].cpu().numpy(), got[3], rtol=1e-7, atol=1e-7 * max(np.abs(want_dense[3]).max(), 1e-300))
    o[2].fill_(7.0); o[3].fill_(7.0)
    m.eval_dual_device_more(nd, t[4].data_ptr(), t[5].data_ptr(), t[6].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st)
    assert m.check() == 0
    assert np.all(o[2].cpu().numpy()[~live] == 0.0) and np.all(o[3].cpu().numpy()[~live] == 0.0)
    np.testing.assert_allclose(o[2].cpu().numpy(), got[2], rtol=1e-10, atol=1e-10 * np.abs(want_dense[2]).max())
    m.close()


@pytest.mark.parametrize("poison", [0, 1])
@pytest.mark.parametrize("cfg", ["c1", "c2", "pencil", "tight"])
def test_dual_device_on_small_scenes(pfc, cfg, poison):
    """pfc_eval_dual_device on scenes the one-workgroup-per-item kernel takes: its value pass hands item records and lists
    to the batched Dual passes, pfc_check reads one counter; pfc_eval_dual_device_more extends it.  "tight": more
    contributing pairs than a fresh handle's hand-over lists hold -- the miss must come back as PFC_ERR_OVERFLOW and the
    re-issue (batched value pass, which grows the lists) must succeed with the same numbers.  Every evaluation equals a fresh handle's host-buffer evaluation."""
    import torch
    rng = np.random.default_rng(43)
    if cfg == "c1":
        w = pfc.configs.c1_boxes()
    elif cfg == "c2":
        w = pfc.configs.c2_box_on_plane(3, montecarlo=True)
    elif cfg == "pencil":
        w = pfc.configs.c3_blob_tool(2, seed=3, n_div_blob=5, n_div_tool=4)
    else:
        w = pfc.configs.c2_box_on_plane(256, n_div=12, montecarlo=True)      # ~93k contributing pairs: a fresh handle's lists hold 65536
    n = w.n_items
    dev = torch.device("cuda", 0)
    T = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    t_ins, t_tw, t_s = T(w.ins_ids, torch.int32), T(w.twist), T(w.s)
    o_w, o_sd = torch.zeros((n, 6), dtype=torch.float64, device=dev), torch.zeros((n, 6), dtype=torch.float64, device=dev)
    o_ct = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    m = pfc.configs.build_scenario(w)
    m.set_option("poison", poison)

    def seeds(nd):
        return (rng.standard_normal((n, nd, 24)) * 1e-2, rng.standard_normal((n, nd, 6)) * 0.1,
                rng.standard_normal((n, nd, 6)) * 1e-3)

    def fresh(pose, sd):
        f = pfc.configs.build_scenario(w)
        f.set_option("dual_reuse", 0)
        f.set_option("fused", 0)
        out = f.force_all_elastic_intersections_dual(pose, w.twist, w.s, *sd, w.ins_ids)
        f.close()
        return out

    def same(got, want, nd):
        assert np.array_equal(got[4], want[4])
        for x, y, tol in ((want[0], got[0], 1e-10), (want[1], got[1], 1e-6), (want[2], got[2], 1e-8), (want[3], got[3], 1e-5)):
            np.testing.assert_allclose(y, x, rtol=tol, atol=tol * max(np.abs(x).max(), 1e-300))

    reissued = 0
    for k in range(4):
        pose = w.pose.copy()
        pose[:, 21:24] += rng.standard_normal((n, 3)) * 1e-4 * k        # another point each time
        t_pose = T(pose)
        sd = seeds(6)
        t = [T(x) for x in sd]
        dw = torch.zeros((n, 6, 6), dtype=torch.float64, device=dev); dsd = torch.zeros_like(dw)
        for attempt in range(40):
            m.eval_dual_device(n, 6, t_ins.data_ptr(), t_pose.data_ptr(), t_tw.data_ptr(), t_s.data_ptr(), t[0].data_ptr(),
                               t[1].data_ptr(), t[2].data_ptr(), o_w.data_ptr(), o_sd.data_ptr(), dw.data_ptr(), dsd.data_ptr(),
                               o_ct.data_ptr(), stream)
            if m.check() == 0:
                break
            reissued += 1
        else:
            raise AssertionError("no success in 40 issues")
        want = fresh(pose, sd)
        same((o_w.cpu().numpy(), o_sd.cpu().numpy(), dw.cpu().numpy(), dsd.cpu().numpy(), o_ct.cpu().numpy()), want, 6)
        for nd in (6, 3):
            sd2 = seeds(nd)
            t2 = [T(x) for x in sd2]
            dw2 = torch.zeros((n, nd, 6), dtype=torch.float64, device=dev); dsd2 = torch.zeros_like(dw2)
            m.eval_dual_device_more(nd, t2[0].data_ptr(), t2[1].data_ptr(), t2[2].data_ptr(), dw2.data_ptr(), dsd2.data_ptr(), stream)
            assert m.check() == 0
            want2 = fresh(pose, sd2)
            same((o_w.cpu().numpy(), o_sd.cpu().numpy(), dw2.cpu().numpy(), dsd2.cpu().numpy(), o_ct.cpu().numpy()), want2, nd)
    if cfg == "tight":
        assert reissued >= 1
    m.close()


@pytest.mark.parametrize("poison", [0, 1])
@pytest.mark.parametrize("cfg", ["c1", "c2", "pencil", "c3pose"])
def test_jacobian_chunks_on_small_scenes(pfc, cfg, poison):
    """The same reuse on the small-scene paths: the first Dual evaluation of a point runs in the all-in-one kernel
    (regularized scenes) or as fused value kernel + batched Dual passes (bristle); a repeat of the point takes the
    hybrid path, which hands its lists over, and the chunks after it run only the Dual passes.  Every call equals a fresh
    handle's evaluation; value evaluations in between end the reuse.  poison = 1: the work lists are refilled with
    entries that must never be followed before every evaluation -- but not before a reuse, whose lists they are (a first
    version did, and the Dual narrowphase followed the poisoned candidates: GPU memory fault, found by scripts/soak.py)."""
    rng = np.random.default_rng(41)
    if cfg == "c1":
        w = pfc.configs.c1_boxes()
    elif cfg == "c2":
        w = pfc.configs.c2_box_on_plane(3, montecarlo=True)
    elif cfg == "pencil":
        w = pfc.configs.c3_blob_tool(2, seed=3, n_div_blob=5, n_div_tool=4)       # bristle, small trees
    else:
        w = pfc.configs.c3_blob_tool(3, seed=4)       # the full-size meshes: beyond the fused kernel, the one-graph path
    n = w.n_items

    def seeds(nd):
        return (rng.standard_normal((n, nd, 24)) * 1e-2, rng.standard_normal((n, nd, 6)) * 0.1,
                rng.standard_normal((n, nd, 6)) * 1e-3)

    def fresh(sd):
        f = pfc.configs.build_scenario(w)
        f.set_option("dual_reuse", 0)
        f.set_option("fused", 0)
        out = f.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *sd, w.ins_ids)
        f.close()
        return out

    m = pfc.configs.build_scenario(w)
    m.set_option("poison", poison)
    flags = []
    for k, nd in enumerate((6, 6, 6, 3, 6)):
        sd = seeds(nd)
        got = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *sd, w.ins_ids)
        flags.append(m.last_dual_reused())
        want = fresh(sd)
        assert np.array_equal(got[4], want[4])
        for x, y, tol in ((want[0], got[0], 1e-10), (want[1], got[1], 1e-6), (want[2], got[2], 1e-8), (want[3], got[3], 1e-5)):
            np.testing.assert_allclose(y, x, rtol=tol, atol=tol * max(np.abs(x).max(), 1e-300))
    assert flags[0] is False and flags[-1] is True and flags[-2] is True, flags
    m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    sd = seeds(6)
    m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *sd, w.ins_ids)
    assert not m.last_dual_reused()
    m.close()


def test_jacobian_chunks_without_ids_and_state(pfc):
    """The reuse test compares the optional inputs too: ins_ids = NULL (item i is instruction i) and s = NULL (no bristle
    instruction) on the small-scene paths, and a call that differs from the previous one ONLY in having ins_ids (same
    mapping) or s (zeros) must still give the same results, reused or not."""
    rng = np.random.default_rng(43)
    w = pfc.configs.c1_boxes()
    n = w.n_items
    ids = np.arange(n, dtype=np.int32)
    assert np.array_equal(w.ins_ids, ids)

    def seeds(nd):
        return (rng.standard_normal((n, nd, 24)) * 1e-2, rng.standard_normal((n, nd, 6)) * 0.1, None)

    f = pfc.configs.build_scenario(w)
    f.set_option("dual_reuse", 0)
    f.set_option("fused", 0)
    m = pfc.configs.build_scenario(w)
    calls = [(None, None), (None, None), (None, None), (ids, None), (ids, None), (ids, np.zeros((n, 6))), (None, np.zeros((n, 6))),
             (None, None)]
    flags = []
    for i_arg, s_arg in calls:
        sd = seeds(6)
        got = m.force_all_elastic_intersections_dual(w.pose, w.twist, s_arg, *sd, i_arg)
        flags.append(m.last_dual_reused())
        want = f.force_all_elastic_intersections_dual(w.pose, w.twist, None, *sd, None)
        assert np.array_equal(got[4], want[4])
        for x, y, tol in ((want[0], got[0], 1e-10), (want[1], got[1], 1e-6), (want[2], got[2], 1e-8), (want[3], got[3], 1e-5)):
            np.testing.assert_allclose(y, x, rtol=tol, atol=tol * max(np.abs(x).max(), 1e-300))
    assert flags[0] is False and flags[2] is True, flags
    f.close(); m.close()


def test_host_chunk_cache_is_tied_to_its_value_pass(pfc):
    """A caller's own pfc_eval_dual_device(Y) + pfc_check between two host-buffer Dual evaluations at X (same item count)
    leaves a checked Dual value pass on the handle -- Y's.  The host-side cache of X (pinned value inputs / outputs) must
    not be paired with it: the second call at X is a full evaluation (last_dual_reused() false) and equals a fresh handle's."""
    import torch
    rng = np.random.default_rng(41)
    w = pfc.configs.c3_blob_tool(700, n_div_blob=6, n_div_tool=5)
    n, nd = w.n_items, 6
    dev = torch.device("cuda", 0)
    T = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)

    def seeds():
        return (rng.standard_normal((n, nd, 24)) * 1e-2, rng.standard_normal((n, nd, 6)) * 0.1,
                rng.standard_normal((n, nd, 6)) * 1e-3)

    def same(a, b):
        assert np.array_equal(a[4], b[4])
        for x, y, tol in ((a[0], b[0], 1e-11), (a[1], b[1], 1e-7), (a[2], b[2], 1e-9), (a[3], b[3], 1e-6)):
            np.testing.assert_allclose(y, x, rtol=tol, atol=tol * max(np.abs(x).max(), 1e-300))

    # Y: other poses of the same scene (the blob turned by a few degrees: other candidates, other wrenches)
    wy = pfc.configs.c3_blob_tool(700, seed=777, n_div_blob=6, n_div_tool=5)
    m = pfc.configs.build_scenario(w)
    sd1, sd2, sdy = seeds(), seeds(), seeds()
    m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *sd1, w.ins_ids)          # host Dual at X
    t_ins, t_pose, t_tw, t_s = T(w.ins_ids, torch.int32), T(wy.pose), T(wy.twist), T(wy.s)
    ty = [T(x) for x in sdy]
    o_w = torch.zeros((n, 6), dtype=torch.float64, device=dev); o_sd = torch.zeros_like(o_w)
    o_ct = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    o_dw = torch.zeros((n, nd, 6), dtype=torch.float64, device=dev); o_dsd = torch.zeros_like(o_dw)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(40):                                                                       # device Dual at Y, checked
        m.eval_dual_device(n, nd, t_ins.data_ptr(), t_pose.data_ptr(), t_tw.data_ptr(), t_s.data_ptr(), ty[0].data_ptr(),
                           ty[1].data_ptr(), ty[2].data_ptr(), o_w.data_ptr(), o_sd.data_ptr(), o_dw.data_ptr(), o_dsd.data_ptr(),
                           o_ct.data_ptr(), stream)
        if m.check() == 0:
            break
    got = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *sd2, w.ins_ids)    # host Dual at X again
    assert not m.last_dual_reused()
    f = pfc.configs.build_scenario(w)
    f.set_option("dual_reuse", 0)
    want = f.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *sd2, w.ins_ids)
    f.close()
    same(want, got)
    # and the chunk after it does reuse (the cache now belongs to the value pass on the device)
    sd3 = seeds()
    got3 = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *sd3, w.ins_ids)
    assert m.last_dual_reused()
    f = pfc.configs.build_scenario(w)
    f.set_option("dual_reuse", 0)
    same(f.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *sd3, w.ins_ids), got3)
    f.close()
    m.close()


def test_pass_b_folded_into_pass_a_equals_the_three_pass_form(pfc):
    """Option dual_fold (default 1, tri-tet scenes on the batched value pass): the 21 patch-stiffness sums are formed in pass A about
    the VALUE pass's cop and moved to the Dual cop per key by the parallel-axis rule (k_dual_eig) instead of in a pass of their
    own over the kept polygons.  Same partials as the three-pass form (option 0) -- on a batch big enough for the batched path,
    first chunk and a further chunk; items whose scaled stiffness has a rounding-level eigenvalue excepted."""
    w = pfc.configs.c3_blob_tool(700, seed=11, n_div_blob=6, n_div_tool=4)
    rng = np.random.default_rng(3)
    n, nd = w.n_items, 6
    dp = rng.standard_normal((n, nd, 24)) * 1e-3; dt = rng.standard_normal((n, nd, 6)) * 1e-2; ds = rng.standard_normal((n, nd, 6)) * 1e-4
    out = {}
    for fold in (1, 0):
        m = pfc.configs.build_scenario(w)
        m.set_option("dual_fold", fold)
        a = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, dp, dt, ds, w.ins_ids)
        b = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, dp * 2.0, dt, ds, w.ins_ids)
        assert m.last_dual_reused() and m.last_parts() != 0
        out[fold] = (a, b)
        m.close()
    for x, y in zip(out[1], out[0]):
        assert np.array_equal(x[4], y[4])
        sw = np.abs(y[2]).max(axis=(1, 2), keepdims=True) + 1e-300
        ss = np.abs(y[3]).max(axis=(1, 2), keepdims=True) + 1e-300
        okw = (np.abs(x[2] - y[2]) <= 1e-8 * sw).all(axis=(1, 2)); oks = (np.abs(x[3] - y[3]) <= 1e-6 * ss).all(axis=(1, 2))
        assert okw.mean() > 0.99 and oks.mean() > 0.98, (okw.mean(), oks.mean())
