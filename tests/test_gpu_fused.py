"""The fused small-scene kernel (csrc/pfc_fused.h: one workgroup per item, one launch per evaluation; what Radau's
stage evaluations of scenes like test/boxes.jl go through, src/radau/radau_functions.jl:64-70) against the CPU oracle
and against the batched launch sequence.  Integer outputs bit-exact, wrenches 1e-9, as for the batched path."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _run(pfc, w, fused=1):
    m = pfc.configs.build_scenario(w)
    m.set_option("fused", fused)
    out = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    return m, out


def _against_oracle(pfc, w, wrench, sdot, counts, sd_tol=1e-6):
    ref = H.oracle_run(pfc, w, debug=False)
    for k, r in enumerate(ref):
        assert np.array_equal(counts[k], r.counts), (k, counts[k], r.counts)
        for name, a, b, tol in (("wrench", wrench[k], r.wrench, TOL), ("sdot", sdot[k], r.sdot, sd_tol)):
            if np.linalg.norm(b) == 0.0:
                assert np.linalg.norm(a) == 0.0, (name, k, a)
            else:
                assert H.rel_err(a, b) < tol, (name, k, a, b)
    return ref


@pytest.mark.parametrize("config", ["c1", "c2", "c4"])
def test_fused_regularized_configs(pfc, config):
    w = {"c1": pfc.configs.c1_boxes, "c2": lambda: pfc.configs.c2_box_on_plane(1),
         "c4": lambda: pfc.configs.c2_box_on_plane(256, montecarlo=True)}[config]()
    m, (wrench, sdot, counts) = _run(pfc, w)
    assert m.last_parts() == 0, "the fused kernel did not run"
    # a lone mid-sized item (C2: 972 tets) takes a small team of workgroups, many items (C4) or tiny ones (C1) one each
    assert m.last_team() == {"c1": 1, "c2": 8, "c4": 1}[config]      # a workgroup per 128 leaves of the pair
    _against_oracle(pfc, w, wrench, sdot, counts)
    st = m.stats()
    assert st["candidates"] == int(counts[:, 1].sum()) and st["node_tests"] == int(counts[:, 0].sum())
    assert st["nonempty"] == int(counts[:, 2].sum()) and st["tractions"] == int(counts[:, 3].sum())
    m.close()


@pytest.mark.parametrize("n_quad", [1, 2])
def test_fused_bristle(pfc, n_quad):
    """Bristle items: three passes inside the kernel (cop; patch stiffness about the cop; friction) with the 6x6 eigen
    on one wave in between."""
    w = pfc.configs.c3_blob_tool(12, n_div_blob=8, n_div_tool=6)
    w.instructions[0].n_quad_rule = n_quad
    m, (wrench, sdot, counts) = _run(pfc, w)
    assert m.last_parts() == 0 and m.last_team() == 16       # 12 items of 2 000 leaves: teams of 16
    _against_oracle(pfc, w, wrench, sdot, counts)
    assert np.count_nonzero(counts[:, 3]) >= 6
    m.set_option("team", 0)                                   # a workgroup per item: the same integers
    w1, s1, c1 = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert m.last_team() == 1 and np.array_equal(c1, counts)
    np.testing.assert_allclose(w1, wrench, rtol=1e-11, atol=1e-11 * np.abs(wrench).max())
    m.close()
    # separated bodies: zero wrench, sdot = -s / tau bit for bit
    w = pfc.configs.c3_blob_tool(3, n_div_blob=6, n_div_tool=5, distance=0.25)
    m, (wrench, sdot, counts) = _run(pfc, w)
    assert m.last_parts() == 0
    ref = H.oracle_run(pfc, w, debug=False)
    for k, r in enumerate(ref):
        assert np.array_equal(counts[k], r.counts) and np.all(wrench[k] == 0.0) and np.array_equal(sdot[k], r.sdot)
    m.close()


@pytest.mark.parametrize("degenerate", [False, True])
def test_fused_fuzz(pfc, degenerate):
    """Random and degenerate (axis-aligned, lattice) poses of small meshes, both friction models in one evaluation."""
    from test_gpu_parity import _fuzz_workload
    rng = np.random.default_rng(191 + int(degenerate))
    w = _fuzz_workload(pfc, rng, 256, degenerate, tet_tet=False)
    m, (wrench, sdot, counts) = _run(pfc, w)
    assert m.last_parts() == 0
    ref = H.oracle_run(pfc, w, debug=False)
    n_contact = 0
    for k, r in enumerate(ref):
        assert np.array_equal(counts[k], r.counts), (k, counts[k], r.counts)
        if np.linalg.norm(r.wrench) > 0:
            assert H.rel_err(wrench[k], r.wrench) < TOL, (k, wrench[k], r.wrench)
            n_contact += 1
        else:
            assert np.linalg.norm(wrench[k]) == 0.0
    assert n_contact > 40
    m.close()


def test_fused_equals_batched(pfc):
    for w in (pfc.configs.c1_boxes(), pfc.configs.c2_box_on_plane(40, montecarlo=True),
              pfc.configs.c3_blob_tool(5, n_div_blob=7, n_div_tool=5)):
        m1, (w1, s1, c1) = _run(pfc, w, fused=1)
        m0, (w0, s0, c0) = _run(pfc, w, fused=0)
        assert m1.last_parts() == 0 and m0.last_parts() == 1
        assert np.array_equal(c1, c0)
        np.testing.assert_allclose(w1, w0, rtol=1e-11, atol=1e-11 * max(np.abs(w0).max(), 1e-300))
        np.testing.assert_allclose(s1, s0, rtol=1e-6, atol=1e-6 * max(np.abs(s0).max(), 1e-300))
        m1.close(); m0.close()


def test_fused_item_that_does_not_fit_falls_back(pfc):
    """More candidate pairs than the kernel's LDS list holds: the evaluation is transparently re-issued on the batched
    path (pfc_eval), the fused kernel stays off for a while and comes back."""
    w = pfc.configs.c3_blob_tool(3, n_div_blob=8, n_div_tool=6, distance=0.04)
    ref = H.oracle_run(pfc, w, debug=False)
    # with a team of workgroups per item (round 3) every workgroup holds its own share of the candidates: the scene fits
    mt, (wrench, sdot, counts) = _run(pfc, w)
    assert counts[:, 1].max() > 4096, counts[:, 1]
    assert mt.last_parts() == 0 and mt.last_team() == 16
    for k, r in enumerate(ref):
        assert np.array_equal(counts[k], r.counts)
        assert H.rel_err(wrench[k], r.wrench) < TOL
    mt.close()
    # a workgroup per item (option team = 0): the list overflows, the evaluation ends on the batched path
    m = pfc.configs.build_scenario(w)
    m.set_option("team", 0)
    wrench, sdot, counts = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert m.last_parts() == 1           # ended on the batched path
    for k, r in enumerate(ref):
        assert np.array_equal(counts[k], r.counts)
        assert H.rel_err(wrench[k], r.wrench) < TOL
    # a scene that fits, same handle: batched while the back-off lasts, fused again afterwards
    w2 = pfc.configs.c3_blob_tool(3, n_div_blob=8, n_div_tool=6)
    paths = []
    for _ in range(70):
        wr2, sd2, ct2 = m.force_all_elastic_intersections(w2.pose, w2.twist, w2.s, w2.ins_ids)
        paths.append(m.last_parts())
    assert paths[0] == 1 and paths[-1] == 0
    ref2 = H.oracle_run(pfc, w2, debug=False)
    for k, r in enumerate(ref2):
        assert np.array_equal(ct2[k], r.counts) and H.rel_err(wr2[k], r.wrench) < TOL
    m.close()


def test_fused_errors_and_device_buffers(pfc):
    import torch
    L = pfc._lib
    w = pfc.configs.c2_box_on_plane(8, montecarlo=True)
    m = pfc.configs.build_scenario(w)
    bad = w.pose.copy(); bad[3, 5] = np.inf
    with pytest.raises(L.PFCError) as ei:
        m.force_all_elastic_intersections(bad, w.twist, w.s, w.ins_ids)
    assert ei.value.status == L.ERR_NONFINITE
    with pytest.raises(L.PFCError) as ei:
        m.force_all_elastic_intersections(w.pose, w.twist, w.s, np.full(8, 3, dtype=np.int32))
    assert ei.value.status == L.ERR_BAD_ARG
    wr, sd, ct = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert m.last_parts() == 0
    # pfc_eval_device + pfc_check: buffers resident in HBM, caller's stream
    dev = torch.device("cuda", 0)
    n = w.n_items
    d = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    d_ins, d_pose, d_tw, d_s = d(w.ins_ids, torch.int32), d(w.pose, torch.float64), d(w.twist, torch.float64), d(w.s, torch.float64)
    d_w = torch.zeros((n, 6), dtype=torch.float64, device=dev); d_sd = torch.zeros_like(d_w)
    d_ct = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    for counts_ptr in (d_ct.data_ptr(), 0):
        m.eval_device(n, d_ins.data_ptr(), d_pose.data_ptr(), d_tw.data_ptr(), d_s.data_ptr(), d_w.data_ptr(), d_sd.data_ptr(),
                      counts_ptr, torch.cuda.current_stream().cuda_stream)
        assert m.check() == 0 and m.last_parts() == 0
        np.testing.assert_allclose(d_w.cpu().numpy(), wr, rtol=1e-12, atol=1e-12 * np.abs(wr).max())
        assert np.array_equal(d_ct.cpu().numpy(), ct)
        assert m.stats()["candidates"] == int(ct[:, 1].sum())
    # debug views belong to the batched path
    with pytest.raises(L.PFCError):
        m.debug_stiffness(0)
    m.close()


@pytest.mark.parametrize("n_dir", [1, 6, 7, 16])
def test_fused_dual_regularized(pfc, O, n_dir):
    """Value AND Dual passes inside the fused kernel (all-regularized small scenes: what Radau's Jacobian evaluations of
    test/boxes.jl-like scenes go through) against the Dual oracle and against the batched Dual path (option fused = 0)."""
    from helpers import oracle_ins, oracle_meshes
    from test_oracle_dual import tangents
    rng = np.random.default_rng(40 + n_dir)
    for w in (pfc.configs.c1_boxes(), pfc.configs.c2_box_on_plane(6, montecarlo=True, n_div=3)):
        n = w.n_items
        dq = rng.standard_normal((n, n_dir, 6)) * np.array([1, 1, 1, 0.05, 0.05, 0.05])
        d_twist = rng.standard_normal((n, n_dir, 6)) * np.array([1, 1, 1, 0.1, 0.1, 0.1])
        d_pose = np.zeros((n, n_dir, 24))
        for k in range(n):
            d_pose[k] = tangents(w.pose[k][:9].reshape(3, 3, order="F"), w.pose[k][9:12], dq[k])
        res = []
        for fused in (1, 0):
            m = pfc.configs.build_scenario(w)
            m.set_option("fused", fused)
            for _ in range(2):
                out = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, None, w.ins_ids)
            res.append((out, m.last_parts()))
            m.close()
        (a, pa), (b, pb) = res
        assert pb != 0
        if n_dir <= 7:
            assert pa == 0, pa                                  # the fused kernel really ran the first variant
        assert np.array_equal(a[4], b[4])
        np.testing.assert_allclose(a[0], b[0], rtol=1e-11, atol=1e-11 * np.abs(b[0]).max())
        np.testing.assert_allclose(a[2], b[2], rtol=1e-9, atol=1e-9 * np.abs(b[2]).max())
        assert np.all(a[3] == 0.0) and np.all(a[1] == 0.0)
        om = oracle_meshes(w)
        for k in range(n):
            c = w.instructions[int(w.ins_ids[k])]
            st, rw, rs, rdw, rdsd = O.evaluate_dual(om[c.id_1], om[c.id_2], oracle_ins(pfc, c), w.pose[k], w.twist[k], w.s[k],
                                                    d_pose[k], d_twist[k], np.zeros((n_dir, 6)))
            assert st == 0
            sw = max(np.abs(rdw).max(), 1e-300)
            assert np.abs(a[2][k] - rdw).max() <= 1e-6 * sw, (k, np.abs(a[2][k] - rdw).max() / sw)


def test_fused_dual_falls_back_when_an_item_has_too_many_polygons(pfc):
    w = pfc.configs.c2_box_on_plane(3, montecarlo=True)          # ~220 polygons x 6 directions per item: past the in-kernel limit
    n, nd = w.n_items, 6
    rng = np.random.default_rng(3)
    d_pose = rng.standard_normal((n, nd, 24)) * 1e-2
    d_twist = rng.standard_normal((n, nd, 6)) * 0.1
    m = pfc.configs.build_scenario(w)
    out = [m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, None, w.ins_ids) for _ in range(3)]
    m.close()
    f = pfc.configs.build_scenario(w)
    f.set_option("fused", 0)
    ref = f.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, None, w.ins_ids)
    f.close()
    for o in out:
        assert np.array_equal(o[4], ref[4])
        np.testing.assert_allclose(o[2], ref[2], rtol=1e-9, atol=1e-9 * np.abs(ref[2]).max())


@pytest.mark.parametrize("model", ["regularized", "bristle"])
def test_fused_tet_tet(pfc, model):
    """Volume-volume items (equal-pressure plane, clip_plane_tet, zero_small_coordinates, quad clip) in the fused kernel."""
    w = pfc.configs.vol_vol(8, n_div=5, model=model)
    m, (wrench, sdot, counts) = _run(pfc, w)
    assert m.last_parts() == 0
    ref = H.oracle_run(pfc, w, debug=False)
    assert sum(int(r.counts[3] > 0) for r in ref) >= 10
    for k, r in enumerate(ref):
        assert np.array_equal(counts[k], r.counts), (k, counts[k], r.counts)
        if np.linalg.norm(r.wrench) > 0:
            assert H.rel_err(wrench[k], r.wrench) < 1e-8, (k, wrench[k], r.wrench)
        else:
            assert np.linalg.norm(wrench[k]) == 0.0
    m.close()


@pytest.mark.parametrize("degenerate", [False, True])
def test_fused_fuzz_with_tet_tet(pfc, degenerate):
    from test_gpu_parity import _fuzz_workload
    rng = np.random.default_rng(291 + int(degenerate))
    w = _fuzz_workload(pfc, rng, 256, degenerate, tet_tet=True)
    m, (wrench, sdot, counts) = _run(pfc, w)
    # deeply overlapping tet meshes can exceed the kernel's 4 096-candidate list: then the call ends on the batched path
    assert m.last_parts() == (0 if counts[:, 1].max() <= 4096 else 1)
    ref = H.oracle_run(pfc, w, debug=False)
    n_contact = 0
    for k, r in enumerate(ref):
        assert np.array_equal(counts[k], r.counts), (k, int(w.ins_ids[k]), counts[k], r.counts)
        if np.linalg.norm(r.wrench) > 0:
            assert H.rel_err(wrench[k], r.wrench) < TOL, (k, wrench[k], r.wrench)
            n_contact += 1
        else:
            assert np.linalg.norm(wrench[k]) == 0.0
    assert n_contact > 40
    m.close()


@pytest.mark.parametrize("n_poses, model", [(1, "bristle"), (4, "bristle"), (8, "bristle"), (3, "regularized")])
def test_team_of_workgroups_for_big_pairs(pfc, n_poses, model):
    """BASELINE config 3 as written -- ONE 9 680-tet blob against ONE 5 120-triangle tool -- is too big for a lone
    workgroup (110 broadphase iterations; its candidates overflow the LDS list): a TEAM of workgroups per item takes it in
    one launch (k_fused<.., true>: top of the descent redundantly, the stack shared out by rank, candidates clipped where
    found, three team sums for the bristle passes).  Full-size meshes, counts bit-equal to the oracle's, wrench 1e-9; the
    batched path (option team = 0) gives the same integers."""
    w = pfc.configs.c3_blob_tool(n_poses)
    assert w.meta["n_tet"] == 9680 and w.meta["n_tri"] == 5120
    if model == "regularized":
        w.instructions[0].model = "regularized"
    m, (wrench, sdot, counts) = _run(pfc, w)
    assert m.last_parts() == 0 and m.last_team() == min(48, 256 // n_poses), "the team kernel did not run"
    _against_oracle(pfc, w, wrench, sdot, counts)
    st = m.stats()
    assert st["candidates"] == int(counts[:, 1].sum()) and st["node_tests"] == int(counts[:, 0].sum())
    # repeated evaluations on the same handle (the arrival counters only grow), other poses in between
    w2 = pfc.configs.c3_blob_tool(n_poses, seed=5)
    if model == "regularized":
        w2.instructions[0].model = "regularized"
    for _ in range(3):
        a = m.force_all_elastic_intersections(w2.pose, w2.twist, w2.s, w2.ins_ids)
        b = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        assert m.last_parts() == 0
        assert np.array_equal(b[2], counts)
        np.testing.assert_allclose(b[0], wrench, rtol=1e-11, atol=1e-11 * np.abs(wrench).max())
    _against_oracle(pfc, w2, *a)
    m.set_option("team", 0)
    c = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert m.last_parts() == 1
    assert np.array_equal(c[2], counts)
    np.testing.assert_allclose(c[0], wrench, rtol=1e-10, atol=1e-10 * np.abs(wrench).max())
    m.close()


@pytest.mark.parametrize("model", ["regularized", "bristle"])
def test_small_teams_on_tet_tet_and_mixed_items(pfc, model):
    """Teams of the one-launch kernel on volume-volume (tet-tet) items: compliant box on the compliant half-plane tet and two
    compliant spheres of 1 280 tets each (k_fused<true, true>: the plane / tet polygon instead of the triangle, the
    trivial reject behind it); items of very different size in one launch (the box items finish before they are shared
    out: everything stays with rank 0)."""
    w = pfc.configs.vol_vol(3, n_div=8, model=model)
    m, (wrench, sdot, counts) = _run(pfc, w)
    assert m.last_parts() == 0 and m.last_team() == 20       # 2 560 leaves: a workgroup per 128
    _against_oracle(pfc, w, wrench, sdot, counts)
    assert np.count_nonzero(counts[:, 3]) >= 4
    m.set_option("team", 0)
    w1, s1, c1 = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert m.last_team() == 1 and np.array_equal(c1, counts)
    np.testing.assert_allclose(w1, wrench, rtol=1e-10, atol=1e-10 * np.abs(wrench).max())
    m.close()
