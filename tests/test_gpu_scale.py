"""GPU tests at BASELINE.json's full sizes: oracle parity on the multi-instruction configs (C4, C5) and
size-independent properties of the HIP path on full C3 batches (no oracle needed)."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _check_vs_oracle(pfc, w, m, wrench, sdot, counts, pair_items=(), tol=TOL, oracle_debug=False):
    ref = H.oracle_run(pfc, w, debug=bool(pair_items) or oracle_debug)      # debug: the oracle also returns K (the ṡ tolerance rule below)
    for k, r in enumerate(ref):
        assert np.array_equal(counts[k], r.counts), (k, counts[k], r.counts)
        for name, a, b in (("wrench", wrench[k], r.wrench), ("sdot", sdot[k], r.sdot)):
            if np.linalg.norm(b) == 0.0:
                assert np.linalg.norm(a) == 0.0, (name, k)
            else:
                t = tol
                if name == "sdot" and r.has_K:
                    # ṡ = -(K̄^{-1/2} S⁻¹ w + s)/τ.  decompose_K! clamps eigenvalues at 1e-16 sigma_max (friction.jl:92),
                    # i.e. it amplifies by up to 1e8.  A flat patch has ONE exactly-null direction (harmless: w has no
                    # component along it beyond rounding).  A sliver / edge contact has further eigenvalues at the
                    # rounding level of K itself; there ṡ depends on the last bits of K in the reference as much as here.
                    Kb = np.diag(r.Sinv) @ r.K @ np.diag(r.Sinv)
                    ev = np.linalg.eigvalsh((Kb + Kb.T) / 2)
                    if np.sum(ev < 1e-12 * ev[-1]) >= 2:
                        t = max(tol, 1e-3)
                assert H.rel_err(a, b) < t, (name, k, a, b)
    for k in pair_items:
        gp, gc = H.sorted_pairs(*m.debug_pairs(k))
        rp, rc = H.sorted_pairs(ref[k].pairs, ref[k].clip_n)
        assert np.array_equal(gp, rp) and np.array_equal(gc, rc), k
    return ref


def test_c4_256_scenes(pfc):
    """BASELINE C4: 256 Monte-Carlo box-on-plane scenes (972-tet box, 2-triangle ground), regularized."""
    w = pfc.configs.c2_box_on_plane(256, montecarlo=True)
    m = pfc.configs.build_scenario(w, debug=True)
    wrench, sdot, counts = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    _check_vs_oracle(pfc, w, m, wrench, sdot, counts, pair_items=(0, 17, 255))
    assert np.all(counts[:, 3] > 0)
    m.close()


def test_c5_pile_all_pairs(pfc):
    """BASELINE C5: 64 boxes (108 .. 2352 tets), all 2016 unordered pairs as bristle instructions."""
    w = pfc.configs.c5_pile()
    assert w.n_items == 2016
    m = pfc.configs.build_scenario(w, debug=True)
    wrench, sdot, counts = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    touching = np.nonzero(counts[:, 3] > 0)[0]
    assert 50 < touching.size < 400          # lattice neighbours only
    # Flat face-to-face patches: K has an exact null direction (no tangential stiffness along n̂), decompose_K! clamps
    # the eigenvalue at 1e-16 sigma_max (friction.jl:92) and K̄^{-1/2} amplifies rounding noise along it by 1e8, in the
    # reference as much as here: ṡ is only reproducible to ~1e-8 relative.  The north_star tolerance (1e-6) is asserted.
    _check_vs_oracle(pfc, w, m, wrench, sdot, counts, pair_items=tuple(touching[:3]) + (0,), tol=1e-6)
    m.close()


def test_c5_pile_mode_and_wide_broadphase_workgroups(pfc):
    """Default options (no debug lists): the first C5 evaluation of a handle runs as two halves; it finds at most a quarter of
    the 2 016 items in contact, so the next evaluations of that shape run as ONE launch sequence with 512-thread broadphase
    workgroups (pile_mode, pfc_hip.hip).  All of them against the oracle, and equal among themselves in every integer.  A batch
    of 24 full-size C3 poses takes the 512-thread workgroups from its first evaluation (big trees, fewer than 1 024 items)."""
    w = pfc.configs.c5_pile()
    m = pfc.configs.build_scenario(w)
    first = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert m.last_parts() == 2
    _check_vs_oracle(pfc, w, m, *first, tol=1e-6, oracle_debug=True)
    for _ in range(2):
        again = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        assert m.last_parts() == 1          # pile mode
        assert np.array_equal(again[2], first[2])
        np.testing.assert_allclose(again[0], first[0], rtol=1e-9, atol=1e-9 * np.abs(first[0]).max())
    # a dense evaluation of the same size ends the pile mode: every item in contact (the touching pairs repeated)
    touching = np.nonzero(first[2][:, 3] > 0)[0]
    idx = np.resize(touching, w.n_items)
    dense = m.force_all_elastic_intersections(w.pose[idx], w.twist[idx], w.s[idx], w.ins_ids[idx])
    assert m.last_parts() == 1              # laid out from the previous evaluation's picture
    dense2 = m.force_all_elastic_intersections(w.pose[idx], w.twist[idx], w.s[idx], w.ins_ids[idx])
    assert m.last_parts() == 2              # ... and from its own
    assert np.array_equal(dense[2], dense2[2]) and np.array_equal(dense[2], first[2][idx])
    # a host that alternates batch sizes keeps a picture of each (the last four shapes)
    sub = np.arange(1500)
    for k in range(3):
        part = m.force_all_elastic_intersections(w.pose[sub], w.twist[sub], w.s[sub], w.ins_ids[sub])
        assert m.last_parts() == (2 if k == 0 else 1) and np.array_equal(part[2], first[2][sub])
        full = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        assert m.last_parts() == (2 if k == 0 else 1) and np.array_equal(full[2], first[2])
    m.close()
    w3 = pfc.configs.c3_blob_tool(24, seed=77)
    m3 = pfc.configs.build_scenario(w3)
    m3.set_option("fused", 0)
    out = m3.force_all_elastic_intersections(w3.pose, w3.twist, w3.s, w3.ins_ids)
    assert m3.last_parts() == 1
    _check_vs_oracle(pfc, w3, m3, *out, oracle_debug=True)
    m3.close()


def test_c3_batch_properties(pfc):
    """Full C3 meshes, 96 poses: (1) an item evaluated alone equals the item inside the batch, (2) permuting the
    batch permutes the results, (3) integer outputs are reproducible run to run, (4) doubling Ē doubles the
    normal wrench of a regularized evaluation exactly up to summation order."""
    w = pfc.configs.c3_blob_tool(96)
    m = pfc.configs.build_scenario(w)
    wr, sd, ct = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    wr2, sd2, ct2 = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert np.array_equal(ct, ct2)
    np.testing.assert_allclose(wr2, wr, rtol=1e-11, atol=1e-11 * np.abs(wr).max())
    for k in (0, 41, 95):
        w1, s1, c1 = m.force_all_elastic_intersections(w.pose[k:k + 1], w.twist[k:k + 1], w.s[k:k + 1], w.ins_ids[k:k + 1])
        assert np.array_equal(c1[0], ct[k])
        np.testing.assert_allclose(w1[0], wr[k], rtol=1e-11, atol=1e-11 * np.abs(wr[k]).max())
        np.testing.assert_allclose(s1[0], sd[k], rtol=1e-10, atol=1e-10 * np.abs(sd[k]).max())
    perm = np.random.default_rng(0).permutation(w.n_items)
    wp, sp, cp = m.force_all_elastic_intersections(w.pose[perm], w.twist[perm], w.s[perm], w.ins_ids[perm])
    assert np.array_equal(cp, ct[perm])
    np.testing.assert_allclose(wp, wr[perm], rtol=1e-11, atol=1e-11 * np.abs(wr).max())
    m.close()
    # Ē linearity (regularized: p = ϵ Ē damp, non_friction.jl:262)
    out = []
    for Ebar in (1.0e6, 2.0e6):
        wE = pfc.configs.c3_blob_tool(8)
        wE.instructions[0].model = "regularized"
        wE.meshes[1].Ebar = Ebar
        mE = pfc.configs.build_scenario(wE)
        out.append(mE.force_all_elastic_intersections(wE.pose, wE.twist, wE.s, wE.ins_ids))
        mE.close()
    assert np.array_equal(out[0][2], out[1][2])
    np.testing.assert_allclose(out[1][0], 2.0 * out[0][0], rtol=1e-12, atol=1e-12 * np.abs(out[1][0]).max())


def test_rigid_motion_invariance(pfc):
    """Moving both bodies by the same rigid transform leaves the relative pose unchanged (host helper) and hence
    every output bit-identical in counts; expressed through relative_pose this checks the host/device contract."""
    S, Cf = pfc.scenario, pfc.configs
    w = Cf.c3_blob_tool(4, n_div_blob=10, n_div_tool=8)
    m = Cf.build_scenario(w)
    wr, sd, ct = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    rng = np.random.default_rng(3)
    Q, d = Cf.random_rotation(rng), rng.standard_normal(3)
    pose2 = []
    for k in range(w.n_items):
        R21 = w.pose[k, :9].reshape(3, 3, order="F"); t21 = w.pose[k, 9:12]
        # body 2 at (Q, d) in the world, body 1 = body 2 * x_r2_r1
        pose2.append(S.relative_pose(Q @ R21, Q @ t21 + d, Q, d))
    wr2, sd2, ct2 = m.force_all_elastic_intersections(np.array(pose2), w.twist, w.s, w.ins_ids)
    assert np.array_equal(ct2[:, 1:], ct[:, 1:]) or np.abs(ct2 - ct).max() <= 2      # poses differ by rounding only
    np.testing.assert_allclose(wr2, wr, rtol=1e-8, atol=1e-8 * np.abs(wr).max())
    m.close()


@pytest.mark.parametrize("clip_min", [0, 1])
def test_overflow_growth_is_transparent(pfc, clip_min):
    """Work lists start small and double on overflow (VectorCache semantics); results must not depend on it -- in the
    one-kernel narrowphase and in the clip-only kernel + k_integ form (whose moment records overflow on their own)."""
    w = pfc.configs.c3_blob_tool(300, n_div_blob=8, n_div_tool=6)
    m = pfc.configs.build_scenario(w)                      # fresh handle: minimal capacities, must grow
    m.set_option("clip_min", clip_min)
    wr, sd, ct = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    st = m.stats()
    assert st["candidates"] > 65536 or st["n_items"] == 300
    wr2, sd2, ct2 = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert np.array_equal(ct, ct2)
    np.testing.assert_allclose(wr2, wr, rtol=1e-11, atol=1e-11 * np.abs(wr).max())
    ref = H.oracle_run(pfc, w, items=[0, 150, 299], debug=False)
    for k, r in zip([0, 150, 299], ref):
        assert np.array_equal(ct[k], r.counts)
        assert H.rel_err(wr[k], r.wrench) < TOL
    m.close()


def test_error_paths(pfc):
    """C-ABI error behaviour: NaN pose -> PFC_ERR_NONFINITE ("Non-finite vertex likely"), bad ins id -> BAD_ARG,
    bristle without state -> BAD_ARG, inverted tet -> PFC_ERR_INVERTED_TET."""
    L = pfc._lib
    w = pfc.configs.c3_blob_tool(2, n_div_blob=4, n_div_tool=3)
    m = pfc.configs.build_scenario(w)
    bad = w.pose.copy(); bad[1, 3] = np.nan
    with pytest.raises(L.PFCError) as ei:
        m.force_all_elastic_intersections(bad, w.twist, w.s, w.ins_ids)
    assert ei.value.status == L.ERR_NONFINITE
    with pytest.raises(L.PFCError) as ei:
        m.force_all_elastic_intersections(w.pose, w.twist, w.s, np.array([0, 7], dtype=np.int32))
    assert ei.value.status == L.ERR_BAD_ARG
    with pytest.raises(L.PFCError) as ei:
        m.force_all_elastic_intersections(w.pose, w.twist, None, w.ins_ids)
    assert ei.value.status == L.ERR_BAD_ARG
    wr, _, _ = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)    # handle still usable
    assert np.all(np.isfinite(wr))
    m.close()
    G = pfc.geometry
    box = G.as_tet_emesh(G.emesh_box(0.05))
    tree = G.build_tree(box)
    flipped = G.EMesh.__new__(G.EMesh)
    flipped.point, flipped.tri, flipped.eps = box.point, None, box.eps
    flipped.tet = box.tet[:, [1, 0, 2, 3]].copy()
    m2 = pfc.MechanismScenario()
    m2.add_contact("flipped", flipped, c_prop=pfc.ContactProperties(1.0e6), tree=tree)
    m2.add_contact("tri", G.as_tri_emesh(G.emesh_box(0.05)))
    m2.add_friction_regularize(0, 1, mu_d=0.3)
    with pytest.raises(L.PFCError) as ei:
        m2.finalize()
    assert ei.value.status == L.ERR_INVERTED_TET


def test_scatter_generalized_third_law(pfc):
    """addGeneralizedForcesThirdLaw! (non_friction.jl:267-286) on the device, 256 scenes x (1 free box
    against the rooted ground): wrench -> world frame -> J' w, against the oracle's restatement."""
    from oracle import oracle as O
    w = pfc.configs.c2_box_on_plane(64, montecarlo=True)
    m = pfc.configs.build_scenario(w)
    wrench, _, _ = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    rng = np.random.default_rng(12)
    n, nv, n_body = w.n_items, 6, 2
    x_w_r2 = np.zeros((n, 12))
    for k in range(n):
        R = pfc.configs.random_rotation(rng)
        x_w_r2[k, :9] = R.reshape(-1, order="F"); x_w_r2[k, 9:] = rng.standard_normal(3)
    jac = rng.standard_normal((n_body, nv, 6))
    body_1 = np.full(n, -1, dtype=np.int32)                      # ground: root body, no Jacobian
    body_2 = np.where(np.arange(n) % 3 == 0, 1, 0).astype(np.int32)
    scene = (np.arange(n) // 4).astype(np.int32)
    f = m.scatter_generalized(wrench, x_w_r2, body_1, body_2, jac, scene, n_scene=16)
    ref = O.scatter_generalized(wrench, x_w_r2, body_1, body_2, jac, scene, n_scene=16)
    np.testing.assert_allclose(f, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
    # both bodies movable: third law (+w on body 2, -w on body 1)
    body_1b = np.ones(n, dtype=np.int32)
    f2 = m.scatter_generalized(wrench, x_w_r2, body_1b, np.zeros(n, dtype=np.int32), jac, None, 1)
    ref2 = O.scatter_generalized(wrench, x_w_r2, body_1b, np.zeros(n, dtype=np.int32), jac, None, 1)
    np.testing.assert_allclose(f2, ref2, rtol=1e-11, atol=1e-11 * np.abs(ref2).max())
    m.close()


def test_scatter_generalized_device_resident(pfc):
    """pfc_scatter_generalized_device: the wrenches pfc_eval_device left in HBM are projected on the Jacobians and added to a
    device-resident f_generalized, all on one stream and without a host synchronisation in between ("keep
    f_generalized resident"); overwrite and accumulate modes against the host-buffer entry point."""
    import torch
    w = pfc.configs.c2_box_on_plane(64, montecarlo=True)
    m = pfc.configs.build_scenario(w)
    rng = np.random.default_rng(14)
    n, nv, n_body, n_scene = w.n_items, 6, 2, 16
    x_w_r2 = np.zeros((n, 12))
    for k in range(n):
        R = pfc.configs.random_rotation(rng)
        x_w_r2[k, :9] = R.reshape(-1, order="F"); x_w_r2[k, 9:] = rng.standard_normal(3)
    jac = rng.standard_normal((n_body, nv, 6))
    body_1 = np.full(n, -1, dtype=np.int32)
    body_2 = np.where(np.arange(n) % 3 == 0, 1, 0).astype(np.int32)
    scene = (np.arange(n) // 4).astype(np.int32)
    wrench, _, _ = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    want = m.scatter_generalized(wrench, x_w_r2, body_1, body_2, jac, scene, n_scene=n_scene)
    dev = torch.device("cuda", 0)
    T = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    t_ins, t_pose, t_tw, t_s = T(w.ins_ids, torch.int32), T(w.pose), T(w.twist), T(w.s)
    o_w, o_sd = torch.zeros((n, 6), dtype=torch.float64, device=dev), torch.zeros((n, 6), dtype=torch.float64, device=dev)
    o_ct = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    t_x, t_b1, t_b2, t_sc, t_j = T(x_w_r2), T(body_1, torch.int32), T(body_2, torch.int32), T(scene, torch.int32), T(jac)
    f = torch.full((n_scene, nv), 7.0, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    m.eval_device(n, t_ins.data_ptr(), t_pose.data_ptr(), t_tw.data_ptr(), t_s.data_ptr(), o_w.data_ptr(), o_sd.data_ptr(),
                  o_ct.data_ptr(), st)
    m.scatter_generalized_device(n, o_w.data_ptr(), t_x.data_ptr(), t_b1.data_ptr(), t_b2.data_ptr(), t_sc.data_ptr(), n_scene, nv,
                                 t_j.data_ptr(), f.data_ptr(), False, st)          # same stream: ordered behind the evaluation
    assert m.check() == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(f.cpu().numpy(), want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
    m.scatter_generalized_device(n, o_w.data_ptr(), t_x.data_ptr(), t_b1.data_ptr(), t_b2.data_ptr(), 0, 1, nv, t_j.data_ptr(),
                                 f.data_ptr(), True, st)                           # accumulate, one mechanism: row 0 only
    torch.cuda.synchronize()
    one = m.scatter_generalized(wrench, x_w_r2, body_1, body_2, jac, None, 1)
    got = f.cpu().numpy()
    np.testing.assert_allclose(got[0], want[0] + one[0], rtol=1e-12, atol=1e-12 * np.abs(one).max())
    np.testing.assert_allclose(got[1:], want[1:], rtol=1e-12, atol=1e-12 * np.abs(want).max())
    m.close()


def test_split_evaluation_equals_unsplit(pfc):
    """Batches of >= split_min items run as two concurrent halves on two streams (pfc_set_option "split_min"): same
    per-item integers, same wrenches up to summation order, merged totals, and growth of the work lists of BOTH halves
    from their small initial capacities (the first evaluations overflow and are re-issued inside pfc_eval)."""
    w = pfc.configs.c3_blob_tool(1100, seed=31, n_div_blob=8, n_div_tool=6)
    w.s[:] = np.random.default_rng(5).standard_normal((w.n_items, 6)) * 1e-3
    out = {}
    for split_min in (0, 1024):
        m = pfc.configs.build_scenario(w)
        m.set_option("split_min", split_min)
        wr, sd, ct = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        out[split_min] = (wr, sd, ct, m.stats(), m.last_parts())
        # a second call reuses the grown buffers and the captured graphs of both halves
        wr2, sd2, ct2 = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        assert np.array_equal(ct2, ct)
        np.testing.assert_allclose(wr2, wr, rtol=1e-11, atol=1e-11 * np.abs(wr).max())
        m.close()
    a, b = out[0], out[1024]
    assert a[4] == 1 and b[4] == 2
    assert np.array_equal(a[2], b[2])
    np.testing.assert_allclose(b[0], a[0], rtol=1e-11, atol=1e-11 * np.abs(a[0]).max())
    np.testing.assert_allclose(b[1], a[1], rtol=1e-7, atol=1e-7 * np.abs(a[1]).max())
    for key in ("node_tests", "candidates", "nonempty", "tractions", "n_items"):
        assert a[3][key] == b[3][key], key
    assert np.all(a[2][:, 3] > 0)


def test_two_half_evaluation_on_a_caller_stream(pfc):
    """pfc_eval_device on a stream the CALLER created (a torch pool stream): the two halves run on the library's own two
    streams between a fork and a join on the caller's stream, so work the caller enqueues behind the call sees the results,
    and the results equal those of the library's own stream.  (The first half used to run on the caller's stream, which may
    share a hardware queue with the second half's: no overlap, 4.1 -> 5.1 ms for 8 192 poses.)"""
    import torch
    w = pfc.configs.c3_blob_tool(1100, seed=33, n_div_blob=8, n_div_tool=6)
    dev = torch.device("cuda:0")
    m = pfc.configs.build_scenario(w)
    n = w.n_items
    d = dict(ins=torch.from_numpy(w.ins_ids.astype(np.int32)).to(dev), pose=torch.from_numpy(w.pose).to(dev),
             twist=torch.from_numpy(w.twist).to(dev), s=torch.from_numpy(w.s).to(dev))
    res = {}
    for name, st in (("own", None), ("caller", torch.cuda.Stream())):
        wr = torch.zeros((n, 6), dtype=torch.float64, device=dev)
        sd = torch.zeros((n, 6), dtype=torch.float64, device=dev)
        ct = torch.zeros((n, 4), dtype=torch.int32, device=dev)
        total = torch.zeros(6, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        for _ in range(40):
            m.eval_device(n, d["ins"].data_ptr(), d["pose"].data_ptr(), d["twist"].data_ptr(), d["s"].data_ptr(),
                          wr.data_ptr(), sd.data_ptr(), ct.data_ptr(), st.cuda_stream if st else 0)
            if st is not None:
                with torch.cuda.stream(st):      # enqueued behind the call, before any synchronisation
                    total = wr.sum(dim=0)
            if m.check() == 0:
                break
        else:
            raise AssertionError("work lists kept overflowing")
        assert m.last_parts() == 2
        torch.cuda.synchronize()
        res[name] = (wr.cpu().numpy(), sd.cpu().numpy(), ct.cpu().numpy(), total.cpu().numpy())
    a, b = res["own"], res["caller"]
    assert np.array_equal(a[2], b[2])
    np.testing.assert_allclose(b[0], a[0], rtol=1e-11, atol=1e-11 * np.abs(a[0]).max())
    np.testing.assert_allclose(b[1], a[1], rtol=1e-7, atol=1e-7 * np.abs(a[1]).max())
    np.testing.assert_allclose(b[3], b[0].sum(axis=0), rtol=1e-9, atol=1e-9 * np.abs(b[0]).max())
    m.close()


def test_handles_are_built_and_used_from_concurrent_host_threads(pfc):
    """include/pfc.h: different handles may be used from different host threads.  Every thread builds its own scenario
    (uploads in pfc_finalize) and evaluates new batch shapes (each records a graph) while the others do the same.  The
    uploads used to go through hipMemcpy, i.e. the legacy stream, which the runtime refuses while any thread is capturing
    ("operation would make the legacy stream depend on a capturing blocking stream")."""
    import threading
    w = pfc.configs.c3_blob_tool(96, seed=21, n_div_blob=5, n_div_tool=4)
    ref_m = pfc.configs.build_scenario(w)
    ref_m.set_option("fused", 0)
    sizes = (96, 7, 33, 64, 1)
    ref = {s: ref_m.force_all_elastic_intersections(w.pose[:s], w.twist[:s], w.s[:s], w.ins_ids[:s]) for s in sizes}
    ref_m.close()
    errors = []

    def work(t):
        try:
            for rnd in range(3):
                m = pfc.configs.build_scenario(w)
                m.set_option("fused", 0)       # the batched path: every new shape records a graph
                for s in sizes[t % len(sizes):] + sizes[:t % len(sizes)]:
                    wr, sd, ct = m.force_all_elastic_intersections(w.pose[:s], w.twist[:s], w.s[:s], w.ins_ids[:s])
                    assert np.array_equal(ct, ref[s][2])
                    np.testing.assert_allclose(wr, ref[s][0], rtol=1e-10, atol=1e-10 * np.abs(ref[s][0]).max())
                m.close()
        except Exception as e:
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_alternating_batch_shapes_on_one_handle(pfc):
    """One handle, evaluations of very different sizes back to back (1, 700, 3, 1500 split, 64 items): the append lists
    (candidates, kept polygons and moment records in their 64 regions, contributing pairs) and their counters must
    start every evaluation empty whatever the previous one left behind, on the graph path and across the growth of
    the buffers.  Every evaluation is checked against a fresh handle that only ever saw that one batch."""
    w = pfc.configs.c3_blob_tool(1500, seed=77, n_div_blob=8, n_div_tool=6)
    w.s[:] = np.random.default_rng(9).standard_normal((w.n_items, 6)) * 1e-3
    m = pfc.configs.build_scenario(w)
    for rep in range(2):
        for lo, hi in ((0, 1), (100, 800), (5, 8), (0, 1500), (900, 964)):
            sl = slice(lo, hi)
            wr, sd, ct = m.force_all_elastic_intersections(w.pose[sl], w.twist[sl], w.s[sl], w.ins_ids[sl])
            f = pfc.configs.build_scenario(w)
            wr0, sd0, ct0 = f.force_all_elastic_intersections(w.pose[sl], w.twist[sl], w.s[sl], w.ins_ids[sl])
            f.close()
            assert np.array_equal(ct, ct0), (rep, lo, hi)
            np.testing.assert_allclose(wr, wr0, rtol=1e-11, atol=1e-11 * max(np.abs(wr0).max(), 1e-300))
            np.testing.assert_allclose(sd, sd0, rtol=1e-7, atol=1e-7 * max(np.abs(sd0).max(), 1e-300))
    m.close()


def test_overflowing_lists_never_expose_unwritten_slots(pfc):
    """Option "poison" fills the work lists with entries of item index -1 before every evaluation.  A first evaluation
    that overflows the candidate list AND the seed frontier several times (buffers grow from their small initial
    sizes; every overflowing attempt still runs its narrowphase over the truncated lists) must neither report a slot
    that was read before it was written (PFC_ERR_STATE) nor change a result.  Regression test for appends that skipped
    their whole run on overflow and left the slots below the capacity unwritten (read as wild indices: a GPU fault
    once the allocator recycles memory)."""
    w = pfc.configs.c3_blob_tool(1100, seed=13, n_div_blob=6, n_div_tool=4)
    w.s[:] = np.random.default_rng(3).standard_normal((w.n_items, 6)) * 1e-3
    out = []
    for poison in (0, 1):
        m = pfc.configs.build_scenario(w)
        m.set_option("split_min", 0)
        m.set_option("bfs_levels", 3)        # 1100 x 4^3 seeds > 65536: the frontier overflows as well
        m.set_option("poison", poison)
        wr, sd, ct = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        wr2, sd2, ct2 = m.force_all_elastic_intersections(w.pose[:300], w.twist[:300], w.s[:300], w.ins_ids[:300])
        out.append((wr, ct, wr2, ct2, m.stats()))
        m.close()
    a, b = out
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
    np.testing.assert_allclose(b[0], a[0], rtol=1e-11, atol=1e-11 * np.abs(a[0]).max())
    np.testing.assert_allclose(b[2], a[2], rtol=1e-11, atol=1e-11 * np.abs(a[2]).max())
    assert a[4]["candidates"] > 65536 // 2


def test_soak_mixed_value_and_dual_on_one_handle(pfc):
    """A short version of scripts/soak.py: one long-lived handle, 40 value / Dual evaluations of random sizes around the
    path thresholds (zero-copy <= 512 items, one-graph Dual <= 512 items and 4096 pairs, staged above), each against a
    fresh handle.  Guards the lifetime of everything a replayed graph holds (pinned blocks, Dual buffers, work lists)."""
    rng = np.random.default_rng(2027)
    w = pfc.configs.c3_blob_tool(700, seed=9, n_div_blob=6, n_div_tool=4)
    w.s[:] = rng.standard_normal((w.n_items, 6)) * 1e-3
    nd = 6
    d_pose = rng.standard_normal((w.n_items, nd, 24)) * 1e-2
    d_twist = rng.standard_normal((w.n_items, nd, 6)) * 0.1
    d_s = rng.standard_normal((w.n_items, nd, 6)) * 1e-3
    m = pfc.configs.build_scenario(w)
    for it in range(40):
        size = int(rng.choice([1, 5, 64, 85, 86, 200, 512, 513, 700]))
        lo = int(rng.integers(0, w.n_items - size + 1))
        sl = slice(lo, lo + size)
        f = pfc.configs.build_scenario(w)
        if rng.random() < 0.5:
            args = (w.pose[sl], w.twist[sl], w.s[sl], d_pose[sl], d_twist[sl], d_s[sl], w.ins_ids[sl])
            got, ref = m.force_all_elastic_intersections_dual(*args), f.force_all_elastic_intersections_dual(*args)
            cmp = ((0, 1e-11), (1, 1e-7), (2, 1e-9), (3, 1e-6))
            assert np.array_equal(got[4], ref[4]), (it, size)
        else:
            args = (w.pose[sl], w.twist[sl], w.s[sl], w.ins_ids[sl])
            got, ref = m.force_all_elastic_intersections(*args), f.force_all_elastic_intersections(*args)
            cmp = ((0, 1e-11), (1, 1e-7))
            assert np.array_equal(got[2], ref[2]), (it, size)
        f.close()
        for k, tol in cmp:
            np.testing.assert_allclose(got[k], ref[k], rtol=tol, atol=tol * max(np.abs(ref[k]).max(), 1e-300), err_msg=f"{it} {size} {k}")
    m.close()


def test_max_levels_option_is_clamped(pfc):
    """Option "max_levels": every counter / tail buffer is sized from the finalized tree depth, so a larger value is
    refused (PFC_ERR_BAD_ARG) and a smaller one only caps the seed-expansion levels; results never change."""
    L = pfc._lib
    w = pfc.configs.c3_blob_tool(40, n_div_blob=6, n_div_tool=5)
    m = pfc.configs.build_scenario(w)
    depth_sum = m.MeshCache[0].tree.depth() + m.MeshCache[1].tree.depth()
    wr0, sd0, ct0 = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    for bad in (depth_sum + 2, 10 ** 6, -1):
        with pytest.raises(L.PFCError) as ei:
            m.set_option("max_levels", bad)
        assert ei.value.status == L.ERR_BAD_ARG
    for lv in (1, 2, depth_sum + 1, 0):
        m.set_option("max_levels", lv)
        for bfs in (-1, 3):
            m.set_option("bfs_levels", bfs)
            wr, sd, ct = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
            assert np.array_equal(ct, ct0), (lv, bfs)
            np.testing.assert_allclose(wr, wr0, rtol=1e-11, atol=1e-11 * np.abs(wr0).max())
    m.close()
    # set before finalize: nothing to validate against yet, the effective value is clamped at evaluation time
    m2 = pfc.MechanismScenario()
    for ms in w.meshes:
        m2.add_contact(ms.name, ms.mesh, c_prop=None if ms.Ebar is None else pfc.ContactProperties(ms.Ebar), tree=ms.tree)
    m2.add_friction_bristle(0, 1, mu_d=0.3)
    m2.finalize()
    m2.close()


def test_evaluate_sharded_with_the_hip_evaluator(pfc):
    """parallel.evaluate_sharded (the product sharding + exchange code, world = 1 here; 2 ranks over gloo in
    tests/test_host.py) with MechanismScenario.force_all_elastic_intersections as the evaluator, block and cost-weighted
    shards, against the oracle."""
    P = pfc.parallel
    w = pfc.configs.c2_box_on_plane(48, montecarlo=True)
    m = pfc.configs.build_scenario(w)

    def evaluator(idx):
        return m.force_all_elastic_intersections(w.pose[idx], w.twist[idx], w.s[idx], w.ins_ids[idx])

    ref = H.oracle_run(pfc, w, debug=False)
    cost = np.arange(w.n_items, dtype=np.float64) % 7 + 1.0
    for parts in (P.shard_block(w.n_items, 1), P.shard_by_cost(cost, 1), None):
        wr, sd, ct = P.evaluate_sharded(evaluator, w.n_items, parts)
        for k, r in enumerate(ref):
            assert np.array_equal(ct[k], r.counts)
            assert H.rel_err(wr[k], r.wrench) < TOL
    # a rank's shard of a multi-rank partition, evaluated alone: rows land at the right items
    parts4 = P.shard_by_cost(cost, 4)
    for p in parts4:
        wr, sd, ct = evaluator(p)
        for j, k in enumerate(p):
            assert np.array_equal(ct[j], ref[k].counts) and H.rel_err(wr[j], ref[k].wrench) < TOL
    m.close()


def test_clip_and_integrate_split_equals_the_one_kernel_narrowphase(pfc):
    """Option clip_min: from that many items on, the narrowphase runs as a clip-only kernel (gather, clip, polygon set-up,
    kept polygon) followed by k_integ over the compacted polygons.  Forced on (clip_min = 1) and off (0) on scenes of
    every kind -- regularized boxes, a bristle blob / tool batch, random and degenerate fuzz poses with mixed models and
    tet-tet instructions -- the per-item counters must be equal and the sums agree to reduction-order accuracy; both
    agree with the oracle."""
    import helpers as H
    from test_gpu_parity import _fuzz_workload
    rng = np.random.default_rng(77)
    worlds = [pfc.configs.c1_boxes(), pfc.configs.c2_box_on_plane(5, montecarlo=True),
              pfc.configs.c3_blob_tool(40, seed=9, n_div_blob=7, n_div_tool=5),
              _fuzz_workload(pfc, rng, 300, False, tet_tet=True), _fuzz_workload(pfc, rng, 300, True, tet_tet=True)]
    for w in worlds:
        ref = H.oracle_run(pfc, w, debug=False)
        out = []
        # clip_queue: tri-tet scenarios clip in k_clip_queue (survivors of the trivial reject queued in the ring, 64 dense
        # lanes per clip round; round 3); 0 keeps k_narrow<.., 2 / 3>, which scenarios with tet-tet instructions always use
        for cm, cq in ((0, 1), (1, 2), (1, 0)):        # 2: k_clip_queue whatever the launch size
            m = pfc.configs.build_scenario(w)
            m.set_option("fused", 0)
            m.set_option("clip_min", cm)
            m.set_option("clip_queue", cq)
            out.append(m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids))
            m.close()
        assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][2], out[2][2]), w.name
        assert np.abs(out[0][0] - out[2][0]).max() <= 1e-11 * max(np.abs(out[0][0]).max(), 1e-300), w.name
        for k in range(w.n_items):
            assert np.array_equal(out[1][2][k], ref[k].counts), (w.name, k)
        scale = max(np.abs(out[0][0]).max(), 1e-300)
        assert np.abs(out[0][0] - out[1][0]).max() <= 1e-11 * scale, w.name
        wref = np.array([r.wrench for r in ref])
        assert np.abs(out[1][0] - wref).max() <= 1e-9 * max(np.abs(wref).max(), 1e-300), w.name
        sref = np.array([r.sdot for r in ref])
        assert np.abs(out[1][1] - sref).max() <= 1e-6 * max(np.abs(sref).max(), 1e-300), w.name


@pytest.mark.parametrize("n_scene", [700, 1300])
def test_clip_on_the_compacted_ring(pfc, n_scene):
    """The clip-only kernel of a half of a two-half evaluation (1 300 scenes) and of a launch of 640 .. 1 023 items (700)
    keeps its polygon ring by survivor rank, 48 columns: a round in which more than 48 of the 64 candidates pass the trivial
    reject takes a second pass.  Box-on-plane scenes let 63 % of the candidates through (2 % of the rounds are such) and
    a fuzz batch mixes models, tet-tet items and degenerate poses.  Equal counters and sums to reduction-order accuracy
    against the one-kernel narrowphase (clip_min = 0); oracle on a sample."""
    import helpers as H
    from test_gpu_parity import _fuzz_workload
    rng = np.random.default_rng(79)
    worlds = [pfc.configs.c2_box_on_plane(n_scene, montecarlo=True, n_div=5), _fuzz_workload(pfc, rng, n_scene, False, tet_tet=True)]
    for w in worlds:
        out = []
        for cm, cq in ((0, 1), (-1, 0), (-1, 2)):
            m = pfc.configs.build_scenario(w)
            m.set_option("fused", 0)
            m.set_option("clip_queue", cq)       # 0: the compacted ring of k_narrow<false, 3> for the tri-tet world as well
            if cm >= 0:
                m.set_option("clip_min", cm)
            out.append(m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids))
            if cm < 0 and n_scene > 1024:
                assert m.last_parts() == 2
            m.close()
        assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][2], out[2][2]), w.name
        assert np.abs(out[0][0] - out[2][0]).max() <= 1e-11 * max(np.abs(out[0][0]).max(), 1e-300), w.name
        scale = max(np.abs(out[0][0]).max(), 1e-300)
        assert np.abs(out[0][0] - out[1][0]).max() <= 1e-11 * scale, w.name
        sample = [0, 1, w.n_items // 2, w.n_items - 1] + [int(k) for k in rng.integers(0, w.n_items, 12)]
        ref = H.oracle_run(pfc, w, items=sample, debug=False)
        for k, r in zip(sample, ref):
            assert np.array_equal(out[1][2][k], r.counts), (w.name, k)
            assert np.abs(out[1][0][k] - r.wrench).max() <= 1e-9 * max(np.abs(r.wrench).max(), 1e-300), (w.name, k)


def test_dual_evaluation_over_the_split_narrowphase(pfc):
    """pfc_eval_dual with the value pass in clip-only + k_integ form (the list of contributing candidates is then written
    by k_integ from the candidate index kept with every polygon): partials equal to those of the one-kernel value pass."""
    w = pfc.configs.c3_blob_tool(24, seed=12, n_div_blob=7, n_div_tool=5)
    rng = np.random.default_rng(5)
    n_dir = 3
    dp = rng.standard_normal((w.n_items, n_dir, 24)) * 1e-2
    dt = rng.standard_normal((w.n_items, n_dir, 6))
    ds = rng.standard_normal((w.n_items, n_dir, 6)) * 1e-3
    out = []
    for cm in (0, 1):
        m = pfc.configs.build_scenario(w)
        m.set_option("fused", 0)
        m.set_option("clip_min", cm)
        out.append(m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, dp, dt, ds, w.ins_ids))
        m.close()
    for a, b in zip(out[0], out[1]):
        a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
        assert np.abs(a - b).max() <= 1e-9 * max(np.abs(a).max(), 1e-300)


def test_split_narrowphase_on_many_small_mixed_items(pfc):
    """20 000 fuzz items (four small meshes, regularized / bristle / tet-tet instructions mixed at random, ~100 candidates
    each): more than a million candidates, so the candidate chunks are 512 wide and a 64-polygon piece of k_integ holds
    the runs of several items of different friction models -- the path a big single-instruction batch never takes.  The
    split narrowphase must reproduce the one-kernel narrowphase (itself checked against the oracle on subsets of these
    workloads by the fuzz tests): per-item counters equal, sums to reduction-order accuracy; a sample of items is checked
    against the oracle directly."""
    import helpers as H
    from test_gpu_parity import _fuzz_workload
    rng = np.random.default_rng(4242)
    w = _fuzz_workload(pfc, rng, 20000, False, tet_tet=True)
    out = []
    for cm in (0, 1024):
        m = pfc.configs.build_scenario(w)
        m.set_option("clip_min", cm)
        out.append(m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids))
        if cm:
            assert m.stats()["candidates"] >= 512 * 2048, m.stats()       # wide chunks were in use
        m.close()
    assert np.array_equal(out[0][2], out[1][2])
    scale = np.abs(out[0][0]).max()
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-11 * scale
    # sdot of a bristle item is conditioned like its patch stiffness: 1e-6 per item, except for sliver
    # contacts whose K-bar has two or more eigenvalues at the rounding level of K (there the reference's own sdot moves by
    # as much under a perturbation of K by its backward error, tests/test_sdot_sensitivity.py)
    ds = np.abs(out[0][1] - out[1][1]).max(axis=1)
    rel = ds / np.maximum(np.abs(out[0][1]).max(axis=1), 1e-300)
    loose = np.flatnonzero(rel > 1e-6)
    assert len(loose) <= w.n_items // 500, len(loose)
    for r, k in zip(H.oracle_run(pfc, w, items=loose, debug=True), loose):
        assert r.has_K and rel[k] < 5e-3, (k, rel[k])
        Kb = np.diag(r.Sinv) @ r.K @ np.diag(r.Sinv)
        ev = np.linalg.eigvalsh((Kb + Kb.T) / 2)
        assert np.sum(ev < 1e-12 * ev[-1]) >= 2, (k, rel[k], ev)
    items = rng.choice(w.n_items, 40, replace=False)
    ref = H.oracle_run(pfc, w, items=items, debug=False)
    for r, k in zip(ref, items):
        assert np.array_equal(out[1][2][k], r.counts), k
        assert np.abs(out[1][0][k] - r.wrench).max() <= 1e-9 * max(np.abs(r.wrench).max(), 1e-300), k


def _sampled_items_vs_oracle(pfc, w, wrench, sdot, counts, n_sample, seed):
    """n_sample random items of a batch against the CPU oracle: counts bit-equal, wrench 1e-9, ṡ 1e-6 (1e-3 for items whose
    K̄ has two or more eigenvalues at the rounding level: DESIGN.md §5.7)."""
    pick = np.sort(np.random.default_rng(seed).choice(w.n_items, size=n_sample, replace=False))
    ref = H.oracle_run(pfc, w, items=pick, debug=False)
    for k, r in zip(pick, ref):
        assert np.array_equal(counts[k], r.counts), (k, counts[k], r.counts)
        assert H.rel_err(wrench[k], r.wrench) < TOL, (k, wrench[k], r.wrench)
        t = 1e-6
        if r.has_K:
            Kb = np.diag(r.Sinv) @ r.K @ np.diag(r.Sinv)
            ev = np.linalg.eigvalsh((Kb + Kb.T) / 2)
            if np.sum(ev < 1e-12 * ev[-1]) >= 2:
                t = 1e-3
        assert H.rel_err(sdot[k], r.sdot) < t, (k, sdot[k], r.sdot)
    return pick


@pytest.mark.parametrize("n_poses, parts", [(1100, 2), (700, 1)])
def test_c3_full_size_at_bench_scale_default_options(pfc, n_poses, parts):
    """The path bench.py times, at full C3 size (9 680 tets x 5 120 triangles, bristle) with DEFAULT options: 1 100 poses run
    as two concurrent halves (k_bp_dfs32, clip-only k_narrow<.., 3> on the compacted ring, k_integ, k_fric per half);
    700 poses as one launch sequence with the clip-only narrowphase.  12 sampled items against the CPU oracle."""
    w = pfc.configs.c3_blob_tool(n_poses)
    assert w.meta["n_tet"] == 9680 and w.meta["n_tri"] == 5120
    m = pfc.configs.build_scenario(w)
    wr, sd, ct = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert m.last_parts() == parts
    _sampled_items_vs_oracle(pfc, w, wr, sd, ct, 12, seed=n_poses)
    # a second evaluation (grown lists, replayed graphs): integers reproducible, sums to summation order
    wr2, sd2, ct2 = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert np.array_equal(ct2, ct)
    np.testing.assert_allclose(wr2, wr, rtol=1e-11, atol=1e-11 * np.abs(wr).max())
    assert ct[:, 1].sum() > 1500 * n_poses        # ~1.9 k candidate pairs per pose
    m.close()


def test_bench_two_ranks_share_the_gpu_gloo_rehearsal():
    """bench.py --config C5 --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), with both
    ranks on GPU 0 and the exchange through host memory (PFC_BENCH_BACKEND=gloo): the product sharding code
    (parallel.shard_by_cost, pack_rows, all_gather_rows) around the HIP evaluator in two fresh processes, the JSON line of
    rank 0 with its oracle validation.  The measured configuration is nccl (= RCCL) with one rank per GPU."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PFC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--config", "C5", "--steps", "2",
           "--warmup", "1", "--reps", "1", "--cpu-seconds", "0", "--no-extras"]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["validated_items"] >= 1 and out["validation"].startswith("counts bit-equal"), out.get("validation")


@pytest.mark.parametrize("config", ["C5", "C3"])
def test_bench_one_rank_rccl_exchange_rehearsal(config):
    """The RCCL leg of the multi-GPU step on a one-GPU box: bench.py in a fresh process with a process group of ONE rank over
    the nccl backend and PFC_BENCH_FORCE_EXCHANGE=1, so that the per-step exchange is really issued through RCCL on device
    tensors -- parallel.all_gather_rows ([wrench | sdot | counts] rows, C5) / all_gather_into_tensor ([wrench | sdot], C3) -- with
    the Float64 rows and shapes the multi-rank run uses.  What it cannot show is the wire."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PFC_BENCH_FORCE_EXCHANGE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PFC_BENCH_BACKEND", None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--config", config, "--steps", "2", "--warmup", "1",
           "--reps", "1", "--cpu-seconds", "0", "--no-extras"] + (["--poses", "1200"] if config == "C3" else [])
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["exchange"].startswith("RCCL all-gather"), out["config"]
    assert out["exchange_ms_per_step"] > 0.0 and out["validation"].startswith("counts bit-equal")


def test_two_halves_run_side_by_side_after_an_eager_rccl_init():
    """The two halves of a big evaluation need two hardware queues.  With an RCCL communicator initialised eagerly before
    pfc_create and no collective issued yet (what bench.py does under torch.distributed.run), the runtime put both streams of a
    handle on ONE queue and the step took 25 % longer; make_twin now tests the pair (k_queue_probe) and re-creates the twin's stream
    with another priority when they do not overlap.  scripts/rccl_queue_probe.py in fresh processes: the step after such an
    initialisation must cost what it costs without one, and the pair must not be reported serial."""
    import os
    import re
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ("none", "before-nocoll"):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        p = subprocess.run([sys.executable, os.path.join(root, "scripts", "rccl_queue_probe.py"), mode, "4096"], cwd=root, env=env,
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("mode")][-1]
        res[mode] = (float(re.search(r": ([0-9.]+) ms per step", line).group(1)), line)
        assert "parts 2" in line and "serial" not in line, line
    assert res["before-nocoll"][0] < 1.10 * res["none"][0], res


@pytest.mark.gpu
def test_host_paths_without_bar_resident_inputs(tmp_path):
    """On a device with a large BAR the host-buffer entry points write their inputs into device memory and let the kernels work
    in place up to 4 096 items / (item, direction) pairs; without one (PFC_NO_BAR_INPUTS=1 stands in for it) they read pinned
    host memory in place up to 512 and stage copies above.  Both forms, in fresh processes, over the sizes on either side of
    those limits (value and Dual): the same counts, results equal to the summation-order noise of the atomics."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for tag, extra in (("bar", {}), ("nobar", {"PFC_NO_BAR_INPUTS": "1"})):
        f = str(tmp_path / (tag + ".npz"))
        env = dict(os.environ, **extra)
        env.pop("PFC_LIB", None)
        p = subprocess.run([sys.executable, os.path.join(root, "scripts", "host_paths_dump.py"), f], cwd=root, env=env,
                           capture_output=True, text=True, timeout=400)
        assert p.returncode == 0, p.stderr[-2000:]
        got[tag] = np.load(f)
    assert sorted(got["bar"].files) == sorted(got["nobar"].files) and len(got["bar"].files) >= 70
    for k in got["bar"].files:
        a, b = got["bar"][k], got["nobar"][k]
        if k.endswith("counts"):
            assert np.array_equal(a, b), k
        else:
            scale = np.abs(a).max(axis=-1, keepdims=True) + 1e-300
            assert np.all(np.abs(a - b) <= 1e-11 * scale + 1e-13), (k, float(np.max(np.abs(a - b) / scale)))
