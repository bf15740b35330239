"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): candidate pair indices and clipped-polygon vertex counts bit-exact; Float64
wrenches / ṡ within 1e-6 relative.  The HIP path's sums are order-nondeterministic (wave reductions + FP64
atomics) so the tolerance actually asserted is TOL_WRENCH below; observed agreement is ~1e-13.
"""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

TOL_WRENCH = 1e-6      # north_star tolerance, relative to the 6-vector's norm
TOL_TIGHT = 1e-9       # what we additionally expect from a reordered Float64 sum of < 1e6 terms


def _eval(pfc, w, debug=True):
    m = pfc.configs.build_scenario(w, debug=debug)
    wrench, sdot, counts = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    return m, wrench, sdot, counts


def _assert_item_parity(m, k, ref, wrench, sdot, counts, tol=TOL_WRENCH):
    # counts: {node tests, candidates, non-empty, traction points} -- all integer, all exact
    assert np.array_equal(counts[k], ref.counts), (k, counts[k], ref.counts)
    pairs, clip_n = m.debug_pairs(k)
    gp, gc = H.sorted_pairs(pairs, clip_n)
    rp, rc = H.sorted_pairs(ref.pairs, ref.clip_n)
    assert np.array_equal(gp, rp), f"item {k}: candidate pair set differs"
    assert np.array_equal(gc, rc), f"item {k}: clip vertex counts differ"
    for name, a, b in (("wrench", wrench[k], ref.wrench), ("sdot", sdot[k], ref.sdot)):
        if np.linalg.norm(b) == 0.0:
            assert np.linalg.norm(a) == 0.0, (name, k, a)
        else:
            assert H.rel_err(a, b) < tol, (name, k, a, b)


def test_device_arithmetic_is_correctly_rounded(pfc):
    """Division, sqrt and fma on the device must be bitwise IEEE (the clip predicates depend on it)."""
    w = pfc.configs.c1_boxes()
    m = pfc.configs.build_scenario(w)
    rng = np.random.default_rng(1)
    n = 1 << 16
    x = rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8, n)
    y = rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8, n)
    out = np.zeros(3 * n)
    import ctypes as C
    dp = C.POINTER(C.c_double)
    rc = pfc._lib.lib().pfc_selftest_math(m._h, n, x.ctypes.data_as(dp), y.ctypes.data_as(dp), out.ctypes.data_as(dp))
    assert rc == 0
    assert np.array_equal(out[:n], x / y)
    assert np.array_equal(out[n:2 * n], np.sqrt(np.abs(x)))
    import math
    idx = rng.integers(0, n, 2000)
    fm = np.array([math.fma(float(x[i]), float(y[i]), float(x[i])) for i in idx]) if hasattr(math, "fma") else None
    if fm is not None:
        assert np.array_equal(out[2 * n:][idx], fm)
    m.close()


def test_c1_boxes(pfc):
    w = pfc.configs.c1_boxes()
    m, wrench, sdot, counts = _eval(pfc, w)
    ref = H.oracle_run(pfc, w)
    for k, r in enumerate(ref):
        _assert_item_parity(m, k, r, wrench, sdot, counts, TOL_TIGHT)
    m.close()


def test_c2_box_on_plane(pfc):
    w = pfc.configs.c2_box_on_plane(1)
    m, wrench, sdot, counts = _eval(pfc, w)
    ref = H.oracle_run(pfc, w)
    _assert_item_parity(m, 0, ref[0], wrench, sdot, counts, TOL_TIGHT)
    assert counts[0, 3] > 0
    m.close()


def test_c4_montecarlo_scenes(pfc):
    w = pfc.configs.c2_box_on_plane(64, montecarlo=True)
    m, wrench, sdot, counts = _eval(pfc, w)
    ref = H.oracle_run(pfc, w)
    for k, r in enumerate(ref):
        _assert_item_parity(m, k, r, wrench, sdot, counts, TOL_TIGHT)
    m.close()


@pytest.mark.parametrize("n_quad", [1, 2])
def test_c3_reduced_bristle(pfc, n_quad):
    w = pfc.configs.c3_blob_tool(6, n_div_blob=8, n_div_tool=6)
    w.instructions[0].n_quad_rule = n_quad
    m, wrench, sdot, counts = _eval(pfc, w)
    ref = H.oracle_run(pfc, w)
    for k, r in enumerate(ref):
        _assert_item_parity(m, k, r, wrench, sdot, counts, TOL_TIGHT)
        if r.has_K:
            K, Kis, Sinv, cop = m.debug_stiffness(k)
            assert H.rel_err(K, r.K) < TOL_TIGHT
            assert H.rel_err(Kis, r.Kbar_inv_sqrt) < 1e-8
            assert H.rel_err(Sinv, r.Sinv) < TOL_TIGHT
            assert H.rel_err(cop, r.cop) < TOL_TIGHT
    m.close()


def test_c3_full_size_poses(pfc):
    """BASELINE C3 at full mesh size (9 680 tets x 5 120 triangles), 4 poses, against the oracle."""
    w = pfc.configs.c3_blob_tool(4)
    m, wrench, sdot, counts = _eval(pfc, w)
    ref = H.oracle_run(pfc, w)
    for k, r in enumerate(ref):
        _assert_item_parity(m, k, r, wrench, sdot, counts, TOL_TIGHT)
    # traction points themselves (TractionCache), compared as a sorted multiset
    t_gpu = m.debug_tractions(0)
    t_ref = ref[0].trac
    assert t_gpu.shape == t_ref.shape
    key = lambda t: np.lexsort(tuple(t[:, c] for c in range(7, -1, -1)))
    assert np.array_equal(t_gpu[key(t_gpu)], t_ref[key(t_ref)]), "traction points are not bit-identical"
    m.close()


def test_no_contact_and_empty(pfc):
    """Separated bodies: zero wrench, ṡ = -s/τ (friction.jl:77-81); and n_items = 0."""
    w = pfc.configs.c3_blob_tool(3, n_div_blob=6, n_div_tool=5, distance=0.25)
    m, wrench, sdot, counts = _eval(pfc, w)
    ref = H.oracle_run(pfc, w)
    for k, r in enumerate(ref):
        assert counts[k, 3] == 0 and r.counts[3] == 0
        assert np.array_equal(counts[k], r.counts)
        assert np.all(wrench[k] == 0.0)
        assert np.array_equal(sdot[k], r.sdot)          # -(1/τ) s : one rounding, bit-exact
    w0, s0, c0 = m.force_all_elastic_intersections(np.zeros((0, 24)), np.zeros((0, 6)), np.zeros((0, 6)),
                                                   np.zeros(0, dtype=np.int32))
    assert w0.shape == (0, 6)
    m.close()


@pytest.mark.parametrize("model", ["regularized", "bristle"])
def test_tet_tet_vol_vol(pfc, model):
    """Volume-volume contact (non_friction.jl:166-194): equal-pressure plane, clip_plane_tet, zero_small_coordinates,
    quad / triangle Sutherland-Hodgman; compliant box on the compliant half-plane (test_vol_vol.jl geometry) and two
    compliant spheres of different stiffness."""
    w = pfc.configs.vol_vol(6, n_div=5, model=model)
    m, wrench, sdot, counts = _eval(pfc, w)
    ref = H.oracle_run(pfc, w)
    assert sum(int(r.counts[3] > 0) for r in ref) >= 8
    for k, r in enumerate(ref):
        _assert_item_parity(m, k, r, wrench, sdot, counts, 1e-8)
    t_gpu = m.debug_tractions(1)
    t_ref = ref[1].trac
    assert t_gpu.shape == t_ref.shape and t_ref.shape[0] > 0
    key = lambda t: np.lexsort(tuple(t[:, c] for c in range(7, -1, -1)))
    assert np.array_equal(t_gpu[key(t_gpu)], t_ref[key(t_ref)]), "traction points are not bit-identical"
    m.close()


def test_mixed_tri_tet_and_tet_tet_instructions(pfc):
    """One scenario holding tri-tet and tet-tet instructions with both friction models in one evaluation."""
    G, Cf = pfc.geometry, pfc.configs
    sph = G.emesh_sphere(0.1, 4)
    meshes = [Cf.MeshSpec("tet_a", G.as_tet_emesh(sph), G.build_tree(G.as_tet_emesh(sph)), 1.0e6),
              Cf.MeshSpec("tri_b", G.as_tri_emesh(sph), G.build_tree(G.as_tri_emesh(sph)), None),
              Cf.MeshSpec("tet_c", G.as_tet_emesh(sph), G.build_tree(G.as_tet_emesh(sph)), 2.0e6)]
    ins = [Cf.InsSpec(1, 0, "bristle"), Cf.InsSpec(2, 0, "regularized"), Cf.InsSpec(1, 2, "regularized"),
           Cf.InsSpec(0, 2, "bristle")]
    rng = np.random.default_rng(8)
    ids, pose, twist, s = [], [], [], []
    for k in range(12):
        R1, R2 = Cf.random_rotation(rng), Cf.random_rotation(rng)
        u = rng.standard_normal(3); u /= np.linalg.norm(u)
        ids.append(k % 4)
        pose.append(pfc.relative_pose(R1, 0.19 * u, R2, np.zeros(3)))
        twist.append(rng.uniform(-0.5, 0.5, 6))
        s.append(rng.standard_normal(6) * 1e-2)
    w = Cf.Workload("mixed", meshes, ins, np.asarray(ids, dtype=np.int32), np.array(pose), np.array(twist), np.array(s))
    m, wrench, sdot, counts = _eval(pfc, w)
    ref = H.oracle_run(pfc, w)
    for k, r in enumerate(ref):
        _assert_item_parity(m, k, r, wrench, sdot, counts, 1e-8)
    m.close()


def test_host_supplied_trees_with_improper_and_skewed_boxes(pfc):
    """A host may pass its own flattened tree.  Leaf boxes whose R is a reflection (same box, one axis
    negated) cannot be held as a unit quaternion, so every node pair that involves them is left undecided by the
    single-precision broadphase and settled by the exact Float64 test: the candidate sets must still be the reference's
    bit for bit.  Half of the leaves are flipped so that both paths are mixed inside the same waves."""
    w = pfc.configs.c3_blob_tool(6, seed=21, n_div_blob=8, n_div_tool=6)
    for ms in w.meshes:
        t = ms.tree
        leaves = np.nonzero(t.leaf != pfc.geometry.INTERNAL)[0]
        flip = leaves[::2]
        t.R[flip, 3:6] *= -1.0                       # negate the second axis: det R = -1, same box
    m, wrench, sdot, counts = _eval(pfc, w)
    ref = H.oracle_run(pfc, w)
    for k in range(w.n_items):
        _assert_item_parity(m, k, ref[k], wrench, sdot, counts, tol=TOL_TIGHT)
    stats = m.stats()
    assert stats["candidates"] == sum(int(r.counts[1]) for r in ref)
    m.close()


_fuzz_workload = H.fuzz_workload


@pytest.mark.parametrize("degenerate,tet_tet", [(False, False), (True, False), (False, True), (True, True)])
def test_fuzz_random_and_degenerate_poses(pfc, degenerate, tet_tet):
    """600 items of random small-mesh pairs: every integer output, every candidate set and every clipped-polygon vertex
    count bit-exact against the oracle; wrenches at the tight tolerance.  The degenerate half exercises exact zeros."""
    rng = np.random.default_rng(77 + int(degenerate) + 2 * int(tet_tet))
    w = _fuzz_workload(pfc, rng, 600, degenerate, tet_tet)
    m, wrench, sdot, counts = _eval(pfc, w)
    ref = H.oracle_run(pfc, w)
    n_contact = 0
    for k in range(w.n_items):
        # sdot goes through K̄^{-1/2}: tiny / degenerate patches are ill-conditioned there
        assert np.array_equal(counts[k], ref[k].counts), (k, counts[k], ref[k].counts)
        pairs, clip_n = m.debug_pairs(k)
        gp, gc = H.sorted_pairs(pairs, clip_n)
        rp, rc = H.sorted_pairs(ref[k].pairs, ref[k].clip_n)
        assert np.array_equal(gp, rp) and np.array_equal(gc, rc), k
        if np.linalg.norm(ref[k].wrench) > 0:
            assert H.rel_err(wrench[k], ref[k].wrench) < TOL_TIGHT, (k, wrench[k], ref[k].wrench)
            n_contact += 1
        else:
            assert np.linalg.norm(wrench[k]) == 0.0
    assert n_contact > 100
    m.close()


def test_bound_evaluation_equals_the_general_entry_point(pfc):
    """MechanismScenario.bind: persistent input / output buffers, one foreign call per evaluation; inputs changed in place
    are what the next call evaluates."""
    w = pfc.configs.c1_boxes()
    m = pfc.configs.build_scenario(w)
    wr0, sd0, ct0 = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    b = m.bind(w.pose, w.twist, w.s, w.ins_ids)
    wr, sd, ct = b()
    assert np.array_equal(ct, ct0) and np.array_equal(wr, wr0) and np.array_equal(sd, sd0)
    # a second point: the same change through both entry points
    pose2 = w.pose.copy(); pose2[:, 11] -= 1.0e-3; pose2[:, 23] += 1.0e-3      # t21 / t12 along z (a rigid shift of frame 2)
    tw2 = w.twist * 0.5
    wr1, sd1, ct1 = m.force_all_elastic_intersections(pose2, tw2, w.s, w.ins_ids)
    b.pose[:] = pose2; b.twist[:] = tw2
    wr, sd, ct = b()
    assert np.array_equal(ct, ct1) and np.array_equal(wr, wr1) and np.array_equal(sd, sd1)
    m.close()
