"""The reference's OBB separating-axis tests (test/test_obb/test_intersection.jl:39-104) run through the device
broadphase itself: pfc_eval on host-supplied trees whose root boxes are the test's boxes.

A one-node tree makes the root pair leaf x leaf (Float32 filter radius 320 u S, quaternion path); a three-node tree
(root + two tiny leaves at the centre) makes it internal x internal (radius 24 u S, identity quaternions).  The
verdict of the root pair is read from the item's counters: leaf trees -> candidates (1 / 0), internal trees -> node
tests (5 = root hit + its 4 child pairs, 1 = root separated).  Touching configurations are probed at 1 -/+ tol for
tol = 1e-3 (decided by the Float32 filter), 1e-6 (the reference's tolerance), 1e-9 and 1e-12 (inside the filter's
error radius: the pair must come back "undecided" and be settled by the exact Float64 test), with the filter on
(k_bp_dfs32), off (option no_filter = 1: k_bp_dfs, all Float64) and through the seed-expansion kernel (k_bp_expand).  Every verdict must equal the oracle's
BB_BB_intersect (src/obb/bb_intersection.jl:2-74) and, for tol >= 1e-9, the geometric truth.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as Orc

pytestmark = pytest.mark.gpu

TOLS = (1.0e-3, 1.0e-6, 1.0e-9, 1.0e-12)


def _rot_between(u, v):
    u = u / np.linalg.norm(u); v = v / np.linalg.norm(v)
    c = float(u @ v)
    if c > 1 - 1e-15:
        return np.eye(3)
    if c < -1 + 1e-15:
        a = np.cross(u, [1.0, 0, 0]) if abs(u[0]) < 0.9 else np.cross(u, [0, 1.0, 0])
        a /= np.linalg.norm(a)
        return 2 * np.outer(a, a) - np.eye(3)
    w = np.cross(u, v)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return np.eye(3) + K + K @ K / (1 + c)


def _axis_angle(th, a):
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _cases():
    """(e1, e2, R_a_b, t_a_b, expected) for the face-corner (6 x 8 x 2) and edge-edge (6 x 15 x 5 x 3) touching
    configurations at 1 -/+ tol."""
    out = []
    e1, e2 = np.array([1.0, 2.0, 3.0]), np.array([2.1, 2.2, 2.3])
    faces = [np.eye(3)[k // 2] * (-1.0 if k % 2 == 0 else 1.0) for k in range(6)]
    corners = [np.array([sx, sy, sz], dtype=float) for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)]
    for f in faces:
        for c in corners:
            for d1, d2 in ((f, c), (c, f)):
                R = _rot_between(d2, d1)
                sep = d1 * e1 + R @ (d2 * e2)
                for tol in TOLS:
                    out.append((0, R, sep * (1 - tol), 1, tol))
                    out.append((0, R, sep * (1 + tol), 0, tol))
    rng = np.random.default_rng(3)
    edges = [np.array(e, dtype=float) for e in ([0, -1, -1], [0, 1, -1], [0, -1, 1], [0, 1, 1], [-1, 0, -1], [1, 0, -1])]
    for e in edges:
        for th in rng.random(15) * 2 * np.pi:
            for extra in np.arange(0, 2 * np.pi + 1e-9, np.pi / 2):
                for ax in np.eye(3):
                    R = _axis_angle(th, e) @ _axis_angle(extra, ax)
                    sep = e * 2.0
                    for tol in TOLS:
                        out.append((1, R, sep * (1 - tol), 1, tol))
                        out.append((1, R, sep * (1 + tol), 0, tol))
    return out


def _tree(pfc, e, internal):
    G = pfc.geometry
    I9 = np.eye(3).reshape(-1)
    if not internal:
        return G.OBBTree(np.zeros((1, 3)), np.asarray(e, dtype=float).reshape(1, 3), I9.reshape(1, 9),
                         np.full((1, 2), -1, dtype=np.int32), np.zeros(1, dtype=np.int32))
    c = np.zeros((3, 3))
    ee = np.array([e, [1e-3] * 3, [1e-3] * 3], dtype=float)
    return G.OBBTree(c, ee, np.tile(I9, (3, 1)), np.array([[1, 2], [-1, -1], [-1, -1]], dtype=np.int32),
                     np.array([G.INTERNAL, 0, 1], dtype=np.int32))


def _meshes(pfc, n_elem):
    """n_elem far-apart dummy elements per mesh: the broadphase only looks at the boxes of the host-supplied tree."""
    G = pfc.geometry
    tri_pts, tris, tet_pts, tets, eps = [], [], [], [], []
    for k in range(n_elem):
        o = np.array([100.0 * (k + 1), 0.0, 0.0])
        tri_pts += [o, o + [1, 0, 0], o + [0, 1, 0]]
        tris.append([3 * k, 3 * k + 1, 3 * k + 2])
        tet_pts += [o, o + [1, 0, 0], o + [0, 1, 0], o + [0, 0, 1]]
        tets.append([4 * k, 4 * k + 1, 4 * k + 2, 4 * k + 3])
        eps += [0.0, 0.0, 0.0, 1.0]
    return G.EMesh(np.array(tri_pts), tri=np.array(tris)), G.EMesh(np.array(tet_pts), tet=np.array(tets), eps=np.array(eps))


# kernel that tests the root pair: the Float32 workgroup descent with its cooperative exact settle (default), the
# all-Float64 descent (option no_filter), or the level-synchronous seed expansion (what a batch of >= 1 024 items over
# small trees starts with; Float64 bb_bb_intersect)
PATHS = {"dfs32": dict(no_filter=0, bfs_levels=0), "dfs64": dict(no_filter=1, bfs_levels=0),
         "expand": dict(no_filter=0, bfs_levels=1)}


@pytest.mark.parametrize("path", list(PATHS))
@pytest.mark.parametrize("internal", [False, True])
def test_sat_touching_through_the_device_broadphase(pfc, path, internal):
    cases = _cases()
    ext = ((np.array([1.0, 2.0, 3.0]), np.array([2.1, 2.2, 2.3])), (np.ones(3), np.ones(3)))
    n_elem = 2 if internal else 1
    tri, tet = _meshes(pfc, n_elem)
    S = pfc.scenario
    m = S.MechanismScenario()
    for k, (ea, eb) in enumerate(ext):
        i1 = m.add_contact(f"a{k}", tri, tree=_tree(pfc, ea, internal))
        i2 = m.add_contact(f"b{k}", tet, c_prop=S.ContactProperties(1.0e6), tree=_tree(pfc, eb, internal))
        m.add_friction_regularize(i1, i2, mu_d=0.3)
    m.finalize()
    for k, v in PATHS[path].items():
        m.set_option(k, v)
    n = len(cases)
    ids = np.array([c[0] for c in cases], dtype=np.int32)
    pose = np.zeros((n, 24))
    for k, (_, R, t, _, _) in enumerate(cases):
        # the broadphase reads x_r1_r2 = (R_a_b, t_a_b) (src/obb/tree_types.jl:43-50); x_r2_r1 is its inverse
        pose[k, 12:21] = R.reshape(-1, order="F"); pose[k, 21:24] = t
        pose[k, 0:9] = R.T.reshape(-1, order="F"); pose[k, 9:12] = -(R.T @ t)
    wrench, sdot, counts = m.force_all_elastic_intersections(pose, np.zeros((n, 6)), np.zeros((n, 6)), ids)
    if internal:
        assert set(np.unique(counts[:, 0])) <= {1, 5}
        hit = counts[:, 0] == 5
        assert np.all(counts[:, 1] == 0)
    else:
        assert np.all(counts[:, 0] == 1)
        hit = counts[:, 1] == 1
    # oracle verdict of the same root pair
    L = Orc.lib()
    dp = C.POINTER(C.c_double)
    P = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)
    z, I = np.zeros(3), np.eye(3).reshape(-1)
    n_und_expected = 0
    for k, (kind, R, t, expect, tol) in enumerate(cases):
        ea, eb = ext[kind]
        ref = L.pfo_bb_bb_intersect(P(z), P(ea), P(I), P(z), P(eb), P(I), P(R.reshape(-1, order="F")), P(t))
        assert bool(hit[k]) == bool(ref), (k, kind, tol, expect, counts[k])
        if tol >= 1e-9:       # the reference's own assertion (test_intersection.jl:86-88,98-103), tightened to 1e-9
            assert bool(hit[k]) == bool(expect), (k, kind, tol)
        n_und_expected += tol <= 1e-9
    if path == "dfs32":
        # the Float32 filter must have handed the near-touching pairs to the exact test (statistics word [7] of the stamps view)
        out = (C.c_longlong * 16)()
        assert pfc._lib.lib().pfc_debug_stamps(m._h, out) == 0
        assert out[7] >= n_und_expected, (out[7], n_und_expected)
    m.close()


def test_filtered_and_exact_broadphase_agree_on_a_contact_scene(pfc):
    """Option no_filter = 1 (k_bp_dfs, all Float64) against the default Float32 + exact-settle kernel on a reduced C3
    batch: identical per-item counters (node tests, candidates, non-empty pairs, traction points) and candidate sets."""
    import helpers as H
    w = pfc.configs.c3_blob_tool(8, seed=5, n_div_blob=8, n_div_tool=6)
    ref = H.oracle_run(pfc, w)
    res = []
    for nf in (0, 1):
        m = pfc.configs.build_scenario(w, debug=True)
        m.set_option("no_filter", nf)
        wrench, sdot, counts = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        pairs = [H.sorted_pairs(*m.debug_pairs(k)) for k in range(w.n_items)]
        res.append((wrench, counts, pairs))
        m.close()
    assert np.array_equal(res[0][1], res[1][1])
    for k in range(w.n_items):
        assert np.array_equal(res[0][1][k], ref[k].counts)
        rp, rc = H.sorted_pairs(ref[k].pairs, ref[k].clip_n)
        for nf in (0, 1):
            assert np.array_equal(res[nf][2][k][0], rp) and np.array_equal(res[nf][2][k][1], rc)
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-9, atol=1e-9 * np.abs(res[0][0]).max())


@pytest.mark.parametrize("n_pose", [400, 8800])
def test_poses_that_are_not_rotations_are_settled_exactly(pfc, n_pose):
    """(8 800 poses: two half-launches of 4 400 items, which run the depth-first kernel in its 128-thread form.)
    The Float32 filter composes the pose as a quaternion, which only represents proper rotations.  A pose whose 3x3
    block is scaled, sheared or a reflection fails the per-item check (pose_quat) and every node test of that item goes
    to the exact Float64 test, which uses the matrix as given -- like the reference's BB_BB_intersect
    (src/obb/bb_intersection.jl:2-12), whose verdict the oracle supplies."""
    ea, eb = np.array([1.0, 2.0, 3.0]), np.array([2.1, 2.2, 2.3])
    tri, tet = _meshes(pfc, 1)
    S = pfc.scenario
    m = S.MechanismScenario()
    i1 = m.add_contact("a", tri, tree=_tree(pfc, ea, False))
    i2 = m.add_contact("b", tet, c_prop=S.ContactProperties(1.0e6), tree=_tree(pfc, eb, False))
    m.add_friction_regularize(i1, i2, mu_d=0.3)
    m.finalize()
    m.set_option("bfs_levels", 0)
    rng = np.random.default_rng(11)
    Rs, ts = [], []
    for k in range(n_pose):
        R = _axis_angle(rng.random() * 2 * np.pi, rng.standard_normal(3))
        kind = k % 4
        if kind == 0:
            R = R * (1.0 + 10.0 ** rng.uniform(-6, -1))              # scaled
        elif kind == 1:
            R = R @ np.diag([1.0, 1.0, -1.0])                          # reflection
        elif kind == 2:
            R = R + 10.0 ** rng.uniform(-6, -2) * rng.standard_normal((3, 3))   # sheared
        # kind 3: proper rotation (control: decided by the filter)
        d = rng.standard_normal(3); d /= np.linalg.norm(d)
        Rs.append(R); ts.append(d * rng.uniform(2.0, 7.5))
    n = len(Rs)
    pose = np.zeros((n, 24))
    for k in range(n):
        pose[k, 12:21] = Rs[k].reshape(-1, order="F"); pose[k, 21:24] = ts[k]
        Ri = np.linalg.inv(Rs[k])
        pose[k, 0:9] = Ri.reshape(-1, order="F"); pose[k, 9:12] = -(Ri @ ts[k])
    wrench, sdot, counts = m.force_all_elastic_intersections(pose, np.zeros((n, 6)), np.zeros((n, 6)), np.zeros(n, dtype=np.int32))
    L = Orc.lib()
    dp = C.POINTER(C.c_double)
    P = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)
    z, I = np.zeros(3), np.eye(3).reshape(-1)
    n_hit = 0
    for k in range(n):
        ref = L.pfo_bb_bb_intersect(P(z), P(ea), P(I), P(z), P(eb), P(I), P(Rs[k].reshape(-1, order="F")), P(ts[k]))
        assert bool(counts[k, 1] == 1) == bool(ref), (k, k % 4, counts[k])
        n_hit += bool(ref)
    assert 0 < n_hit < n
    out = (C.c_longlong * 16)()
    assert pfc._lib.lib().pfc_debug_stamps(m._h, out) == 0
    assert out[7] >= 3 * (n // 4), out[7]          # the three non-rotation kinds went to the exact test
    m.close()


def test_meshes_far_from_their_frame_origins(pfc):
    """The single-precision test forms the centre offset of two boxes from Float32 centres: its error is absolute -- u x the
    distance of the boxes from their frame origins, not u x their size -- and enters the error radius as the per-item term
    ItemRec.bp_eabs (pfc_bp.h, "Error radius E", (0)).  A reduced C3 scene whose meshes sit 60 .. 90 m from their frame
    origins (boxes of centimetres: eabs is a few per cent of a box) must give the oracle's candidate sets and counters
    bit for bit, with visibly more node pairs settled by the exact test than the same scene centred at the origins."""
    import helpers as H
    Cfg, S = pfc.configs, pfc.scenario
    n_pose = 6
    w0 = Cfg.c3_blob_tool(n_pose, seed=21, n_div_blob=8, n_div_tool=6)
    o1, o2 = np.array([61.5, -83.25, 17.0]), np.array([-72.0, 40.5, 66.75])
    tool0, blob0 = w0.meshes[0].mesh, w0.meshes[1].mesh
    G = pfc.geometry
    tool = G.EMesh(tool0.point + o1, tri=tool0.tri)
    blob = G.EMesh(blob0.point + o2, tet=blob0.tet, eps=blob0.eps)
    w1 = Cfg.Workload("C3 reduced, meshes far from their frame origins", [Cfg._mesh("tool", tool), Cfg._mesh("blob", blob, 1.0e6)],
                      w0.instructions, w0.ins_ids.copy(), w0.pose.copy(), w0.twist.copy(), w0.s.copy())
    for k in range(n_pose):
        # the same relative placement of the two surfaces: frame r1 (tool) and frame r2 (blob) move against their meshes
        R21 = w0.pose[k, 0:9].reshape(3, 3, order="F"); t21 = w0.pose[k, 9:12]        # x_r2_r1: p2 = R21 p1 + t21
        t21n = t21 + o2 - R21 @ o1
        R12 = R21.T
        w1.pose[k, 9:12] = t21n
        w1.pose[k, 21:24] = -(R12 @ t21n)
        # ... and the same velocity field: twist of 2 w.r.t. 1 in frame r2, v(p) = v_lin + w x p, about the moved origin
        w1.twist[k, 3:6] = w0.twist[k, 3:6] - np.cross(w0.twist[k, 0:3], o2)
    und = []
    for w in (w0, w1):
        ref = H.oracle_run(pfc, w)
        m = Cfg.build_scenario(w, debug=True)
        m.set_option("fused", 0)          # the batched path: k_bp_dfs32
        wrench, sdot, counts = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        out = (C.c_longlong * 16)()
        assert pfc._lib.lib().pfc_debug_stamps(m._h, out) == 0
        und.append(int(out[7]))
        for k in range(w.n_items):
            assert np.array_equal(counts[k], ref[k].counts), (w.name, k, counts[k], ref[k].counts)
            pairs, clip_n = H.sorted_pairs(*m.debug_pairs(k))
            rp, rc = H.sorted_pairs(ref[k].pairs, ref[k].clip_n)
            assert np.array_equal(pairs, rp) and np.array_equal(clip_n, rc), (w.name, k)
            assert H.rel_err(wrench[k], ref[k].wrench) < 1e-7, (w.name, k)        # 1e-16 x 100 m / 1 cm of cancellation in both
        assert counts[:, 1].sum() > 0 and counts[:, 3].sum() > 0
        m.close()
    assert und[1] > 10 * max(und[0], 1), und      # the absolute term widened the band; verdicts stayed exact
