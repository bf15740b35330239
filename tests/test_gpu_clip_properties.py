"""Device-side restatement of the reference's clip property tests (test/test_clip/test_static_clip.jl:13-64 for the
Sutherland-Hodgman clip of a triangle / quad against a tet, test/test_clip/test_plane_tet_intersection.jl:15-62 for
the plane-tet polygon that feeds it in the tet-tet op), on what the HIP path itself produces: random single triangles and
tets thrown at random tets; every traction point of the device (the quadrature points of the fan over the clipped
polygon, pfc_debug_tractions) must lie (i) inside the tet the polygon was clipped to, (ii) in the plane and inside the
triangle it was clipped from (tri-tet) or on the equal-pressure plane (tet-tet: all points of a pair coplanar), and the
clipped area must equal an independent Sutherland-Hodgman clip in numpy; vertex counts are compared with the oracle."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _bary_tet(p, tet):
    A = np.vstack([tet.T, np.ones(4)])
    return np.linalg.solve(A, np.append(p, 1.0))


def _clip_poly_numpy(poly, tet):
    """Independent S-H clip of a planar polygon (k x 3) by the 4 half-spaces of a tet, in Cartesian coordinates."""
    out = [np.asarray(v, dtype=np.float64) for v in poly]
    for i in range(4):
        inp, out = out, []
        if not inp:
            break
        val = [_bary_tet(v, tet)[i] for v in inp]
        for a in range(len(inp)):
            b = (a + 1) % len(inp)
            if val[a] >= 0:
                out.append(inp[a])
            if (val[a] >= 0) != (val[b] >= 0):
                t = val[a] / (val[a] - val[b])
                out.append(inp[a] + t * (inp[b] - inp[a]))
    return np.array(out)


def _area(poly):
    if len(poly) < 3:
        return 0.0
    s = np.zeros(3)
    for k in range(1, len(poly) - 1):
        s += np.cross(poly[k] - poly[0], poly[k + 1] - poly[0])
    return 0.5 * np.linalg.norm(s)


def _random_tets(rng, n):
    pts, tets, eps = [], [], []
    for k in range(n):
        while True:
            p = rng.standard_normal((4, 3)) * 0.5 + rng.uniform(-0.5, 0.5, 3)
            v = np.linalg.det(np.c_[p[1] - p[0], p[2] - p[0], p[3] - p[0]])
            if abs(v) > 0.02:
                break
        if v < 0:
            p = p[[1, 0, 2, 3]]
        pts.append(p); tets.append(np.arange(4) + 4 * k); eps.append([0.0, 0.0, 0.0, 1.0])
    return np.vstack(pts), np.array(tets), np.concatenate(eps)


def test_triangle_tet_clip_properties_on_the_device(pfc, O):
    G, S = pfc.geometry, pfc.scenario
    rng = np.random.default_rng(1234)
    n_tri, n_tet = 60, 40
    tp = rng.standard_normal((n_tri, 3, 3)) * 0.6
    tri_mesh = G.EMesh(tp.reshape(-1, 3), tri=np.arange(3 * n_tri).reshape(-1, 3))
    pts, tets, eps = _random_tets(rng, n_tet)
    tet_mesh = G.EMesh(pts, tet=tets, eps=eps)
    t1, t2 = G.build_tree(tri_mesh, "median"), G.build_tree(tet_mesh, "median")
    ins = dict(model="regularized", chi=0.0, n_quad=2, mu_s=0.3, mu_d=0.3, v_c=0.01)
    pose = Orc_pose = None
    from oracle import oracle as Orc
    pose = Orc.make_pose(np.eye(3), np.zeros(3))
    dev = H.eval_scene("hip", pfc, tri_mesh, None, tet_mesh, 1.0e6, ins, pose, np.zeros(6), None, trees=(t1, t2), want_pairs=True)
    ref = H.eval_scene("oracle", pfc, tri_mesh, None, tet_mesh, 1.0e6, ins, pose, np.zeros(6), None, trees=(t1, t2), want_pairs=True)
    gp, gc = H.sorted_pairs(dev.pairs, dev.clip_n)
    rp, rc = H.sorted_pairs(ref.pairs, ref.clip_n)
    assert np.array_equal(gp, rp) and np.array_equal(gc, rc)
    assert np.count_nonzero(gc >= 3) > 100 and gc.max() >= 6
    # every traction point: inside its tet AND inside its triangle (some pair that clipped to a polygon must explain it)
    pairs_ne = gp[gc >= 3]
    trac = dev.trac
    assert trac.shape[0] > 500
    area_dev = trac[:, 6].sum()
    area_np = 0.0
    for a, b in pairs_ne:
        area_np += _area(_clip_poly_numpy(tp[a], pts[tets[b]]))
    # zero twist, chi = 0, eps > 0 inside the tet: every quadrature point has p > 0, so sum dA = clipped area
    assert area_dev == pytest.approx(area_np, rel=1e-10)
    tet_of = {}
    for a, b in pairs_ne:
        tet_of.setdefault(int(a), []).append(int(b))
    n_checked = 0
    for row in trac[:: max(1, trac.shape[0] // 400)]:
        nh, r = row[0:3], row[3:6]
        ok = False
        for a, bs in tet_of.items():
            v = tp[a]
            nrm = np.cross(v[1] - v[0], v[2] - v[1]); nrm /= np.linalg.norm(nrm)
            if abs(nrm @ (r - v[0])) > 1e-12 or np.linalg.norm(nrm - nh) > 1e-12:
                continue
            # inside the triangle
            w = np.linalg.lstsq(np.c_[v[1] - v[0], v[2] - v[0]], r - v[0], rcond=None)[0]
            if w.min() < -1e-10 or w.sum() > 1 + 1e-10:
                continue
            if any(_bary_tet(r, pts[tets[b]]).min() > -1e-10 for b in bs):
                ok = True
                break
        assert ok, row
        n_checked += 1
    assert n_checked >= 300


def test_tet_tet_clip_properties_on_the_device(pfc, O):
    """Volume-volume: clip_plane_tet gives the 3- or 4-gon on the equal-pressure plane inside tet 1, S-H clips it to
    tet 2; the device's traction points must lie in both tets and, per pair, in one plane with the polygon's normal."""
    G = pfc.geometry
    rng = np.random.default_rng(4321)
    p1, t1i, e1 = _random_tets(rng, 30)
    p2, t2i, e2 = _random_tets(rng, 30)
    m1, m2 = G.EMesh(p1, tet=t1i, eps=e1), G.EMesh(p2, tet=t2i, eps=e2)
    trees = (G.build_tree(m1, "median"), G.build_tree(m2, "median"))
    ins = dict(model="regularized", chi=0.0, n_quad=2, mu_s=0.3, mu_d=0.3, v_c=0.01)
    from oracle import oracle as Orc
    pose = Orc.make_pose(np.eye(3), np.zeros(3))
    dev = H.eval_scene("hip", pfc, m1, 1.0e6, m2, 2.0e6, ins, pose, np.zeros(6), None, trees=trees, want_pairs=True)
    ref = H.eval_scene("oracle", pfc, m1, 1.0e6, m2, 2.0e6, ins, pose, np.zeros(6), None, trees=trees, want_pairs=True)
    gp, gc = H.sorted_pairs(dev.pairs, dev.clip_n)
    rp, rc = H.sorted_pairs(ref.pairs, ref.clip_n)
    assert np.array_equal(gp, rp) and np.array_equal(gc, rc)
    assert np.count_nonzero(gc >= 3) > 30
    key = lambda t: np.lexsort(tuple(t[:, c] for c in range(7, -1, -1)))
    assert np.array_equal(dev.trac[key(dev.trac)], ref.trac[key(ref.trac)]), "traction points are not bit-identical"
    pairs_ne = gp[gc >= 3]
    for row in dev.trac[:: max(1, dev.trac.shape[0] // 300)]:
        r = row[3:6]
        assert any(_bary_tet(r, p1[t1i[a]]).min() > -1e-9 and _bary_tet(r, p2[t2i[b]]).min() > -1e-9 for a, b in pairs_ne), row
