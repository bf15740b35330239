"""the library's native tree builder (pfc_build_tree, csrc/pfc_tree.cpp; host code, runs without a
GPU) against the pure-Python statement of recursive_top_down and against the structural properties the reference's
eMesh_to_tree guarantees (src/geometry/blob_types.jl:136-190, src/geometry/top_down.jl:10-32)."""
import numpy as np
import pytest

from helpers import oracle_run, sorted_pairs
from tree_reference import build_tree_py


def _meshes(g):
    S = g.emesh_sphere(0.1, 6)
    B = g.emesh_box_div(np.array([0.05, 0.04, 0.03]), 5)
    return [g.as_tet_emesh(S), g.as_tri_emesh(S), g.as_tet_emesh(B), g.as_tri_emesh(B)]


def _leaf_candidates(g, m, i):
    if m.tri is not None:
        return [g.make_obb(m.point[m.tri[i]], 0)]
    p = m.point[m.tet[i]][list(g.tet_perm_by_num(int(np.argmax(np.abs(m.eps[m.tet[i]])))))]
    return [g.make_obb(p, k) for k in range(3)]


def _check_leaf_boxes(g, m, t):
    """Every leaf box is one of make_obb's candidates and has the largest area among them (obb_construction.jl:29-41);
    which of several equal-area candidates wins is decided by the last bit and is not compared."""
    for k in np.nonzero(t.leaf != g.INTERNAL)[0]:
        cands = _leaf_candidates(g, m, int(t.leaf[k]))
        areas = [g.obb_area(c[1]) for c in cands]
        hit = [j for j, (c, e, R) in enumerate(cands)
               if np.allclose(t.c[k], c, atol=1e-13) and np.allclose(t.e[k], e, atol=1e-13)
               and np.allclose(t.R[k].reshape(3, 3, order="F"), R, atol=1e-12)]
        assert hit, f"leaf {k}: box is none of the make_obb candidates"
        assert areas[hit[0]] >= max(areas) * (1 - 1e-12)


def test_native_median_tree_matches_python_statement(pfc):
    g = pfc.geometry
    for m in _meshes(g):
        tn, tp = g.build_tree(m, "median"), build_tree_py(g, m)
        assert np.array_equal(tn.child, tp.child) and np.array_equal(tn.leaf, tp.leaf)
        internal = tn.leaf == g.INTERNAL
        np.testing.assert_array_equal(tn.c[internal], tp.c[internal])      # same arithmetic, same order
        np.testing.assert_array_equal(tn.e[internal], tp.e[internal])
        np.testing.assert_array_equal(tn.R[internal], tp.R[internal])
        _check_leaf_boxes(g, m, tn)


@pytest.mark.parametrize("method", ["blob", "median"])
def test_tree_structure(pfc, method):
    g = pfc.geometry
    for m in _meshes(g):
        t = g.build_tree(m, method)
        elem = m.tri if m.tri is not None else m.tet
        n = elem.shape[0]
        assert t.n_node == 2 * n - 1
        leaves = t.leaf[t.leaf != g.INTERNAL]
        assert np.array_equal(np.sort(leaves), np.arange(n))                # every element exactly once
        internal = np.nonzero(t.leaf == g.INTERNAL)[0]
        assert np.all(t.child[internal] > internal[:, None])                # preorder: parents precede children
        assert np.all(t.child[t.leaf != g.INTERNAL] == -1)
        # containment: every element vertex lies in every ancestor box (internal boxes are axis aligned)
        lo, hi = np.full((t.n_node, 3), np.inf), np.full((t.n_node, 3), -np.inf)
        for k in range(t.n_node - 1, -1, -1):
            if t.leaf[k] != g.INTERNAL:
                P = m.point[elem[t.leaf[k]]]
                lo[k], hi[k] = P.min(axis=0), P.max(axis=0)
            else:
                a, b = t.child[k]
                lo[k], hi[k] = np.minimum(lo[a], lo[b]), np.maximum(hi[a], hi[b])
                assert np.all(t.c[k] - t.e[k] <= lo[k] + 1e-15) and np.all(t.c[k] + t.e[k] >= hi[k] - 1e-15)
                np.testing.assert_allclose(t.c[k] - t.e[k], lo[k], atol=1e-15)   # and are tight
                np.testing.assert_allclose(t.c[k] + t.e[k], hi[k], atol=1e-15)
        _check_leaf_boxes(g, m, t)


def test_blob_tree_quality_and_balance(pfc):
    """The bottom-up phase exists to beat the plain median split: smaller summed internal surface area, and the
    n*log2(2n) term of blobCost (:74-82) keeps the depth near log2(n)."""
    g = pfc.geometry
    m = g.as_tet_emesh(g.emesh_sphere(0.1, 10))
    tb, tm = g.build_tree(m, "blob"), g.build_tree(m, "median")
    def sa(t):
        e = t.e[t.leaf == g.INTERNAL]
        return float((8 * (e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 2] * e[:, 0])).sum())
    assert sa(tb) < sa(tm)
    assert tb.depth() <= 2 * int(np.ceil(np.log2(m.tet.shape[0])))


def test_single_element_and_disconnected(pfc):
    g = pfc.geometry
    pt = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]])
    one = g.EMesh(pt, None, np.array([[0, 1, 2, 3]], dtype=np.int32), np.array([0, 0, 0, 1.0]))
    t = g.build_tree(one)
    assert t.n_node == 1 and t.leaf[0] == 0
    np.testing.assert_array_equal(t.R[0], np.eye(3).reshape(9))             # keeps its AABB (:139-146)
    np.testing.assert_allclose(t.c[0], [0.5, 0.5, 0.5]); np.testing.assert_allclose(t.e[0], [0.5, 0.5, 0.5])
    # two tet islands: the queue empties with two blobs, recursive_top_down joins them (:160-164)
    two = g.EMesh(np.vstack([pt, pt + 5.0]), None, np.array([[0, 1, 2, 3], [4, 5, 6, 7]], dtype=np.int32),
                  np.array([0, 0, 0, 1.0, 0, 0, 0, 1.0]))
    t = g.build_tree(two)
    assert t.n_node == 3 and sorted(t.leaf[1:].tolist()) == [0, 1]


def test_tree_errors_follow_reference(pfc):
    g = pfc.geometry
    pt = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0], [1, 1, 1.0]])
    open_surface = g.EMesh(pt, np.array([[0, 1, 2], [0, 1, 3]], dtype=np.int32), None, None)
    with pytest.raises(ValueError, match="disconnected mesh"):
        g.build_tree(open_surface)
    fan = g.EMesh(pt, np.array([[0, 1, 2], [0, 1, 3], [1, 0, 4]], dtype=np.int32), None, None)
    with pytest.raises(ValueError, match="three triangles share the same edge"):
        g.build_tree(fan)
    tets = np.array([[0, 1, 2, 3], [0, 2, 1, 3]], dtype=np.int32)          # second one inverted
    with pytest.raises(ValueError, match="inverted"):
        g.build_tree(g.EMesh(pt, None, tets, np.array([0, 0, 0, 1.0, 0])))
    assert g.build_tree(open_surface, "median").n_node == 3                  # pure top-down needs no adjacency


def test_contact_result_independent_of_tree(pfc, O):
    """Candidate lists depend on the tree, the contact does not: same intersecting pairs, same wrench."""
    cfg = pfc.configs
    res = {}
    for method in ("blob", "median"):
        w = cfg.c3_blob_tool(1, seed=5, n_div_blob=6, n_div_tool=4)
        for ms in w.meshes:
            ms.tree = pfc.geometry.build_tree(ms.mesh, method)
        out = oracle_run(pfc, w, [0])[0]
        pairs, clip_n = sorted_pairs(out.pairs, out.clip_n)
        res[method] = (out.wrench, pairs[clip_n > 0])
    a, b = res["blob"], res["median"]
    assert np.array_equal(a[1], b[1]) and a[1].shape[0] > 0
    np.testing.assert_allclose(a[0], b[0], rtol=1e-11, atol=1e-11 * np.abs(b[0]).max())


def test_reference_spoon_and_swept_pencil_meshes(pfc):
    """The reference's real test geometry: test/data/spoon.obj (fixture tests/golden/spoon_quads.npz: vertex / face data only) is a
    closed, outward-oriented manifold of 2 502 quads; the pencil of test/pencil.jl:198-200 (create_swept_mesh, 12 sides, two
    segments, zero-radius tip) has fewer than 100 surface triangles and is closed too.  Both go through the
    blob and the median tree builder (pfc_build_tree), whose leaves are the triangles."""
    from collections import Counter
    G, Cf = pfc.geometry, pfc.configs
    spoon = Cf.spoon_emesh()
    pen = Cf.pencil_emesh()
    pen_tri = G.as_tri_emesh(pen)
    assert (spoon.n_point, spoon.n_tri) == (2504, 5004)
    assert (pen.n_point, pen_tri.n_tri, pen.n_tet) == (29, 48, 72) and pen_tri.n_tri < 100
    assert pen.eps.min() == 0.0 and pen.eps.max() == 1.0
    for msh, vol_ref in ((spoon, None), (pen_tri, np.pi * 0.0035 ** 2 * (0.147 + 0.013 / 3))):
        e = Counter()
        for t in msh.tri:
            for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
                e[(int(a), int(b))] += 1
        assert max(e.values()) == 1 and all((b, a) in e for (a, b) in e), "not a closed, consistently oriented manifold"
        P = msh.point[msh.tri]
        vol = np.einsum("ij,ij->i", P[:, 0], np.cross(P[:, 1], P[:, 2])).sum() / 6.0
        assert vol > 0.0
        if vol_ref is not None:      # a 12-gon circumscribing the circle of radius r: area factor n tan(pi / n) / pi
            assert abs(vol / (vol_ref * 12 * np.tan(np.pi / 12) / np.pi) - 1.0) < 1e-9
        for method in ("blob", "median"):
            tr = G.build_tree(msh, method)
            assert tr.n_node == 2 * msh.n_tri - 1
            assert sorted(int(v) for v in tr.leaf[tr.leaf >= 0]) == list(range(msh.n_tri))
