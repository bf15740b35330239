"""Pure-Python statement of the median-split OBB tree (test comparator for the library's native builder,
pfc_build_tree).  Test infrastructure: not part of the product package."""
import numpy as np


def build_tree_py(G, m):
    """G: the product's geometry module (EMesh / OBBTree containers and the leaf-box fitters)."""
    EMesh, OBBTree, INTERNAL, fit_tri_obb, fit_tet_obb = G.EMesh, G.OBBTree, G.INTERNAL, G.fit_tri_obb, G.fit_tet_obb
    """Pure-Python statement of the "median" tree (test comparator for pfc_build_tree): recursive_top_down
    (src/geometry/top_down.jl:10-32) over leaf AABBs, every internal box = OBB(child_1.box, child_2.box)
    (src/obb/box_types.jl:11-15), leaves then re-fitted tight (src/geometry/blob_types.jl:170,175-190) unless the
    mesh has a single element (:139-146)."""
    if m.tri is not None and m.tet is not None:
        raise ValueError("Cannot create tree for eMesh{Tri,Tet}; use as_tri_emesh or as_tet_emesh first")
    elem = m.tri if m.tri is not None else m.tet
    n_leaf = elem.shape[0]
    if n_leaf == 0:
        raise ValueError("empty mesh")
    P = m.point[elem]                                   # (n_leaf, k, 3)
    lo, hi = P.min(axis=1), P.max(axis=1)
    lc, le = (hi + lo) * 0.5, (hi - lo) * 0.5           # calc_obb: centre/extent of the leaf AABB

    def union(a, b):                                    # OBB(a, b) for axis-aligned boxes
        mn = np.minimum(np.minimum(a[0] - a[1], a[0] + a[1]), np.minimum(b[0] - b[1], b[0] + b[1]))
        mx = np.maximum(np.maximum(a[0] - a[1], a[0] + a[1]), np.maximum(b[0] - b[1], b[0] + b[1]))
        return (mx + mn) * 0.5, (mx - mn) * 0.5

    def rec(ts):                                        # ts: list of subtrees (box, leaf, children)
        n = len(ts)
        if n == 1:
            return ts[0]
        if n == 2:
            return (union(ts[0][0], ts[1][0]), INTERNAL, (ts[0], ts[1]))
        box = ts[0][0]
        for t in ts:
            box = union(box, t[0])
        ax = int(np.argmax(box[1]))
        perm = np.argsort(np.array([t[0][0][ax] for t in ts]), kind="stable")
        n_mid = -(-n // 2)
        a = rec([ts[i] for i in perm[:n_mid - 1]])
        b = rec([ts[i] for i in perm[n_mid - 1:]])
        return (union(a[0], b[0]), INTERNAL, (a, b))

    import sys
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 10000))
    try:
        root = rec([((lc[i], le[i]), i, None) for i in range(n_leaf)])
    finally:
        sys.setrecursionlimit(old)

    C, E, CH, LF = [], [], [], []
    stack = [(root, -1, 0)]
    while stack:                                        # preorder flattening, node 0 = root
        t, parent, side = stack.pop()
        k = len(C)
        C.append(t[0][0]); E.append(t[0][1]); CH.append([-1, -1]); LF.append(t[1])
        if parent >= 0:
            CH[parent][side] = k
        if t[1] == INTERNAL:
            stack.append((t[2][1], k, 1))
            stack.append((t[2][0], k, 0))
    C, E = np.asarray(C), np.asarray(E)
    R = np.tile(np.eye(3).reshape(9), (len(C), 1))
    CH, LF = np.asarray(CH, dtype=np.int32), np.asarray(LF, dtype=np.int32)
    if n_leaf > 1:                                      # tight_fit_leaves!
        for k in np.nonzero(LF != INTERNAL)[0]:
            i = int(LF[k])
            if m.tri is not None:
                c, e, Rm = fit_tri_obb(m.point[m.tri[i]])
            else:
                c, e, Rm = fit_tet_obb(m.point[m.tet[i]], m.eps[m.tet[i]])
            C[k], E[k], R[k] = c, e, Rm.T.reshape(9)    # column-major
    return OBBTree(np.ascontiguousarray(C), np.ascontiguousarray(E), np.ascontiguousarray(R), CH, LF)
