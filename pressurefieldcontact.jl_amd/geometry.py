"""Host-side mesh authoring and OBB-tree construction (build time, not on the hot path).

This is the host-side mirror of what the reference's ``Geometry`` / ``Binary_BB_Trees`` modules hand to the
contact hot path: an ``eMesh`` (points, tri or tet indices, per-vertex normalised penetration extent eps) and a
binary OBB tree whose internal boxes are axis aligned and whose leaves are tight-fitted oriented boxes.

Reference conventions followed (all paths relative to /root/reference):
  * eMesh fields and invariants ............ src/geometry/mesh.jl:10-46  (eps = 0 on the surface, > 0 inside,
                                             every tet has positive ``volume``)
  * half plane / box / sphere primitives ... src/geometry/mesh.jl:430-442, :527-575, :449-525
  * triangle subdivision ................... src/geometry/mesh.jl:362-421
  * surface-of-a-tet-mesh .................. src/geometry/mesh.jl:63-78 (drop the largest-|eps| vertex)
  * median-split top-down tree ............. src/geometry/top_down.jl:10-32
  * merged (internal) boxes are AABBs ...... src/obb/box_types.jl:11-15, src/obb/util.jl:17-51
  * tight leaf boxes ....................... src/obb/obb_construction.jl:13-41, src/geometry/blob_types.jl:175-190
  * single-element mesh keeps an AABB leaf . src/geometry/blob_types.jl:139-146

The reference's agglomerative "blob" builder (src/geometry/blob_types.jl:136-173) depends on Julia Dict /
PriorityQueue iteration order and is not reproducible; only its top-down fallback is restated here.  The tree
is an *input* of the hot path (the Julia host would pass its own), so candidate sets are compared between the
oracle and the HIP path on the same tree, never against a particular builder.

Indices are 0-based here (the reference is 1-based).  The flattened tree uses ``leaf == INTERNAL`` (-9999, the
reference's sentinel, src/obb/tree_types.jl:11,56) for internal nodes and the element index for leaves.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

INTERNAL = -9999


# ----------------------------------------------------------------------------------------------------------------
# eMesh
# ----------------------------------------------------------------------------------------------------------------
def tet_volume(p: np.ndarray) -> np.ndarray:
    """Signed tet volume, same algebraic form as src/math_kernel/geometry_kernel.jl:22-38.  p: (...,4,3)."""
    a, b, c, d = p[..., 0, :], p[..., 1, :], p[..., 2, :], p[..., 3, :]
    a1, a2, a3 = a[..., 0], a[..., 1], a[..., 2]
    b1, b2, b3 = b[..., 0], b[..., 1], b[..., 2]
    c1, c2, c3 = c[..., 0], c[..., 1], c[..., 2]
    d1, d2, d3 = d[..., 0], d[..., 1], d[..., 2]
    v = (b1 - a1) * (c2 * d3 - c3 * d2)
    v = (b2 - a2) * (c3 * d1 - c1 * d3) + v
    v = (b3 - a3) * (c1 * d2 - c2 * d1) + v
    v = (c1 - d1) * (a3 * b2 - a2 * b3) + v
    v = (c2 - d2) * (a1 * b3 - a3 * b1) + v
    v = (c3 - d3) * (a2 * b1 - a1 * b2) + v
    return v * (1.0 / 6.0)


@dataclass
class EMesh:
    """eMesh{Tri?,Tet?}: src/geometry/mesh.jl:10-46."""
    point: np.ndarray                      # (n_pt, 3) float64
    tri: Optional[np.ndarray] = None       # (n_tri, 3) int32, 0-based
    tet: Optional[np.ndarray] = None       # (n_tet, 4) int32, 0-based
    eps: Optional[np.ndarray] = None       # (n_pt,) float64

    def __post_init__(self):
        self.point = np.ascontiguousarray(self.point, dtype=np.float64).reshape(-1, 3)
        if self.tri is not None:
            self.tri = np.ascontiguousarray(self.tri, dtype=np.int32).reshape(-1, 3)
        if self.tet is not None:
            self.tet = np.ascontiguousarray(self.tet, dtype=np.int32).reshape(-1, 4)
            if self.eps is None:
                raise ValueError("a tet eMesh needs eps")
            self.eps = np.ascontiguousarray(self.eps, dtype=np.float64).reshape(-1)
            if self.eps.shape[0] != self.point.shape[0]:
                raise ValueError("length(eps) != length(point)")
            if self.eps.size:
                if not (0.0 < self.eps.max()):
                    raise ValueError("normalized penetration extent must be non-negative")
                if self.eps.min() != 0.0:
                    raise ValueError("normalized penetration extent must be zero on the surface")
            if self.tet.shape[0] and not np.all(0.0 < tet_volume(self.point[self.tet])):
                raise ValueError("inverted tetrahedron")
        elif self.eps is not None:
            raise ValueError("eps given without tets")
        if self.tri is None and self.tet is None:
            raise ValueError("a whole lot of nothing")

    @property
    def n_point(self): return self.point.shape[0]
    @property
    def n_tri(self): return 0 if self.tri is None else self.tri.shape[0]
    @property
    def n_tet(self): return 0 if self.tet is None else self.tet.shape[0]

    def transformed(self, R: Optional[np.ndarray] = None, t: Optional[np.ndarray] = None) -> "EMesh":
        p = self.point
        if R is not None:
            p = p @ np.asarray(R, dtype=np.float64).T
        if t is not None:
            p = p + np.asarray(t, dtype=np.float64)
        return EMesh(p, None if self.tri is None else self.tri.copy(),
                     None if self.tet is None else self.tet.copy(),
                     None if self.eps is None else self.eps.copy())


def tet_perm_by_num(n: int) -> tuple:
    """src/obb/util.jl:60-66 (0-based): even permutation that moves vertex n last."""
    return ((1, 3, 2, 0), (3, 0, 2, 1), (0, 3, 1, 2), (0, 1, 2, 3))[n]


def as_tet_emesh(m: EMesh) -> EMesh:
    return EMesh(m.point.copy(), None, m.tet.copy(), m.eps.copy())


def as_tri_emesh(m: EMesh) -> EMesh:
    """src/geometry/mesh.jl:61-78.  For a Tri+Tet mesh keep the triangles; for a tet-only mesh take, per tet,
    the face opposite the largest-|eps| vertex and delete opposing duplicate pairs (mesh_repair!, :280-361)."""
    if m.tri is not None:
        return EMesh(m.point.copy(), m.tri.copy(), None, None)
    tris = []
    for k in range(m.n_tet):
        it = m.tet[k]
        imax = int(np.argmax(np.abs(m.eps[it])))
        perm = tet_perm_by_num(imax)
        tris.append([it[perm[0]], it[perm[1]], it[perm[2]]])
    tris = np.asarray(tris, dtype=np.int32)
    key = {}
    for k, t in enumerate(tris):
        key.setdefault(tuple(sorted(int(v) for v in t)), []).append(k)
    drop = [k for ks in key.values() if len(ks) == 2 for k in ks]
    keep = np.setdiff1d(np.arange(len(tris)), np.asarray(drop, dtype=np.int64))
    return EMesh(m.point.copy(), tris[keep], None, None)


def emesh_half_plane(plane_w: float = 1.0) -> EMesh:
    """src/geometry/mesh.jl:430-442: one tet, top face = unit-circle triangle at z=0, apex at -plane_w."""
    th = (0.0, 2 * np.pi / 3, 4 * np.pi / 3)
    pts = [[np.cos(t), np.sin(t), 0.0] for t in th] + [[0.0, 0.0, -1.0 * plane_w]]
    tri = [[0, 1, 2]]
    tet = [[3, 0, 1, 2]]
    eps = [0.0, 0.0, 0.0, plane_w]
    return EMesh(np.array(pts), np.array(tri), np.array(tet), np.array(eps))


_BOX_FACES = ((0, 2, 4, 6), (1, 5, 3, 7), (0, 4, 1, 5), (2, 3, 6, 7), (0, 1, 2, 3), (4, 6, 5, 7))


def _box_tri() -> np.ndarray:
    """src/geometry/mesh.jl:527-548 (output_box_ind): two outward triangles per face."""
    tri = []
    for f in _BOX_FACES:
        tri.append([f[0], f[2], f[3]])
        tri.append([f[0], f[3], f[1]])
    return np.asarray(tri, dtype=np.int32)


def emesh_box(r=1.0, c=(0.0, 0.0, 0.0)) -> EMesh:
    """src/geometry/mesh.jl:550-575: 8 corners + centre; 12 surface triangles; 12 tets (centre, tri)."""
    r = np.ones(3) * np.asarray(r, dtype=np.float64)
    pts = np.array([[-1, -1, -1], [1, -1, -1], [-1, 1, -1], [1, 1, -1],
                    [-1, -1, 1], [1, -1, 1], [-1, 1, 1], [1, 1, 1], [0, 0, 0]], dtype=np.float64)
    tri = _box_tri()
    tet = np.concatenate([np.full((12, 1), 8, dtype=np.int32), tri], axis=1)
    eps = np.array([0.0] * 8 + [1.0])
    return EMesh(pts * r + np.asarray(c, dtype=np.float64), tri, tet, eps)


def _sub_div_triangle(p: np.ndarray, n_div: int):
    """src/geometry/mesh.jl:366-411: barycentric lattice of one triangle, same vertex/triangle enumeration."""
    def n_end(n): return (n + 1) * n // 2
    def n_start(n): return 1 + n_end(n - 1)
    tri = []
    for k in range(1, n_div + 1):
        for kk in range(k):
            i1 = n_start(k) + kk
            i2 = i1 + k
            tri.append((i1, i2, i2 + 1))
        for kk in range(k - 1):
            i1 = n_start(k) + kk
            i2 = i1 + k + 1
            tri.append((i1, i2, i2 - k))
    pts = []
    for n_vert in range(1, n_end(n_div + 1) + 1):
        i_end_layer, i_layer = 1, 1
        while i_end_layer < n_vert:
            i_layer += 1
            i_end_layer += i_layer
        ext = 0.0 if n_vert == 1 else (i_end_layer - n_vert) / (i_layer - 1)
        f1 = (n_div - i_layer + 1) / n_div
        f2 = (1 - f1) * ext
        f3 = 1 - f1 - f2
        pts.append(p[0] * f1 + p[1] * f2 + p[2] * f3)
    return np.asarray(pts), np.asarray(tri, dtype=np.int64) - 1


def sub_div_surface(point: np.ndarray, tri: np.ndarray, n_div: int):
    """src/geometry/mesh.jl:362-421 (sub_div_mesh): subdivide every triangle n_div x n_div and merge coincident
    points (the reference merges with a BallTree in mesh_repair!; here by a rounded-coordinate key)."""
    all_p, all_t, off = [], [], 0
    for t in tri:
        p, tt = _sub_div_triangle(point[t], n_div)
        all_p.append(p)
        all_t.append(tt + off)
        off += p.shape[0]
    P = np.concatenate(all_p)
    T = np.concatenate(all_t)
    scale = np.abs(P).max() if P.size else 1.0
    key = np.round(P / (scale * 1e-9)).astype(np.int64)
    _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")          # keep first-seen ordering of the merged points
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    return P[first[order]], rank[inv.reshape(-1)][T].astype(np.int32)


def volumize_about(point: np.ndarray, tri: np.ndarray, centre=(0.0, 0.0, 0.0)) -> EMesh:
    """src/geometry/mesh.jl:497-508: one tet (centre, tri) per surface triangle, eps = 1 at the centre."""
    n = point.shape[0]
    pts = np.concatenate([point, np.asarray(centre, dtype=np.float64).reshape(1, 3)])
    tet = np.concatenate([np.full((tri.shape[0], 1), n, dtype=np.int32), tri.astype(np.int32)], axis=1)
    eps = np.concatenate([np.zeros(n), [1.0]])
    return EMesh(pts, tri.astype(np.int32), tet, eps)


def _icosahedron():
    """src/geometry/mesh.jl:450-495 (make_icosahedron): 12 vertices, 20 outward faces."""
    phi = (1 + np.sqrt(5.0)) / 2
    v = []
    for s1 in (-1.0, 1.0):
        for s2 in (-1.0, 1.0):
            v.append([0.0, s1, phi * s2])
            v.append([s1, phi * s2, 0.0])
            v.append([phi * s2, 0.0, s1])
    v = np.asarray(v)
    d = np.linalg.norm(v[:, None, :] - v[None, :, :], axis=2)
    b = d == 2.0
    faces = []
    for i1 in range(12):
        for i2 in range(i1 + 1, 12):
            for i3 in range(i2 + 1, 12):
                if b[i1, i2] and b[i2, i3] and b[i1, i3]:
                    n = np.cross(v[i2] - v[i1], v[i3] - v[i2])
                    c = v[i1] + v[i2] + v[i3]
                    faces.append((i1, i2, i3) if np.dot(n, c) > 0 else (i1, i3, i2))
    return v, np.asarray(faces, dtype=np.int32)


def emesh_sphere(rad=1.0, n_div: int = 4) -> EMesh:
    """src/geometry/mesh.jl:449-525: subdivided icosahedron projected to the sphere, 20*n_div^2 triangles,
    volumised about the centre."""
    v, f = _icosahedron()
    P, T = sub_div_surface(v, f, n_div)
    P = P / np.linalg.norm(P, axis=1, keepdims=True)
    P = P * (np.ones(3) * np.asarray(rad, dtype=np.float64))
    return volumize_about(P, T)


def emesh_box_div(r=1.0, n_div: int = 1, c=(0.0, 0.0, 0.0)) -> EMesh:
    """Box whose 12 surface triangles are each subdivided n_div x n_div (12*n_div^2 tets sharing the centre
    apex).  Same convention as emesh_box (eps = 0 on the surface, 1 at the centre); n_div = 1 is emesh_box up
    to vertex order.  Used for the BASELINE.md C2/C4/C5 synthetic bodies."""
    base = emesh_box(1.0)
    P, T = sub_div_surface(base.point[:8], base.tri, n_div)
    r = np.ones(3) * np.asarray(r, dtype=np.float64)
    m = volumize_about(P * r, T)
    return m.transformed(t=np.asarray(c, dtype=np.float64)) if np.any(np.asarray(c) != 0) else m


# ---- swept meshes (src/geometry/mesh_create_swept.jl) and mesh repair (src/geometry/mesh.jl:235-361) -------------------
def _tri_area(p: np.ndarray) -> np.ndarray:
    return 0.5 * np.linalg.norm(np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 1]), axis=1)


def _remove_unused_points(point, tri, tet, eps):
    """mesh_remove_unused_points! (mesh.jl:302-319): points keep their order."""
    used = np.zeros(point.shape[0], dtype=bool)
    for e in (tri, tet):
        if e is not None and e.size:
            used[e.reshape(-1)] = True
    new_key = np.cumsum(used) - 1
    tri = None if tri is None else new_key[tri]
    tet = None if tet is None else new_key[tet]
    return point[used], tri, tet, (None if eps is None else eps[used])


def mesh_repair(point, tri, tet, eps):
    """mesh_repair! (mesh.jl:235-240): drop unused points, merge points closer than 0.499 x the shortest element side onto the
    lowest index of their cluster (mesh_inplace_rekey!, :266-300; the reference asks a BallTree, here a k-d tree), drop unused
    points again, delete opposing duplicate triangle pairs (delete_triangles!, :322-361).  Returns (point, tri, tet, eps)."""
    from scipy.spatial import cKDTree
    point, tri, tet, eps = _remove_unused_points(point, tri, tet, eps)
    d_min = np.inf
    for e in (tri, tet):
        if e is not None and e.size:
            v = point[e]
            for a in range(e.shape[1]):
                for b in range(a):
                    d_min = min(d_min, float(np.linalg.norm(v[:, a] - v[:, b], axis=1).min()))
    if np.isfinite(d_min):
        near = cKDTree(point).query_ball_point(point, d_min * 0.499)
        new_key = np.asarray([min(ix) for ix in near], dtype=np.int64)
        tri = None if tri is None else new_key[tri]
        tet = None if tet is None else new_key[tet]
        point, tri, tet, eps = _remove_unused_points(point, tri, tet, eps)
    if tri is not None and tri.size:
        seen = {}
        for k, t in enumerate(tri):
            seen.setdefault(tuple(sorted(int(v) for v in t)), []).append(k)
        drop = []
        for ks in seen.values():
            if len(ks) == 2:
                drop += ks
            elif len(ks) >= 3:
                raise ValueError("something is wrong")
        if drop:
            tri = np.delete(tri, np.asarray(sorted(drop)), axis=0)
    return point, tri, tet, eps


def f_swept_triv(theta: float):
    """mesh_create_swept.jl:20-24: a straight path along +y; returns (position, a normal of the path, its direction)."""
    n1 = np.array([0.0, 0.0, -1.0]); n2 = np.array([0.0, 1.0, 0.0])
    return n2 * theta, n1, n2


def _angle_axis(phi: float, axis: np.ndarray, v: np.ndarray) -> np.ndarray:
    """AngleAxis(phi, axis...) * v (Rodrigues)."""
    k = axis / np.linalg.norm(axis)
    return v * np.cos(phi) + np.cross(k, v) * np.sin(phi) + k * np.dot(k, v) * (1.0 - np.cos(phi))


def create_swept_mesh(fun_gen, lr, rad, n_side: int = 4, is_open: bool = True, rot_half: bool = True) -> EMesh:
    """create_swept_mesh (mesh_create_swept.jl:73-104) with add_rot_sym_segment! (:25-58): per path segment and side one
    7-point wedge -- 2 surface triangles (+ an end cap where the path is open), 4 tets, eps = 1 on the path and 0 on the
    surface and at open ends -- then remove_degenerate! (mesh.jl:242-255: elements below 1e-6 of the largest go, e.g. the
    zero-radius tip of test/pencil.jl:199) and mesh_repair!."""
    lr = np.asarray(lr, dtype=np.float64)
    rad = np.zeros(lr.size) + np.asarray(rad, dtype=np.float64)
    if rad.size != lr.size:
        raise ValueError("the length of lr and length of rad must be the same")
    d_phi = 2 * np.pi / n_side
    rad = rad / np.cos(d_phi / 2)
    pts, tri, tet, eps = [], [], [], []
    n_th = lr.size - 1
    for k_th in range(n_th):
        for k_phi in range(1, n_side + 1):
            phi0 = d_phi * (k_phi - 0.5 * (1.0 if rot_half else 0.0))
            phi1 = phi0 + d_phi
            open0, open1 = (is_open and k_th == 0), (is_open and k_th == n_th - 1)
            p1, x1, y1 = fun_gen(float(lr[k_th]))
            p2, x2, y2 = fun_gen(float(lr[k_th + 1]))
            p3 = (p1 + p2) * 0.5
            p4 = p1 + _angle_axis(phi0, y1, x1) * rad[k_th]
            p6 = p1 + _angle_axis(phi1, y1, x1) * rad[k_th]
            p5 = p2 + _angle_axis(phi0, y2, x2) * rad[k_th + 1]
            p7 = p2 + _angle_axis(phi1, y2, x2) * rad[k_th + 1]
            o = len(pts) - 1                      # the reference's indices are 1-based
            pts += [p1, p2, p3, p4, p5, p6, p7]
            tet += [[o + 1, o + 3, o + 4, o + 6], [o + 3, o + 2, o + 5, o + 7], [o + 3, o + 4, o + 6, o + 7], [o + 4, o + 3, o + 5, o + 7]]
            tri += [[o + 4, o + 6, o + 7], [o + 4, o + 7, o + 5]]
            e = [1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0]
            if open0:
                e[0] = 0.0
                tri.append([o + 1, o + 6, o + 4])
            if open1:
                e[1] = 0.0
                tri.append([o + 2, o + 5, o + 7])
            eps += e
    P = np.asarray(pts); T = np.asarray(tri, dtype=np.int64); Q = np.asarray(tet, dtype=np.int64); E = np.asarray(eps)
    vol = tet_volume(P[Q]); Q = Q[~(vol < vol.max() * 1.0e-6)]
    ar = _tri_area(P[T]); T = T[~(ar < ar.max() * 1.0e-6)]
    P, T, Q, E = mesh_repair(P, T, Q, E)
    return EMesh(P, T, Q, E)


def emesh_from_quads(point: np.ndarray, quad: np.ndarray, scale: float = 1.0) -> EMesh:
    """A surface eMesh from a quad mesh file's data (test/spoon.jl:39-41 loads test/data/spoon.obj through MeshIO into a
    triangle mesh and scales it by 0.01): every quad (a, b, c, d) becomes the triangles (a, b, c), (a, c, d)."""
    quad = np.asarray(quad, dtype=np.int64).reshape(-1, 4)
    tri = np.concatenate([quad[:, [0, 1, 2]], quad[:, [0, 2, 3]]], axis=1).reshape(-1, 3)
    return EMesh(np.asarray(point, dtype=np.float64) * scale, tri, None, None)


def emesh_ground(half_w: float = 1.0, z: float = 0.0) -> EMesh:
    """Open rigid ground patch: 2 triangles forming a (2*half_w)^2 square at height z, normal +z
    (BASELINE.md C2).  The reference's blob builder cannot build a tree for an open mesh
    (src/geometry/blob_types.jl:62-69,156); build_tree(..., "median") can."""
    p = np.array([[-half_w, -half_w, z], [half_w, -half_w, z], [half_w, half_w, z], [-half_w, half_w, z]])
    return EMesh(p, np.array([[0, 1, 2], [0, 2, 3]]), None, None)


# ----------------------------------------------------------------------------------------------------------------
# OBB fitting (leaf boxes)
# ----------------------------------------------------------------------------------------------------------------
def _normalize(v):
    return v * (1.0 / np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]))


def _vector_area(a, b, c):
    return np.cross(b - a, c - b) * 0.5


def make_obb(p: np.ndarray, i_start: int):
    """src/obb/obb_construction.jl:13-26.  p: (3|4, 3); i_start is 0-based in {0,1,2}.  Returns (c, e, R) with
    R's columns = box axes (e1 = an edge of the first three vertices, e3 = their normal)."""
    e1 = _normalize(p[(i_start + 1) % 3] - p[i_start])
    e3 = _normalize(_vector_area(p[0], p[1], p[2]))
    e2 = np.cross(e3, e1)
    pr = np.stack([p @ e1, p @ e2, p @ e3], axis=1)
    pmin, pmax = pr.min(axis=0), pr.max(axis=0)
    c = (pmax + pmin) * 0.5
    e = (pmax - pmin) * 0.5
    R = np.stack([e1, e2, e3], axis=1)
    return R @ c, e, R


def obb_area(e):
    """src/obb/extensions.jl:2."""
    return 8 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0])


def fit_tri_obb(p: np.ndarray):
    return make_obb(p, 0)


def fit_tet_obb(p: np.ndarray, eps_tet: np.ndarray):
    """src/obb/obb_construction.jl:29-41: permute so the largest-|eps| vertex is last, try the three base
    edges, keep the box with the LARGEST surface area (sic, :35-40)."""
    if not (0.0 < tet_volume(p)):
        raise ValueError("inverted tet")
    p = p[list(tet_perm_by_num(int(np.argmax(np.abs(eps_tet)))))]
    boxes = [make_obb(p, k) for k in range(3)]
    a = [obb_area(b[1]) for b in boxes]
    if max(a[1], a[2]) <= a[0]:
        return boxes[0]
    if max(a[0], a[2]) <= a[1]:
        return boxes[1]
    return boxes[2]


# ----------------------------------------------------------------------------------------------------------------
# Flattened binary OBB tree
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class OBBTree:
    """Flattened bin_BB_Tree (src/obb/tree_types.jl:1-16).  Node 0 is the root.  R is stored column-major
    (R[:, 0:3] = first column) exactly like the linear indexing R[1..9] in src/obb/bb_intersection.jl:21-26."""
    c: np.ndarray        # (n, 3)
    e: np.ndarray        # (n, 3)
    R: np.ndarray        # (n, 9) column-major
    child: np.ndarray    # (n, 2) int32, -1 for leaves
    leaf: np.ndarray     # (n,) int32, element index, INTERNAL for internal nodes

    @property
    def n_node(self): return self.c.shape[0]
    @property
    def n_leaf(self): return int(np.sum(self.leaf != INTERNAL))

    def depth(self) -> int:
        d = np.zeros(self.n_node, dtype=np.int64)
        best = 0
        for k in range(self.n_node):          # preorder: parents precede children
            if self.leaf[k] == INTERNAL:
                d[self.child[k, 0]] = d[k] + 1
                d[self.child[k, 1]] = d[k] + 1
            best = max(best, int(d[k]))
        return best


def build_tree(m: EMesh, method: str = "blob") -> OBBTree:
    """eMesh_to_tree (src/geometry/blob_types.jl:136-173) through the library's native builder (pfc_build_tree,
    csrc/pfc_tree.cpp).  method "blob": bottom-up blob merging + top-down over the remaining blobs (the reference's
    builder); "median": the pure median-split top-down tree (src/geometry/top_down.jl:10-32)."""
    import ctypes as C
    from . import _lib
    if m.tri is not None and m.tet is not None:
        raise ValueError("Cannot create tree for eMesh{Tri,Tet}; use as_tri_emesh or as_tet_emesh first")
    elem = m.tri if m.tri is not None else m.tet
    if elem is None or elem.shape[0] == 0:
        raise ValueError("empty mesh")
    code = {"blob": 0, "median": 1}[method]
    pt = np.ascontiguousarray(m.point, dtype=np.float64)
    el = np.ascontiguousarray(elem, dtype=np.int32)
    eps = None if m.eps is None else np.ascontiguousarray(m.eps, dtype=np.float64)
    n = 2 * el.shape[0] - 1
    c, e, R = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 9))
    child, leaf = np.zeros((n, 2), dtype=np.int32), np.zeros(n, dtype=np.int32)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    L = _lib.lib()
    rc = L.pfc_build_tree(pt.shape[0], pt.ctypes.data_as(dp), el.shape[0], el.shape[1], el.ctypes.data_as(ip),
                          None if eps is None else eps.ctypes.data_as(dp), code, c.ctypes.data_as(dp),
                          e.ctypes.data_as(dp), R.ctypes.data_as(dp), child.ctypes.data_as(ip), leaf.ctypes.data_as(ip))
    if rc < 0:
        raise ValueError(L.pfc_tree_last_error().decode())
    assert rc == n
    return OBBTree(c, e, R, child, leaf)
