"""The reference's own analytic known-answer and property tests, restated against the CPU oracle.

These pin oracle/pfc_oracle.c to the reference: the reference is pure Julia, there is no Julia
toolchain here and no golden vectors exist in its tree, so every pin is analytic or a property, exactly as in the
reference's test suite.  Each test names the reference test it restates (paths relative to the reference repo).
"""
import ctypes as C

import numpy as np
import pytest

dp = C.POINTER(C.c_double)


def P(a):
    return np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)


# ---------------------------------------------------------------------------------------------------------------
# test/test_friction.jl:17-31  calc_clamped_piecewise
# ---------------------------------------------------------------------------------------------------------------
def test_calc_clamped_piecewise(O):
    f = O.lib().pfo_calc_clamped_piecewise
    x1, x2, y1, y2 = 0.3, 0.5, 1.1, 0.1
    eps = np.finfo(float).eps
    assert f(x1 - 0.1, x1, x2, y1, y2) == pytest.approx(y1)
    assert f(x1, x1, x2, y1, y2) == pytest.approx(y1)
    assert f(x1 + 10 * eps, x1, x2, y1, y2) == pytest.approx(y1)
    assert f((x1 + x2) / 2, x1, x2, y1, y2) == pytest.approx((y1 + y2) / 2)
    assert f(x2 - 10 * eps, x1, x2, y1, y2) == pytest.approx(y2)
    assert f(x2, x1, x2, y1, y2) == pytest.approx(y2)
    assert f(x2 + 0.1, x1, x2, y1, y2) == pytest.approx(y2)


# ---------------------------------------------------------------------------------------------------------------
# test/test_friction.jl:33-90  traction laws against the independent piecewise re-derivation
# ---------------------------------------------------------------------------------------------------------------
def _verify_bri(mu_s, mu_d, Ts, p_dA):
    m = np.linalg.norm(Ts)
    T_s, T_d = mu_s * Ts / m * p_dA, mu_d * Ts / m * p_dA
    if m <= mu_s:
        return Ts * p_dA
    if m <= 2 * mu_s:
        return T_s
    if 3 * mu_s <= m:
        return T_d
    wd = (m - 2 * mu_s) / (3 * mu_s - 2 * mu_s)
    return T_s * (1 - wd) + T_d * wd


def _verify_reg(mu_s, mu_d, v_c, vt, p_dA):
    m = np.linalg.norm(vt)
    T_s, T_d = -mu_s * vt / m * p_dA, -mu_d * vt / m * p_dA
    if m <= v_c:
        return -mu_s * vt / v_c * p_dA
    if m <= 2 * v_c:
        return T_s
    if 3 * v_c <= m:
        return T_d
    wd = (m - 2 * v_c) / (3 * v_c - 2 * v_c)
    return T_s * (1 - wd) + T_d * wd


def test_traction_bristle(O):
    mu_s, mu_d, p_dA = 1.1, 0.3, 0.133
    d = np.array([1.0, 2.0, 3.0]) / np.linalg.norm([1.0, 2.0, 3.0])
    for mag in np.linspace(0.0, 4 * mu_s, 100)[1:]:
        Ts = mag * d
        out = np.zeros(3)
        O.lib().pfo_traction_bristle(mu_s, mu_d, P(Ts), p_dA, P(out))
        np.testing.assert_allclose(out, _verify_bri(mu_s, mu_d, Ts, p_dA), rtol=1e-12, atol=1e-15)
    out = np.ones(3)
    O.lib().pfo_traction_bristle(mu_s, mu_d, P(np.zeros(3)), p_dA, P(out))
    assert np.all(out == 0.0)


def test_traction_regularized(O):
    mu_s, mu_d, v_c, p_dA = 1.1, 0.3, 1.0e-4, 0.133
    d = np.array([1.0, 2.0, 3.0]) / np.linalg.norm([1.0, 2.0, 3.0])
    for mag in np.linspace(0.0, 4 * v_c, 100)[1:]:
        vt = mag * d
        out = np.zeros(3)
        O.lib().pfo_traction_regularized(mu_s, mu_d, v_c, P(vt), p_dA, P(out))
        np.testing.assert_allclose(out, _verify_reg(mu_s, mu_d, v_c, vt, p_dA), rtol=1e-12, atol=1e-18)


# ---------------------------------------------------------------------------------------------------------------
# test/test_math_kernel/test_utility.jl:11-13, test_vector_projections.jl:7-24, test_geometry_kernel.jl:12-24
# ---------------------------------------------------------------------------------------------------------------
def test_weight_poly(O):
    p1, p2 = np.array([1.0, 2.0, 3.0]), np.array([2.0, 3.0, 4.0])
    out = np.zeros(3)
    O.lib().pfo_weight_poly(3, P(p1), P(p2), 1.0, 0.0, P(out)); assert np.array_equal(out, p2)
    O.lib().pfo_weight_poly(3, P(p1), P(p2), 0.0, 1.0, P(out)); assert np.array_equal(out, p1)
    O.lib().pfo_weight_poly(3, P(p1), P(p2), -0.7, 0.7, P(out)); assert np.array_equal(out, (p1 + p2) * 0.5)


def test_vector_projections(O):
    L = O.lib()
    n = np.array([0.0, 0.0, 1.0])
    out = np.zeros(3)
    for v, want in (([1.0, 0, 0], [1.0, 0, 0]), ([0, 0, 1.0], [0, 0, 0]), ([0, 1.0, 1.0], [0, 1.0, 0])):
        L.pfo_vec_sub_vec_proj(P(v), P(n), P(out))
        assert np.array_equal(out, np.array(want, dtype=float))
    a = np.array([1.0, 2.0, 3.0, 4.0])
    assert L.pfo_a_dot_one_pad_b(P(a), P([2.0, 0, 0])) == 6.0
    assert L.pfo_a_dot_one_pad_b(P(a), P([1.0, 2.0, 0])) == 9.0
    assert L.pfo_a_dot_one_pad_b(P(a), P([1.0, 2.0, 3.0])) == 18.0


def test_geometry_kernel(O, pfc):
    L = O.lib()
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=float)
    n = np.zeros(3)
    L.pfo_triangle_normal(P(v[:3]), P(n))
    np.testing.assert_allclose(n, [0, 0, 1])
    assert L.pfo_triangle_area(P(v[:3]), P(n)) == pytest.approx(0.5)
    assert L.pfo_volume(P(v)) == pytest.approx(1 / 6)
    assert float(pfc.geometry.tet_volume(v)) == pytest.approx(1 / 6)


# ---------------------------------------------------------------------------------------------------------------
# test/test_clip/test_quadrature.jl:2-29 (triangle rules reachable from the hot path: 1 and 2)
# ---------------------------------------------------------------------------------------------------------------
def test_quadrature(O):
    for rule, npt in ((1, 1), (2, 3)):
        z = np.zeros(9); w = np.zeros(3)
        assert O.lib().pfo_tri_quad_rule(rule, P(z), P(w)) == npt
        z = z[:3 * npt].reshape(npt, 3); w = w[:npt]
        assert np.sum(w) == pytest.approx(1.0)
        np.testing.assert_allclose(z.sum(axis=1), 1.0)
        np.testing.assert_allclose((w[:, None] * z).sum(axis=0), 1 / 3)
    z = np.zeros(9); w = np.zeros(3)
    O.lib().pfo_tri_quad_rule(2, P(z), P(w))
    z = np.sort(z.reshape(3, 3), axis=1)
    assert np.array_equal(z[0], z[1]) and np.array_equal(z[1], z[2])
    assert O.lib().pfo_tri_quad_rule(6, P(z), P(w)) < 0          # "rule 6 throws"


# ---------------------------------------------------------------------------------------------------------------
# test/test_clip/test_poly_eight.jl:2-27  centroid exact values incl. degenerate; :29-63 zero_small_coordinates
# ---------------------------------------------------------------------------------------------------------------
def test_poly_centroid(O):
    p1, p2, p3, p4 = map(np.array, ([0.0, 0, 0], [1.0, 0, 0], [1.0, 1, 0], [0.0, 1, 0]))
    n = np.array([0.0, 0, 1.0])

    def cen(nv, verts):
        v = np.array(list(verts) + [verts[0]] * (8 - len(verts)), dtype=float)
        c = np.zeros(3)
        a = O.lib().pfo_poly_centroid(nv, P(v), P(n), P(c))
        return a, c

    a, c = cen(4, [p1, p2, p3, p4]); assert a == 1.0 and np.array_equal(c, [0.5, 0.5, 0.0])
    a, c = cen(8, [p1, p2, p3, p4, p1, p1, p1, p1]); assert a == 1.0 and np.array_equal(c, [0.5, 0.5, 0.0])
    a, c = cen(5, [p1, p2, p2, p3, p4]); assert a == 1.0 and np.array_equal(c, [0.5, 0.5, 0.0])
    a, c = cen(3, [p1, p2, p4]); assert a == 0.5 and np.array_equal(c, [1 / 3, 1 / 3, 0.0])
    a, c = cen(3, [p1, p2, p2]); assert a == 0.0 and not np.any(np.isnan(c))


def test_zero_small_coordinates(O):
    rng = np.random.default_rng(0)
    for n in range(1, 9):
        for iv in range(n):
            for ic in range(4):
                A = rng.random((8, 4)) + 0.5
                A[iv, ic] = (rng.random() - 0.5) * 3.0e-15
                B = A.copy()
                O.lib().pfo_zero_small_coordinates(n, P(B))
                want = A.copy(); want[iv, ic] = 0.0
                assert np.array_equal(B[:n], want[:n])


# ---------------------------------------------------------------------------------------------------------------
# helpers shared by the clip property tests (src/clip/test_utility.jl:2-30)
# ---------------------------------------------------------------------------------------------------------------
def _roll_tet(O, rng):
    while True:
        v = rng.standard_normal((4, 3))
        if O.lib().pfo_volume(P(v)) >= 0.25:
            return v


def _as_mat_one_pad(v):
    A = np.ones((4, 4))
    A[:3, :] = v.T
    return A


def _inv4(O, A):
    B = np.zeros(16)
    assert O.lib().pfo_inv4(P(A.reshape(-1, order="F")), P(B)) == 0
    return B.reshape(4, 4, order="F")


def test_inv4_matches_lapack(O):
    rng = np.random.default_rng(5)
    for _ in range(200):
        A = _as_mat_one_pad(_roll_tet(O, rng))
        np.testing.assert_allclose(_inv4(O, A), np.linalg.inv(A), rtol=1e-10, atol=1e-12)


def _clip_plane_tet(O, plane, A):
    out = np.zeros(12)
    n = O.lib().pfo_clip_plane_tet(P(plane), P(A.reshape(-1, order="F")), P(out))
    return out.reshape(4, 3)[:n]


def _tri_normal(a, b, c):
    n = np.cross(b - a, c - b)
    return n / np.linalg.norm(n)


# ---------------------------------------------------------------------------------------------------------------
# test/test_clip/test_plane_tet_intersection.jl:15-62
# ---------------------------------------------------------------------------------------------------------------
def test_clip_plane_tet_properties(O):
    rng = np.random.default_rng(11)
    n0 = n3 = n4 = 0
    for _ in range(100):
        v = _roll_tet(O, rng)
        A = _as_mat_one_pad(v)
        iA = _inv4(O, A)
        for _ in range(100):
            nh = rng.standard_normal(3); nh /= np.linalg.norm(nh)
            plane = np.array([nh[0], nh[1], nh[2], rng.standard_normal()])
            c = _clip_plane_tet(O, plane, A)
            d = v @ nh + plane[3]
            small2 = lambda q: np.sum(np.abs(iA @ np.append(q, 1.0)) < 1.0e-14) >= 2
            if len(c) == 3:
                assert (np.sum(d < 0) == 1) or (np.sum(0 < d) == 1)
                np.testing.assert_allclose(_tri_normal(c[0], c[1], c[2]), nh, atol=1e-7)
                assert all(small2(q) for q in c)
                n3 += 1
            elif len(c) == 4:
                assert np.sum(d < 0) == 2 and np.sum(0 < d) == 2
                for k in range(4):
                    np.testing.assert_allclose(_tri_normal(c[k], c[(k + 1) % 4], c[(k + 2) % 4]), nh, atol=1e-6)
                assert all(small2(q) for q in c)
                n4 += 1
            else:
                assert np.all(d <= 0) or np.all(0 <= d)
                n0 += 1
            for q in c:                                      # verify_inplane
                assert abs(q @ nh + plane[3]) <= 1.0e-14 * 50
    assert n0 and n3 and n4


# ---------------------------------------------------------------------------------------------------------------
# test/test_clip/test_static_clip.jl:13-64   random planar quads x random tets until >= 3 octagons were seen
# ---------------------------------------------------------------------------------------------------------------
def _min_area(poly, nh, r):
    if len(poly) == 0:
        return -np.inf
    m = np.inf
    for k in range(len(poly)):
        a, b = poly[k], poly[(k + 1) % len(poly)]
        m = min(m, float(nh @ (np.cross(b - a, r - b) * 0.5)))
    return m


def test_static_clip_properties(O):
    rng = np.random.default_rng(2)
    tol = 1.0e-13
    n_hits = np.zeros(9, dtype=int)
    n_empty = 0
    it = 0
    while n_hits[8] <= 2 and it < 400000:
        it += 1
        # make_4_sided
        while True:
            v0 = _roll_tet(O, rng)
            nh0 = rng.standard_normal(3); nh0 /= np.linalg.norm(nh0)
            quad = _clip_plane_tet(O, np.array([*nh0, rng.standard_normal()]), _as_mat_one_pad(v0))
            if len(quad) == 4:
                break
        nh = _tri_normal(quad[0], quad[1], quad[2])
        plane = np.array([*nh, -nh @ quad[0]])
        tet = _roll_tet(O, rng)
        A = _as_mat_one_pad(tet)
        iA = _inv4(O, A)
        z_in = (iA @ np.c_[quad, np.ones(4)].T).T
        z_out = np.zeros(32)
        n = O.lib().pfo_clip_in_tet_coordinates(4, P(z_in), P(z_out))
        assert n >= 0
        r_clip = (A @ z_out.reshape(8, 4)[:n].T).T[:, :3] if n else np.zeros((0, 3))
        for q in r_clip:
            assert abs(q @ nh + plane[3]) < 2000 * tol
        for _ in range(30 if n else 3):
            x = rng.standard_normal(3)
            x = x - (x @ nh + plane[3]) * nh
            min_z = np.min(iA @ np.append(x, 1.0))
            a_orig = _min_area(quad, nh, x)
            a_clip = _min_area(r_clip, nh, x)
            if tol < a_clip:
                assert -tol < min_z and -tol < a_orig
            else:
                assert min_z < tol or a_orig < tol
        if n == 0:
            n_empty += 1
        else:
            n_hits[n] += 1
    assert n_hits[8] >= 2, n_hits
    assert n_empty > 1000
    assert n_hits[1] == 0 and n_hits[2] == 0


def test_clip_triangle_inside_and_outside(O):
    """clip() fast paths: a triangle fully inside is returned unchanged (static_clip.jl:40,45-46); fully behind
    one face gives the empty polygon (:44); a vertex exactly on a face is 'inside' (<= vs <)."""
    z_in = np.array([[0.2, 0.3, 0.1, 0.4], [0.3, 0.2, 0.4, 0.1], [0.25, 0.25, 0.25, 0.25]])
    z_out = np.zeros(32)
    assert O.lib().pfo_clip_in_tet_coordinates(3, P(z_in), P(z_out)) == 3
    assert np.array_equal(z_out.reshape(8, 4)[:3], z_in)
    assert np.array_equal(z_out.reshape(8, 4)[3:], np.tile(z_in[0], (5, 1)))       # unused slots = vertex 1
    z_in = np.array([[-0.2, 0.5, 0.3, 0.4], [-0.1, 0.2, 0.5, 0.4], [0.0, 0.3, 0.3, 0.4]])
    assert O.lib().pfo_clip_in_tet_coordinates(3, P(z_in), P(z_out)) == 0
    z_in = np.array([[0.0, 0.5, 0.3, 0.2], [0.1, 0.2, 0.5, 0.2], [0.2, 0.3, 0.3, 0.2]])
    assert O.lib().pfo_clip_in_tet_coordinates(3, P(z_in), P(z_out)) == 3
    assert O.lib().pfo_clip_in_tet_coordinates(5, P(np.zeros((5, 4))), P(z_out)) < 0    # "something is wrong"


# ---------------------------------------------------------------------------------------------------------------
# test/test_obb/test_intersection.jl:39-104   face-corner and edge-edge touching at 1 -/+ 1e-6
# ---------------------------------------------------------------------------------------------------------------
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


def _sat(O, ea, eb, R, t):
    z = np.zeros(3); I = np.eye(3).reshape(-1, order="F")
    return O.lib().pfo_bb_bb_intersect(P(z), P(ea), P(I), P(z), P(eb), P(I), P(np.asarray(R).reshape(-1, order="F")),
                                       P(t))


def test_obb_face_corner(O):
    """face_corner_test (test_intersection.jl:39-52): a corner of one box touches the centre of a face of the other;
    called as (face, corner) and as (corner, face) for all 6 x 8 combinations."""
    e1, e2, tol = np.array([1.0, 2.0, 3.0]), np.array([2.1, 2.2, 2.3]), 1.0e-6
    faces = [np.eye(3)[k // 2] * (-1.0 if k % 2 == 0 else 1.0) for k in range(6)]
    corners = [np.array([sx, sy, sz], dtype=float) for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)]

    def check(dir_1, dir_2):
        v1, v2 = dir_1 * e1, dir_2 * e2
        R = _rot_between(dir_2, dir_1)              # rotation_between(dir_2, dir_1)
        sep = v1 + R @ v2
        assert _sat(O, e1, e2, R, sep * (1 - tol)) == 1
        assert _sat(O, e1, e2, R, sep * (1 + tol)) == 0

    for f in faces:
        for c in corners:
            check(f, c)
            check(c, f)


def test_obb_edge_edge(O):
    rng = np.random.default_rng(3)
    one = np.ones(3)
    tol = 1.0e-6
    edges = [np.array(e, dtype=float) for e in ([0, -1, -1], [0, 1, -1], [0, -1, 1], [0, 1, 1], [-1, 0, -1],
                                                [1, 0, -1])]

    def axis_angle(th, a):
        a = a / np.linalg.norm(a)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K

    def rx(t): return axis_angle(t, np.array([1.0, 0, 0]))
    def ry(t): return axis_angle(t, np.array([0, 1.0, 0]))
    def rz(t): return axis_angle(t, np.array([0, 0, 1.0]))

    for e in edges:
        for th in rng.random(15) * 2 * np.pi:
            for extra in np.arange(0, 2 * np.pi + 1e-9, np.pi / 2):
                for Rb in (rx(extra), ry(extra), rz(extra)):
                    R = axis_angle(th, e) @ Rb
                    sep = e * 2.0
                    assert _sat(O, one, one, R, sep * (1 - tol)) == 1
                    assert _sat(O, one, one, R, sep * (1 + tol)) == 0


# ---------------------------------------------------------------------------------------------------------------
# test/test_friction.jl:163-176   decompose_K! identities on a random PD matrix
# ---------------------------------------------------------------------------------------------------------------
def test_decompose_K(O):
    rng = np.random.default_rng(9)
    for _ in range(20):
        A = rng.standard_normal((6, 6))
        U, s, _ = np.linalg.svd(A)
        M = np.diag([1.0, 1.0, 1.0, 1000, 1000, 1000])
        K = M @ (U @ np.diag(s) @ U.T) @ M
        K = (K + K.T) / 2
        magic = 1.0e-2
        Kis = np.zeros(36); Sinv = np.zeros(6)
        O.lib().pfo_decompose_K(P(K.reshape(-1, order="F")), magic, P(Kis), P(Sinv))
        Kis = Kis.reshape(6, 6, order="F")
        Kbar = np.linalg.matrix_power(np.linalg.inv(Kis), 2)
        t1, t2 = np.trace(Kbar[:3, :3]), np.trace(Kbar[3:, 3:])
        assert t1 == pytest.approx(t2 * magic ** 2, rel=1e-8)
        S = np.diag(1.0 / Sinv)
        np.testing.assert_allclose(S @ Kbar @ S, K, rtol=1e-7, atol=1e-7 * np.abs(K).max())
