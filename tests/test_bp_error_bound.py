"""The error radius of the broadphase's Float32 filter (pfc_bp.h, "Error radius E"), sampled: a NumPy emulation of
test_pair_f32 (same operations in the same order; a fused multiply-add is emulated as float32(float64 product + float64
addend)) against the Float64 15-axis expressions of the reference (src/obb/bb_intersection.jl:29-72) on random leaf x
leaf and internal x internal pairs.  The bound is derived, not fitted: the test only shows that the observed worst case
stays below it (and by how much).  The centre offset is formed in Float32 from Float32 centres (round 3): its error is
absolute -- 5.1 u (|c_a|_1 + |c_b|_1 + max |t_i|) -- and enters the radius as eabs = 24 u x that sum; the sampled scenes put
the boxes up to 1 000 box sizes away from their frame origins.  CPU only."""
import numpy as np

U = 2.0 ** -24
f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def quat_to_R64(q):
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - z * w); R[..., 0, 2] = 2 * (x * z + y * w)
    R[..., 1, 0] = 2 * (x * y + z * w); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - x * w)
    R[..., 2, 0] = 2 * (x * z - y * w); R[..., 2, 1] = 2 * (y * z + x * w); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def quat_to_R32(q):
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    x2, y2, z2 = x + x, y + y, z + z
    xx, yy, zz, xy, xz, yz = x * x2, y * y2, z * z2, x * y2, x * z2, y * z2
    wx, wy, wz = w * x2, w * y2, w * z2
    one = f32(1.0)
    R = np.empty(q.shape[:-1] + (3, 3), dtype=f32)
    R[..., 0, 0] = one - (yy + zz); R[..., 0, 1] = xy - wz; R[..., 0, 2] = xz + wy
    R[..., 1, 0] = xy + wz; R[..., 1, 1] = one - (xx + zz); R[..., 1, 2] = yz - wx
    R[..., 2, 0] = xz - wy; R[..., 2, 1] = yz + wx; R[..., 2, 2] = one - (xx + yy)
    return R


def quat_mul32(a, b, conj_a):
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    if conj_a:
        ax, ay, az = -ax, -ay, -az
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    r = np.empty_like(a)
    r[..., 0] = fma(-az, bz, fma(-ay, by, fma(-ax, bx, aw * bw)))
    r[..., 1] = fma(-az, by, fma(ay, bz, fma(ax, bw, aw * bx)))
    r[..., 2] = fma(az, bx, fma(ay, bw, fma(-ax, bz, aw * by)))
    r[..., 3] = fma(az, bw, fma(-ay, bx, fma(ax, by, aw * bz)))
    return r


def quat_rot_inv32(q, v):
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    c0 = fma(-w, v[..., 0], fma(y, v[..., 2], -(z * v[..., 1])))
    c1 = fma(-w, v[..., 1], fma(z, v[..., 0], -(x * v[..., 2])))
    c2 = fma(-w, v[..., 2], fma(x, v[..., 1], -(y * v[..., 0])))
    d0, d1, d2 = fma(y, c2, -(z * c1)), fma(z, c0, -(x * c2)), fma(x, c1, -(y * c0))
    two = np.full_like(d0, 2.0)
    return np.stack([fma(two, d0, v[..., 0]), fma(two, d1, v[..., 1]), fma(two, d2, v[..., 2])], axis=-1)


def axes(ea, eb, t, R, aR, f):
    """the 15 values d = |T.L| - (r_a + r_b); f: fused multiply-add of the precision at hand"""
    d = []
    for i in range(3):
        rb = f(aR[..., i, 2], eb[..., 2], f(aR[..., i, 1], eb[..., 1], aR[..., i, 0] * eb[..., 0]))
        d.append(np.abs(t[..., i]) - (ea[..., i] + rb))
    for j in range(3):
        tl = np.abs(f(R[..., 2, j], t[..., 2], f(R[..., 1, j], t[..., 1], R[..., 0, j] * t[..., 0])))
        ra = f(aR[..., 2, j], ea[..., 2], f(aR[..., 1, j], ea[..., 1], aR[..., 0, j] * ea[..., 0]))
        d.append(tl - (ra + eb[..., j]))
    i100, i221 = (1, 0, 0), (2, 2, 1)
    for m, (u, v) in enumerate(((1, 2), (0, 2), (0, 1))):       # rows of the cross block: t[v] R[u] - t[u] R[v]
        for j in range(3):
            if m == 0:
                tl = np.abs(f(t[..., 2], R[..., 1, j], -(t[..., 1] * R[..., 2, j])))
                ra = f(ea[..., 1], aR[..., 2, j], ea[..., 2] * aR[..., 1, j])
            elif m == 1:
                tl = np.abs(f(t[..., 0], R[..., 2, j], -(t[..., 2] * R[..., 0, j])))
                ra = f(ea[..., 0], aR[..., 2, j], ea[..., 2] * aR[..., 0, j])
            else:
                tl = np.abs(f(t[..., 1], R[..., 0, j], -(t[..., 0] * R[..., 1, j])))
                ra = f(ea[..., 0], aR[..., 1, j], ea[..., 1] * aR[..., 0, j])
            rb = f(eb[..., i100[j]], aR[..., m, i221[j]], eb[..., i221[j]] * aR[..., m, i100[j]])
            d.append(tl - (ra + rb))
    return np.stack(d, axis=-1)


def run(n, leaf, seed):
    rng = np.random.default_rng(seed)

    def rand_q(k):
        q = rng.standard_normal((k, 4))
        # a share of near-axis rotations: the worst cases of the quaternion -> matrix map sit where components vanish
        q[: k // 4] *= rng.choice([1.0, 1e-3, 1e-6], size=(k // 4, 4))
        return q / np.linalg.norm(q, axis=-1, keepdims=True)

    ident = np.tile(np.array([1.0, 0.0, 0.0, 0.0]), (n, 1))
    qa, qb, qp = (rand_q(n) if leaf else ident), (rand_q(n) if leaf else ident), rand_q(n)
    Ra, Rb, Rp = quat_to_R64(qa), quat_to_R64(qb), quat_to_R64(qp)
    qaf, qbf, qpf = qa.astype(f32), qb.astype(f32), qp.astype(f32)
    for qf, Rref in ((qaf, Ra), (qbf, Rb), (qpf, Rp)):       # the checks pfc_add_mesh / pose_quat apply
        q64 = qf.astype(np.float64)
        assert np.abs(quat_to_R64(q64) - Rref).max() <= 4 * U
        assert np.abs((q64 ** 2).sum(-1) - 1).max() <= 2.25 * U
    scale = 10.0 ** rng.uniform(-3, 1, (n, 1))
    ea, eb = scale * rng.uniform(0.05, 1.0, (n, 3)), scale * rng.uniform(0.05, 1.0, (n, 3))
    # centre offsets around touching distance (that is where the sign of d matters), some far apart
    v_goal = scale * rng.standard_normal((n, 3)) * rng.choice([0.3, 1.0, 3.0], size=(n, 1))
    # box centres in their mesh frames, up to 1 000 box sizes from the origin; the pose's translation puts the boxes at v_goal
    far = scale * 10.0 ** rng.uniform(-1, 3, (n, 1))
    ca, cb = far * rng.standard_normal((n, 3)), far * rng.standard_normal((n, 3))
    tp = v_goal - np.einsum("nij,nj->ni", Rp, cb) + ca
    v64 = np.einsum("nij,nj->ni", Rp, cb) + (tp - ca)
    babs = np.abs(ca).sum(-1) + np.abs(cb).sum(-1) + np.abs(tp).max(-1)
    # reference: Float64 throughout, abs_R = |R| + 1e-14
    Rt = np.einsum("nki,nkl,nlj->nij", Ra, Rp, Rb)
    t64 = np.einsum("nki,nk->ni", Ra, v64)
    f64 = lambda a, b, c: a * b + c
    d_ref = axes(ea, eb, t64, Rt, np.abs(Rt) + 1e-14, f64)
    # device path
    eaf, ebf = ea.astype(f32), eb.astype(f32)
    Rpf, caf, cbf, tpf = Rp.astype(f32), ca.astype(f32), cb.astype(f32), tp.astype(f32)
    vf = np.stack([fma(Rpf[:, i, 2], cbf[:, 2], fma(Rpf[:, i, 1], cbf[:, 1], fma(Rpf[:, i, 0], cbf[:, 0], tpf[:, i] - caf[:, i])))
                   for i in range(3)], axis=-1)
    dv = np.abs(vf.astype(np.float64) - v64).max(-1) / babs
    q = quat_mul32(quat_mul32(qaf, qpf, True), qbf, False)
    Rf = quat_to_R32(q)
    tf = quat_rot_inv32(qaf, vf)
    d_f = axes(eaf, ebf, tf, Rf, np.abs(Rf), fma)
    S = (np.abs(vf).sum(-1) + eaf.sum(-1) + ebf.sum(-1)).astype(np.float64)
    dR = np.abs(Rf.astype(np.float64) - Rt).max() / U
    err = np.abs(d_f.astype(np.float64) - d_ref).max(-1)
    k = 320.0 if leaf else 24.0
    # the part of the error the relative radius k u S does not cover, in units of u Babs (radius: 24)
    dd_abs = (np.maximum(err - k * U * S, 0.0) / babs).max() / U
    # and the whole error against the whole radius
    dd = (err / (k * U * S + 24.0 * U * babs)).max()
    return dR, dd, dd_abs, dv.max() / U


def test_leaf_pairs_stay_inside_320u():
    dR, dd, dd_abs, dv = run(200_000, True, 1)
    print(f"leaf x leaf: worst |dR| = {dR:.1f} u (bound 143), worst |dv| = {dv:.2f} u Babs (bound 5.1), worst error / radius = {dd:.3f}, "
          f"beyond 320 u S: {dd_abs:.2f} u Babs (eabs 24)")
    assert dR < 143.0 and dv < 5.1 and dd < 1.0 and dd_abs < 24.0


def test_axis_aligned_pairs_stay_inside_24u():
    dR, dd, dd_abs, dv = run(200_000, False, 2)
    print(f"internal x internal: worst |dR| = {dR:.1f} u (bound 9), worst |dv| = {dv:.2f} u Babs (bound 5.1), worst error / radius = {dd:.3f}, "
          f"beyond 24 u S: {dd_abs:.2f} u Babs (eabs 24)")
    assert dR < 9.0 and dv < 5.1 and dd < 1.0 and dd_abs < 24.0
