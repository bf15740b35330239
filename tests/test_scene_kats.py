"""Scene-level analytic tests of the reference, run through BOTH implementations: the CPU oracle (``-m "not gpu"``)
and the HIP library through its C ABI (``-m gpu``).  They pin the whole path (broadphase -> clip -> quadrature ->
friction) to the reference's own known answers; the reference holds no golden vectors, so these analytic tests are the
only reference-held pins there are, and they bear on the device code directly.  Paths relative to the reference
repository."""
import numpy as np
import pytest

import helpers as H

BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]


def _box_plane(pfc, box_rad):
    """Rigid 12-triangle box resting on the compliant half-plane tet (the geometry of test_normal.jl:2-25 and
    test_friction.jl:92-131); the plane is attached to the world so frame r2 = world, frame r1 = the box body."""
    G = pfc.geometry
    plane = G.as_tet_emesh(G.emesh_half_plane())
    box = G.as_tri_emesh(G.emesh_box(box_rad)).transformed(t=[0, 0, box_rad])
    return box, plane


def _pose(trans):
    from oracle import oracle as Orc
    return Orc.make_pose(np.eye(3), trans)


def _bristle(chi, n_quad, mu, tau, k_bar, magic=1.0e-3):
    return dict(model="bristle", chi=chi, n_quad=n_quad, mu_s=mu, mu_d=mu, tau=tau, k_bar=k_bar, magic=magic)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("k_quad_rule", [1, 2])
def test_normal_wrench_is_exact(pfc, backend, k_quad_rule):
    """test/test_normal.jl:2-49: wrench = -[r x f; f], f_z = Ē * pene * 4 * box_rad^2, box shifted by (0.1, 0.2).
    normal_wrench(b) is recomputed from the TractionCache view, as the reference test does (:34)."""
    box_rad, p_pos, Ebar = 0.05, (0.1, 0.2), 1.0e9
    pene = 0.1 * box_rad
    box, plane = _box_plane(pfc, box_rad)
    r = H.eval_scene(backend, pfc, box, None, plane, Ebar, _bristle(0.6, k_quad_rule, 0.3, 0.03, 1.0e6),
                     _pose([p_pos[0], p_pos[1], -pene]), np.zeros(6), np.zeros(6))
    assert r.status == 0
    f3 = np.array([0.0, 0.0, Ebar * pene / 1.0 * box_rad ** 2 * 4])
    a3 = np.cross([p_pos[0], p_pos[1], 0.0], f3)
    check = -np.concatenate([a3, f3])
    assert r.trac.shape[0] > 0
    np.testing.assert_allclose(H.normal_wrench_from_tractions(r.trac), check, rtol=1e-8, atol=1e-8 * np.linalg.norm(check))
    # s = 0 and zero twist: the bristle friction force vanishes, so the total wrench is the normal wrench
    np.testing.assert_allclose(r.wrench, check, rtol=1e-8, atol=1e-8 * np.linalg.norm(check))


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("n_quad_rule", [1, 2])
def test_regularized_friction_sign(pfc, backend, n_quad_rule):
    """test/test_friction.jl:92-143: the box travels at v_tol in +y and is pushed with 0.999x / 1.001x the friction
    strength; v̇_y (index 11 of xx) must be negative / positive.  Single free box with identity orientation, so
    v̇_y = (τ_ext_y + F_contact_y) / m with F_contact on the box = -wrench_lin (the wrench is the one on body 2 = the
    plane; third law, non_friction.jl:270-272)."""
    box_rad, Ebar, mu_d, v_tol, rho, d = 0.05, 1.0e9, 0.3, 1.0e-4, 400.0, 0.09
    mag_grav = 9.8054
    mass = rho * d * 6 * (2 * box_rad) ** 2          # shell inertia: density x thickness x surface area
    mg = mag_grav * mass
    pene = mg / (Ebar * 4 * box_rad ** 2)
    box, plane = _box_plane(pfc, box_rad)
    ins = dict(model="regularized", chi=0.5, n_quad=n_quad_rule, mu_s=mu_d, mu_d=mu_d, v_c=v_tol)
    # twist_r2_r1_r2: velocity of body 2 (plane, at rest) relative to body 1 (box moving +y): linear = -v_box
    twist = np.array([0, 0, 0, 0.0, -v_tol, 0.0])
    r = H.eval_scene(backend, pfc, box, None, plane, Ebar, ins, _pose([0.0, 0.0, -pene]), twist, None)
    assert r.status == 0 and r.counts[3] > 0
    # normal force balances gravity by construction of pene
    assert -r.wrench[5] == pytest.approx(mg, rel=1e-6)
    f_contact_y = -r.wrench[4]                         # on the box
    # |v_t| = v_c: T = -μs v_t / v_c on the bottom face; the ±x side faces dip pene into the plane and add
    # O(pene / box_rad) ~ 2e-5 of extra friction, far inside the test's 1e-3 margin
    assert f_contact_y == pytest.approx(-mu_d * mg, rel=1e-4)
    for coe, sign in ((0.999, -1), (1.001, +1)):
        vdot_y = (mg * mu_d * coe + f_contact_y) / mass
        assert np.sign(vdot_y) == sign


@pytest.mark.parametrize("backend", BACKENDS)
def test_bristle_stiffness_is_analytic(pfc, backend):
    """test/test_friction.jl:178-237: rigid half-plane surface (tri) against a small compliant box (tet),
    K_55 ≈ hol_rad^2 * 4 * k̄ * Ē * pene / hol_rad within 1 %, and K_44 ≈ K_55 (spatialStiffness view)."""
    G = pfc.geometry
    box_rad = 0.05
    hol_rad = 0.2 * box_rad
    Ebar, k_bar, tau = 1.0e9, 1.0e6, 0.03
    part = G.as_tri_emesh(G.emesh_half_plane(1.0))
    hol = G.as_tet_emesh(G.emesh_box(hol_rad)).transformed(t=[0, 0, hol_rad])
    pene = hol_rad * 0.001
    # body 2 (the box, prismatic along z) sits at z = -pene; body 1 (part) at the origin
    pose = pfc.scenario.relative_pose(np.eye(3), np.zeros(3), np.eye(3), [0, 0, -pene])
    r = H.eval_scene(backend, pfc, part, None, hol, Ebar, _bristle(0.6, 2, 0.3, tau, k_bar), pose, np.zeros(6), np.zeros(6))
    assert r.status == 0 and r.has_K
    K_ana = hol_rad ** 2 * 4 * k_bar * (Ebar * (pene / hol_rad))
    S = np.diag(1.0 / r.Sinv)
    K2 = S @ np.linalg.inv(r.Kbar_inv_sqrt @ r.Kbar_inv_sqrt) @ S
    assert K2[3, 3] == pytest.approx(K2[4, 4], rel=1e-6)
    assert 0.99 * K_ana < K2[4, 4] < 1.01 * K_ana
    np.testing.assert_allclose(K2, r.K, rtol=1e-6, atol=1e-6 * np.abs(r.K).max())
    # the reference reads K straight from spatialStiffness.K (:232-236)
    assert r.K[3, 3] == pytest.approx(r.K[4, 4], rel=1e-6)
    assert 0.99 * K_ana < r.K[4, 4] < 1.01 * K_ana


@pytest.mark.parametrize("backend", BACKENDS)
def test_stiffness_translation_invariance(pfc, backend):
    """test/test_friction.jl:239-266: K about the cop is invariant under a translation t of the box over the
    half-plane, and the cop moves by t."""
    G = pfc.geometry
    box_rad = 0.05
    plane = G.as_tet_emesh(G.emesh_half_plane())
    box = G.as_tri_emesh(G.emesh_box(box_rad))
    ins = _bristle(2.2, 2, 1.0, 0.05, 1.0e4)

    def calc_it(t):
        tb = np.asarray(t) + np.array([0.0, 0.0, 0.99 * box_rad])
        pose = pfc.scenario.relative_pose(np.eye(3), tb, np.eye(3), np.zeros(3))
        w_box = np.array([0.4, 0.3, 1.0])                      # body-frame angular velocity, R = I
        tw_box = np.concatenate([w_box, -np.cross(w_box, tb)])  # twist about the world origin
        twist = pfc.scenario.relative_twist(np.eye(3), np.zeros(3), tw_box, np.zeros(6))
        r = H.eval_scene(backend, pfc, box, None, plane, 1.0e6, ins, pose, twist, np.zeros(6))
        assert r.status == 0 and r.has_K
        return r.K, r.cop

    t = np.array([0.35, 0.10, 0.0])
    K_t, cop_t = calc_it(t)
    K_0, cop_0 = calc_it(np.zeros(3))
    np.testing.assert_allclose(K_t, K_0, rtol=1e-7, atol=1e-7 * np.abs(K_0).max())
    np.testing.assert_allclose(cop_t, cop_0 + t, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("backend", BACKENDS)
def test_no_contact_bristle_decay(pfc, backend):
    """no_contact!(::Bristle): ṡ = -(1/τ) s (friction.jl:77-81); regularized: nothing."""
    box, plane = _box_plane(pfc, 0.05)
    s = np.array([1.0, -2.0, 3.0, 0.5, 0.25, -4.0])
    r = H.eval_scene(backend, pfc, box, None, plane, 1.0e9, _bristle(0.5, 2, 0.3, 0.04, 1.0e4), _pose([0.0, 0.0, 0.01]),
                     np.zeros(6), s)
    assert r.counts[3] == 0 and np.all(r.wrench == 0)
    assert np.array_equal(r.sdot, -(1 / 0.04) * s)


@pytest.mark.parametrize("backend", BACKENDS)
def test_tet_tet_normal_force(pfc, backend):
    """Volume-volume contact (non_friction.jl:166-194; test_vol_vol.jl geometry): compliant box on the compliant
    half-plane, both Ē = 1e6.  Checked through force balance properties: pure -z force on the plane, zero lateral
    force, torque = r x f, and the flat-punch magnitude on the equal-pressure plane."""
    G = pfc.geometry
    box_rad, Ebar = 0.05, 1.0e6
    plane = G.as_tet_emesh(G.emesh_half_plane())
    box = G.as_tet_emesh(G.emesh_box(box_rad))
    pene = 0.002
    ins = dict(model="regularized", chi=0.5, n_quad=2, mu_s=0.0, mu_d=0.0, v_c=0.01)
    out = []
    for shift in ((0.0, 0.0), (0.1, 0.2)):
        pose = pfc.scenario.relative_pose(np.eye(3), [shift[0], shift[1], box_rad - pene], np.eye(3), np.zeros(3))
        r = H.eval_scene(backend, pfc, box, Ebar, plane, Ebar, ins, pose, np.zeros(6), None)
        assert r.status == 0 and r.counts[3] > 0
        out.append(r.wrench)
        f = r.wrench[3:]
        assert f[2] < 0 and abs(f[0]) < 1e-9 * abs(f[2]) and abs(f[1]) < 1e-9 * abs(f[2])
        np.testing.assert_allclose(r.wrench[:3], np.cross([shift[0], shift[1], 0.0], f), atol=1e-9 * abs(f[2]))
    # pressure on the equal-pressure plane: plane field ϵ = depth (plane_w = 1), box field ϵ = depth / box_rad;
    # both pressures agree at depth z* = pene / (1 + box_rad) below the plane surface; force = p * area
    z_star = pene / (1 + box_rad)
    f_ana = Ebar * z_star * 4 * box_rad ** 2
    # the bottom pyramid's cross-section shrinks by (1 - h / box_rad)^2 at the height h of the equal-pressure plane and
    # the side tets carry their own field near the rim, so the flat-punch value holds to O(pene / box_rad)
    assert -out[0][5] == pytest.approx(f_ana, rel=5e-3)
    np.testing.assert_allclose(out[0][3:], out[1][3:], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("backend", BACKENDS)
def test_tet_tet_frictionless_spin_is_conserved(pfc, backend):
    """test/test_vol_vol.jl: a compliant box spinning about z (w_z = 1.14) on the compliant half-plane with μd = 0 and
    χ = 0 keeps its spin over 5 s of simulation (`data_state[end, 9] ≈ w_z_0`).  At the boundary of this path that is:
    the contact wrench of the spinning box has no component about z -- neither at the plane's origin nor about the box's
    axis when the box sits off-centre -- and equals the wrench of the box at rest (χ = 0: no damping)."""
    G = pfc.geometry
    box_rad, Ebar, w_z = 0.05, 1.0e6, 1.14
    plane = G.as_tet_emesh(G.emesh_half_plane())
    box = G.as_tet_emesh(G.emesh_box(box_rad))
    ins = dict(model="regularized", chi=0.0, n_quad=2, mu_s=0.0, mu_d=0.0, v_c=0.01)
    for shift in ((0.0, 0.0), (0.07, -0.03)):
        pose = pfc.scenario.relative_pose(np.eye(3), [shift[0], shift[1], 2 * box_rad - box_rad - 0.001], np.eye(3), np.zeros(3))
        # relative twist of the box about its own vertical axis, expressed at the plane frame's origin: v = -w x c
        c = np.array([shift[0], shift[1], 0.0])
        tw = np.concatenate([[0.0, 0.0, w_z], -np.cross([0.0, 0.0, w_z], c)])
        spin = H.eval_scene(backend, pfc, box, Ebar, plane, Ebar, ins, pose, tw, None)
        rest = H.eval_scene(backend, pfc, box, Ebar, plane, Ebar, ins, pose, np.zeros(6), None)
        assert spin.status == 0 and spin.counts[3] > 0
        f = spin.wrench[3:]
        tau_axis = spin.wrench[:3] - np.cross(c, f)          # torque about the box's axis
        assert abs(spin.wrench[2]) < 1e-9 * abs(f[2]) * box_rad and abs(tau_axis[2]) < 1e-9 * abs(f[2]) * box_rad
        np.testing.assert_allclose(spin.wrench, rest.wrench, rtol=1e-12, atol=1e-12 * abs(f[2]))

