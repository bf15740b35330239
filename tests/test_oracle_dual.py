"""The Dual-number (value + partials) oracle, oracle/pfc_oracle_dual.cpp, pinned by central differences of the value
oracle (which the reference's own analytic tests pin, tests/test_oracle_kat.py / test_scene_kats.py).  The Dual
path is what Radau's Jacobian evaluation runs (src/mechanism_scenario.jl:187, src/radau/radau_functions.jl:2-40)."""
import numpy as np
import pytest

from helpers import oracle_ins, oracle_meshes


def rodrigues(v):
    th = np.linalg.norm(v)
    if th < 1e-300:
        return np.eye(3)
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def pose_of(R0, t0, q):
    """x_r2_r1 = (exp(q[:3]) R0, t0 + q[3:]) and its inverse, packed like pfc_eval's pose[24]."""
    R = rodrigues(q[:3]) @ R0
    t = t0 + q[3:]
    Ri = R.T
    return np.concatenate([R.reshape(-1, order="F"), t, Ri.reshape(-1, order="F"), -Ri @ t])


def tangents(R0, t0, dirs, h=1e-6):
    return np.stack([(pose_of(R0, t0, h * d) - pose_of(R0, t0, -h * d)) / (2 * h) for d in dirs])


def fd_check(O, pfc, w, k, rng, n_dir=6, h=1e-6, rtol=2e-5, sdot_rtol=None):
    om = oracle_meshes(w)
    c = w.instructions[int(w.ins_ids[k])]
    ins = oracle_ins(pfc, c)
    m1, m2 = om[c.id_1], om[c.id_2]
    R0 = w.pose[k][:9].reshape(3, 3, order="F"); t0 = w.pose[k][9:12]
    dq = rng.standard_normal((n_dir, 6)) * np.array([1, 1, 1, 0.05, 0.05, 0.05])
    dtw = rng.standard_normal((n_dir, 6)) * np.array([1, 1, 1, 0.1, 0.1, 0.1])
    ds = rng.standard_normal((n_dir, 6)) * 1e-3
    d_pose = tangents(R0, t0, dq)
    st, wr, sd, dw, dsd = O.evaluate_dual(m1, m2, ins, w.pose[k], w.twist[k], w.s[k], d_pose, dtw, ds)
    assert st == 0
    ref = O.evaluate(m1, m2, ins, w.pose[k], w.twist[k], w.s[k], debug=False)
    np.testing.assert_allclose(wr, ref.wrench, rtol=1e-12, atol=1e-12 * np.abs(ref.wrench).max())
    np.testing.assert_allclose(sd, ref.sdot, rtol=1e-9, atol=1e-9 * max(np.abs(ref.sdot).max(), 1e-300))
    assert ref.counts[3] > 0, "scene must be in contact"
    for d in range(n_dir):
        f = []
        for sgn in (+1, -1):
            r = O.evaluate(m1, m2, ins, pose_of(R0, t0, sgn * h * dq[d]), w.twist[k] + sgn * h * dtw[d],
                           w.s[k] + sgn * h * ds[d], debug=False)
            f.append((r.wrench, r.sdot))
        fw = (f[0][0] - f[1][0]) / (2 * h); fs = (f[0][1] - f[1][1]) / (2 * h)
        assert np.linalg.norm(dw[d] - fw) <= rtol * np.linalg.norm(fw) + 1e-9 * np.linalg.norm(ref.wrench), (d, dw[d], fw)
        if sdot_rtol is not None:
            assert np.linalg.norm(dsd[d] - fs) <= sdot_rtol * np.linalg.norm(fs), (d, dsd[d], fs)
    return dw, dsd


def test_dual_regularized_box_on_plane(pfc, O):
    w = pfc.configs.c2_box_on_plane(4, montecarlo=True)
    rng = np.random.default_rng(1)
    for k in range(4):
        fd_check(O, pfc, w, k, rng)


def test_dual_bristle_blob_tool(pfc, O):
    w = pfc.configs.c3_blob_tool(3, seed=11, n_div_blob=6, n_div_tool=4)
    rng = np.random.default_rng(2)
    for k in range(3):
        w.s[k] = rng.standard_normal(6) * 1e-3
        fd_check(O, pfc, w, k, rng, sdot_rtol=1e-3)


def test_dual_tet_tet(pfc, O):
    rng = np.random.default_rng(3)
    for model in ("regularized", "bristle"):
        w = pfc.configs.vol_vol(2, n_div=3, model=model)
        for k in range(w.n_items):
            if model == "bristle":
                w.s[k] = rng.standard_normal(6) * 1e-3
            fd_check(O, pfc, w, k, rng, sdot_rtol=1e-3 if model == "bristle" else None)


def test_dual_no_contact_bristle(pfc, O):
    """no_contact!(::Bristle) (friction.jl:77-81): sdot = -s / tau, so d(sdot) = -ds / tau."""
    w = pfc.configs.c3_blob_tool(1, seed=11, n_div_blob=4, n_div_tool=3)
    om = oracle_meshes(w)
    c = w.instructions[0]
    ins = oracle_ins(pfc, c)
    pose = pose_of(np.eye(3), np.array([0.0, 0.0, 5.0]), np.zeros(6))
    ds = np.eye(6)
    st, wr, sd, dw, dsd = O.evaluate_dual(om[c.id_1], om[c.id_2], ins, pose, np.zeros(6), np.ones(6),
                                          np.zeros((6, 24)), np.zeros((6, 6)), ds)
    assert st == 0 and np.all(wr == 0) and np.all(dw == 0)
    np.testing.assert_allclose(sd, -np.ones(6) / c.tau)
    np.testing.assert_allclose(dsd, -ds / c.tau)
