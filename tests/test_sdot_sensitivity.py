"""ṡ of sliver / edge contacts: the looser tolerance is demonstrated, not asserted.

ṡ = -(1/τ) (K̄^{-1/2} S⁻¹ w_fric + s) (src/contact_algorithms_friction.jl:134) and decompose_K! clamps the eigenvalues
of K̄ at 1e-16 σ_max (:92), i.e. K̄^{-1/2} amplifies by up to 1e8 along near-null directions.  A flat patch has ONE
exactly-null direction (harmless); patches whose K̄ has two or more eigenvalues at the rounding level of K (slivers, edge
contacts: a few in the C5 pile) make ṡ depend on the last bits of K *in the reference's own arithmetic*.  Shown here on
the oracle alone: perturbing K̄ by eps ||K̄|| (the backward error of any stable eigen-solver) moves the oracle's ṡ of those items by 1e-6 .. 1e-2
relative (median 3e-4), while the typical other item moves by ~2e-9.  The HIP-vs-oracle difference on the same items (GPU test) stays
inside that band, and inside 1e-6 everywhere else -- which is why tests/test_gpu_scale.py asserts 1e-3 for exactly the
items with >= 2 eigenvalues below 1e-12 σ_max and 1e-6 (the north_star tolerance) for all others."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from oracle import oracle as Orc

_dp = C.POINTER(C.c_double)


def _sdot_from_K(K, magic, w_fric_cop, s, tau):
    Kis = np.zeros(36); Sinv = np.zeros(6)
    Kc = np.ascontiguousarray(K.reshape(-1, order="F"))
    Orc.lib().pfo_decompose_K(Kc.ctypes.data_as(_dp), float(magic), Kis.ctypes.data_as(_dp), Sinv.ctypes.data_as(_dp))
    Kis = Kis.reshape(6, 6, order="F")
    return -(1.0 / tau) * (Kis @ (Sinv * w_fric_cop) + s)


def _near_null(r):
    Kb = np.diag(r.Sinv) @ r.K @ np.diag(r.Sinv)
    ev = np.linalg.eigvalsh((Kb + Kb.T) / 2)
    return int(np.sum(ev < 1e-12 * ev[-1]))


def _ulp_sensitivity(r, c, s, n_trial=6, seed=0):
    """Largest relative change of the oracle's own ṡ when K̄ = S⁻¹ K S⁻¹ is perturbed by eps ||K̄||: the backward error of
    ANY stable symmetric eigen-solver (LAPACK's in the reference, Jacobi here), i.e. the decomposition the reference
    returns is the exact one of such a neighbour.  (Entry-wise +/- 1 ulp of K alone does not reach the near-null block.)"""
    rng = np.random.default_rng(seed)
    base = _sdot_from_K(r.K, c.magic, r.wrench_fric_cop, s, c.tau)
    S = np.diag(1.0 / r.Sinv)
    Kb = np.diag(r.Sinv) @ r.K @ np.diag(r.Sinv)
    scale = np.finfo(np.float64).eps * np.abs(Kb).max()
    worst = 0.0
    for _ in range(n_trial):
        G = rng.choice([-1.0, 1.0], size=(6, 6))
        G = np.triu(G) + np.triu(G, 1).T
        Kp = S @ (Kb + scale * G) @ S
        Kp = (Kp + Kp.T) / 2
        worst = max(worst, H.rel_err(_sdot_from_K(Kp, c.magic, r.wrench_fric_cop, s, c.tau), base))
    return base, worst


@pytest.fixture(scope="module")
def pile(pfc):
    w = pfc.configs.c5_pile()
    ref = H.oracle_run(pfc, w, debug=True)
    touching = [k for k, r in enumerate(ref) if r.has_K]
    return w, ref, touching


def test_oracle_sdot_is_as_sensitive_as_the_tolerance_says(pfc, pile):
    w, ref, touching = pile
    sliver, regular = [], []
    for k in touching:
        r, c = ref[k], w.instructions[int(w.ins_ids[k])]
        base, sens = _ulp_sensitivity(r, c, w.s[k])
        # the formula above IS the oracle's ṡ (sanity of the experiment)
        assert H.rel_err(base, r.sdot) < 1e-9, k
        (sliver if _near_null(r) >= 2 else regular).append(sens)
    assert len(sliver) >= 3 and len(regular) >= 20
    print("sliver", np.sort(sliver)); print("regular", np.percentile(regular, [50, 90, 100]))
    # a rounding-level perturbation of K̄ moves the reference-order ṡ of sliver patches by far more than any tight
    # tolerance could absorb ...
    assert np.median(sliver) > 1e-6, sliver
    # ... while the typical patch does not notice (median ~2e-9; the exactly-null direction of a flat patch, whose computed
    # eigenvalue is pure rounding and gets clamped, makes a few of them move by up to ~6e-5 as well)
    assert np.median(regular) < 1e-8, np.median(regular)
    assert np.median(sliver) > 100 * np.median(regular)


@pytest.mark.gpu
def test_hip_sdot_difference_stays_inside_the_oracles_own_noise(pfc, pile):
    w, ref, touching = pile
    m = pfc.configs.build_scenario(w)
    wrench, sdot, counts = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    m.close()
    n_sliver = 0
    for k in touching:
        r, c = ref[k], w.instructions[int(w.ins_ids[k])]
        diff = H.rel_err(sdot[k], r.sdot)
        if _near_null(r) >= 2:
            _, sens = _ulp_sensitivity(r, c, w.s[k], n_trial=12)
            n_sliver += 1
            # within a small multiple of what a rounding-level perturbation of K̄ does to the oracle itself.  K is a sum over
            # the item's n traction points, taken in another order on the device: the two sums differ by ~sqrt(n) eps, not
            # by one eps (item 1804: three eigenvalues at the clamp, sens 2.4e-8 per eps, 270 points, difference 2.8e-6)
            noise = sens * max(1.0, np.sqrt(float(r.counts[3])))
            assert diff <= max(50.0 * noise, 1e-9), (k, diff, sens, noise)
            assert diff < 1e-3
        else:
            assert diff < 1e-6, (k, diff)
    assert n_sliver >= 3
