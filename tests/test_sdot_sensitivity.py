"""ṡ of sliver / edge contacts: the looser tolerance is demonstrated, not asserted.

ṡ = -(1/τ) (K̄^{-1/2} S⁻¹ w_fric + s) (src/contact_algorithms_friction.jl:134) and decompose_K! clamps the eigenvalues
of K̄ at 1e-16 σ_max (:92), i.e. K̄^{-1/2} amplifies by up to 1e8 along near-null directions.  A flat patch has ONE
exactly-null direction (harmless); patches whose K̄ has two or more eigenvalues at the rounding level of K (slivers, edge
contacts: a few in the C5 pile) make ṡ depend on the last bits of K *in the reference's own arithmetic*.  Shown here on
the oracle alone: perturbing K by one unit in the last place moves the oracle's ṡ of those items by up to ~1e-5
relative PER ULP (K differs by several ulp between two summation orders), while the other items move by < 2e-7 (median 1e-13).  The HIP-vs-oracle difference on the same items (GPU test) stays
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
    """Largest relative change of the oracle's own ṡ under symmetric +/- 1 ulp perturbations of K."""
    rng = np.random.default_rng(seed)
    base = _sdot_from_K(r.K, c.magic, r.wrench_fric_cop, s, c.tau)
    worst = 0.0
    for _ in range(n_trial):
        sign = rng.choice([-1.0, 1.0], size=(6, 6))
        sign = np.triu(sign) + np.triu(sign, 1).T
        Kp = np.where(sign > 0, np.nextafter(r.K, np.inf), np.nextafter(r.K, -np.inf))
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
    # one ulp of K moves the reference-order ṡ of sliver patches by far more than any tight tolerance could absorb ...
    assert np.median(sliver) > 1e-7, sliver
    assert max(sliver) > 1e-6, sliver
    # ... and every other patch inside the 1e-6 asserted for C5 (the single exactly-null direction of a flat patch costs
    # up to ~2e-8 per ulp; most patches sit at 1e-13)
    assert max(regular) < 2e-7, max(regular)
    assert np.median(regular) < 1e-10


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
            # within a small multiple of what ONE ulp of K does to the oracle itself (K differs by a few ulp: reordered sums)
            assert diff <= max(200.0 * sens, 1e-9), (k, diff, sens)
            assert diff < 1e-3
        else:
            assert diff < 1e-6, (k, diff)
    assert n_sliver >= 3
