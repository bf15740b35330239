"""How much of the path depends on the ONE rounding the oracle cannot pin: the 4x4 inverse of calc_ζ_transforms
(src/contact_algorithms_non_friction.jl:158-162), which the reference takes from StaticArrays 0.10.3 `inv(::SMatrix{4,4})`
(not vendored; no Julia here), and on the reference's numeric environment, `set_zero_subnormals(true)`
(test/runtests.jl:13).

Every scene family the parity suite uses is evaluated by the oracle under three inverses -- 1: explicit cofactor expansion
x 1/det (the published StaticArrays 0.10.3 form; default of oracle and HIP path), 0: adjugate from 2x2 minors x 1/det
(rounds 1-2), 2: Gauss-Jordan with partial pivoting -- and with flush-to-zero on.  Compared per item against the default:
candidate pairs (the broadphase never sees the inverse), per-pair clipped-polygon vertex counts (every predicate of
src/clip/static_clip.jl:43-45,140,150,162,176,188 sits downstream of the inverse), traction-point counts, wrench.

Result (asserted below, table in DESIGN.md section 2): on every BASELINE config (C1, C2, C4 x 256, C5 x 2 016, reduced and
full-size C3), on tet-tet scenes and on 1 200 random-pose fuzz items -- 1.3 M candidate pairs -- NO predicate outcome
depends on the inverse and flush-to-zero changes no bit; wrenches move by <= 1e-9 relative (items with near-cancelling
patches; typically 1e-13).  Only the deliberately degenerate fuzz (axis-aligned rotations, lattice offsets: triangles
lying exactly in tet faces) is exposed: there 1-3 % of the pairs change their vertex count with the inverse -- coplanar
ties that the reference's own result depends on in the same way.
"""
import numpy as np
import pytest

import helpers as H

VARIANTS = (("minors", 0, 0), ("gauss-jordan", 2, 0), ("cofactor + FTZ", 1, 1))


def _families(pfc):
    Cf = pfc.configs
    yield "C1 boxes", Cf.c1_boxes(), True
    yield "C2 box on plane", Cf.c2_box_on_plane(), True
    yield "C4 256 scenes", Cf.c2_box_on_plane(256, montecarlo=True), True
    yield "C5 pile 2016", Cf.c5_pile(), True
    yield "C3 reduced x 64", Cf.c3_blob_tool(64, n_div_blob=8, n_div_tool=6), True
    yield "C3 full size x 4", Cf.c3_blob_tool(4), True
    yield "tet-tet x 16", Cf.vol_vol(8, model="bristle"), True
    for tt in (False, True):
        yield f"fuzz random tt={int(tt)}", H.fuzz_workload(pfc, np.random.default_rng(77 + 2 * int(tt)), 600, False, tt), True
    for tt in (False, True):
        yield f"fuzz degenerate tt={int(tt)}", H.fuzz_workload(pfc, np.random.default_rng(78 + 2 * int(tt)), 600, True, tt), False


def _compare(base, alt):
    """-> (items whose candidate list differs, pairs whose vertex count differs, items whose traction count differs,
    worst relative wrench difference, [(item, pair index, n_base, n_alt) ...] first few)"""
    d_pairs = d_clip = d_trac = 0
    werr = 0.0
    first = []
    for k, (a, b) in enumerate(zip(base, alt)):
        if not np.array_equal(a.pairs, b.pairs):
            d_pairs += 1
            continue
        diff = np.nonzero(a.clip_n != b.clip_n)[0]
        d_clip += int(diff.size)
        for j in diff[:2]:
            if len(first) < 6:
                first.append((k, int(j), int(a.clip_n[j]), int(b.clip_n[j])))
        d_trac += int(a.counts[3] != b.counts[3])
        if np.linalg.norm(a.wrench) > 0:
            werr = max(werr, H.rel_err(b.wrench, a.wrench))
    return d_pairs, d_clip, d_trac, werr, first


@pytest.fixture(scope="module")
def exposure(pfc, O):
    L = O.lib()
    rows = []
    try:
        for name, w, generic in _families(pfc):
            L.pfo_set_inv4_variant(1); L.pfo_set_ftz(0)
            base = H.oracle_run(pfc, w, debug=True)
            n_pairs = sum(len(r.clip_n) for r in base)
            res = {}
            for label, var, ftz in VARIANTS:
                L.pfo_set_inv4_variant(var); L.pfo_set_ftz(ftz)
                res[label] = _compare(base, H.oracle_run(pfc, w, debug=True))
            rows.append((name, generic, w.n_items, n_pairs, res))
    finally:
        L.pfo_set_inv4_variant(1); L.pfo_set_ftz(0)
    return rows


def test_variants_are_inverses(O):
    """The three forms invert the same matrices to a few ulp (x_r2_ζ2 = [v1 v2 v3 v4; 1 1 1 1] of random tets)."""
    import ctypes as C
    L = O.lib()
    rng = np.random.default_rng(3)
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    try:
        for _ in range(200):
            A = np.ones((4, 4)); A[:3, :] = rng.standard_normal((3, 4)) * 0.1
            outs = []
            for v in (0, 1, 2):
                L.pfo_set_inv4_variant(v)
                B = np.zeros(16)
                assert L.pfo_inv4(P(np.ascontiguousarray(A.reshape(-1, order="F"))), P(B)) == 0
                outs.append(B.reshape(4, 4, order="F"))
            ref = np.linalg.inv(A)
            for B in outs:
                np.testing.assert_allclose(B, ref, rtol=0, atol=1e-11 * np.abs(ref).max())
    finally:
        L.pfo_set_inv4_variant(1)
    assert L.pfo_set_inv4_variant(3) != 0 and L.pfo_set_inv4_variant(1) == 0


def test_no_predicate_of_the_baseline_configs_depends_on_the_inverse_or_on_ftz(exposure):
    total = 0
    for name, generic, n_items, n_pairs, res in exposure:
        if not generic:
            continue
        total += n_pairs
        for label, (d_pairs, d_clip, d_trac, werr, first) in res.items():
            assert d_pairs == 0, (name, label)
            assert d_clip == 0 and d_trac == 0, (name, label, d_clip, d_trac, first)
            assert werr < 1e-8, (name, label, werr)           # measured: <= 7e-10 (C5), typically 1e-13
        d = res["cofactor + FTZ"]
        assert d[3] == 0.0, (name, "flush-to-zero changed a wrench bit", d[3])
    assert total > 1_000_000


def test_degenerate_lattice_poses_are_exposed_and_by_how_much(exposure):
    """Axis-aligned rotations and lattice offsets put triangles exactly into tet faces: zero ties everywhere.  Flush-to-zero
    still changes nothing; the inverse decides 1-3 % of the vertex counts (the reference's own outcome there depends on
    StaticArrays' rounding just the same).  The candidate lists never differ."""
    seen = 0
    for name, generic, n_items, n_pairs, res in exposure:
        if generic:
            continue
        seen += 1
        for label, (d_pairs, d_clip, d_trac, werr, first) in res.items():
            assert d_pairs == 0
            if "FTZ" in label:
                assert d_clip == 0 and d_trac == 0 and werr == 0.0
            else:
                assert 0 < d_clip < 0.05 * n_pairs, (name, label, d_clip, n_pairs)
    assert seen == 2


def test_print_exposure_table(exposure, capsys):
    """The table DESIGN.md section 2 quotes (python -m pytest tests/test_inv4_exposure.py -s -k table)."""
    with capsys.disabled():
        print("\nfamily | items | candidate pairs | " + " | ".join(f"{l}: pairs with another vertex count / items with another traction count / worst wrench diff" for l, _, _ in VARIANTS))
        for name, generic, n_items, n_pairs, res in exposure:
            cells = [f"{res[l][1]} / {res[l][2]} / {res[l][3]:.1e}" for l, _, _ in VARIANTS]
            print(f"{name} | {n_items} | {n_pairs} | " + " | ".join(cells))
