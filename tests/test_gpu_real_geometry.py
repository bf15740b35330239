"""Real (non-synthetic) geometry through the path ("lets real reference scenes run"; BASELINE config 5: "pile of 64
compliant boxes (pencil/spoon-scale meshes)"): the reference's spoon -- the vertex / face data of test/data/spoon.obj as the
fixture tests/golden/spoon_quads.npz, 2 502 quads split into 5 004 triangles, scaled as test/spoon.jl:39-41 does -- and its swept
12-sided pencil (src/geometry/mesh_create_swept.jl:73-104, test/pencil.jl:198-200), trees from pfc_build_tree (blob and median),
against the compliant finger pad of test/pencil.jl:188-190; and both of them on top of the C5 pile."""
import numpy as np
import pytest

import helpers as H
from test_gpu_scale import _check_vs_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("method", ["blob", "median"])
def test_spoon_and_pencil_against_finger_pads(pfc, method):
    """32 random poses (16 spoon, 16 pencil), bristle friction: per item counters, candidate (i_1, i_2) sets and clip-vertex
    counts bit-equal to the oracle's, wrench and sdot to the north_star tolerance; on the default path as well (the fused
    kernel: these are pencil.jl-sized scenes)."""
    w = pfc.configs.spoon_pencil_pads(32)
    assert w.meta["n_tri_spoon"] == 5004 and w.meta["n_tri_pencil"] == 48 and w.meta["n_tet_pad"] == 320
    if method == "median":
        for ms in w.meshes:
            ms.tree = pfc.geometry.build_tree(ms.mesh, "median")
    m = pfc.configs.build_scenario(w, debug=True)
    wrench, sdot, counts = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    # counters, candidate sets and clip-vertex counts of EVERY item; wrench 1e-6 (asserted tighter below), sdot 1e-6 except for
    # patches whose scaled stiffness has two or more eigenvalues at its rounding level (grazing pad contacts: the 1e-3 rule of
    # tests/test_gpu_scale.py)
    ref = _check_vs_oracle(pfc, w, m, wrench, sdot, counts, pair_items=tuple(range(w.n_items)), tol=1e-6)
    assert sum(r.counts[3] > 0 for r in ref) >= 20
    for k, r in enumerate(ref):
        if r.counts[3] > 0:      # (a grazing pad with a few dozen traction points is a nearly cancelling sum: 1e-7)
            assert H.rel_err(wrench[k], r.wrench) < (1e-9 if r.counts[3] >= 200 else 1e-7), k
    m.set_option("debug", 0)
    a = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert m.last_parts() == 0, "pencil.jl-sized items are expected on the one-launch path"
    assert np.array_equal(a[2], counts)
    for k, r in enumerate(ref):
        if r.counts[3] > 0:
            assert H.rel_err(a[0][k], r.wrench) < (1e-9 if r.counts[3] >= 200 else 1e-7), k
            # (the items the 1e-3 rule applied to above -- patches with two or more rounding-level eigenvalues -- get it here too)
            assert H.rel_err(a[1][k], r.sdot) < (1e-6 if H.rel_err(sdot[k], r.sdot) < 1e-6 else 2e-3), k
    m.close()


def test_pile_with_the_spoon_and_the_pencil_on_top(pfc):
    """BASELINE config 5 as worded: the 64-box pile plus the spoon (rigid surface) and the pencil (compliant) lying on its top
    layer -- 2 145 bristle instructions; every item against the oracle."""
    w = pfc.configs.c5_pile(pencil_spoon=True)
    assert w.n_items == 2016 + 64 + 64 + 1
    m = pfc.configs.build_scenario(w, debug=True)
    wrench, sdot, counts = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    extra = np.nonzero(counts[2016:, 3] > 0)[0] + 2016
    assert extra.size >= 4, "the spoon and the pencil do not touch the pile"
    _check_vs_oracle(pfc, w, m, wrench, sdot, counts, pair_items=tuple(int(k) for k in extra[:3]), tol=1e-6)
    m.set_option("debug", 0)
    b = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)       # default options (pile mode after the first evaluation)
    c = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert np.array_equal(b[2], counts) and np.array_equal(c[2], counts)
    m.close()
