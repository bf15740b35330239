"""Option "fixed_order": bit-reproducible evaluations.

The reference sums an instruction's traction points in ONE order (candidate list order, src/contact_algorithms_non_friction.jl:136-143,
then fan and quadrature order), so two calcXd! on the same state give the same bits.  The default GPU path does not: the broadphase
appends candidate runs in the order its workgroups finish and the per-item sums are atomics.  That is harmless up to rounding --
except where decompose_K! (src/contact_algorithms_friction.jl:85-117) clamps an eigenvalue that is zero in exact arithmetic (a flat
patch: 4 of the 331 touching pairs of BASELINE config 5) at 1e-16 sigma_max (:92): there the last bits of K decide the branch and two
identical evaluations differ by up to tens of percent in the partials of those items.  With the option on, the candidate list is
sorted (csrc/pfc_sort.hip), an item's per-chunk records are added in chunk order (k_integ_fixed, k_shift_fixed), the Dual passes'
sums leave as records added in list order (FixedSink, k_fixed_reduce) and k_dual_eig decomposes the value pass's K: every output is
the same bit pattern in every run."""
import numpy as np
import pytest

import helpers as H
from test_gpu_dual import run_case
from test_gpu_scale import _check_vs_oracle
from test_oracle_dual import tangents

pytestmark = pytest.mark.gpu


def _workload(pfc, cfg):
    C = pfc.configs
    if cfg == "c5":
        return C.c5_pile()
    if cfg == "c5ps":
        return C.c5_pile(pencil_spoon=True)
    if cfg == "c3x24":
        return C.c3_blob_tool(24)
    if cfg == "c4":
        return C.c2_box_on_plane(256, montecarlo=True)
    if cfg == "volvol":
        return C.vol_vol(24, n_div=4, model="bristle")
    if cfg == "c1":
        return C.c1_boxes()
    raise KeyError(cfg)


def _seeds(w, n_dir, seed):
    rng = np.random.default_rng(seed)
    n = w.n_items
    dq = rng.standard_normal((n, n_dir, 6)) * np.array([1, 1, 1, 0.05, 0.05, 0.05])
    d_pose = np.zeros((n, n_dir, 24))
    for k in range(n):
        d_pose[k] = tangents(w.pose[k][:9].reshape(3, 3, order="F"), w.pose[k][9:12], dq[k])
    return d_pose, rng.standard_normal((n, n_dir, 6)) * 0.1, rng.standard_normal((n, n_dir, 6)) * 1e-3


@pytest.mark.parametrize("cfg", ["c5", "c5ps", "c3x24", "c4", "volvol", "c1"])
def test_two_fresh_handles_give_the_same_bits(pfc, cfg):
    """Value and Dual evaluation (6 directions, then 3 further directions on the kept value pass) on two fresh handles and twice on
    one of them: counters, wrench, sdot and every partial bit-equal -- including the flat-patch pairs of the pile, whose partials
    differ by O(1) between two default evaluations."""
    w = _workload(pfc, cfg)
    d6 = _seeds(w, 6, 5)
    d3 = _seeds(w, 3, 6)
    outs = []
    for rep in range(2):
        m = pfc.configs.build_scenario(w)
        m.set_option("fixed_order", 1)
        res = []
        for again in range(2 if rep == 0 else 1):
            res.append(m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids))
            res.append(m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *d6, w.ins_ids))
            res.append(m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *d3, w.ins_ids))      # a further chunk of the Jacobian
        outs.append(res)
        m.close()
    assert (outs[0][0][2][:, 3] > 0).any()
    ref = outs[0][:3]
    for other in (outs[0][3:], outs[1]):
        for a, b in zip(ref, other):
            for x, y in zip(a, b):
                assert np.array_equal(x, y)
    # the Dual evaluation's values are the value evaluation's (same value pass, same order)
    assert np.array_equal(ref[0][0], ref[1][0]) and np.array_equal(ref[0][1], ref[1][1])


def test_fixed_order_pile_against_the_oracle(pfc):
    """BASELINE config 5 with the option on: counters bit-equal, wrench / sdot at the tolerances of the default path
    (tests/test_gpu_scale.py::test_c5_pile_all_pairs)."""
    w = pfc.configs.c5_pile()
    m = pfc.configs.build_scenario(w)
    m.set_option("fixed_order", 1)
    wrench, sdot, counts = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert 50 < int((counts[:, 3] > 0).sum()) < 400
    _check_vs_oracle(pfc, w, m, wrench, sdot, counts, tol=1e-6, oracle_debug=True)
    m.close()


@pytest.mark.parametrize("cfg", ["blob", "c4", "volvol_reg", "volvol_bri", "c1"])
def test_fixed_order_dual_against_the_dual_oracle(pfc, O, cfg):
    """The Dual path with the option on (sorted lists, record sinks, K of the value pass in k_dual_eig) against the Dual oracle:
    tri-tet bristle (the folded pass), regularized, tet-tet (three passes) and the reference's box scene."""
    C = pfc.configs
    w = {"blob": lambda: C.c3_blob_tool(12, seed=3, n_div_blob=6, n_div_tool=4),
         "c4": lambda: C.c2_box_on_plane(24, montecarlo=True),
         "volvol_reg": lambda: C.vol_vol(6, n_div=3, model="regularized"),
         "volvol_bri": lambda: C.vol_vol(6, n_div=3, model="bristle"),
         "c1": lambda: C.c1_boxes()}[cfg]()
    run_case(pfc, O, w, 6, 7, options={"fixed_order": 1})


def test_fixed_order_goes_off_again(pfc):
    """The option means the batched path (no one-launch kernel, no split) whatever the other path options say -- also
    when they are set while it is on -- and a small scene is evaluated by the one-launch kernel again once it is cleared."""
    w = pfc.configs.c1_boxes()
    m = pfc.configs.build_scenario(w)
    a = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert m.last_parts() == 0     # the one-launch kernel
    m.set_option("fixed_order", 1)
    m.set_option("fused", 1); m.set_option("graph", 1); m.set_option("split_min", 2)
    b = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert m.last_parts() == 1     # the batched launch sequence, one part
    b2 = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert all(np.array_equal(x, y) for x, y in zip(b, b2))
    m.set_option("split_min", 1025)
    m.set_option("fixed_order", 0)
    c = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    assert m.last_parts() == 0
    for x, y in zip(a, b):
        assert np.allclose(x, y, rtol=1e-9, atol=1e-12)
    for x, y in zip(a, c):
        assert np.allclose(x, y, rtol=1e-9, atol=1e-12)
    m.close()


def test_fixed_order_on_a_multi_device_handle(pfc):
    """The option reaches every shard of a multi-device handle (here {0, 0}): with the same item ranges -- the first evaluation of a
    handle cuts them by the leaf counts -- two fresh handles give the same bits, value and Dual."""
    w = pfc.configs.c5_pile()
    d6 = _seeds(w, 6, 5)
    outs = []
    for rep in range(2):
        m = pfc.configs.build_scenario(w, devices=[0, 0])
        m.set_option("fixed_order", 1)
        outs.append((m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids),
                     m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *d6, w.ins_ids)))
        assert m.last_shards() == 2
        m.close()
    for a, b in zip(outs[0], outs[1]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_fixed_order_growth_errors_and_poisoned_lists(pfc):
    """With the option on: the first evaluation of a fresh handle (work lists, record lists and sort buffers grow, the evaluation is
    re-issued) returns the bits of every later one, also from poisoned work lists; a non-finite pose is reported and the next
    evaluation is clean; three sampled items against the oracle."""
    L = pfc._lib
    w = pfc.configs.c3_blob_tool(300, n_div_blob=8, n_div_tool=6)
    m = pfc.configs.build_scenario(w)
    m.set_option("fixed_order", 1)
    first = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)      # minimal capacities: must grow
    assert m.stats()["candidates"] > 65536
    second = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    m.set_option("poison", 1)
    third = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    m.set_option("poison", 0)
    bad = w.pose.copy(); bad[7, 3] = np.nan
    with pytest.raises(L.PFCError) as ei:
        m.force_all_elastic_intersections(bad, w.twist, w.s, w.ins_ids)
    assert ei.value.status == L.ERR_NONFINITE
    fourth = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    for other in (second, third, fourth):
        for x, y in zip(first, other):
            assert np.array_equal(x, y)
    ref = H.oracle_run(pfc, w, items=[0, 150, 299], debug=False)
    for k, r in zip([0, 150, 299], ref):
        assert np.array_equal(first[2][k], r.counts)
        assert H.rel_err(first[0][k], r.wrench) < 1e-9
    m.close()


def test_fixed_order_sort_follows_the_list_length(pfc):
    """The sort covers a power of two above twice the previous evaluation's candidates, not the list's capacity (a reference-sized
    scene sorts 1 024 keys in one launch): a handle that has seen two items and is then given three hundred must notice that the
    list outgrew the covered part, evaluate again over the whole list, and return the bits a fresh handle returns -- and the other
    way round."""
    w = pfc.configs.c3_blob_tool(300, n_div_blob=8, n_div_tool=6)
    fresh = {}
    for n in (2, 300):
        f = pfc.configs.build_scenario(w)
        f.set_option("fixed_order", 1)
        fresh[n] = f.force_all_elastic_intersections(w.pose[:n], w.twist[:n], w.s[:n], w.ins_ids[:n])
        f.close()
    m = pfc.configs.build_scenario(w)
    m.set_option("fixed_order", 1)
    for n in (2, 2, 300, 300, 2, 2, 300):
        got = m.force_all_elastic_intersections(w.pose[:n], w.twist[:n], w.s[:n], w.ins_ids[:n])
        for x, y in zip(got, fresh[n]):
            assert np.array_equal(x, y), n
    m.close()


def test_fixed_order_first_evaluation_on_dirty_memory(pfc):
    """The first evaluation of a fresh handle overflows its minimal candidate list; the per-item counts then say more than the list
    holds, and cutting the list into per-item segments by them would leave slots nobody writes -- whatever the freshly allocated
    buffers held -- to be read as candidates.  Device memory is filled with a wild pattern first (fresh allocations are usually zero
    pages, which hides this); the evaluation must come back clean and with the bits of a second handle."""
    import torch
    junk = [torch.full((2 ** 28,), 0x7F7F7F7F, dtype=torch.int32, device="cuda") for _ in range(4)]      # 4 GiB
    torch.cuda.synchronize()
    del junk
    torch.cuda.empty_cache()
    w = pfc.configs.c5_pile()
    outs = []
    for rep in range(2):
        m = pfc.configs.build_scenario(w)
        m.set_option("fixed_order", 1)
        outs.append(m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids))
        assert m.stats()["candidates"] > 65536      # (the list started smaller than that)
        m.close()
    for x, y in zip(*outs):
        assert np.array_equal(x, y)


def test_fixed_order_item_with_more_candidates_than_a_segment_sort_takes(pfc):
    """The candidate list is put in order per item (a bitonic network in LDS, up to 4 096 candidates); an item with more is reported,
    the handle sorts the whole list from then on (rocPRIM) and the evaluation is re-issued: deep overlaps of the full-size C3 pair
    (tens of thousands of candidates per item) against the oracle, two handles bit-equal, value and Dual."""
    w = pfc.configs.c3_blob_tool(3, distance=0.12)
    d6 = _seeds(w, 6, 5)
    outs = []
    for rep in range(2):
        m = pfc.configs.build_scenario(w)
        m.set_option("fixed_order", 1)
        v = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        assert v[2][:, 1].max() > 4096
        d = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *d6, w.ins_ids)
        v2 = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        outs.append(tuple(v) + tuple(d) + tuple(v2))
        _check_vs_oracle(pfc, w, m, *v, tol=1e-9, oracle_debug=True)
        m.close()
    for x, y in zip(*outs):
        assert np.array_equal(x, y)
    for x, y in zip(outs[0][:3], outs[0][8:]):
        assert np.array_equal(x, y)
