"""Frozen vectors (tests/golden/oracle_vectors.npz, provenance in tests/golden/make_golden.py: this repository's CPU
oracle, NOT the Julia reference).  CPU: the oracle still produces them.  GPU: the HIP path matches them."""
import os
import sys

import numpy as np
import pytest

import helpers as H

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as MG   # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "oracle_vectors.npz"))


def _close(a, b, tol):
    scale = max(float(np.abs(b).max()), 1e-300)
    return float(np.abs(a - b).max()) <= tol * scale


def test_oracle_reproduces_golden(pfc, O):
    for name, w in MG.cases(pfc):
        ref = H.oracle_run(pfc, w)
        assert np.array_equal(np.array([r.counts for r in ref], dtype=np.int32), GOLD[name + "/counts"]), name
        assert np.array_equal(np.array([int(np.sum(r.clip_n)) for r in ref]), GOLD[name + "/n_clip_vertices"]), name
        # same source, same flags: bit-identical unless the compiler changes; a loose bound keeps the test portable
        assert _close(np.array([r.wrench for r in ref]), GOLD[name + "/wrench"], 1e-12), name
        assert _close(np.array([r.sdot for r in ref]), GOLD[name + "/sdot"], 1e-6), name


@pytest.mark.gpu
def test_hip_matches_golden(pfc):
    for name, w in MG.cases(pfc):
        m = pfc.configs.build_scenario(w, debug=True)
        d_pose, d_twist, d_s = MG.dual_seeds(w)
        wr, sd, dw, dsd, ct = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, d_s, w.ins_ids)
        assert np.array_equal(ct, GOLD[name + "/counts"]), name
        wr2, sd2, ct2 = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        n_clip = [int(np.sum(m.debug_pairs(k)[1])) for k in range(w.n_items)]
        assert np.array_equal(np.array(n_clip), GOLD[name + "/n_clip_vertices"]), name
        assert _close(wr, GOLD[name + "/wrench"], 1e-9) and _close(wr2, GOLD[name + "/wrench"], 1e-9), name
        assert _close(sd, GOLD[name + "/sdot"], 1e-6), name
        assert _close(dw, GOLD[name + "/d_wrench"], 1e-6), name
        assert _close(dsd, GOLD[name + "/d_sdot"], 1e-5), name
        m.close()
