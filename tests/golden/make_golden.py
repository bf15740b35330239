"""Regenerates tests/golden/oracle_vectors.npz.

PROVENANCE: these vectors are outputs of THIS repository's CPU oracle (oracle/pfc_oracle.c and pfc_oracle_dual.cpp) on
seeded synthetic scenes -- NOT outputs of the Julia reference, which cannot run in the build container or on the GPU
box (no Julia toolchain).  The reference's own tests hold no golden vectors for this path; what pins the
oracle to the reference are the restated analytic / property tests in tests/test_oracle_*.py.  The fixture exists so
that (a) a change of the oracle's results is noticed, and (b) the HIP path is also checked against numbers that were
frozen at commit time rather than recomputed by the same run.

usage: python tests/golden/make_golden.py        (needs the oracle library only, no GPU)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def cases(pfc):
    cf = pfc.configs
    yield "c1_boxes", cf.c1_boxes()
    yield "c2_box_on_plane", cf.c2_box_on_plane(1)
    yield "c4_montecarlo_6", cf.c2_box_on_plane(6, montecarlo=True)
    w = cf.c3_blob_tool(3, seed=99, n_div_blob=6, n_div_tool=4)
    w.s[:] = np.random.default_rng(3).standard_normal(w.s.shape) * 1e-3
    yield "c3_reduced_bristle", w
    yield "vol_vol_regularized", cf.vol_vol(2, n_div=3, model="regularized")
    yield "vol_vol_bristle", cf.vol_vol(2, n_div=3, model="bristle")


def dual_seeds(w):
    from test_oracle_dual import tangents
    rng = np.random.default_rng(11)
    n = w.n_items
    dq = rng.standard_normal((n, 3, 6)) * np.array([1, 1, 1, 0.05, 0.05, 0.05])
    d_pose = np.stack([tangents(w.pose[k][:9].reshape(3, 3, order="F"), w.pose[k][9:12], dq[k]) for k in range(n)])
    return d_pose, rng.standard_normal((n, 3, 6)) * 0.1, rng.standard_normal((n, 3, 6)) * 1e-3


def main():
    import pfc_pkg
    import helpers as H
    from oracle import oracle as O
    pfc = pfc_pkg.load()
    out = {}
    for name, w in cases(pfc):
        ref = H.oracle_run(pfc, w)
        out[name + "/wrench"] = np.array([r.wrench for r in ref])
        out[name + "/sdot"] = np.array([r.sdot for r in ref])
        out[name + "/counts"] = np.array([r.counts for r in ref], dtype=np.int32)
        out[name + "/n_clip_vertices"] = np.array([int(np.sum(r.clip_n)) for r in ref], dtype=np.int64)
        om = H.oracle_meshes(w)
        d_pose, d_twist, d_s = dual_seeds(w)
        dw, dsd = [], []
        for k in range(w.n_items):
            c = w.instructions[int(w.ins_ids[k])]
            st, _, _, a, b = O.evaluate_dual(om[c.id_1], om[c.id_2], H.oracle_ins(pfc, c), w.pose[k], w.twist[k], w.s[k],
                                             d_pose[k], d_twist[k], d_s[k])
            assert st == 0
            dw.append(a); dsd.append(b)
        out[name + "/d_wrench"] = np.array(dw)
        out[name + "/d_sdot"] = np.array(dsd)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
