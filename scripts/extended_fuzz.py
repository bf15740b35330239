import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import helpers as H
import test_gpu_parity as T
import pfc_pkg
pfc = pfc_pkg.load()
bad = 0
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12      # 4 x 400 fuzz items per seed, each against the CPU oracle
fixed = "--fixed" in sys.argv      # option fixed_order (debug lists off: the clip-only kernel + k_integ_fixed / k_shift_fixed / k_fixed_reduce)
for seed in range(200, 200 + n_seeds):
    for degenerate in (False, True):
        for tet_tet in (False, True):
            rng = np.random.default_rng(seed)
            w = T._fuzz_workload(pfc, rng, 400, degenerate, tet_tet)
            if fixed:
                m = pfc.configs.build_scenario(w)
                m.set_option("fixed_order", 1)
                wrench, sdot, counts = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
                again = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
                if not all(np.array_equal(x, y) for x, y in zip((wrench, sdot, counts), again)):
                    bad += 1
                    print("NOT REPRODUCED", seed, degenerate, tet_tet)
            else:
                m, wrench, sdot, counts = T._eval(pfc, w)
            ref = H.oracle_run(pfc, w)
            for k in range(w.n_items):
                ok = np.array_equal(counts[k], ref[k].counts)
                if np.linalg.norm(ref[k].wrench) > 0:
                    ok = ok and H.rel_err(wrench[k], ref[k].wrench) < T.TOL_TIGHT
                else:
                    ok = ok and np.linalg.norm(wrench[k]) == 0.0
                if not ok:
                    bad += 1
                    print("MISMATCH", seed, degenerate, tet_tet, k, counts[k], ref[k].counts)
            m.close()
    print("seed", seed, "done, mismatches so far", bad, flush=True)
print(("option fixed_order: " if fixed else "") + f"{n_seeds * 4 * 400} fuzz items ({n_seeds} seeds x regular / degenerate x tri-tet / tet-tet), total mismatches", bad)
