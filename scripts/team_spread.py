"""With build/variants/team_spread.so (scripts/variants/team_spread.py): spread of the ranks' broadphase / pass-0 end times of team
evaluations.  usage: PFC_LIB=build/variants/team_spread.so PFC_ALLOW_DIAGNOSTIC=1 python scripts/team_spread.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
for name, mk in (("c3 single", lambda s: pfc.configs.c3_blob_tool(1, seed=s)), ("c2", lambda s: pfc.configs.c2_box_on_plane(1, seed=s))):
    for seed in (20260103, 101, 102, 103):
        w = mk(seed)
        m = pfc.configs.build_scenario(w)
        for _ in range(30):
            out = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        st = (C.c_longlong * 16)()
        pfc._lib.lib().pfc_debug_stamps(m._h, st)
        v = [int(x) for x in st]
        print("%-10s seed %9d team %2d: broadphase ends %.1f .. %.1f us, pass 0 ends %.1f .. %.1f us after block 0's start; node tests per rank %d .. %d (total %d); kernel %.1f us" %
              (name, seed, m.last_team(), v[12] / 100, v[13] / 100, v[14] / 100, v[15] / 100, v[10], v[11], out[2][0][0], (v[9] - v[0]) / 100), flush=True)
        m.close()
