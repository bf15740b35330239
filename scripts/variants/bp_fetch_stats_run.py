import ctypes as C, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
for name, w, L in (("c3 full x16", pfc.configs.c3_blob_tool(16), -1), ("c3 full x128", pfc.configs.c3_blob_tool(128), -1), ("c5", pfc.configs.c5_pile(), -1)):
    m = pfc.configs.build_scenario(w)
    m.set_option("fused", 0); m.set_option("team", 0); m.set_option("split_min", 0)
    for _ in range(3): m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    out = (C.c_longlong * 16)()
    pfc._lib.lib().pfc_debug_stamps(m._h, out)
    v = [int(x) for x in out]; it = max(v[11], 1)
    print("%-14s iterations %d seeds %d | per iteration (cycles): fetch %.0f  test %.0f  ballots+barrier %.0f  push+barrier(+flush) %.0f  pairs %.1f | fetch > 4000: %d (mean %.0f) max %d" % (
        name, v[11], v[14], v[8] / it, v[9] / it, v[10] / it, v[13] / it, v[12] / it, v[3], v[4] / max(v[3], 1), v[5]), flush=True)
