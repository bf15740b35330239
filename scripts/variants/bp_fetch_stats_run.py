import ctypes as C, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
cfg = sys.argv[1]; L = int(sys.argv[2])
w = pfc.configs.c5_pile() if cfg == "c5" else pfc.configs.c3_blob_tool(int(sys.argv[3]))
m = pfc.configs.build_scenario(w)
m.set_option("bfs_levels", L); m.set_option("split_min", 0); m.set_option("fused", 0)
for _ in range(3): m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
out = (C.c_longlong * 16)()
pfc._lib.lib().pfc_debug_stamps(m._h, out)
v = [int(x) for x in out]
it = max(v[11], 1)
print(cfg, "L", L, "iterations", v[11], "seeds", v[14], "mean fetch %.0f" % (v[8] / it), "| fetch > 4000 cycles: %d iterations, mean %.0f" % (v[3], v[4] / max(v[3], 1)), "| max", v[5], "| mean FIRST iteration of a seed %.0f" % (v[6] / max(v[14], 1)))
tb = v[8] + v[9] + v[10] + v[13]
print("   per iteration: fetch %.0f  test %.0f  ballots+barrier %.0f  push+barrier(+flush) %.0f  pairs %.1f" % (v[8] / it, v[9] / it, v[10] / it, v[13] / it, v[12] / it))
