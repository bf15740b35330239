import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
for name, w in (("c5", pfc.configs.c5_pile()), ("c3 x128", pfc.configs.c3_blob_tool(128)), ("c3 x16", pfc.configs.c3_blob_tool(16))):
    m = pfc.configs.build_scenario(w)
    m.set_option("fused", 0); m.set_option("team", 0); m.set_option("split_min", 0)
    for _ in range(3):
        try: m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        except Exception: pass
    m.set_option("profile", 1)
    acc = {}
    for _ in range(10):
        try: m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        except Exception: pass
        for k, v in m.stage_ms().items(): acc[k] = acc.get(k, 0) + v / 10
    print(os.environ.get("PFC_LIB", "product").split("/")[-1], name, "final %.1f us  setup %.1f us" % (acc["final"] * 1e3, acc["setup"] * 1e3), flush=True)
