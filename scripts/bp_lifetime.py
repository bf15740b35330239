"""Diagnostic: lifetimes of the broadphase workgroups of one launch (span, first exit, mean lifetime, where the time goes).
Needs the variant build  EXTRA_FLAGS=-DPFC_STAMPS bash scripts/mkvar.sh $PWD/scripts/variants/bp_lifetime.py bplife  and
PFC_LIB=.../build/variants/bplife.so PFC_ALLOW_DIAGNOSTIC=1.  usage: bp_lifetime.py [poses] [split_min]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
w = pfc.configs.c3_blob_tool(n)
m = pfc.configs.build_scenario(w)
if len(sys.argv) > 2: m.set_option("split_min", int(sys.argv[2]))
for _ in range(3):
    m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
out = (C.c_longlong * 16)()
pfc._lib.lib().pfc_debug_stamps(m._h, out)
v = [int(x) & 0xFFFFFFFFFFFFFFFF for x in out]
M = 0xFFFFFFFFFFFFFFFF
nwg, nseed = v[1], v[2]
life = v[0] * 10e-3   # us total
first_in = (~v[4]) & M; last_out = v[3]; first_out = (~v[6]) & M
print("workgroups", nwg, "seeds", nseed, "seeds/wg %.1f" % (nseed / nwg))
print("kernel span (first entry -> last exit) %.1f us; first exit after %.1f us; mean lifetime %.1f us" % ((last_out - first_in) * 0.01, (first_out - first_in) * 0.01, life / nwg))
clk = 2400.0
for k, nm in ((8, "iterations"), (9, "settle iterations"), (10, "seed set-up"), (13, "flush"), (11, "seed end + ticket")):
    print("  %-20s %8.1f us per workgroup (%.1f %% of lifetime)" % (nm, v[k] / clk / nwg, 100.0 * v[k] / clk / life))
