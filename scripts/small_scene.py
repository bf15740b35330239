"""Diagnostic: N host-buffer evaluations of one small config (c1 | c2 | c3 | c4 | c3r = reduced bristle) for a kernel trace
or, with a -DPFC_STAMPS build (scripts/build_stamps.sh + PFC_LIB=...), the phase stamps of the fused kernel's block 0.
usage: small_scene.py <config> [n_evals] [fused 0|1]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fused = int(sys.argv[3]) if len(sys.argv) > 3 else 1
w = {"c1": pfc.configs.c1_boxes, "c2": lambda: pfc.configs.c2_box_on_plane(1), "c3": lambda: pfc.configs.c3_blob_tool(1),
     "c3r": lambda: pfc.configs.c3_blob_tool(4, n_div_blob=8, n_div_tool=6),
     "pencil": lambda: pfc.configs.c3_blob_tool(8, seed=3, n_div_blob=4, n_div_tool=2, distance=0.195),
     "c4": lambda: pfc.configs.c2_box_on_plane(256, montecarlo=True)}[cfg]()
m = pfc.configs.build_scenario(w)
m.set_option("fused", fused)
for _ in range(10):
    m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
# median over blocks of 25 evaluations: a process sees ONE ~36 ms stall some 100-150 launches in (runtime / power-state
# housekeeping, with polling and with hipStreamSynchronize alike), which a plain mean over a few hundred calls would smear
blocks = []
for _ in range(max(n // 25, 1)):
    t0 = time.perf_counter()
    for _ in range(25):
        out = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
    blocks.append((time.perf_counter() - t0) / 25)
dt = float(np.median(blocks))
print(f"{cfg} fused={fused} path={m.last_parts()}: {dt*1e6:.1f} us/eval (python caller), counts[0]={out[2][0]}")
if pfc._lib.lib().pfc_build_info() & 1:
    st = (C.c_longlong * 16)()
    pfc._lib.lib().pfc_debug_stamps(m._h, st)
    v = [int(x) for x in st]
    names = ["item load", "node cache", "broadphase", "clip round 0", "integrate", "reduce+passes", "epilogue"]
    v[7] = v[9]      # slot 7 of the stamps view carries a broadphase statistic
    print("block 0 phases (us):", ", ".join(f"{nm} {(v[k+1]-v[k])/100:.2f}" for k, nm in enumerate(names)), f"| total {(v[7]-v[0])/100:.2f} | node tests {v[8]} | shader clock {(v[11]-v[10])/max(v[7]-v[0],1)*100:.0f} MHz")
if pfc._lib.lib().pfc_build_info() & 1 and m.last_parts() == 0 and m.last_team() > 1:
    # team kernel (k_fused<.., true>): slots 12..15 are rank 0's stamps behind team sum 0, team sum 1, the eigen-decomposition, team sum 2
    if v[13] > v[12]:      # bristle: stamps behind the exchange, the shift to the cop, the eigen-decomposition, the friction pass + team sum
        print("team of %d (rank 0, us): pass 0 end -> exchange %.2f, -> moments at the cop %.2f, -> eigen %.2f, -> friction pass + team sum %.2f, -> end %.2f" % (m.last_team(), (v[12]-v[5])/100, (v[13]-v[12])/100, (v[14]-v[13])/100, (v[15]-v[14])/100, (v[9]-v[15])/100))
    else:
        print("team of %d (rank 0, us): pass 0 end -> exchange %.2f" % (m.last_team(), (v[12]-v[5])/100))
elif pfc._lib.lib().pfc_build_info() & 1:
    it = max(v[15] & 0xFFFF, 1)
    print(f"broadphase iterations {it}: cycles/iteration pop+fetch {v[12]/it:.0f}, test {v[13]/it:.0f}, ballot+barrier {v[14]/it:.0f}, push+barrier {(v[15]>>16)/it:.0f}")
m.close()
