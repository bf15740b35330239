import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pfc_pkg
pfc = pfc_pkg.load()
rng = np.random.default_rng(11)
w = pfc.configs.c5_pile()
n = w.n_items; nd = 2
d_pose = rng.standard_normal((n, nd, 24)) * 1e-2
d_twist = rng.standard_normal((n, nd, 6)) * 0.1
d_s = rng.standard_normal((n, nd, 6)) * 1e-3
outs = []
for k in range(3):
    f = pfc.configs.build_scenario(w)
    outs.append(f.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, d_pose, d_twist, d_s, w.ins_ids))
    f.close()
a, b = outs[0], outs[1]
dw = np.abs(a[2] - b[2]); scale = np.abs(b[2]).max()
print("two fresh handles, same inputs: max |d_wrench diff| / max |d_wrench| = %.3e;  wrench %.3e" % (dw.max() / scale, np.abs(a[0] - b[0]).max() / np.abs(b[0]).max()))
it = np.unravel_index(np.argmax(dw), dw.shape)
print("worst item", it[0], "counts", a[4][it[0]], "d_wrench a", a[2][it[0], it[1]], "b", b[2][it[0], it[1]])
# per-item relative differences
per = dw.reshape(n, -1).max(1) / np.maximum(np.abs(b[2]).reshape(n, -1).max(1), 1e-300)
print("items with relative difference > 1e-6:", int((per > 1e-6).sum()), "of", int((a[4][:, 3] > 0).sum()), "in contact; > 1e-2:", int((per > 1e-2).sum()))
