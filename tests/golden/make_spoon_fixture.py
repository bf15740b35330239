"""Writes tests/golden/spoon_quads.npz: the vertex and face DATA of the reference's test mesh test/data/spoon.obj (2 504
vertices, 2 502 quads; used by test/spoon.jl:16,39-41) -- a data file the reference's own tests hold, kept as a fixture so that
real (non-synthetic) geometry runs through the path on the GPU box, where /root/reference does not exist.

usage: python tests/golden/make_spoon_fixture.py [/root/reference/test/data/spoon.obj]"""
import os
import sys

import numpy as np

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/test/data/spoon.obj"
v, f = [], []
for ln in open(src):
    t = ln.split()
    if not t:
        continue
    if t[0] == "v":
        v.append([float(x) for x in t[1:4]])
    elif t[0] == "f":
        f.append([int(x.split("/")[0]) - 1 for x in t[1:]])      # OBJ indices are 1-based; vertex index only
assert all(len(q) == 4 for q in f), "the spoon is an all-quad mesh"
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "spoon_quads.npz")
np.savez_compressed(out, point=np.asarray(v, dtype=np.float64), quad=np.asarray(f, dtype=np.int32))
print(out, len(v), "vertices,", len(f), "quads")
