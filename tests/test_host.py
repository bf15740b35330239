"""CPU tests of the host logic: mesh generators, OBB-tree builder, scenario API mirror, C-ABI exports, and the
multi-process sharding + all-gather path (gloo, world_size 2).  No compute call on libpfc_hip (no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- geometry (reference: test/test_geometry/test_mesh.jl, test_blob.jl; src/geometry/mesh.jl) --------------------
def test_box_and_half_plane(pfc):
    G = pfc.geometry
    b = G.emesh_box(0.05)
    assert (b.n_point, b.n_tri, b.n_tet) == (9, 12, 12)
    assert np.all(G.tet_volume(b.point[b.tet]) > 0)
    assert np.sum(G.tet_volume(b.point[b.tet])) == pytest.approx(0.1 ** 3)
    # outward triangles: area-vector sum is zero, volume from the divergence theorem is positive
    P = b.point[b.tri]
    av = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 1]) * 0.5
    np.testing.assert_allclose(av.sum(axis=0), 0, atol=1e-15)
    assert np.sum(np.einsum("ij,ij->i", P.mean(axis=1), av)) / 3 == pytest.approx(0.1 ** 3)
    assert b.eps.min() == 0.0 and b.eps.max() == 1.0
    hp = G.emesh_half_plane()
    assert (hp.n_point, hp.n_tri, hp.n_tet) == (4, 1, 1)
    assert G.tet_volume(hp.point[hp.tet])[0] > 0


def test_sphere_counts_and_volume(pfc):
    G = pfc.geometry
    for n_div in (1, 2, 4):
        s = G.emesh_sphere(0.1, n_div)
        assert s.n_tri == 20 * n_div ** 2 and s.n_tet == s.n_tri
        assert s.n_point == 10 * n_div ** 2 + 2 + 1          # closed triangulated sphere + the centre
        np.testing.assert_allclose(np.linalg.norm(s.point[:-1], axis=1), 0.1, rtol=1e-12)
        assert np.all(G.tet_volume(s.point[s.tet]) > 0)
    s = G.emesh_sphere(1.0, 8)
    assert np.sum(G.tet_volume(s.point[s.tet])) == pytest.approx(4 / 3 * np.pi, rel=0.02)


def test_box_div_and_surface_extraction(pfc):
    G = pfc.geometry
    b = G.emesh_box_div(0.05, 9)
    assert b.n_tet == 972 and b.n_point == 6 * 81 + 2 + 1
    assert np.sum(G.tet_volume(b.point[b.tet])) == pytest.approx(0.1 ** 3)
    surf = G.as_tri_emesh(G.as_tet_emesh(b))       # face opposite the largest-eps vertex of every tet
    assert surf.n_tri == 972
    P = surf.point[surf.tri]
    av = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 1]) * 0.5
    assert np.sum(np.linalg.norm(av, axis=1)) == pytest.approx(6 * 0.1 ** 2)
    np.testing.assert_allclose(av.sum(axis=0), 0, atol=1e-15)


def test_emesh_validation(pfc):
    G = pfc.geometry
    b = G.emesh_box(1.0)
    with pytest.raises(ValueError, match="inverted"):
        G.EMesh(b.point, None, b.tet[:, [1, 0, 2, 3]], b.eps)
    with pytest.raises(ValueError, match="zero on the surface"):
        G.EMesh(b.point, None, b.tet, b.eps + 0.5)
    with pytest.raises(ValueError):
        G.EMesh(b.point, None, None, None)
    with pytest.raises(ValueError, match="Tri,Tet"):
        G.build_tree(b)


def test_tree_structure(pfc):
    """test/test_geometry/test_blob.jl:2-18 on the sphere(320): leaf count, id set, depth < 1.3 log2 n; plus
    containment: every internal AABB contains the vertices below it, every leaf OBB contains its element."""
    G = pfc.geometry
    s = G.as_tet_emesh(G.emesh_sphere(1.0, 4))
    t = G.build_tree(s)
    n = s.n_tet
    assert n == 320 and t.n_leaf == n and t.n_node == 2 * n - 1
    assert sorted(t.leaf[t.leaf != G.INTERNAL].tolist()) == list(range(n))
    assert t.depth() < 1.3 * np.log2(n) + 1
    # containment, bottom-up
    below = [None] * t.n_node
    for k in range(t.n_node - 1, -1, -1):            # preorder numbering: children have larger indices
        if t.leaf[k] != G.INTERNAL:
            pts = s.point[s.tet[t.leaf[k]]]
            R = t.R[k].reshape(3, 3, order="F")
            loc = (pts - t.c[k]) @ R
            assert np.all(np.abs(loc) <= t.e[k] * (1 + 1e-12) + 1e-15)
            np.testing.assert_allclose(R.T @ R, np.eye(3), atol=1e-12)
            below[k] = pts
        else:
            assert np.array_equal(t.R[k], np.eye(3).reshape(9))
            pts = np.concatenate([below[t.child[k, 0]], below[t.child[k, 1]]])
            assert np.all(np.abs(pts - t.c[k]) <= t.e[k] * (1 + 1e-12) + 1e-15)
            below[k] = pts
    # tight tet OBB keeps the box of LARGEST surface area (src/obb/obb_construction.jl:35-40)
    p = s.point[s.tet[0]]; e = s.eps[s.tet[0]]
    c, ext, R = G.fit_tet_obb(p, e)
    perm = list(G.tet_perm_by_num(int(np.argmax(np.abs(e)))))
    areas = [G.obb_area(G.make_obb(p[perm], k)[1]) for k in range(3)]
    assert G.obb_area(ext) == max(areas)


def test_single_element_tree_keeps_aabb(pfc):
    G = pfc.geometry
    hp = G.as_tet_emesh(G.emesh_half_plane())
    t = G.build_tree(hp)
    assert t.n_node == 1 and t.leaf[0] == 0
    assert np.array_equal(t.R[0], np.eye(3).reshape(9))
    two = G.build_tree(G.emesh_ground(), "median")       # open patch: no adjacency for the blob builder
    assert two.n_node == 3 and two.leaf.tolist() == [G.INTERNAL, 0, 1]


# ---- scenario API mirror (reference: src/mechanism_scenario.jl) ----------------------------------------------------
def test_friction_defaults_and_canonical_order(pfc):
    S = pfc.scenario
    assert S.determine_mu_s_mu_d(None, None) == (0.5, 0.5)           # default_χ, sic (:350)
    assert S.determine_mu_s_mu_d(None, 0.3) == (0.3, 0.3)
    with pytest.raises(ValueError):
        S.determine_mu_s_mu_d(0.2, None)
    with pytest.raises(ValueError, match="something is wrong"):
        S.determine_mu_s_mu_d(0.2, 0.3)
    G = pfc.geometry
    m = pfc.MechanismScenario()
    i_plane = m.add_contact("plane", G.as_tet_emesh(G.emesh_half_plane()), c_prop=S.ContactProperties(1.0e6))
    i_box = m.add_contact("box", G.as_tri_emesh(G.emesh_box(0.05)))
    c = m.add_friction_regularize(i_plane, i_box, mu_d=0.2, chi=2.2)   # (tet, tri) is swapped to (tri, tet) :402-408
    assert (c.id_1, c.id_2) == (i_box, i_plane)
    assert c.friction_model.v_c == 0.01 and c.friction_model.v_mu_s == 0.02 and c.friction_model.v_mu_d == 0.03
    b = m.add_friction_bristle(i_box, i_plane, mu_d=0.3)
    assert (b.friction_model.tau, b.friction_model.k_bar, b.friction_model.magic) == (0.05, 1.0e4, 1.0e-3)
    assert b.friction_model.Ts_mu_s == 0.6 and b.friction_model.Ts_mu_d == pytest.approx(0.9)
    assert b.friction_model.bristle_id == 0 and m.n_bristle == 1
    assert m.find_mesh_id("box") == i_box
    with pytest.raises(KeyError):
        m.find_mesh_id("nope")
    with pytest.raises(ValueError, match="quadrature"):
        m.add_friction_regularize(i_box, i_plane, mu_d=0.2, n_quad_rule=3)
    with pytest.raises(ValueError, match="cannot be 0"):
        m.add_friction_bristle(i_box, i_plane, mu_d=0.0)
    with pytest.raises(TypeError):
        m.add_friction_regularize(i_box, i_box, mu_d=0.2)
    with pytest.raises(ValueError, match="ContactProperties for triangular"):
        m.add_contact("bad", G.as_tri_emesh(G.emesh_box(0.05)), c_prop=S.ContactProperties(1.0e6))
    with pytest.raises(ValueError, match="nothing as ContactProperties"):
        m.add_contact("bad", G.as_tet_emesh(G.emesh_box(0.05)))
    with pytest.raises(ValueError, match="unexpected range"):
        S.ContactProperties(1.0)
    with pytest.raises(RuntimeError, match="finalize"):
        m.force_all_elastic_intersections(np.zeros((1, 24)), np.zeros((1, 6)))


def test_relative_pose_and_twist(pfc):
    S, Cf = pfc.scenario, pfc.configs
    rng = np.random.default_rng(4)
    R1, R2 = Cf.random_rotation(rng), Cf.random_rotation(rng)
    t1, t2 = rng.standard_normal(3), rng.standard_normal(3)
    p = S.relative_pose(R1, t1, R2, t2)
    R21 = p[:9].reshape(3, 3, order="F"); t21 = p[9:12]
    R12 = p[12:21].reshape(3, 3, order="F"); t12 = p[21:24]
    x = rng.standard_normal(3)                        # a point in frame 1
    xw = R1 @ x + t1
    np.testing.assert_allclose(R21 @ x + t21, R2.T @ (xw - t2), atol=1e-14)
    np.testing.assert_allclose(R12 @ (R21 @ x + t21) + t12, x, atol=1e-14)
    # twist: velocity of a material point of body 2 relative to body 1, expressed in frame 2
    tw1, tw2 = rng.standard_normal(6), rng.standard_normal(6)
    tw = S.relative_twist(R2, t2, tw1, tw2)
    vw = (tw2[3:] + np.cross(tw2[:3], xw)) - (tw1[3:] + np.cross(tw1[:3], xw))
    x2 = R2.T @ (xw - t2)
    np.testing.assert_allclose(tw[3:] + np.cross(tw[:3], x2), R2.T @ vw, atol=1e-13)


def test_workload_generators(pfc):
    Cf = pfc.configs
    w = Cf.c1_boxes()
    assert w.n_items == 4 and [c.model for c in w.instructions] == ["regularized"] * 4
    w = Cf.c2_box_on_plane(3, montecarlo=True)
    assert w.meshes[1].mesh.n_tet == 972 and w.meshes[0].mesh.n_tri == 2 and w.pose.shape == (3, 24)
    w2 = Cf.c2_box_on_plane(3, montecarlo=True)
    assert np.array_equal(w.pose, w2.pose) and np.array_equal(w.twist, w2.twist)      # seeded, reproducible
    w = Cf.c3_blob_tool(2, n_div_blob=4, n_div_tool=3)
    assert w.meshes[1].mesh.n_tet == 320 and w.meshes[0].mesh.n_tri == 180 and w.s.shape == (2, 6)
    w = Cf.c5_pile(n_side=2, n_divs=(1, 2))
    assert w.n_items == 28 and len(w.meshes) == 16


# ---- C ABI ---------------------------------------------------------------------------------------------------------
def test_c_abi_exports_every_declared_symbol(pfc):
    """include/pfc.h <-> libpfc_hip.so <-> the ctypes table must agree symbol for symbol."""
    hdr = open(os.path.join(ROOT, "include", "pfc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pfc_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(pfc._lib.SIGNATURES), declared ^ set(pfc._lib.SIGNATURES)
    L = pfc._lib.lib()                                # binds every symbol; AttributeError if one is missing
    for name in declared:
        assert hasattr(L, name)
    assert L.pfc_version() == 100
    out = subprocess.run(["nm", "-D", "--defined-only", pfc._lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (pfc_[a-z_0-9]+)", out))
    assert declared <= exported


def test_the_package_ships_exactly_one_shared_library(pfc):
    """Variant / diagnostic builds (scripts/mkvar.sh, build_stamps.sh, elimination.sh) go to build/variants/, never into the
    package: whatever PFC_LIB could point at inside csrc/ is the product library, whose pfc_build_info() is 0."""
    import glob
    so = sorted(glob.glob(os.path.join(os.path.dirname(pfc._lib.LIB_PATH), "**", "*.so"), recursive=True))
    assert so == [pfc._lib.LIB_PATH], so
    if not os.environ.get("PFC_LIB"):
        assert pfc._lib.lib().pfc_build_info() == 0


def test_multi_device_handle_needs_devices_too(pfc):
    """pfc_create_multi without a usable HIP device fails like pfc_create (no CPU fallback); bad device lists are refused."""
    import ctypes as C
    import torch
    L = pfc._lib.lib()
    h = C.c_void_p()
    assert L.pfc_create_multi(None, 2, C.byref(h)) == pfc._lib.ERR_BAD_ARG and not h
    dv = (C.c_int * 2)(0, 0)
    assert L.pfc_create_multi(dv, 0, C.byref(h)) == pfc._lib.ERR_BAD_ARG and not h
    if not torch.cuda.is_available():
        assert L.pfc_create_multi(dv, 2, C.byref(h)) == pfc._lib.ERR_HIP and not h
        m = pfc.MechanismScenario(devices=[0, 0])
        G = pfc.geometry
        i1 = m.add_contact("plane", G.as_tet_emesh(G.emesh_half_plane()), c_prop=pfc.ContactProperties(1.0e6))
        i2 = m.add_contact("box", G.as_tri_emesh(G.emesh_box(0.05)))
        m.add_friction_regularize(i1, i2, mu_d=0.2)
        with pytest.raises(pfc._lib.PFCError) as ei:
            m.finalize()
        assert ei.value.status == pfc._lib.ERR_HIP


def test_no_cpu_fallback(pfc):
    """Without a HIP device the product path must fail loudly (pfc_create -> PFC_ERR_HIP), not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    m = pfc.MechanismScenario()
    G = pfc.geometry
    i1 = m.add_contact("plane", G.as_tet_emesh(G.emesh_half_plane()), c_prop=pfc.ContactProperties(1.0e6))
    i2 = m.add_contact("box", G.as_tri_emesh(G.emesh_box(0.05)))
    m.add_friction_regularize(i1, i2, mu_d=0.2)
    with pytest.raises(pfc._lib.PFCError) as ei:
        m.finalize()
    assert ei.value.status == pfc._lib.ERR_HIP
    # the product package never imports, loads or calls the oracle
    bad = re.compile(r"^\s*(from|import)\s+\S*oracle|libpfc_oracle|pfo_|pfc_oracle")
    pfc_dir = os.path.join(ROOT, "pressurefieldcontact.jl_amd")
    for dirpath, _, files in os.walk(pfc_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                for line in open(os.path.join(dirpath, f)):
                    assert not bad.search(line), (f, line)


# ---- multi-process sharding (gloo, world_size 2) ------------------------------------------------------------------
def test_shard_partitions(pfc):
    Pl = pfc.parallel
    parts = Pl.shard_block(10, 4)
    assert [p.tolist() for p in parts] == [[0, 1], [2, 3, 4], [5, 6], [7, 8, 9]]
    cost = [5, 1, 1, 1, 4, 4, 2, 2]
    parts = Pl.shard_by_cost(cost, 3)
    assert sorted(np.concatenate(parts).tolist()) == list(range(8))
    loads = [sum(cost[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= 2


_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import pfc_pkg, helpers as H
pfc = pfc_pkg.load()
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
w = pfc.configs.c2_box_on_plane(5, montecarlo=True, n_div=3)
def evaluator(idx):
    rs = H.oracle_run(pfc, w, items=idx, debug=False)
    return (np.array([r.wrench for r in rs]), np.array([r.sdot for r in rs]), np.array([r.counts for r in rs]))
parts = pfc.parallel.shard_by_cost([3, 1, 1, 1, 2], dist.get_world_size())
wrench, sdot, counts = pfc.parallel.evaluate_sharded(evaluator, w.n_items, parts)
ref = H.oracle_run(pfc, w, debug=False)
assert np.array_equal(wrench, np.array([r.wrench for r in ref]))
assert np.array_equal(counts, np.array([r.counts for r in ref]))
w2, _, _ = pfc.parallel.evaluate_sharded(evaluator, w.n_items)        # default block partition
assert np.array_equal(w2, wrench)
# the planned exchange (parallel.RowExchange: what bench.py --config C4 / C5 calls per step) gives the same rows
rank = dist.get_rank()
plan = pfc.parallel.RowExchange(parts, w.n_items, "cpu")
mw, ms, mc = evaluator(parts[rank])
for _ in range(2):      # the buffers are reused from call to call
    rows = plan(torch.as_tensor(mw), torch.as_tensor(ms), torch.as_tensor(mc.astype(np.int32)))
    pw, ps, pc = pfc.parallel.unpack_rows(rows)
    assert np.array_equal(pw, wrench) and np.array_equal(ps, sdot) and np.array_equal(pc, counts)
dist.barrier()
dist.destroy_process_group()
print("rank", os.environ["RANK"], "ok")
'''


def test_sharded_evaluation_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


def _build_c_example(tmp_path):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "box_on_plane")
    csrc = os.path.join(root, "pressurefieldcontact.jl_amd", "csrc")
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                    os.path.join(root, "examples", "box_on_plane.c"), "-L", csrc, "-lpfc_hip", f"-Wl,-rpath,{csrc}", "-lm",
                    "-o", exe], check=True)
    return exe


def test_c_example_builds_and_refuses_without_gpu(pfc, tmp_path):
    """examples/box_on_plane.c binds the C ABI from plain C (what a Julia ccall shim binds).  Without a HIP device it
    must stop at pfc_create: there is no CPU fallback behind the ABI."""
    import subprocess
    import torch
    exe = _build_c_example(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked test")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_c_example_matches_analytic_normal_wrench(pfc, tmp_path):
    """test/test_normal.jl:2-49 through the C ABI from a C program: exact normal wrench of a box on the half-plane."""
    import subprocess
    r = subprocess.run([_build_c_example(tmp_path), "200"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "us per pfc_eval" in r.stdout          # the timing loop ran its 200 repeated evaluations without an error


def test_array_addresses_through_the_buffer_protocol(pfc):
    """scenario._addr: the address the foreign call gets is the array's first element, for writable arrays (buffer protocol)
    and for read-only and empty ones (ndarray.ctypes)."""
    import sys
    sc = sys.modules[pfc.MechanismScenario.__module__]
    a = np.arange(72, dtype=np.float64).reshape(3, 24)
    assert sc._addr(a) == a.ctypes.data
    v = a[1:]                      # a C-contiguous view that does not start at the buffer's first byte
    assert v.flags.c_contiguous and sc._addr(v) == v.ctypes.data == a.ctypes.data + 24 * 8
    r = a.copy(); r.flags.writeable = False
    assert sc._addr(r) == r.ctypes.data
    e = np.zeros((0, 6))
    assert sc._addr(e) == e.ctypes.data
    i = np.arange(5, dtype=np.int32)
    assert sc._ia(i)[1] == i.ctypes.data and sc._da(a)[1] == a.ctypes.data
    assert sc._da([[1.0, 2.0]])[0].dtype == np.float64      # a list is converted first
