"""Shared test helpers: run a configs.Workload through the CPU oracle."""
import numpy as np

from oracle import oracle as O


def oracle_meshes(w):
    return [O.OracleMesh(ms.mesh, ms.tree, ms.Ebar or 0.0) for ms in w.meshes]


def oracle_ins(pfc, c):
    mu_s, mu_d = pfc.scenario.determine_mu_s_mu_d(c.mu_s, c.mu_d)
    if c.model == "regularized":
        return O.make_ins(c.chi, c.n_quad_rule, O.REGULARIZED, mu_s, mu_d, v_c=c.v_tol)
    return O.make_ins(c.chi, c.n_quad_rule, O.BRISTLE, mu_s, mu_d, tau=c.tau, k_bar=c.k_bar, magic=c.magic)


def oracle_run(pfc, w, items=None, debug=True):
    """Per-item oracle evaluation (force_single_elastic_intersection!).  Returns a list of EvalResult."""
    om = oracle_meshes(w)
    oi = [oracle_ins(pfc, c) for c in w.instructions]
    out = []
    for k in (range(w.n_items) if items is None else items):
        c = w.instructions[int(w.ins_ids[k])]
        out.append(O.evaluate(om[c.id_1], om[c.id_2], oi[int(w.ins_ids[k])], w.pose[k], w.twist[k], w.s[k],
                              debug=debug))
    return out


def sorted_pairs(pairs, clip_n):
    """Canonical order for comparing candidate sets: sort by (i_1, i_2)."""
    pairs = np.asarray(pairs).reshape(-1, 2)
    clip_n = np.asarray(clip_n).reshape(-1)
    order = np.lexsort((pairs[:, 1], pairs[:, 0]))
    return pairs[order], clip_n[order]


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    scale = max(np.linalg.norm(b), 1e-300)
    return float(np.linalg.norm(a - b) / scale)


# ----------------------------------------------------------------------------------------------------------------
# One (mesh_1, mesh_2, instruction, pose) scene through either backend: "oracle" (CPU restatement) or "hip" (the
# library, through the C ABI, debug views on).  The reference's scene-level tests read m.float.bodyBodyCache after
# one calcXd (test/test_normal.jl:31-41, test/test_friction.jl:228-236,251-256); both backends return the same view.
# ----------------------------------------------------------------------------------------------------------------
class SceneResult:
    __slots__ = ("status", "wrench", "sdot", "counts", "trac", "has_K", "K", "Kbar_inv_sqrt", "Sinv", "cop", "pairs", "clip_n")


def normal_wrench_from_tractions(trac):
    """normal_wrench(b) (src/contact_algorithms_normal.jl:2-15) over a TractionCache dump (n 3, r 3, dA, p)."""
    pdA = trac[:, 6] * trac[:, 7]
    lin = (pdA[:, None] * trac[:, 0:3]).sum(axis=0)
    ang = np.cross(trac[:, 3:6], pdA[:, None] * trac[:, 0:3]).sum(axis=0)
    return np.concatenate([ang, lin])


def eval_scene(backend, pfc, mesh_1, Ebar_1, mesh_2, Ebar_2, ins, pose, twist, s=None, trees=None, options=None,
               want_pairs=False):
    """ins: dict(model="regularized"|"bristle", chi, n_quad, mu_s, mu_d, v_c | tau, k_bar, magic).
    trees: optional (OBBTree, OBBTree) supplied by the host instead of the library's builder."""
    G = pfc.geometry
    t1, t2 = trees if trees is not None else (G.build_tree(mesh_1), G.build_tree(mesh_2))
    r = SceneResult()
    if backend == "oracle":
        m1, m2 = O.OracleMesh(mesh_1, t1, Ebar_1 or 0.0), O.OracleMesh(mesh_2, t2, Ebar_2 or 0.0)
        if ins["model"] == "regularized":
            oi = O.make_ins(ins["chi"], ins["n_quad"], O.REGULARIZED, ins["mu_s"], ins["mu_d"], v_c=ins["v_c"])
        else:
            oi = O.make_ins(ins["chi"], ins["n_quad"], O.BRISTLE, ins["mu_s"], ins["mu_d"], tau=ins["tau"],
                            k_bar=ins["k_bar"], magic=ins["magic"])
        e = O.evaluate(m1, m2, oi, pose, twist, s)
        r.status, r.wrench, r.sdot, r.counts, r.trac = e.status, e.wrench, e.sdot, e.counts, e.trac
        r.has_K, r.K, r.Kbar_inv_sqrt, r.Sinv, r.cop = e.has_K, e.K, e.Kbar_inv_sqrt, e.Sinv, e.cop
        r.pairs, r.clip_n = e.pairs, e.clip_n
        return r
    assert backend == "hip"
    S = pfc.scenario
    m = S.MechanismScenario()
    i1 = m.add_contact("mesh_1", mesh_1, c_prop=None if mesh_1.tri is not None else S.ContactProperties(Ebar_1), tree=t1)
    i2 = m.add_contact("mesh_2", mesh_2, c_prop=S.ContactProperties(Ebar_2), tree=t2)
    if ins["model"] == "regularized":
        m.add_friction_regularize(i1, i2, mu_s=ins["mu_s"], mu_d=ins["mu_d"], chi=ins["chi"], v_tol=ins["v_c"],
                                  n_quad_rule=ins["n_quad"])
    else:
        m.add_friction_bristle(i1, i2, tau=ins["tau"], k_bar=ins["k_bar"], mu_s=ins["mu_s"], mu_d=ins["mu_d"],
                               chi=ins["chi"], n_quad_rule=ins["n_quad"], magic=ins["magic"])
    m.finalize()
    m.set_option("debug", 1)
    for k, v in (options or {}).items():
        m.set_option(k, v)
    pose = np.asarray(pose, dtype=np.float64).reshape(1, 24)
    twist = np.asarray(twist, dtype=np.float64).reshape(1, 6)
    s_in = np.zeros((1, 6)) if s is None else np.asarray(s, dtype=np.float64).reshape(1, 6)
    wrench, sdot, counts = m.force_all_elastic_intersections(pose, twist, s_in)
    r.status, r.wrench, r.sdot, r.counts = 0, wrench[0], sdot[0], counts[0]
    r.trac = m.debug_tractions(0)
    r.pairs, r.clip_n = m.debug_pairs(0) if want_pairs else (None, None)
    st = m.debug_stiffness(0) if ins["model"] == "bristle" else None
    r.has_K = st is not None
    r.K, r.Kbar_inv_sqrt, r.Sinv, r.cop = st if st is not None else (None, None, None, None)
    m.close()
    return r


def fuzz_workload(pfc, rng, n_items, degenerate, tet_tet=False):
    """Random small meshes in random relative poses: a box surface (12 triangles), a sphere surface and a tet box /
    tet sphere with random scales; poses put the surfaces at random depths through the tets.  degenerate: poses
    snapped to axis-aligned rotations and lattice offsets, so that triangle vertices and edges land exactly on tet
    faces (the strict / non-strict inside tests of static_clip.jl:140-188 and the zero ties of the trivial reject)."""
    G, Cf = pfc.geometry, pfc.configs
    box_tet = G.as_tet_emesh(G.emesh_box_div(np.array([0.5, 0.5, 0.5]), 2))
    sph_tet = G.as_tet_emesh(G.emesh_sphere(0.5, 3))
    box_tri = G.as_tri_emesh(G.emesh_box(np.array([0.25, 0.25, 0.25])))
    sph_tri = G.as_tri_emesh(G.emesh_sphere(0.3, 2))
    meshes = [Cf._mesh("box_tri", box_tri), Cf._mesh("sph_tri", sph_tri), Cf._mesh("box_tet", box_tet, 1.0e6),
              Cf._mesh("sph_tet", sph_tet, 2.0e6)]
    ins = [Cf.InsSpec(0, 2, "regularized", chi=0.5, mu_d=0.3, v_tol=1e-2), Cf.InsSpec(1, 2, "bristle", chi=0.3, mu_d=0.4),
           Cf.InsSpec(0, 3, "bristle", chi=0.5, mu_d=0.3, n_quad_rule=1), Cf.InsSpec(1, 3, "regularized", chi=0.1, mu_d=0.2, v_tol=1e-3)]
    if tet_tet:    # volume-volume instructions in the same scenario: the TT kernel variants serve all six
        ins += [Cf.InsSpec(2, 3, "regularized", chi=0.5, mu_d=0.3, v_tol=1e-2), Cf.InsSpec(3, 2, "bristle", chi=0.4, mu_d=0.3)]
    ids = rng.integers(0, len(ins), n_items).astype(np.int32)
    pose, twist, s = [], [], []
    quarter = [np.eye(3), np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]]), np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0.0]]),
               np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0.0]])]
    for k in range(n_items):
        if degenerate:
            R = quarter[rng.integers(0, 4)] @ quarter[rng.integers(0, 4)]
            t = rng.integers(-3, 4, 3) * 0.125
        else:
            R = Cf.random_rotation(rng)
            t = rng.uniform(-0.6, 0.6, 3)
        pose.append(pfc.scenario.relative_pose(R, t, np.eye(3), np.zeros(3)))
        twist.append(rng.standard_normal(6) * np.array([1, 1, 1, 0.1, 0.1, 0.1]))
        s.append(rng.standard_normal(6) * 1e-3)
    return Cf.Workload("fuzz", meshes, ins, ids, np.array(pose), np.array(twist), np.array(s))
