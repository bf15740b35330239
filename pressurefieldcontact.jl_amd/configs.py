"""Synthetic inputs of BASELINE.md §4 (configs C1..C5), built with the reference's mesh conventions.

Every function returns plain host data (meshes, trees, instruction parameters, per-item pose / twist / state
arrays); ``build_scenario`` turns that into a finalized MechanismScenario (HIP).  The same host data feeds the
CPU oracle in the tests and in bench.py's cpu_baseline leg.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import geometry as G
from .scenario import ContactProperties, MechanismScenario, relative_pose, relative_twist


@dataclass
class MeshSpec:
    name: str
    mesh: G.EMesh
    tree: G.OBBTree
    Ebar: Optional[float]        # None for a rigid triangle mesh


@dataclass
class InsSpec:
    id_1: int
    id_2: int
    model: str                   # "regularized" | "bristle"
    chi: float = 0.5
    n_quad_rule: int = 2
    mu_s: Optional[float] = None
    mu_d: Optional[float] = 0.3
    v_tol: float = 0.01
    tau: float = 0.05
    k_bar: float = 1.0e4
    magic: float = 1.0e-3


@dataclass
class Workload:
    name: str
    meshes: List[MeshSpec]
    instructions: List[InsSpec]
    ins_ids: np.ndarray          # (n_items,) int32
    pose: np.ndarray             # (n_items, 24)
    twist: np.ndarray            # (n_items, 6)
    s: np.ndarray                # (n_items, 6)
    meta: dict = field(default_factory=dict)

    @property
    def n_items(self): return int(self.ins_ids.shape[0])


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]])


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])


def random_rotation(rng) -> np.ndarray:
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _rng(seed: int, stream: int):
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed, spawn_key=(stream,))))


def _mesh(name, mesh, Ebar=None, method: str = "blob") -> MeshSpec:
    return MeshSpec(name, mesh, G.build_tree(mesh, method), Ebar)


# ----------------------------------------------------------------------------------------------------------------
def c1_boxes() -> Workload:
    """C1 = test/boxes.jl:18-45 restricted to its tri-tet instructions: plane(tet) / box_1(tri) and
    box_2(tet) / box_3(tri); the box_1/box_2 and box_3/box_4 instructions pair a tri mesh with a tet mesh too
    (rigid boxes are tri, compliant are tet), so all four are tri-tet.  Boxes are stacked 1 mm into each other."""
    r = 0.05
    plane = G.as_tet_emesh(G.emesh_half_plane())
    box_tri = G.as_tri_emesh(G.emesh_box(r))
    box_tet = G.as_tet_emesh(G.emesh_box(r))
    meshes = [_mesh("plane", plane, 1.0e6), _mesh("box_1", box_tri), _mesh("box_2", box_tet, 1.0e6),
              _mesh("box_3", box_tri), _mesh("box_4", box_tet, 1.0e6)]
    # add_friction_regularize!(plane, box_1, μd=0, χ=2.2) etc.; add_friction! puts the tri mesh first
    ins = [InsSpec(1, 0, "regularized", chi=2.2, mu_d=0.0), InsSpec(1, 2, "regularized", chi=0.2, mu_d=0.2),
           InsSpec(3, 2, "regularized", chi=0.2, mu_d=0.2), InsSpec(3, 4, "regularized", chi=0.2, mu_d=0.2)]
    pen = 0.001
    z = [0.0, r - pen, 3 * r - 2 * pen, 5 * r - 3 * pen, 7 * r - 4 * pen]      # plane, box_1..4 centre heights
    wz = [0.0, 1.0, 2.0, 3.0, 4.0]                                           # test/boxes.jl:41-44
    Rw = [np.eye(3)] + [rot_z(0.1 * k) for k in range(1, 5)]
    pose, twist = [], []
    for c in ins:
        b1, b2 = c.id_1, c.id_2
        t1, t2 = np.array([0, 0, z[b1]]), np.array([0, 0, z[b2]])
        pose.append(relative_pose(Rw[b1], t1, Rw[b2], t2))
        tw1 = np.array([0, 0, wz[b1], 0, 0, 0.0]); tw2 = np.array([0, 0, wz[b2], 0, 0, 0.0])
        twist.append(relative_twist(Rw[b2], t2, tw1, tw2))
    n = len(ins)
    return Workload("C1 boxes", meshes, ins, np.arange(n, dtype=np.int32), np.array(pose), np.array(twist),
                    np.zeros((n, 6)))


def c2_box_on_plane(n_scenes: int = 1, seed: int = 20260101, n_div: int = 9, montecarlo: bool = False) -> Workload:
    """C2 (n_scenes = 1) and C4 (n_scenes = 256, montecarlo = True): 972-tet box (half-width 0.05, 12 x 9^2 tets
    around a centre apex, Ē = 1e6) on a rigid 2-triangle ground (2 m square, z = 0), regularized friction
    μd = 0.3, v_tol = 1e-2, χ = 0.5, quadrature rule 2.  C2: penetration 2 mm, random yaw, at rest.
    C4: penetration U(0.5, 3) mm, roll/pitch U(-2°, 2°), yaw U(0, 2π), linear velocity U(-0.1, 0.1) m/s,
    angular velocity U(-1, 1) rad/s, PCG64 seed 20260101, stream = scene index."""
    r = 0.05
    ground = G.emesh_ground(1.0)
    box = G.as_tet_emesh(G.emesh_box_div(r, n_div))
    meshes = [_mesh("ground", ground, method="median"),      # open patch: the blob builder refuses it, as upstream
              _mesh("box", box, 1.0e6)]
    ins = [InsSpec(0, 1, "regularized", chi=0.5, mu_d=0.3, v_tol=1.0e-2)]
    pose, twist = [], []
    for k in range(n_scenes):
        g = _rng(seed, k)
        if montecarlo:
            pen = g.uniform(0.5e-3, 3.0e-3)
            roll, pitch = np.deg2rad(g.uniform(-2, 2, size=2))
            yaw = g.uniform(0, 2 * np.pi)
            lin = g.uniform(-0.1, 0.1, size=3); ang = g.uniform(-1, 1, size=3)
        else:
            pen, roll, pitch, yaw = 2.0e-3, 0.0, 0.0, g.uniform(0, 2 * np.pi)
            lin = np.zeros(3); ang = np.zeros(3)
        Rb = rot_z(yaw) @ rot_y(pitch) @ rot_x(roll)
        tb = np.array([g.uniform(-0.3, 0.3), g.uniform(-0.3, 0.3), r - pen]) if montecarlo else np.array([0, 0, r - pen])
        pose.append(relative_pose(np.eye(3), np.zeros(3), Rb, tb))
        # body twist of the box in world coordinates about the world origin: v_O = v_c - ω x c
        tw_box = np.concatenate([ang, lin - np.cross(ang, tb)])
        twist.append(relative_twist(Rb, tb, np.zeros(6), tw_box))
    n = n_scenes
    name = "C4 256 box-on-plane scenes" if montecarlo else "C2 box on plane"
    return Workload(name, meshes, ins, np.zeros(n, dtype=np.int32), np.array(pose), np.array(twist), np.zeros((n, 6)),
                    {"n_div": n_div})


def c3_blob_tool(n_poses: int = 1, seed: int = 20260103, n_div_blob: int = 22, n_div_tool: int = 16,
                 distance: float = 0.19) -> Workload:
    """C3: compliant blob = sphere mesh n_div 22 (9 680 tets, r = 0.1, Ē = 1e6) against a rigid tool = sphere
    surface n_div 16 (5 120 triangles, r = 0.1), centre distance 0.19 (1 cm overlap), random tool rotation and
    approach direction per pose; bristle friction τ 0.05, k̄ 1e4, μd 0.3, magic 1e-3, χ 0.5, quadrature rule 2,
    s ~ N(0, 1e-4).  n_poses independent Monte-Carlo poses form one batch (one item each)."""
    blob = G.as_tet_emesh(G.emesh_sphere(0.1, n_div_blob))
    tool = G.as_tri_emesh(G.emesh_sphere(0.1, n_div_tool))
    meshes = [_mesh("tool", tool), _mesh("blob", blob, 1.0e6)]
    ins = [InsSpec(0, 1, "bristle", chi=0.5, mu_d=0.3, tau=0.05, k_bar=1.0e4, magic=1.0e-3)]
    pose, twist, s = [], [], []
    for k in range(n_poses):
        g = _rng(seed, k)
        Rt = random_rotation(g)
        u = g.standard_normal(3); u /= np.linalg.norm(u)
        tt = distance * u
        pose.append(relative_pose(Rt, tt, np.eye(3), np.zeros(3)))
        ang = g.uniform(-1, 1, size=3); lin = g.uniform(-0.1, 0.1, size=3)
        tw_tool = np.concatenate([ang, lin - np.cross(ang, tt)])
        twist.append(relative_twist(np.eye(3), np.zeros(3), tw_tool, np.zeros(6)))
        s.append(g.standard_normal(6) * 1.0e-2)          # variance 1e-4
    n = n_poses
    return Workload("C3 10k-tet blob x 5k-tri tool, bristle", meshes, ins, np.zeros(n, dtype=np.int32),
                    np.array(pose), np.array(twist), np.array(s),
                    {"n_tet": blob.n_tet, "n_tri": tool.n_tri, "distance": distance})


import os as _os

SPOON_FIXTURE = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests", "golden", "spoon_quads.npz")


def spoon_emesh(path: Optional[str] = None) -> G.EMesh:
    """The reference's spoon (test/data/spoon.obj: 2 504 vertices, 2 502 quads, a closed manifold; loaded and scaled by 0.01 as
    test/spoon.jl:39-41 does) from the fixture tests/golden/spoon_quads.npz (vertex / face data only,
    tests/golden/make_spoon_fixture.py): 5 004 triangles, 17.7 cm long."""
    d = np.load(path or SPOON_FIXTURE)
    return G.emesh_from_quads(d["point"], d["quad"], 0.01)


def pencil_emesh() -> G.EMesh:
    """The pencil of test/pencil.jl:31-33,198-200: a 12-sided swept mesh along +y, 16 cm long, radius 3.5 mm, with a 13 mm
    cone tip (create_swept_mesh, src/geometry/mesh_create_swept.jl:73-104): 48 surface triangles, 72 tets."""
    m = G.create_swept_mesh(G.f_swept_triv, [0.0, 0.013, 0.16], [0.0, 0.0035, 0.0035], 12, True, rot_half=True)
    return m.transformed(t=np.array([0.0, 0.0, 0.0035]))      # transform!(eM_penci, SVector(0, 0, penci_rad)), pencil.jl:200


def spoon_pencil_pads(n_poses: int = 32, seed: int = 20260105) -> Workload:
    """Real geometry through the path ("lets real reference scenes run"): the reference's spoon surface and its
    swept pencil surface, each against the compliant finger pad of test/pencil.jl:188-190 (sphere n_div 4 of radius 3.5 mm
    stretched by (2, 1, 2): 320 tets, Ebar 1e6), bristle friction mu_d 0.5, chi 0.6 (pencil.jl:214-215).  Per pose the pad sits
    on a random surface triangle of the spoon / pencil, pressed in by U(0.2, 1.2) mm along the triangle's normal, in a random
    orientation, with a random relative twist.  Items alternate spoon, pencil."""
    spoon = spoon_emesh()
    pencil = G.as_tri_emesh(pencil_emesh())
    pad = G.as_tet_emesh(G.emesh_sphere(0.0035, 4)).transformed(R=np.diag([2.0, 1.0, 2.0]))
    meshes = [_mesh("spoon", spoon), _mesh("pencil", pencil), _mesh("pad", pad, 1.0e6)]
    ins = [InsSpec(0, 2, "bristle", chi=0.6, mu_d=0.5), InsSpec(1, 2, "bristle", chi=0.6, mu_d=0.5)]
    ids, pose, twist, s = [], [], [], []
    for k in range(n_poses):
        g = _rng(seed, k)
        which = k % 2
        srf = spoon if which == 0 else pencil
        t = srf.point[srf.tri[int(g.integers(0, srf.n_tri))]]
        cen = t.mean(axis=0)
        nrm = np.cross(t[1] - t[0], t[2] - t[1]); nrm /= np.linalg.norm(nrm)
        Rp = random_rotation(g)
        # the pad's extent along the contact normal in this orientation: support of the ellipsoid (7, 3.5, 7) mm
        ext = np.linalg.norm(np.array([0.007, 0.0035, 0.007]) * (Rp.T @ nrm))
        tp = cen + nrm * (ext - g.uniform(0.2e-3, 1.2e-3))
        ids.append(which)
        pose.append(relative_pose(np.eye(3), np.zeros(3), Rp, tp))       # body 1 = the rigid surface (world), body 2 = the pad
        ang, lin = g.uniform(-1, 1, size=3), g.uniform(-0.05, 0.05, size=3)
        twist.append(relative_twist(Rp, tp, np.zeros(6), np.concatenate([ang, lin - np.cross(ang, tp)])))
        s.append(g.standard_normal(6) * 1.0e-2)
    return Workload("spoon / pencil against finger pads", meshes, ins, np.asarray(ids, dtype=np.int32), np.array(pose),
                    np.array(twist), np.array(s), {"n_tri_spoon": spoon.n_tri, "n_tri_pencil": pencil.n_tri, "n_tet_pad": pad.n_tet})


def c5_pile(n_side: int = 4, seed: int = 20260102, n_divs=(3, 5, 14), overlap: float = 0.02, pencil_spoon: bool = False) -> Workload:
    """C5: n_side^3 compliant boxes (12*n_div^2 tets each, n_div cycling through n_divs) plus their surface
    triangle meshes on a jittered lattice with 2 % overlap; every unordered body pair is a bristle instruction
    (mesh_1 = surface triangles of body i, mesh_2 = tets of body j), 2 016 instructions for 64 bodies.
    pencil_spoon: BASELINE config 5 as worded ("pencil/spoon-scale meshes"): the reference's spoon (rigid surface, 5 004
    triangles) and its swept pencil (compliant: 48 surface triangles, 72 tets) are laid on top of the pile, 2 mm / 1 mm into the
    top layer; they add the pairs spoon-box (x n_body), pencil-box (x n_body: pencil surface against the box's tets) and
    spoon-pencil (spoon surface against the pencil's tets) -- 2 145 instructions for 64 boxes."""
    r = 0.05
    g = _rng(seed, 0)
    protos = {}
    for nd in n_divs:
        vol = G.emesh_box_div(r, nd)
        protos[nd] = (_mesh(f"tri_nd{nd}", G.as_tri_emesh(vol)), _mesh(f"tet_nd{nd}", G.as_tet_emesh(vol), 1.0e6))
    meshes, body = [], []
    pitch = 2 * r * (1 - overlap)
    n_body = n_side ** 3
    for b in range(n_body):
        nd = n_divs[b % len(n_divs)]
        ix, iy, iz = b % n_side, (b // n_side) % n_side, b // (n_side * n_side)
        t = np.array([ix, iy, iz]) * pitch + g.uniform(-0.002, 0.002, size=3)
        R = rot_z(g.uniform(-0.05, 0.05)) @ rot_y(g.uniform(-0.05, 0.05)) @ rot_x(g.uniform(-0.05, 0.05))
        tw = np.concatenate([g.uniform(-1, 1, size=3), g.uniform(-0.1, 0.1, size=3)])
        mt, mv = protos[nd]
        meshes.append(MeshSpec(f"b{b}_tri", mt.mesh, mt.tree, None))
        meshes.append(MeshSpec(f"b{b}_tet", mv.mesh, mv.tree, 1.0e6))
        body.append((R, t, tw))
    ins, pose, twist, s = [], [], [], []
    for i in range(n_body):
        for j in range(i + 1, n_body):
            ins.append(InsSpec(2 * i, 2 * j + 1, "bristle", chi=0.5, mu_d=0.3))
            Ri, ti, twi = body[i]; Rj, tj, twj = body[j]
            pose.append(relative_pose(Ri, ti, Rj, tj))
            twist.append(relative_twist(Rj, tj, twi, twj))
            s.append(g.standard_normal(6) * 1.0e-2)
    if pencil_spoon:
        top = (n_side - 1) * pitch + r
        mid = 0.5 * (n_side - 1) * pitch
        spoon = spoon_emesh()
        pen = pencil_emesh()
        sp_spec = _mesh("spoon_tri", spoon)
        pen_tri, pen_tet = _mesh("pencil_tri", G.as_tri_emesh(pen)), _mesh("pencil_tet", G.as_tet_emesh(pen), 1.0e6)
        i_sp = len(meshes); meshes.append(sp_spec)
        i_pt = len(meshes); meshes.append(pen_tri)
        i_pv = len(meshes); meshes.append(pen_tet)
        Rs, ts = rot_z(0.3), np.array([mid, mid - 0.01, top - 0.002 - float(spoon.point[:, 2].min())])
        tws = np.concatenate([g.uniform(-1, 1, size=3), g.uniform(-0.1, 0.1, size=3)])
        Rq, tq = rot_z(-0.2), np.array([0.03, 0.6 * pitch, top - 0.001])      # the pencil lies on z in [0, 2 r_pencil]
        twq = np.concatenate([g.uniform(-1, 1, size=3), g.uniform(-0.1, 0.1, size=3)])
        for j in range(n_body):
            Rj, tj, twj = body[j]
            ins.append(InsSpec(i_sp, 2 * j + 1, "bristle", chi=0.5, mu_d=0.3))
            pose.append(relative_pose(Rs, ts, Rj, tj)); twist.append(relative_twist(Rj, tj, tws, twj)); s.append(g.standard_normal(6) * 1.0e-2)
        for j in range(n_body):
            Rj, tj, twj = body[j]
            ins.append(InsSpec(i_pt, 2 * j + 1, "bristle", chi=0.5, mu_d=0.3))
            pose.append(relative_pose(Rq, tq, Rj, tj)); twist.append(relative_twist(Rj, tj, twq, twj)); s.append(g.standard_normal(6) * 1.0e-2)
        ins.append(InsSpec(i_sp, i_pv, "bristle", chi=0.5, mu_d=0.3))
        pose.append(relative_pose(Rs, ts, Rq, tq)); twist.append(relative_twist(Rq, tq, tws, twq)); s.append(g.standard_normal(6) * 1.0e-2)
    n = len(ins)
    return Workload(f"C5 pile of {n_body} boxes" + (" + spoon + pencil" if pencil_spoon else ""), meshes, ins,
                    np.arange(n, dtype=np.int32), np.array(pose), np.array(twist), np.array(s), {"n_body": n_body})


def vol_vol(n_poses: int = 4, seed: int = 20260104, n_div: int = 4, model: str = "regularized") -> Workload:
    """Volume-volume (tet-tet) contact, src/contact_algorithms_non_friction.jl:166-194: (a) the test_vol_vol.jl
    geometry, a compliant box (12 tets) on the compliant half-plane tet, and (b) two compliant spheres
    (20 n_div^2 tets each, different Ē) overlapping by 5 % of the radius, random relative pose per item."""
    r = 0.05
    plane = G.as_tet_emesh(G.emesh_half_plane())
    box = G.as_tet_emesh(G.emesh_box(r))
    sph = G.as_tet_emesh(G.emesh_sphere(0.1, n_div))
    meshes = [_mesh("plane", plane, 1.0e6), _mesh("box", box, 1.0e6), _mesh("sphere_a", sph, 1.0e6),
              _mesh("sphere_b", sph, 3.0e6)]
    kw = dict(chi=0.5, mu_d=0.3)
    ins = [InsSpec(1, 0, model, **kw), InsSpec(2, 3, model, **kw)]
    ids, pose, twist, s = [], [], [], []
    for k in range(n_poses):
        g = _rng(seed, k)
        # box on plane: body 1 = box, body 2 = plane (world)
        Rb = rot_z(g.uniform(0, 2 * np.pi)) @ rot_y(np.deg2rad(g.uniform(-2, 2))) @ rot_x(np.deg2rad(g.uniform(-2, 2)))
        tb = np.array([g.uniform(-0.2, 0.2), g.uniform(-0.2, 0.2), r - g.uniform(0.5e-3, 3e-3)])
        ang, lin = g.uniform(-1, 1, size=3), g.uniform(-0.1, 0.1, size=3)
        ids.append(0)
        pose.append(relative_pose(Rb, tb, np.eye(3), np.zeros(3)))
        twist.append(relative_twist(np.eye(3), np.zeros(3), np.concatenate([ang, lin - np.cross(ang, tb)]), np.zeros(6)))
        s.append(g.standard_normal(6) * 1.0e-2)
        # sphere on sphere
        Ra, Rs = random_rotation(g), random_rotation(g)
        u = g.standard_normal(3); u /= np.linalg.norm(u)
        ta = 0.195 * u
        ang, lin = g.uniform(-1, 1, size=3), g.uniform(-0.1, 0.1, size=3)
        ids.append(1)
        pose.append(relative_pose(Ra, ta, Rs, np.zeros(3)))
        twist.append(relative_twist(Rs, np.zeros(3), np.concatenate([ang, lin - np.cross(ang, ta)]), np.zeros(6)))
        s.append(g.standard_normal(6) * 1.0e-2)
    return Workload("vol-vol (tet-tet)", meshes, ins, np.asarray(ids, dtype=np.int32), np.array(pose), np.array(twist),
                    np.array(s))


# ----------------------------------------------------------------------------------------------------------------
def build_scenario(w: Workload, device: int = 0, debug: bool = False, devices=None) -> MechanismScenario:
    """Workload -> finalized MechanismScenario on the HIP device (devices: a list of devices, pfc_create_multi)."""
    m = MechanismScenario(device=device, devices=devices)
    for ms in w.meshes:
        m.add_contact(ms.name, ms.mesh, c_prop=None if ms.Ebar is None else ContactProperties(ms.Ebar), tree=ms.tree)
    for c in w.instructions:
        if c.model == "regularized":
            m.add_friction_regularize(c.id_1, c.id_2, mu_s=c.mu_s, mu_d=c.mu_d, chi=c.chi, v_tol=c.v_tol,
                                      n_quad_rule=c.n_quad_rule)
        else:
            m.add_friction_bristle(c.id_1, c.id_2, tau=c.tau, k_bar=c.k_bar, mu_s=c.mu_s, mu_d=c.mu_d, chi=c.chi,
                                   n_quad_rule=c.n_quad_rule, magic=c.magic)
    m.finalize()
    if debug:
        m.set_option("debug", 1)
    return m
