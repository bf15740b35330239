"""ctypes binding of oracle/libpfc_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module (oracle/pfc_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpfc_oracle.so")
REGULARIZED, BRISTLE = 0, 1
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class _Mesh(C.Structure):
    _fields_ = [("n_pt", C.c_int), ("n_tri", C.c_int), ("n_tet", C.c_int), ("n_node", C.c_int),
                ("pt", _dp), ("tri", _ip), ("tet", _ip), ("eps", _dp), ("Ebar", C.c_double),
                ("node_c", _dp), ("node_e", _dp), ("node_R", _dp), ("node_child", _ip), ("node_leaf", _ip)]


class _Ins(C.Structure):
    _fields_ = [("chi", C.c_double), ("n_quad", C.c_int), ("model", C.c_int), ("mu_s", C.c_double),
                ("mu_d", C.c_double), ("v_c", C.c_double), ("tau", C.c_double), ("k_bar", C.c_double),
                ("magic", C.c_double)]


class _Trac(C.Structure):
    _fields_ = [("n", C.c_double * 3), ("r", C.c_double * 3), ("dA", C.c_double), ("p", C.c_double)]


class _Debug(C.Structure):
    _fields_ = [("n_pair", C.c_int), ("cap_pair", C.c_int), ("pair", _ip), ("clip_n", _ip),
                ("n_trac", C.c_int), ("cap_trac", C.c_int), ("trac", C.POINTER(_Trac)),
                ("n_node_tests", C.c_longlong), ("has_K", C.c_int),
                ("K", C.c_double * 36), ("Kbar_inv_sqrt", C.c_double * 36), ("Sinv", C.c_double * 6),
                ("cop", C.c_double * 3), ("wrench_normal", C.c_double * 6), ("wrench_fric_cop", C.c_double * 6),
                ("Delta", C.c_double * 6)]


def build(force: bool = False) -> str:
    src = [os.path.join(HERE, f) for f in ("pfc_oracle.c", "pfc_oracle_dual.cpp", "pfc_oracle.h", "Makefile")]
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src)
    if force or stale:
        subprocess.run(["make", "-C", HERE, "-B" if force else "-s", "libpfc_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.pfo_debug_new.restype = C.POINTER(_Debug)
        L.pfo_debug_free.argtypes = [C.POINTER(_Debug)]
        L.pfo_eval.restype = C.c_int
        L.pfo_eval.argtypes = [C.POINTER(_Mesh), C.POINTER(_Mesh), C.POINTER(_Ins), _dp, _dp, _dp, _dp, _dp, _ip,
                               C.POINTER(_Debug)]
        L.pfo_eval_bp.restype = C.c_int
        L.pfo_eval_bp.argtypes = [C.POINTER(_Mesh), C.POINTER(_Mesh), C.POINTER(_Ins), _dp, _dp, _dp, _dp, _dp, _dp, _ip, C.POINTER(_Debug)]
        L.pfo_eval_dual_bp.restype = C.c_int
        L.pfo_eval_dual_bp.argtypes = [C.POINTER(_Mesh), C.POINTER(_Mesh), C.POINTER(_Ins), _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp,
                                       _dp, _dp, _dp, _dp]
        L.pfo_eval_batch.restype = C.c_int
        L.pfo_eval_batch.argtypes = [C.c_int, C.POINTER(_Mesh), C.POINTER(_Ins), _ip, _ip, _ip, _dp, _dp, _dp, _dp, _dp,
                                     _ip, C.c_int]
        L.pfo_max_threads.restype = C.c_int
        L.pfo_eval_dual.restype = C.c_int
        L.pfo_eval_dual.argtypes = [C.POINTER(_Mesh), C.POINTER(_Mesh), C.POINTER(_Ins), _dp, _dp, _dp, C.c_int, _dp, _dp, _dp,
                                    _dp, _dp, _dp, _dp]
        L.pfo_scatter_generalized.argtypes = [C.c_int, _dp, _dp, _ip, _ip, _ip, C.c_int, _dp, _dp]
        L.pfo_calc_clamped_piecewise.restype = C.c_double
        L.pfo_calc_clamped_piecewise.argtypes = [C.c_double] * 5
        L.pfo_traction_regularized.argtypes = [C.c_double, C.c_double, C.c_double, _dp, C.c_double, _dp]
        L.pfo_traction_bristle.argtypes = [C.c_double, C.c_double, _dp, C.c_double, _dp]
        L.pfo_weight_poly.argtypes = [C.c_int, _dp, _dp, C.c_double, C.c_double, _dp]
        L.pfo_a_dot_one_pad_b.restype = C.c_double
        L.pfo_a_dot_one_pad_b.argtypes = [_dp, _dp]
        L.pfo_vec_sub_vec_proj.argtypes = [_dp, _dp, _dp]
        L.pfo_volume.restype = C.c_double
        L.pfo_volume.argtypes = [_dp]
        L.pfo_triangle_area.restype = C.c_double
        L.pfo_triangle_area.argtypes = [_dp, _dp]
        L.pfo_triangle_normal.argtypes = [_dp, _dp]
        L.pfo_clip_in_tet_coordinates.restype = C.c_int
        L.pfo_clip_in_tet_coordinates.argtypes = [C.c_int, _dp, _dp]
        L.pfo_clip_plane_tet.restype = C.c_int
        L.pfo_clip_plane_tet.argtypes = [_dp, _dp, _dp]
        L.pfo_zero_small_coordinates.argtypes = [C.c_int, _dp]
        L.pfo_poly_centroid.restype = C.c_double
        L.pfo_poly_centroid.argtypes = [C.c_int, _dp, _dp, _dp]
        L.pfo_inv4.restype = C.c_int
        L.pfo_inv4.argtypes = [_dp, _dp]
        L.pfo_set_inv4_variant.restype = C.c_int
        L.pfo_set_inv4_variant.argtypes = [C.c_int]
        L.pfo_set_ftz.restype = C.c_int
        L.pfo_set_ftz.argtypes = [C.c_int]
        L.pfo_bb_bb_intersect.restype = C.c_int
        L.pfo_bb_bb_intersect.argtypes = [_dp] * 8
        L.pfo_decompose_K.argtypes = [_dp, C.c_double, _dp, _dp]
        L.pfo_tri_quad_rule.restype = C.c_int
        L.pfo_tri_quad_rule.argtypes = [C.c_int, _dp, _dp]
        _lib = L
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


class OracleMesh:
    """Holds a pfo_mesh view of (EMesh, OBBTree, Ebar); keeps the numpy buffers alive."""

    def __init__(self, mesh, tree, Ebar: float = 0.0):
        self._keep = []
        m = _Mesh()
        m.n_pt = mesh.n_point
        p, m.pt = _d(mesh.point); self._keep.append(p)
        if mesh.tri is not None:
            t, m.tri = _i(mesh.tri); self._keep.append(t); m.n_tri = mesh.n_tri
        if mesh.tet is not None:
            t, m.tet = _i(mesh.tet); self._keep.append(t); m.n_tet = mesh.n_tet
            e, m.eps = _d(mesh.eps); self._keep.append(e)
        m.Ebar = float(Ebar)
        m.n_node = tree.n_node
        for name, arr, conv in (("node_c", tree.c, _d), ("node_e", tree.e, _d), ("node_R", tree.R, _d),
                                ("node_child", tree.child, _i), ("node_leaf", tree.leaf, _i)):
            a, ptr = conv(arr); self._keep.append(a); setattr(m, name, ptr)
        self.c = m


def make_ins(chi, n_quad, model, mu_s, mu_d, v_c=0.0, tau=0.0, k_bar=0.0, magic=0.0):
    return _Ins(float(chi), int(n_quad), int(model), float(mu_s), float(mu_d), float(v_c), float(tau),
                float(k_bar), float(magic))


class EvalResult:
    __slots__ = ("status", "wrench", "sdot", "counts", "pairs", "clip_n", "trac", "K", "Kbar_inv_sqrt", "Sinv",
                 "cop", "wrench_normal", "wrench_fric_cop", "Delta", "has_K")


def evaluate(m1: OracleMesh, m2: OracleMesh, ins: _Ins, pose, twist, s=None, debug: bool = True, bp_pose=None) -> EvalResult:
    """One force_single_elastic_intersection!.  pose: 24 doubles (see pfc_oracle.h).  bp_pose: the pose the tree descent
    culls with (m.float's, non_friction.jl:94-101), default: pose."""
    L = lib()
    pose_a, pose_p = _d(pose)
    bp_a, bp_p = _d(pose if bp_pose is None else bp_pose)
    tw_a, tw_p = _d(twist)
    s_a, s_p = _d(np.zeros(6) if s is None else s)
    wrench = np.zeros(6); sdot = np.zeros(6); counts = np.zeros(4, dtype=np.int32)
    dbg = L.pfo_debug_new() if debug else None
    try:
        st = L.pfo_eval_bp(C.byref(m1.c), C.byref(m2.c), C.byref(ins), pose_p, bp_p, tw_p, s_p,
                           wrench.ctypes.data_as(_dp), sdot.ctypes.data_as(_dp), counts.ctypes.data_as(_ip), dbg)
        r = EvalResult()
        r.status, r.wrench, r.sdot, r.counts = st, wrench, sdot, counts
        r.pairs = r.clip_n = r.trac = None
        r.has_K = False
        if debug:
            d = dbg.contents
            n = d.n_pair
            r.pairs = np.ctypeslib.as_array(d.pair, shape=(max(n, 1) * 2,))[:2 * n].reshape(n, 2).copy()
            r.clip_n = np.ctypeslib.as_array(d.clip_n, shape=(max(n, 1),))[:n].copy()
            nt = d.n_trac
            if nt:
                raw = np.ctypeslib.as_array(C.cast(d.trac, _dp), shape=(nt * 8,)).reshape(nt, 8).copy()
            else:
                raw = np.zeros((0, 8))
            r.trac = raw                                   # columns: n(3) r(3) dA p
            r.has_K = bool(d.has_K)
            r.K = np.array(d.K).reshape(6, 6, order="F")
            r.Kbar_inv_sqrt = np.array(d.Kbar_inv_sqrt).reshape(6, 6, order="F")
            r.Sinv = np.array(d.Sinv); r.cop = np.array(d.cop)
            r.wrench_normal = np.array(d.wrench_normal); r.wrench_fric_cop = np.array(d.wrench_fric_cop)
            r.Delta = np.array(d.Delta)
        return r
    finally:
        if debug:
            L.pfo_debug_free(dbg)


def make_pose(R21, t21) -> np.ndarray:
    """pose[24] from x_r2_r1 = (R21, t21); x_r1_r2 = inv (R', -R' t) as RigidBodyDynamics' inv(Transform3D)."""
    R21 = np.asarray(R21, dtype=np.float64).reshape(3, 3)
    t21 = np.asarray(t21, dtype=np.float64).reshape(3)
    R12 = R21.T
    t12 = -(R12 @ t21)
    return np.concatenate([R21.reshape(-1, order="F"), t21, R12.reshape(-1, order="F"), t12])


def evaluate_batch(meshes, ins_list, ins_m1, ins_m2, ins_ids, pose, twist, s, n_threads: int = 1):
    """pfo_eval_batch: all items of a workload, serially (n_threads = 1) or over OpenMP threads.
    meshes: list of OracleMesh; ins_list: list of _Ins; ins_m1 / ins_m2: mesh ids per instruction."""
    L = lib()
    n = int(np.asarray(ins_ids).shape[0])
    marr = (_Mesh * len(meshes))(*[m.c for m in meshes])
    iarr = (_Ins * len(ins_list))(*ins_list)
    m1_a, m1_p = _i(ins_m1); m2_a, m2_p = _i(ins_m2); id_a, id_p = _i(ins_ids)
    po_a, po_p = _d(pose); tw_a, tw_p = _d(twist); s_a, s_p = _d(s)
    wrench = np.zeros((n, 6)); sdot = np.zeros((n, 6)); counts = np.zeros((n, 4), dtype=np.int32)
    st = L.pfo_eval_batch(n, marr, iarr, m1_p, m2_p, id_p, po_p, tw_p, s_p, wrench.ctypes.data_as(_dp),
                          sdot.ctypes.data_as(_dp), counts.ctypes.data_as(_ip), int(n_threads))
    return st, wrench, sdot, counts


def scatter_generalized(wrench, x_w_r2, body_1, body_2, jac, scene=None, n_scene: int = 1):
    """pfo_scatter_generalized: addGeneralizedForcesThirdLaw! over all items; jac (n_body, nv, 6)."""
    L = lib()
    w_a, w_p = _d(wrench); x_a, x_p = _d(x_w_r2); b1_a, b1_p = _i(body_1); b2_a, b2_p = _i(body_2); j_a, j_p = _d(jac)
    nv = int(np.shape(jac)[1])
    sc_p = None
    if scene is not None:
        sc_a, sc_p = _i(scene)
    f = np.zeros((n_scene, nv))
    L.pfo_scatter_generalized(w_a.size // 6, w_p, x_p, b1_p, b2_p, sc_p, nv, j_p, f.ctypes.data_as(_dp))
    return f


def evaluate_dual(m1: OracleMesh, m2: OracleMesh, ins: _Ins, pose, twist, s, d_pose, d_twist, d_s, bp_pose=None):
    """pfo_eval_dual[_bp]: one force_single_elastic_intersection! on Duals.  d_pose (n_dir, 24), d_twist (n_dir, 6),
    d_s (n_dir, 6); bp_pose: the pose the pair list is taken at (m.float's), default pose.
    Returns (status, wrench, sdot, d_wrench (n_dir, 6), d_sdot (n_dir, 6))."""
    L = lib()
    bp_a, bp_p = _d(pose if bp_pose is None else bp_pose)
    pose_a, pose_p = _d(pose); tw_a, tw_p = _d(twist); s_a, s_p = _d(np.zeros(6) if s is None else s)
    dp_a, dp_p = _d(d_pose); dt_a, dt_p = _d(d_twist)
    n_dir = dp_a.size // 24
    ds_a, ds_p = _d(np.zeros((n_dir, 6)) if d_s is None else d_s)
    wrench = np.zeros(6); sdot = np.zeros(6); dw = np.zeros((n_dir, 6)); dsd = np.zeros((n_dir, 6))
    st = L.pfo_eval_dual_bp(C.byref(m1.c), C.byref(m2.c), C.byref(ins), pose_p, bp_p, tw_p, s_p, n_dir, dp_p, dt_p, ds_p,
                         wrench.ctypes.data_as(_dp), sdot.ctypes.data_as(_dp), dw.ctypes.data_as(_dp),
                         dsd.ctypes.data_as(_dp))
    return st, wrench, sdot, dw, dsd
