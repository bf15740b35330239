"""Host-side mirror of the reference's scenario API for the contact hot path.

Same names, argument meaning, defaults and error behaviour as the reference's Julia API (snake_case instead of
the trailing ``!``), restricted to what the hot path needs:

  MechanismScenario()                      src/mechanism_scenario.jl:166-199
  add_contact(name, e_mesh, c_prop=)       :298-314      -> MeshCache (src/structs.jl:33-45)
  add_friction_regularize(id_1, id_2, ..)  :365-377      -> ContactInstructions (:36-49) with Regularized (:22-34)
  add_friction_bristle(id_1, id_2, ..)     :384-399      -> ContactInstructions with Bristle (:5-20)
  finalize()                               :206-231      -> uploads meshes/trees/instructions (pfc_finalize)
  force_all_elastic_intersections(...)     src/contact_algorithms_non_friction.jl:60-84 -> pfc_eval

The rigid-body side of calcXd! (RigidBodyDynamics: poses, twists, Jacobians, mass matrix, third-law scatter)
stays with the host integrator; this class takes the per-instruction relative pose / twist / bristle state that
refreshBodyBodyTransform! / refreshBodyBodyCache! (:103-134) produce and returns the per-instruction wrench, ṡ and
counters.  All compute runs in libpfc_hip (HIP, gfx950); nothing here has a CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Union

import numpy as np

from . import _lib
from .geometry import EMesh, OBBTree, build_tree

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def default_chi() -> float:
    return 0.5          # src/mechanism_scenario.jl:347


def default_mu() -> float:
    return 0.3          # :348


def determine_mu_s_mu_d(mu_s: Optional[float], mu_d: Optional[float]):
    """src/mechanism_scenario.jl:350-356 (the (nothing, nothing) case returns default_χ twice, sic)."""
    if mu_s is None and mu_d is None:
        return default_chi(), default_chi()
    if mu_d is None:
        raise ValueError("need to specify μd")
    if mu_s is None:
        return float(mu_d), float(mu_d)
    if not (mu_d <= mu_s):
        raise ValueError("something is wrong")
    return float(mu_s), float(mu_d)


@dataclass(frozen=True)
class ContactProperties:
    """src/structs.jl:9-15."""
    Ebar: float

    def __post_init__(self):
        if not (1.0e4 <= self.Ebar <= 3.0e11):
            raise ValueError("E_effective in unexpected range.")


@dataclass(frozen=True)
class Regularized:
    """src/mechanism_scenario.jl:22-34."""
    v_c: float
    mu_s: float
    mu_d: float

    @property
    def v_mu_s(self): return 2 * self.v_c
    @property
    def v_mu_d(self): return 3 * self.v_c


@dataclass(frozen=True)
class Bristle:
    """src/mechanism_scenario.jl:5-20."""
    bristle_id: int
    tau: float
    k_bar: float
    mu_s: float
    mu_d: float
    magic: float

    @property
    def Ts_mu_s(self): return 2 * self.mu_s
    @property
    def Ts_mu_d(self): return 3 * self.mu_s


@dataclass
class MeshCache:
    """src/structs.jl:33-45 (BodyID / FrameID stay with the host mechanism)."""
    name: str
    mesh: EMesh
    tree: OBBTree
    c_prop: Optional[ContactProperties]

    @property
    def is_tri(self): return self.mesh.tri is not None
    @property
    def is_tet(self): return self.mesh.tet is not None


@dataclass(frozen=True)
class ContactInstructions:
    """src/mechanism_scenario.jl:36-49."""
    id_1: int
    id_2: int
    chi: float
    friction_model: Union[Regularized, Bristle]
    n_quad_rule: int


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


_addressof, _char_from_buffer = C.addressof, C.c_char.from_buffer


def _addr(a):
    """address of a C-contiguous array's first element: through the buffer protocol (0.35 us) where the array is writable,
    through ndarray.ctypes (0.95 us: it builds a helper object per call) where it is not -- seven of these per evaluation
    were two thirds of the general entry point's overhead over BoundEvaluation"""
    try:
        return _addressof(_char_from_buffer(a))
    except (TypeError, ValueError):      # read-only or empty
        return a.ctypes.data


def _da(a):
    """float64 C-contiguous view (no copy when it already is one) and its address"""
    if not (type(a) is np.ndarray and a.dtype == np.float64 and a.flags.c_contiguous):
        a = np.ascontiguousarray(a, dtype=np.float64)
    return a, _addr(a)


def _ia(a):
    if not (type(a) is np.ndarray and a.dtype == np.int32 and a.flags.c_contiguous):
        a = np.ascontiguousarray(a, dtype=np.int32)
    return a, _addr(a)


class BoundEvaluation:
    """force_all_elastic_intersections on buffers that live as long as this object: the inputs are copied once into
    `pose` (n,24), `twist` (n,6), `s` (n,6 or None) -- change them IN PLACE between calls -- and every call overwrites
    `wrench` (n,6), `sdot` (n,6), `counts` (n,4) and returns them.  One foreign call per evaluation with prebuilt
    arguments: the ~10 us of array checks and allocations of the general entry point (a quarter of a one-box scene's
    40 us) are paid once here."""

    def __init__(self, scenario: "MechanismScenario", pose, twist, s=None, ins_ids=None):
        self._m = scenario
        self.pose = np.array(pose, dtype=np.float64, order="C").reshape(-1, 24)
        n = self.pose.shape[0]
        self.twist = np.array(twist, dtype=np.float64, order="C").reshape(-1, 6)
        if self.twist.shape[0] != n:
            raise ValueError("twist must have 6 entries per item")
        self.s = None
        if s is not None:
            self.s = np.array(s, dtype=np.float64, order="C").reshape(-1, 6)
            if self.s.shape[0] != n:
                raise ValueError("s must have 6 entries per item")
        self.ins_ids = None
        if ins_ids is not None:
            self.ins_ids = np.array(ins_ids, dtype=np.int32, order="C").reshape(-1)
            if self.ins_ids.size != n:
                raise ValueError("ins_ids must have one entry per item")
        self.wrench = np.zeros((n, 6)); self.sdot = np.zeros((n, 6)); self.counts = np.zeros((n, 4), dtype=np.int32)
        self._fn = _lib.lib().pfc_eval_addr
        # (the handle is NOT captured: after MechanismScenario.close() a call must raise, not hand a freed handle to the library)
        self._args = (n, None if self.ins_ids is None else self.ins_ids.ctypes.data, self.pose.ctypes.data,
                      self.twist.ctypes.data, None if self.s is None else self.s.ctypes.data, self.wrench.ctypes.data,
                      self.sdot.ctypes.data, self.counts.ctypes.data)

    def __call__(self):
        h = self._m._h
        if h is None:
            raise RuntimeError("the scenario of this BoundEvaluation has been closed")
        rc = self._fn(h, *self._args)
        if rc != 0:
            self._m._check(rc)
        return self.wrench, self.sdot, self.counts


class MechanismScenario:
    """Contact part of MechanismScenario{T} (src/mechanism_scenario.jl:166-199), backed by a pfc_handle."""

    def __init__(self, device: int = 0, devices: Optional[Sequence[int]] = None):
        """device: the HIP device of the scenario; devices: a list of HIP devices instead (pfc_create_multi: the one host
        process uses them all, items are cut into ranges per device inside the library)."""
        self.MeshCache: List[MeshCache] = []
        self.ContactInstructions: List[ContactInstructions] = []
        self.n_bristle = 0
        self.device = device
        self.devices = None if devices is None else [int(d) for d in devices]
        self._h = None
        self._finalized = False

    # ---- scenario construction ---------------------------------------------------------------------------------
    def add_contact(self, name: str, e_mesh: EMesh, c_prop: Optional[ContactProperties] = None,
                    tree: Optional[OBBTree] = None) -> int:
        """add_contact! (:298-314).  Returns the mesh id (0-based)."""
        if self._finalized:
            raise RuntimeError("add_contact after finalize")
        # verify_eMesh_ContactProperties (:301-306)
        if e_mesh.tri is not None and e_mesh.tet is not None:
            raise ValueError("eMesh has triangles and tets. Use as_tri_eMesh or as_tet_eMesh to convert eMesh.")
        if e_mesh.tri is not None and c_prop is not None:
            raise ValueError("Using ContactProperties for triangular eMesh")
        if e_mesh.tet is not None and c_prop is None:
            raise ValueError("Using nothing as ContactProperties for tet eMesh")
        if tree is None:
            tree = build_tree(e_mesh)           # eMesh_to_tree (:309)
        self.MeshCache.append(MeshCache(name, e_mesh, tree, c_prop))
        return len(self.MeshCache) - 1

    def find_mesh_id(self, name: str) -> int:
        """src/utility.jl:22-33."""
        ids = [k for k, m in enumerate(self.MeshCache) if m.name == name]
        if len(ids) > 1:
            raise KeyError("multiple")
        if not ids:
            raise KeyError(f"no mesh found by name: {name}")
        return ids[0]

    def _add_friction(self, id_1: int, id_2: int, model, chi: float, n_quad_rule: int) -> ContactInstructions:
        """add_friction! (:402-416): id_1 becomes the triangle mesh (or a tet mesh), id_2 is always a tet mesh."""
        m_1, m_2 = self.MeshCache[id_1], self.MeshCache[id_2]
        if m_1.is_tet and m_2.is_tri:
            id_1, id_2 = id_2, id_1
            m_1, m_2 = m_2, m_1
        if not m_2.is_tet:
            raise TypeError("no method matching add_friction!(tri, tri): one mesh must be a tet mesh")
        if not (1 <= n_quad_rule <= 2):
            raise ValueError("only quadrature rules 1 (first order) and 2 (second? order) are currently implemented")
        c = ContactInstructions(id_1, id_2, float(chi), model, int(n_quad_rule))
        self.ContactInstructions.append(c)
        return c

    def add_friction_regularize(self, mesh_id_1: int, mesh_id_2: int, mu_s=None, mu_d=None, chi: float = None,
                                v_tol: float = 0.01, n_quad_rule: int = 2) -> ContactInstructions:
        """add_friction_regularize! (:365-377)."""
        if self._finalized:
            raise RuntimeError("add_friction after finalize")
        chi = default_chi() if chi is None else chi
        mu_s, mu_d = determine_mu_s_mu_d(mu_s, mu_d)
        return self._add_friction(mesh_id_1, mesh_id_2, Regularized(float(v_tol), mu_s, mu_d), chi, n_quad_rule)

    def add_friction_bristle(self, mesh_id_1: int, mesh_id_c: int, tau: float = 0.05, k_bar: float = 1.0e4,
                             mu_s=None, mu_d=None, chi: float = None, n_quad_rule: int = 2,
                             magic: float = 1.0e-3) -> ContactInstructions:
        """add_friction_bristle! (:384-399)."""
        if self._finalized:
            raise RuntimeError("add_friction after finalize")
        chi = default_chi() if chi is None else chi
        mu_s, mu_d = determine_mu_s_mu_d(mu_s, mu_d)
        if not (0 < mu_d):
            raise ValueError("μd cannot be 0 for bristle friction")
        b = Bristle(self.n_bristle, float(tau), float(k_bar), mu_s, mu_d, float(magic))
        c = self._add_friction(mesh_id_1, mesh_id_c, b, chi, n_quad_rule)
        self.n_bristle += 1
        return c

    # ---- device ------------------------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != _lib.OK:
            msg = _lib.lib().pfc_last_error(self._h).decode() if self._h else ""
            raise _lib.PFCError(rc, msg)

    def _id(self, rc: int) -> int:
        if rc < 0:
            self._check(-rc)
        return rc

    def finalize(self):
        """finalize! (:206-231): upload every MeshCache and ContactInstructions, build device tables."""
        if self._finalized:
            raise RuntimeError("finalize called twice")
        L = _lib.lib()
        h = C.c_void_p()
        if self.devices is not None:
            dv = (C.c_int * len(self.devices))(*self.devices)
            rc = L.pfc_create_multi(dv, len(self.devices), C.byref(h))
        else:
            rc = L.pfc_create(self.device, C.byref(h))
        if rc != _lib.OK:
            raise _lib.PFCError(rc, "pfc_create failed: no usable HIP device (there is no CPU fallback)")
        self._h = h
        for m in self.MeshCache:
            keep = []
            p, pp = _d(m.mesh.point); keep.append(p)
            tri_p = tet_p = eps_p = None
            n_tri = n_tet = 0
            if m.is_tri:
                a, tri_p = _i(m.mesh.tri); keep.append(a); n_tri = m.mesh.n_tri
            else:
                a, tet_p = _i(m.mesh.tet); keep.append(a); n_tet = m.mesh.n_tet
                a, eps_p = _d(m.mesh.eps); keep.append(a)
            t = m.tree
            c, cp = _d(t.c); e, ep = _d(t.e); R, Rp = _d(t.R); ch, chp = _i(t.child); lf, lfp = _i(t.leaf)
            self._id(L.pfc_add_mesh(h, m.mesh.n_point, pp, n_tri, tri_p, n_tet, tet_p, eps_p,
                                    m.c_prop.Ebar if m.c_prop else 0.0, t.n_node, cp, ep, Rp, chp, lfp))
        for c in self.ContactInstructions:
            f = c.friction_model
            if isinstance(f, Regularized):
                par, model = [f.mu_s, f.mu_d, f.v_c, 0, 0, 0, 0, 0], _lib.REGULARIZED
            else:
                par, model = [f.mu_s, f.mu_d, f.tau, f.k_bar, f.magic, 0, 0, 0], _lib.BRISTLE
            a, ap = _d(par)
            self._id(L.pfc_add_instruction(h, c.id_1, c.id_2, c.chi, c.n_quad_rule, model, ap))
        self._check(L.pfc_finalize(h))
        self._finalized = True

    def close(self):
        if self._h is not None:
            _lib.lib().pfc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int):
        self._check(_lib.lib().pfc_set_option(self._h, name.encode(), int(value)))

    # ---- evaluation --------------------------------------------------------------------------------------------
    def force_all_elastic_intersections(self, pose, twist, s=None, ins_ids: Optional[Sequence[int]] = None):
        """forceAllElasticIntersections! without the RigidBodyDynamics parts (host buffers, synchronous).

        pose (n,24), twist (n,6), s (n,6) or None, ins_ids (n,) or None (item i = instruction i).
        Returns (wrench (n,6), sdot (n,6), counts (n,4))."""
        if not self._finalized:
            raise RuntimeError("finalize the scenario first")
        pose_a, pose_p = _da(pose)
        n = pose_a.size // 24
        tw_a, tw_p = _da(twist)
        if tw_a.size != 6 * n:
            raise ValueError("twist must have 6 entries per item")
        s_p = None
        if s is not None:
            s_a, s_p = _da(s)
            if s_a.size != 6 * n:
                raise ValueError("s must have 6 entries per item")
        id_p = None
        if ins_ids is not None:
            id_a, id_p = _ia(ins_ids)
            if id_a.size != n:
                raise ValueError("ins_ids must have one entry per item")
        wrench = np.zeros((n, 6)); sdot = np.zeros((n, 6)); counts = np.zeros((n, 4), dtype=np.int32)
        rc = _lib.lib().pfc_eval_addr(self._h, n, id_p, pose_p, tw_p, s_p, _addr(wrench), _addr(sdot), _addr(counts))
        if rc != 0:
            self._check(rc)
        return wrench, sdot, counts

    def bind(self, pose, twist, s=None, ins_ids: Optional[Sequence[int]] = None) -> "BoundEvaluation":
        """Persistent buffers for a scene that is evaluated again and again (what TypedElasticBodyBodyCache's preallocated
        arrays are to calcXd!, src/mechanism_scenario.jl:78-97): see BoundEvaluation."""
        if not self._finalized:
            raise RuntimeError("finalize the scenario first")
        return BoundEvaluation(self, pose, twist, s, ins_ids)

    def force_all_elastic_intersections_dual(self, pose, twist, s, d_pose, d_twist, d_s=None,
                                             ins_ids: Optional[Sequence[int]] = None, bp_pose=None):
        """The evaluation on Dual numbers (MechanismScenario.dual, src/mechanism_scenario.jl:187): values plus, for
        each of n_dir seed directions, the partials.  d_pose (n, n_dir, 24), d_twist (n, n_dir, 6), d_s (n, n_dir, 6)
        or None.  bp_pose (n, 24) or None: the pose the broadphase culls with -- the reference takes m.float's
        (calcTriTetIntersections!, src/contact_algorithms_non_friction.jl:94-101), i.e. the pose of the last Float64
        evaluation; None = pose.  Returns (wrench, sdot, d_wrench (n, n_dir, 6), d_sdot (n, n_dir, 6), counts)."""
        if not self._finalized:
            raise RuntimeError("finalize the scenario first")
        pose_a, pose_p = _da(pose)
        n = pose_a.size // 24
        tw_a, tw_p = _da(twist)
        dp_a, dp_p = _da(d_pose)
        if n == 0 or dp_a.size % (24 * n) != 0:
            raise ValueError("d_pose must be (n, n_dir, 24)")
        n_dir = dp_a.size // (24 * n)
        dt_a, dt_p = _da(d_twist)
        if tw_a.size != 6 * n or dt_a.size != 6 * n * n_dir:
            raise ValueError("twist must be (n, 6) and d_twist (n, n_dir, 6)")
        s_p = ds_p = id_p = None
        if s is not None:
            s_a, s_p = _da(s)
            if s_a.size != 6 * n:
                raise ValueError("s must have 6 entries per item")
        if d_s is not None:
            ds_a, ds_p = _da(d_s)
            if ds_a.size != 6 * n * n_dir:
                raise ValueError("d_s must be (n, n_dir, 6)")
        if ins_ids is not None:
            id_a, id_p = _ia(ins_ids)
            if id_a.size != n:
                raise ValueError("ins_ids must have one entry per item")
        wrench = np.zeros((n, 6)); sdot = np.zeros((n, 6)); counts = np.zeros((n, 4), dtype=np.int32)
        dw = np.zeros((n, n_dir, 6)); dsd = np.zeros((n, n_dir, 6))
        if bp_pose is not None:
            bp_a, bp_p = _da(bp_pose)
            if bp_a.size != 24 * n:
                raise ValueError("bp_pose must have 24 entries per item")
            rc = _lib.lib().pfc_eval_dual_bp_addr(self._h, n, n_dir, id_p, pose_p, bp_p, tw_p, s_p, dp_p, dt_p, ds_p, _addr(wrench),
                                                  _addr(sdot), _addr(dw), _addr(dsd), _addr(counts))
        else:
            rc = _lib.lib().pfc_eval_dual_addr(self._h, n, n_dir, id_p, pose_p, tw_p, s_p, dp_p, dt_p, ds_p, _addr(wrench),
                                               _addr(sdot), _addr(dw), _addr(dsd), _addr(counts))
        if rc != 0:
            self._check(rc)
        return wrench, sdot, dw, dsd, counts

    def scatter_generalized(self, wrench, x_w_r2, body_1, body_2, jac, scene=None, n_scene: int = 1):
        """addGeneralizedForcesThirdLaw! for all items (non_friction.jl:267-286) on the device.
        wrench (n,6); x_w_r2 (n,12) = R col-major + t; body_1/body_2 (n,) body ids (-1: no Jacobian);
        jac (n_body, nv, 6): per body and velocity coordinate [angular 3; linear 3]; returns f (n_scene, nv)."""
        w_a, w_p = _d(wrench); x_a, x_p = _d(x_w_r2)
        b1_a, b1_p = _i(body_1); b2_a, b2_p = _i(body_2)
        j_a, j_p = _d(jac)
        n = w_a.size // 6
        n_body, nv = (int(np.shape(jac)[0]), int(np.shape(jac)[1])) if np.ndim(jac) == 3 else (0, 0)
        sc_p = None
        if scene is not None:
            sc_a, sc_p = _i(scene)
        f = np.zeros((n_scene, nv))
        self._check(_lib.lib().pfc_scatter_generalized(self._h, n, w_p, x_p, b1_p, b2_p, sc_p, n_scene, n_body, nv, j_p,
                                                       f.ctypes.data_as(_dp)))
        return f

    def scatter_generalized_device(self, n_items: int, d_wrench: int, d_x_w_r2: int, d_body_1: int, d_body_2: int, d_scene: int,
                                   n_scene: int, nv: int, d_jac: int, d_f: int, accumulate: bool = False, stream: int = 0):
        """pfc_scatter_generalized_device: raw device addresses, asynchronous on `stream`; f_generalized stays in HBM."""
        self._check(_lib.lib().pfc_scatter_generalized_device(self._h, int(n_items), d_wrench, d_x_w_r2, d_body_1, d_body_2,
                                                              d_scene or None, int(n_scene), int(nv), d_jac, d_f,
                                                              1 if accumulate else 0, stream or None))

    def eval_device(self, n_items: int, d_ins_ids: int, d_pose: int, d_twist: int, d_s: int, d_wrench: int,
                    d_sdot: int, d_counts: int, stream: int = 0):
        """pfc_eval_device: raw device addresses (e.g. torch.Tensor.data_ptr()); asynchronous."""
        self._check(_lib.lib().pfc_eval_device(self._h, int(n_items), d_ins_ids or None, d_pose, d_twist,
                                               d_s or None, d_wrench, d_sdot, d_counts or None, stream or None))

    def eval_dual_device(self, n_items: int, n_dir: int, d_ins_ids: int, d_pose: int, d_twist: int, d_s: int, d_dpose: int,
                         d_dtwist: int, d_ds: int, d_wrench: int, d_sdot: int, d_dwrench: int, d_dsdot: int, d_counts: int,
                         stream: int = 0, d_bp_pose: int = 0):
        """pfc_eval_dual_device[_bp]: raw device addresses; asynchronous; follow with check() (re-issue on ERR_OVERFLOW)."""
        self._check(_lib.lib().pfc_eval_dual_device_bp(self._h, int(n_items), int(n_dir), d_ins_ids or None, d_pose, d_bp_pose or None,
                                                       d_twist, d_s or None, d_dpose, d_dtwist, d_ds or None, d_wrench, d_sdot,
                                                       d_dwrench, d_dsdot, d_counts or None, stream or None))

    def eval_dual_device_more(self, n_dir: int, d_dpose: int, d_dtwist: int, d_ds: int, d_dwrench: int, d_dsdot: int,
                              stream: int = 0):
        """pfc_eval_dual_device_more: further seed directions at the point of the previous eval_dual_device evaluation
        (the chunks of one Jacobian); only the Dual passes run.  Follow with check()."""
        self._check(_lib.lib().pfc_eval_dual_device_more(self._h, int(n_dir), d_dpose, d_dtwist, d_ds or None, d_dwrench,
                                                         d_dsdot, stream or None))

    def check(self) -> int:
        """pfc_check: synchronise; returns the status (PFC_ERR_OVERFLOW means: re-issue, buffers were grown)."""
        rc = _lib.lib().pfc_check(self._h)
        if rc not in (_lib.OK, _lib.ERR_OVERFLOW):
            self._check(rc)
        return rc

    def stats(self) -> dict:
        out = (C.c_longlong * 8)()
        self._check(_lib.lib().pfc_get_stats(self._h, out))
        k = ("node_tests", "candidates", "nonempty", "tractions", "levels", "frontier_peak", "status", "n_items")
        return dict(zip(k, [int(v) for v in out]))

    def last_parts(self) -> int:
        """1, or 2 if the last checked evaluation ran as two concurrent halves (option split_min); 0 if it ran as the
        single fused small-scene kernel (option fused)."""
        return int(_lib.lib().pfc_last_parts(self._h))

    def last_shards(self) -> int:
        """Devices that took part in the last evaluation (1 for a single-device scenario)."""
        return int(_lib.lib().pfc_last_shards(self._h))

    def last_team(self) -> int:
        """Workgroups per item of the last checked evaluation if it ran as one fused kernel (> 1: a team per item), else 0."""
        return int(_lib.lib().pfc_last_team(self._h))

    def last_dual_reused(self) -> bool:
        """True if the last Dual evaluation ran only its Dual passes on the previous one's value pass."""
        return bool(_lib.lib().pfc_last_dual_reused(self._h))

    def stage_ms(self) -> dict:
        out = (C.c_float * 6)()
        self._check(_lib.lib().pfc_get_stage_ms(self._h, out))
        k = ("setup", "broadphase", "narrowphase", "bristle", "final", "total")
        return dict(zip(k, [float(v) for v in out]))

    # ---- debug views (m.float.bodyBodyCache of the reference's tests) -------------------------------------------
    def debug_pairs(self, item: int):
        L = _lib.lib()
        n = self._id(L.pfc_debug_pairs(self._h, item, None, None, 0))
        pairs = np.zeros((max(n, 1), 2), dtype=np.int32); clip_n = np.zeros(max(n, 1), dtype=np.int32)
        self._id(L.pfc_debug_pairs(self._h, item, pairs.ctypes.data_as(_ip), clip_n.ctypes.data_as(_ip), n))
        return pairs[:n], clip_n[:n]

    def debug_tractions(self, item: int) -> np.ndarray:
        L = _lib.lib()
        n = self._id(L.pfc_debug_tractions(self._h, item, None, 0))
        buf = np.zeros((max(n, 1), 8))
        self._id(L.pfc_debug_tractions(self._h, item, buf.ctypes.data_as(_dp), n))
        return buf[:n]

    def debug_stiffness(self, item: int):
        K = np.zeros(36); Kis = np.zeros(36); Sinv = np.zeros(6); cop = np.zeros(3)
        n = self._id(_lib.lib().pfc_debug_stiffness(self._h, item, K.ctypes.data_as(_dp), Kis.ctypes.data_as(_dp),
                                                    Sinv.ctypes.data_as(_dp), cop.ctypes.data_as(_dp)))
        if n == 0:
            return None
        return K.reshape(6, 6, order="F"), Kis.reshape(6, 6, order="F"), Sinv, cop


# ---- host helpers standing in for the RigidBodyDynamics calls of refreshBodyBodyTransform!/Cache! -----------------
def relative_pose(R_w1, t_w1, R_w2, t_w2) -> np.ndarray:
    """pose[24] for bodies with world poses x_rw_r1 = (R_w1, t_w1), x_rw_r2 = (R_w2, t_w2):
    x_r2_rw = inv(x_rw_r2); x_r2_r1 = x_r2_rw * x_rw_r1; x_r1_r2 = inv(x_r2_r1)
    (src/contact_algorithms_non_friction.jl:109-113; inv(Transform3D) = (R', -R' t))."""
    R_w1 = np.asarray(R_w1, dtype=np.float64).reshape(3, 3); t_w1 = np.asarray(t_w1, dtype=np.float64).reshape(3)
    R_w2 = np.asarray(R_w2, dtype=np.float64).reshape(3, 3); t_w2 = np.asarray(t_w2, dtype=np.float64).reshape(3)
    R_2w = R_w2.T
    t_2w = -(R_2w @ t_w2)
    R21 = R_2w @ R_w1
    t21 = R_2w @ t_w1 + t_2w
    R12 = R21.T
    t12 = -(R12 @ t21)
    return np.concatenate([R21.reshape(-1, order="F"), t21, R12.reshape(-1, order="F"), t12])


def relative_twist(R_w2, t_w2, twist_w1, twist_w2) -> np.ndarray:
    """twist_r2_r1_r2 = transform(-twist_w_r1 + twist_w_r2, x_r2_rw) (:125-128); twists are [angular; linear]
    expressed in world about the world origin (RigidBodyDynamics convention)."""
    R_w2 = np.asarray(R_w2, dtype=np.float64).reshape(3, 3); t_w2 = np.asarray(t_w2, dtype=np.float64).reshape(3)
    tw = np.asarray(twist_w2, dtype=np.float64) - np.asarray(twist_w1, dtype=np.float64)
    R = R_w2.T
    t = -(R @ t_w2)
    ang = R @ tw[:3]
    lin = R @ tw[3:] + np.cross(t, ang)
    return np.concatenate([ang, lin])
