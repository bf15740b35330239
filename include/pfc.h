/*
 * pfc.h — C ABI of libpfc_hip: the MI355X (gfx950) implementation of PressureFieldContact.jl's per-contact-pair
 * hot path (OBB-BVH culling -> tet/triangle clipping -> pressure + friction wrench integration).
 *
 * The reference (pure Julia) has no FFI for this path; its only substitution hook is the `de::Function` field of
 * MechanismScenario (src/mechanism_scenario.jl:175,181, invoked at src/radau/radau_functions.jl:9,67).  A
 * replacement `calcXd_hip!` is calcXd! (src/contact_algorithms_non_friction.jl:18-38) with
 * `forceAllElasticIntersections!` (:60-68) replaced by ONE call of pfc_eval() for all contact instructions; the
 * Julia `ccall` stubs are in INTEGRATION.md.  Each entry point below names the reference code it replaces
 * (paths relative to the reference repository root).
 *
 * Conventions
 *   - All indices crossing the ABI are 0-based (the Julia shim subtracts 1).
 *   - All matrices are column-major (Julia / StaticArrays order).
 *   - Host-pointer entry points copy in/out; the caller owns its buffers.  The library owns all device memory.
 *   - Every function returns a pfc_status (0 = ok, >0 = error) unless documented to return an id or a count
 *     (>= 0) in which case errors are returned as -(pfc_status).
 *   - A handle is not re-entrant (the reference scenario is not either: shared m.TT_Cache / tm.bodyBodyCache,
 *     src/contact_algorithms_non_friction.jl:95,120); different handles may be used from different threads.
 *   - The library never falls back to a CPU path: without a usable HIP device pfc_create() fails.
 */
#ifndef PFC_H
#define PFC_H

#ifdef __cplusplus
extern "C" {
#endif

#define PFC_VERSION 100
#define PFC_INTERNAL_NODE (-9999) /* leaf sentinel of internal nodes: src/obb/tree_types.jl:11,56 */

typedef enum {
    PFC_OK = 0,
    PFC_ERR_NONFINITE = 1, /* error("Non-finite vertex likely"): src/clip/static_clip.jl:52 ; singular tet */
    PFC_ERR_OVERFLOW = 2,  /* a device work list overflowed; capacities were grown, re-issue the evaluation */
    PFC_ERR_BAD_ARG = 3,   /* bad id / size / NULL pointer; "something is wrong": src/clip/static_clip.jl:13 */
    PFC_ERR_NOMEM = 4,
    PFC_ERR_HIP = 5,       /* HIP runtime failure, see pfc_last_error() */
    PFC_ERR_STATE = 6,     /* call order violated (e.g. add_mesh after finalize) */
    PFC_ERR_INVERTED_TET = 7 /* error("inverted tetrahedron"): src/geometry/mesh.jl:28 */
} pfc_status;

typedef enum { PFC_REGULARIZED = 0, PFC_BRISTLE = 1 } pfc_friction_model;

typedef struct pfc_context *pfc_handle;

/* Library/ABI version (PFC_VERSION of the build). */
int pfc_version(void);
/* 0 for the product build.  Diagnostic builds report themselves: bit 0 = in-kernel phase stamps (-DPFC_STAMPS, graph
 * replay off), bits 8..15 = elimination variant (-DPFC_EXP=n: one phase of the narrowphase compiled out, WRONG results),
 * bit 16 = an A/B variant built from patched sources (scripts/mkvar.sh, -DPFC_VARIANT).
 * The Python binding refuses to load a non-product build unless PFC_ALLOW_DIAGNOSTIC=1 is set. */
int pfc_build_info(void);

/* MechanismScenario() (src/mechanism_scenario.jl:181-198): creates an empty scenario bound to HIP device
 * `device`.  Fails with PFC_ERR_HIP when no device is usable. */
int pfc_create(int device, pfc_handle *out);
/*
 * The same scenario over SEVERAL devices of the node for the one host process the reference is (its calcXd! loops over the
 * contact instructions in a single Julia process, src/contact_algorithms_non_friction.jl:60-68; SURVEY section 8(b):
 * "pfc_create(device_mask)").  devices[0..n_devices) are HIP device ordinals (an ordinal may appear more than once: several
 * shard contexts on one device).  Every other entry point takes the handle unchanged:
 *   - pfc_add_mesh / pfc_add_instruction / pfc_finalize replicate meshes, trees and instructions on every device;
 *   - an evaluation cuts its items into contiguous ranges, one per device, balanced by cost (node tests + candidates of the
 *     previous evaluation of the same item list; the leaf-count product of the two meshes the first time), SURVEY 8(e).  Items
 *     are independent, there is no collective:
 *       host-pointer entry points (pfc_eval, pfc_eval_dual, pfc_eval_dual_bp): one library thread per device evaluates its
 *       range straight from / into the caller's arrays;
 *       device-pointer entry points (pfc_eval_device, pfc_eval_dual_device[_bp|_more], pfc_scatter_generalized_device): the
 *       buffers live on devices[0]; the other devices get their ranges by peer copies (xGMI) and write their results back the
 *       same way, ordered against `stream` by events; pfc_check synchronises every device.
 *   - option "multi_min" (default 8): with fewer than this many items per device fewer devices take part
 *     (pfc_last_shards() tells how many did); every other option is forwarded to all devices;
 *   - pfc_get_stats sums over the devices; pfc_debug_* go to the device that evaluated the item; pfc_last_parts / _team and
 *     pfc_get_stage_ms report the first device's.
 * Results are those of the single-device handle (counters bit-equal, sums up to their order).
 */
int pfc_create_multi(const int *devices, int n_devices, pfc_handle *out);
int pfc_last_shards(pfc_handle h);   /* devices that took part in the last evaluation (1 for a single-device handle) */
void pfc_destroy(pfc_handle h);
const char *pfc_last_error(pfc_handle h);

/*
 * add_contact! -> MeshCache(name, eMesh, tree, body, c_prop) -> addMesh! (src/mechanism_scenario.jl:298-314,258;
 * src/structs.jl:33-45).  Uploads one eMesh (src/geometry/mesh.jl:10-46) and its flattened bin_BB_Tree{OBB}
 * (src/obb/tree_types.jl:1-16, src/obb/box_types.jl:4-9).  Exactly one of tri / tet is non-NULL.
 *   xyz      n_pt x 3        vertex coordinates in the mesh frame
 *   tri      n_tri x 3       (or NULL)      tet  n_tet x 4 (or NULL)      eps  n_pt (tet meshes only)
 *   Ebar     ContactProperties.Ē (tet meshes; ignored for tri meshes)
 *   nodes    n_node entries, node 0 = root: c (x3), e (x3), R (x9 column-major), child (x2), leaf (element index,
 *            or PFC_INTERNAL_NODE)
 * Returns the mesh id (>= 0) or -(pfc_status).
 */
int pfc_add_mesh(pfc_handle h, int n_pt, const double *xyz, int n_tri, const int *tri, int n_tet, const int *tet,
                 const double *eps, double Ebar, int n_node, const double *node_c, const double *node_e,
                 const double *node_R, const int *node_child, const int *node_leaf);

/*
 * add_friction_regularize! / add_friction_bristle! -> ContactInstructions (src/mechanism_scenario.jl:365-416,
 * :36-49).  id_1 is the triangle mesh (or a tet mesh), id_2 is always a tet mesh (:402-416).
 *   params (PFC_REGULARIZED): [mu_s, mu_d, v_tol]                       (Regularized, :22-34)
 *   params (PFC_BRISTLE):     [mu_s, mu_d, tau, k_bar, magic]           (Bristle, :5-20)
 * n_quad in {1, 2} (:45).  Returns the instruction id (>= 0) or -(pfc_status).
 */
int pfc_add_instruction(pfc_handle h, int id_1, int id_2, double chi, int n_quad, int model, const double *params);

/* finalize! (src/mechanism_scenario.jl:206-231): meshes and instructions become immutable, device tables are
 * built (per-tet zeta transforms of calc_ζ_transforms, src/contact_algorithms_non_friction.jl:158-162, are
 * precomputed here because meshes never change afterwards). */
int pfc_finalize(pfc_handle h);

/*
 * forceAllElasticIntersections! minus the RigidBodyDynamics parts (src/contact_algorithms_non_friction.jl:60-84):
 * evaluates n_items (instruction, pose) items.  ins_ids == NULL means item i uses instruction i.
 *   pose   n_items x 24  x_r2_r1 (R 9 col-major, t 3) then x_r1_r2 (R 9, t 3)   (refreshBodyBodyTransform!, :103-115)
 *   twist  n_items x 6   twist_r2_r1_r2 = [angular; linear]                     (refreshBodyBodyCache!, :125-128)
 *   s      n_items x 6   bristle deflection state (ignored for regularized items; may be NULL if none is bristle)
 *   wrench n_items x 6   OUT wrench on body 2 in frame r2 about its origin, [angular; linear]; zeros if no contact
 *   sdot   n_items x 6   OUT bristle state derivative (friction.jl:134; no contact: -s/tau, :77-81); zeros if regularized
 *   counts n_items x 4   OUT {OBB node tests, candidate pairs, pairs with a non-empty polygon, traction points}
 *                        (may be NULL)
 * Synchronous.  Work-list overflows are handled internally (grow + re-run), mirroring VectorCache doubling
 * (src/obb/vector_cache.jl:13-17).
 */
int pfc_eval(pfc_handle h, int n_items, const int *ins_ids, const double *pose, const double *twist,
             const double *s, double *wrench, double *sdot, int *counts);

/*
 * Same evaluation with every buffer resident in device memory (HBM) and no host synchronisation: kernels are
 * enqueued on `stream` (a hipStream_t; NULL = the handle's own stream).  d_ins_ids may be NULL.  Call
 * pfc_check() afterwards: it synchronises and returns PFC_ERR_OVERFLOW (after growing the work lists) if the
 * evaluation must be re-issued.
 */
int pfc_eval_device(pfc_handle h, int n_items, const int *d_ins_ids, const double *d_pose, const double *d_twist,
                    const double *d_s, double *d_wrench, double *d_sdot, int *d_counts, void *stream);
int pfc_check(pfc_handle h);

/*
 * The same evaluation on ForwardDiff.Dual numbers: what forceAllElasticIntersections! does when calcXd! runs on
 * MechanismScenario.dual (Dual{Nothing,Float64,N_chunk}, src/mechanism_scenario.jl:187) for Radau's Jacobian
 * (src/radau/radau_functions.jl:2-40).  Values as pfc_eval; in addition, for each of n_dir (1..16) seed directions,
 * the partials of every input and of every output:
 *   d_pose   n_items x n_dir x 24   partials of pose (same packing as pose)
 *   d_twist  n_items x n_dir x 6    d_s  n_items x n_dir x 6 (or NULL = zeros)
 *   d_wrench n_items x n_dir x 6    OUT partials of wrench     d_sdot  n_items x n_dir x 6   OUT partials of sdot
 * The candidate pairs come from the value pass (the intersection does not depend on partials,
 * src/contact_algorithms_non_friction.jl:95); every branch compares values, as ForwardDiff's comparisons do.  The one
 * step that is not the reference's operation sequence is eigen!(Hermitian{Dual}) (src/contact_algorithms_friction.jl:88,
 * GenericLinearAlgebra): the partials of K̄^{-1/2} are its analytic Frechet derivative ("Dual path").
 * An (item, direction) whose 36 seed components (d_pose 24, d_twist 6, d_s 6) are all zero has zero partials by
 * linearity and is not evaluated: the cost of a chunk follows the instructions its seeded state variables touch.
 * Host buffers, synchronous.
 */
int pfc_eval_dual(pfc_handle h, int n_items, int n_dir, const int *ins_ids, const double *pose, const double *twist,
                  const double *s, const double *d_pose, const double *d_twist, const double *d_s, double *wrench,
                  double *sdot, double *d_wrench, double *d_sdot, int *counts);

/*
 * pfc_eval_dual with the broadphase run on a pose of its own.  The reference culls with the transforms of m.float's state
 * whatever scenario is being evaluated (calcTriTetIntersections!: `refreshBodyBodyTransform!(m, m.float, c_ins)`,
 * src/contact_algorithms_non_friction.jl:94-101), and m.float holds what the last Float64 calcXd! left there
 * (src/extensions.jl:21) -- in Radau's Jacobian that is NOT value.(x_dual) in general.  bp_pose (n_items x 24, the packing of
 * pose; only the x_r1_r2 half is read) is that pose: typically the `pose` array of the host's last pfc_eval.  NULL: the
 * candidates come from pose itself, i.e. pfc_eval_dual.  The narrowphase, and every output, is evaluated at pose.
 */
int pfc_eval_dual_bp(pfc_handle h, int n_items, int n_dir, const int *ins_ids, const double *pose, const double *bp_pose,
                     const double *twist, const double *s, const double *d_pose, const double *d_twist, const double *d_s,
                     double *wrench, double *sdot, double *d_wrench, double *d_sdot, int *counts);

/*
 * pfc_eval_dual with every buffer resident in device memory and no host synchronisation (the Dual sibling of
 * pfc_eval_device; Radau's Jacobian evaluations are half of its calls, src/radau/radau_functions.jl:2-14): value pass and
 * Dual passes are enqueued back to back on `stream`.  The Dual polygons kept between the passes are sized from the
 * previous Dual evaluation of the handle; pfc_check() synchronises and returns PFC_ERR_OVERFLOW if a work list or that
 * speculation fell short (buffers have grown: re-issue).  d_ds may be NULL (zeros); d_ins_ids and d_counts may be NULL.
 */
int pfc_eval_dual_device(pfc_handle h, int n_items, int n_dir, const int *d_ins_ids, const double *d_pose,
                         const double *d_twist, const double *d_s, const double *d_dpose, const double *d_dtwist,
                         const double *d_ds, double *d_wrench, double *d_sdot, double *d_dwrench, double *d_dsdot,
                         int *d_counts, void *stream);
/* ... with the broadphase pose of pfc_eval_dual_bp in device memory (d_bp_pose n_items x 24, may be NULL). */
int pfc_eval_dual_device_bp(pfc_handle h, int n_items, int n_dir, const int *d_ins_ids, const double *d_pose,
                            const double *d_bp_pose, const double *d_twist, const double *d_s, const double *d_dpose,
                            const double *d_dtwist, const double *d_ds, double *d_wrench, double *d_sdot, double *d_dwrench,
                            double *d_dsdot, int *d_counts, void *stream);

/*
 * Further seed directions AT THE POINT OF THE PREVIOUS pfc_eval_dual_device evaluation of this handle: the chunks of one
 * Jacobian (update of the Radau iteration matrix: ceil(NX / N_chunk) Dual evaluations with the same values and different
 * partials, src/radau/radau_functions.jl:2-14).  Only the Dual passes run; candidates, contributing pairs and per-item
 * results of that evaluation's value pass are reused, its value outputs are not written again.  n_dir may differ from the
 * first call.  Requires that pfc_check() returned PFC_OK for that evaluation and that nothing else was evaluated on the
 * handle since (PFC_ERR_STATE otherwise); follow with pfc_check() (synchronises; no speculation: it cannot ask for a
 * re-issue).  pfc_eval_dual applies the same reuse by itself when the value inputs of a call equal, bit for bit, those
 * of the previous call (option "dual_reuse", default 1).
 */
int pfc_eval_dual_device_more(pfc_handle h, int n_dir, const double *d_dpose, const double *d_dtwist, const double *d_ds,
                              double *d_dwrench, double *d_dsdot, void *stream);

/*
 * eMesh_to_tree (src/geometry/blob_types.jl:136-173) on the host: builds the flattened binary OBB tree that
 * pfc_add_mesh takes.  method PFC_TREE_BLOB follows the reference (bottom-up merging of face/edge-adjacent blobs by
 * marginal cost :74-134, median-split top-down over the remaining blobs src/geometry/top_down.jl:10-32, tight leaf
 * boxes src/obb/obb_construction.jl:13-41); PFC_TREE_MEDIAN skips the bottom-up phase (pure top_down.jl).  Equal-cost
 * merges are ordered by (cost, blob key); Julia's PriorityQueue order among ties is not specified, so trees are
 * quality-equivalent rather than node-identical to Julia-built ones.
 *   elem   n_elem x arity (3: triangles, 4: tets), 0-based; eps per point (tets only, may be NULL for triangles)
 *   node_* OUT arrays sized for 2*n_elem-1 nodes: c (x3), e (x3), R (x9 column-major), child (x2), leaf (x1)
 * Returns the node count (2*n_elem-1) or a negative pfc_status; message via pfc_tree_last_error().  Reference
 * errors kept: "three triangles share the same edge", "three tetrahedrons share the same face", open triangle
 * surfaces ("not implemented error: disconnected mesh", :156), "inverted tet".
 */
enum { PFC_TREE_BLOB = 0, PFC_TREE_MEDIAN = 1 };
int pfc_build_tree(int n_pt, const double *pt, int n_elem, int arity, const int *elem, const double *eps, int method,
                   double *node_c, double *node_e, double *node_R, int *node_child, int *node_leaf);
const char *pfc_tree_last_error(void);

/*
 * addGeneralizedForcesThirdLaw! for every item (src/contact_algorithms_non_friction.jl:267-286): the item's wrench is
 * moved to the world frame with x_rw_r2 (RigidBodyDynamics transform(wrench, .)) and projected on the geometric
 * Jacobians of the two bodies (torque!), f[scene] += J_2' w - J_1' w.
 *   wrench  n_items x 6   [angular; linear], as returned by pfc_eval
 *   x_w_r2  n_items x 12  x_rw_r2: R (9, column-major) then t (3)
 *   body_1, body_2        body index of mesh_1 / mesh_2 per item, or -1 for a body without Jacobian (root: the
 *                         addGeneralizedForcesExternal!(..., jac::Nothing, ...) method, :275-279)
 *   scene   n_items       scene (mechanism) index per item, or NULL for a single mechanism
 *   jac     n_body x nv x 6   per body the 6 x nv geometric Jacobian, column-major (rows 0..2 angular, 3..5 linear)
 *   f_out   n_scene x nv  OUT f_generalized contribution of the contacts (overwritten)
 * Host buffers, synchronous.
 */
int pfc_scatter_generalized(pfc_handle h, int n_items, const double *wrench, const double *x_w_r2, const int *body_1,
                            const int *body_2, const int *scene, int n_scene, int n_body, int nv, const double *jac,
                            double *f_out);

/*
 * pfc_scatter_generalized with every buffer resident in device memory, enqueued on `stream` without a host
 * synchronisation: the wrenches are the ones pfc_eval_device left in HBM, f_generalized stays there for the next consumer
 * ("keep f_generalized resident").  accumulate = 0: d_f (n_scene x nv) is cleared first; 1: the
 * contact forces are added to what d_f holds (the reference adds to the f_generalized of the other force sources,
 * src/contact_algorithms_non_friction.jl:40-52).  Body / scene ids are NOT range-checked here (they are device data): an id
 * outside [-1, n_body) / [0, n_scene) is undefined behaviour, as with any device pointer of the wrong size.  d_scene may
 * be NULL (one mechanism).
 */
int pfc_scatter_generalized_device(pfc_handle h, int n_items, const double *d_wrench, const double *d_x_w_r2, const int *d_body_1,
                                   const int *d_body_2, const int *d_scene, int n_scene, int nv, const double *d_jac, double *d_f,
                                   int accumulate, void *stream);

/* Options: "debug" (1: keep per-pair clip counts and materialise traction points of every item so that the
 * pfc_debug_* calls work), "profile" (1: bracket each stage with HIP events), "max_levels" (0 = automatic),
 * "bfs_levels" (-1 = automatic: level-synchronous seed expansion only until there are >= 2048 seed pairs),
 * "graph" (1 = capture the launch sequence into a hipGraph per evaluation shape and replay it; default 1),
 * "no_filter" (1 = run the whole broadphase in the exact Float64 kernel instead of the Float32 filter + Float64
 * resolver; same candidate set, for A/B checks), "split_min" (default 1025; 0 = never: an evaluation of at least
 * this many items with ins_ids given is run as two concurrent halves on two streams with their own work lists, the
 * vector-ALU-bound broadphase of one half sharing the CUs with the latency-bound narrowphase of the other; results,
 * counters and stream ordering are those of the unsplit call; a SPARSE PILE -- at least 1 024 items over small or mid-sized
 * trees of which at most a quarter were in contact the last time the handle evaluated that many items -- runs as one launch
 * sequence with wider broadphase workgroups instead: pfc_last_parts() tells; the two streams of a handle are tested once for
 * running side by side -- environment PFC_NO_QUEUE_TEST=1 skips the test; pfc_eval writes the inputs of evaluations of up to 4 096
 * items straight into device memory when the device has a large PCIe BAR -- PFC_NO_BAR_INPUTS=1 keeps them in pinned host memory), "dual_reuse" (default 1: pfc_eval_dual compares the
 * value inputs of a call above the small-scene limits with those of the previous call and, if they are bitwise equal,
 * runs only the Dual passes on the previous call's value pass -- the chunks of one Jacobian), "clip_min" (default 384; 0 = never: a launch of at
 * least this many items runs the narrowphase as a clip-only kernel that keeps every clipped polygon, followed by the
 * integration over the compacted polygons; same results up to the order of the sums), "poison" (diagnostic, default 0: before every
 * evaluation the work lists are filled with entries whose item index is -1; the kernels never follow an item index
 * out of range but report it, PFC_ERR_STATE "a work-list slot was read before it was written"), "fused" (default 1:
 * an evaluation of <= 256 items over small trees runs as ONE kernel, one workgroup per item, instead of the batched
 * launch sequence -- the scene sizes Radau evaluates, src/radau/radau_functions.jl:2-14,64-70; same results; 0 = always
 * batched; the debug / profile options imply the batched path), "clip_queue" (default 1: the clip-only kernel of a
 * tri-tet launch queues the candidates that pass the trivial reject in its polygon ring and clips 64 of them at a time;
 * 0 = the lane-per-candidate clip rounds; same results bit for bit), "team" (default 48, at most 48; 0 = never: an evaluation of a few
 * pairs too big for one workgroup -- BASELINE's single 9 680-tet x 5 120-triangle pair -- runs as ONE kernel with a team
 * of up to this many workgroups per item -- while a workgroup of the team has at most ~1 200 leaves of the pair to descend: up
 * to 16 poses of that pair; beyond that the batched path is faster --, and a few mid-sized items -- a 972-tet box on the
 * ground -- with a small team each (one workgroup per 128 leaves, at most 32); same results up to the order of the sums;
 * pfc_last_team().  Team-mates wait for
 * each other inside the launch: several handles evaluating such scenes at the same moment on one device may each get
 * only part of a team resident; the wait is bounded (~65 ms), the evaluation is then re-issued on the batched path.  Inside
 * one process that case does not arise: a device has one team slot, a handle that finds it taken evaluates without a team at
 * once), "team_fault" (diagnostic, default -1: this rank of every team behaves as if its wait for the team had timed out while
 * the others saw it arrive; the evaluation must come back re-issued on the batched path, never with a wrong result),
 * "multi_min" (multi-device handles, see pfc_create_multi), "fused_f32" (default 1: the one-launch kernel runs the batched
 * broadphase's single-precision SAT filter in front of the exact Float64 test, undecided pairs settled in the same iteration;
 * same node tests, candidates and results; 0 = exact test only; items whose pose has a frame axis parallel to one of the other
 * body's -- a box resting on a plane -- run with the filter off by themselves), "dual_fold" (default 1: pass B of a Dual evaluation of
 * tri-tet scenes is formed inside pass A), "fixed_order" (default 0.  1: BIT-REPRODUCIBLE evaluations.  The reference adds an
 * instruction's traction points in one order (src/contact_algorithms_non_friction.jl:136-143), so calcXd! gives the same bits
 * every time; the default path here appends candidates in the order its workgroups finish and sums with atomics, which changes
 * last bits from run to run -- and, where decompose_K! clamps an eigenvalue that is zero in exact arithmetic at 1e-16 sigma_max
 * (a flat patch; src/contact_algorithms_friction.jl:92), whole partials.  With the option on the candidate list is sorted by
 * (item, element of mesh_1, tet of mesh_2), every per-item sum of the value and Dual passes is added in list order, and the Dual
 * eigen-decomposition takes the value of K from the value pass: two evaluations of the same inputs on handles set up the same
 * way return the same bit patterns in wrench, sdot, every partial and every counter.  Batched path only while the option is on, whatever "fused", "team"
 * and "split_min" say (no one-launch kernel, no two-half split: a reference-sized scene costs 100 - 180 us instead of 35 - 90);
 * with "debug" the value pass's sums are the debug kernel's atomics again; costs a
 * radix sort of the candidate list's capacity per evaluation (BASELINE config 5: 0.29 -> 0.48 ms, an 8 192-pose batch 3.9 -> 5.4 ms).  Needs
 * log2(items) + log2(elements of mesh_1) + log2(tets of mesh_2) <= 64 (PFC_ERR_BAD_ARG otherwise) and at most 4 096 x 512
 * candidates per item (PFC_ERR_STATE)). */
int pfc_set_option(pfc_handle h, const char *name, long long value);

/* Totals of the last checked evaluation: out[0..7] = {node tests, candidate pairs, non-empty pairs, traction
 * points, broadphase levels launched, frontier peak, status word, n_items}. */
int pfc_get_stats(pfc_handle h, long long *out8);

/* Per-stage device time of the last evaluation in ms (profile option): out[0..5] = {setup, broadphase,
 * narrowphase, bristle passes (cop + K + eigen + friction), finalisation, total}.  Synchronises.  If the evaluation
 * ran as two concurrent halves (pfc_last_parts() == 2) a stage time is the mean over the two half-launches, each of
 * which processed half of the items while stages of the other half were running; total is the longer half. */
int pfc_get_stage_ms(pfc_handle h, float *out6);
int pfc_last_parts(pfc_handle h);   /* 1, or 2 if the last checked evaluation ran as two concurrent halves; 0: it ran as the
                                     * single fused small-scene kernel (option "fused") */
int pfc_last_team(pfc_handle h);    /* workgroups per item of the last checked evaluation if it ran as one fused kernel (1: a
                                     * workgroup per item; > 1: a team per item, option "team"), else 0 */
int pfc_last_dual_reused(pfc_handle h);   /* 1 if the last Dual evaluation ran only its Dual passes on the value pass of the
                                             previous one (pfc_eval_dual_device_more, or pfc_eval_dual with equal value inputs) */

/*
 * Debug views of the last evaluation (debug option), needed to restate test/test_normal.jl:31-41 and
 * test/test_friction.jl:228-236,251-256 which read m.float.bodyBodyCache.{TractionCache,spatialStiffness}:
 *   pfc_debug_pairs       candidate (i_1, i_2) pairs of an item (the TT_Cache contents, order unspecified) and the
 *                         vertex count of each clipped polygon; returns the number of pairs of that item
 *   pfc_debug_tractions   TractionCache entries of an item, 8 doubles each: n(3) r(3) dA p
 *   pfc_debug_stiffness   spatialStiffness of a bristle item: K, K̄^{-1/2} (column-major 6x6), S^{-1}, and the cop
 * Each returns a count (>= 0) or -(pfc_status); if the count exceeds cap only cap entries were written.
 */
int pfc_debug_pairs(pfc_handle h, int item, int *pairs, int *clip_n, int cap);
int pfc_debug_tractions(pfc_handle h, int item, double *buf, int cap);
int pfc_debug_stiffness(pfc_handle h, int item, double *K36, double *Kbar_inv_sqrt36, double *Sinv6, double *cop3);

/* Diagnostic builds only (-DPFC_STAMPS): cycles the waves spent per phase in the last evaluation; zeros otherwise.
 * out[0..5]  narrowphase {gather+transform, clip, slot reservation, integration, reductions, wave rounds}
 * out[8..12] broadphase  {pop + node loads, SAT, push + flush, iterations, node pairs tested} */
int pfc_debug_stamps(pfc_handle h, long long *out16);

/* Device arithmetic self-test: out[0..n) = x/y, out[n..2n) = sqrt(|x|), out[2n..3n) = fma(x, y, x) computed on the
 * GPU, so tests can check that device division / sqrt / fma are correctly rounded (bitwise = host). */
int pfc_selftest_math(pfc_handle h, int n, const double *x, const double *y, double *out3n);

#ifdef __cplusplus
}
#endif
#endif
