// pfc_hip.hip — C ABI (include/pfc.h) and launch sequence of libpfc_hip (MI355X / gfx950).  See DESIGN.md.
//
// Device code lives in the headers included below (all in namespace pfc):
//   pfc_kernels.h  HBM records, small-vector math, SAT, wave64 helpers (ballot prefix, segmented DPP scans)
//   pfc_bp.h       broadphase: k_bp_expand (seed levels), k_bp_dfs32 (single-precision workgroup descent with the
//                  cooperative exact test), k_bp_dfs (all-Float64 descent, A/B option)         [tree_types.jl:88-111]
//   pfc_np.h       narrowphase: k_narrow (gather, clip in an LDS polygon ring, fan quadrature, pressure, regularized
//                  friction, bristle moments, kept polygons; MODE 2 = clip only for big batches), k_integ (quadrature
//                  and per-item sums over the compacted polygons), k_fric (bristle friction over kept polygons)
//   pfc_fused.h    k_fused: a small scene's whole evaluation in one launch, one workgroup per item
//   pfc_dual.h     the same path on (value, partial) numbers: k_dual_flags / k_dual_select (the pairs a chunk's seeds
//                  touch), k_narrow_dual, k_dual_poly, k_dual_eig, k_dual_final
//   pfc_br.h       k_shift, k_eig (6x6 Jacobi, one wave per item), k_final (+ result packing), k_scatter, k_selftest
//   option "fixed_order" (bit-reproducible evaluations): k_integ_fixed / k_fric_fixed (pfc_np.h), k_shift_fixed (pfc_br.h),
//                  k_fixed_reduce (pfc_dual.h), the FixedSink record lists of accumulate_items / dual_accumulate, and
//                  pfc_sort.hip (a translation unit of its own: rocPRIM radix sort of the candidate list)
//   pfc_multi.h    host code: multi-device handles (pfc_create_multi)
// This file: mesh record preparation (k_prep_tri, k_prep_tet), per-item setup (k_setup_items), work-list management,
// hipGraph capture / replay, the two-half evaluation and every extern "C" entry point.
#include "pfc_kernels.h"

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pfc.h"
#include "pfc_sort.h"

namespace pfc {

// =================================================================================================================
// mesh preparation kernels (pfc_finalize)
// =================================================================================================================
__global__ void k_prep_tri(int n, const double *__restrict__ pt, const int *__restrict__ tri, TriRec *__restrict__ out) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    V3 a = ld3(pt + 3 * tri[3 * k]), b = ld3(pt + 3 * tri[3 * k + 1]), c = ld3(pt + 3 * tri[3 * k + 2]);
    V3 nh = normalize(vector_area(a, b, c));  // triangleNormal, geometry_kernel.jl:10
    TriRec r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = b.x; r.v[4] = b.y; r.v[5] = b.z;
    r.v[6] = c.x; r.v[7] = c.y; r.v[8] = c.z; r.n[0] = nh.x; r.n[1] = nh.y; r.n[2] = nh.z;
    r.pad[0] = r.pad[1] = r.pad[2] = r.pad[3] = 0.0;
    out[k] = r;
}

__global__ void k_prep_tet(int n, const double *__restrict__ pt, const double *__restrict__ eps,
                           const int *__restrict__ tet, TetRec *__restrict__ out, double *__restrict__ out_eps,
                           unsigned *status) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double A[16], e[4];
    TetRec r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int v = tet[4 * k + j];
        A[4 * j] = pt[3 * v]; A[4 * j + 1] = pt[3 * v + 1]; A[4 * j + 2] = pt[3 * v + 2]; A[4 * j + 3] = 1.0;
        r.xrz[3 * j] = A[4 * j]; r.xrz[3 * j + 1] = A[4 * j + 1]; r.xrz[3 * j + 2] = A[4 * j + 2];
        e[j] = eps[v];
        out_eps[4 * (size_t)k + j] = e[j];
    }
    double id = inv4(A, r.xzr);
    if (!(__builtin_fabs(id) <= 1.79769313486231570815e308)) atomicOr(status, kStNonFinite);
    // ϵ_r2 = ϵ2 * x_ζ2_r2 (1x4 times 4x4)
#pragma unroll
    for (int j = 0; j < 4; ++j)
        r.epsr[j] = ((e[0] * r.xzr[4 * j] + e[1] * r.xzr[4 * j + 1]) + e[2] * r.xzr[4 * j + 2]) + e[3] * r.xzr[4 * j + 3];
    out[k] = r;
}

// =================================================================================================================
// per-evaluation setup
// =================================================================================================================
struct EvalArgs {
    int n_items;
    const int *ins_ids;      // may be null
    const double *pose, *twist, *s;
    const double *bp_pose;   // may be null: the pose the broadphase culls with (pfc_eval_dual_bp), same packing as pose
    const InsDev *ins;
    const MeshDev *meshes;
    int n_ins;
    ItemRec *items;
    WorkRec *frontier0;
    int *fcount;             // [max_levels + 2]
    double *acc;             // n_items x kAccStride
    int *icnt;               // n_items x 4
    unsigned *status;
};

// (The record is written field by field: built in registers and stored whole -- 576 bytes -- the kernel spilled 18 VGPRs.)
__global__ void __launch_bounds__(64) k_setup_items(EvalArgs g) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    {   // the accumulators of the workgroup's items, cleared with consecutive lanes on consecutive doubles (44 stores of 512
        // contiguous bytes per wave instead of 44 stores to 64 different cache lines each)
        const int i0 = blockIdx.x * blockDim.x;
        const int n_here = g.n_items - i0 < (int)blockDim.x ? g.n_items - i0 : (int)blockDim.x;
        double *a0 = g.acc + (size_t)i0 * kAccStride;
        for (int k = threadIdx.x; k < n_here * kAccStride; k += blockDim.x) a0[k] = 0.0;
    }
    if (i >= g.n_items) return;
    int id = g.ins_ids ? g.ins_ids[i] : i;
    if (id < 0 || id >= g.n_ins) {
        atomicOr(g.status, kStBadIns);
        id = 0;
    }
    const InsDev in = g.ins[id];
    const MeshDev m1 = g.meshes[in.m1], m2 = g.meshes[in.m2];
    const double *p = g.pose + 24 * (size_t)i;
    // x_r1_r2 of the broadphase: the pose's own, or the one of m.float's state when the evaluation runs on Duals
    // (calcTriTetIntersections!, non_friction.jl:94-101; k_repose puts the pose's own back once the broadphase is through)
    const double *pb = (g.bp_pose ? g.bp_pose + 24 * (size_t)i : p) + 12;
    ItemRec &r = g.items[i];
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 24; ++k) finite &= (__builtin_fabs(p[k]) <= 1.79769313486231570815e308);
#pragma unroll
    for (int k = 0; k < 9; ++k) r.R21[k] = p[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) r.t21[k] = p[9 + k];
    {
        double R12[9], t12[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) { R12[k] = pb[k]; r.R12[k] = R12[k]; finite &= (__builtin_fabs(R12[k]) <= 1.79769313486231570815e308); }
#pragma unroll
        for (int k = 0; k < 3; ++k) { t12[k] = pb[9 + k]; r.t12[k] = t12[k]; finite &= (__builtin_fabs(t12[k]) <= 1.79769313486231570815e308); }
        float q[4];
        r.pose_exact = pose_quat(R12, q) ? 0 : 1;
        r.q12[0] = q[0]; r.q12[1] = q[1]; r.q12[2] = q[2]; r.q12[3] = q[3];
        // absolute part of the single-precision test's error radius (pfc_bp.h, "Error radius E", (0)); a NaN / huge pose gives
        // NaN / inf here and every test of the item is settled exactly
        const double tm = fmax(fmax(__builtin_fabs(t12[0]), __builtin_fabs(t12[1])), __builtin_fabs(t12[2]));
        r.bp_eabs = (float)(1.4306e-6 * ((m1.cmax + m2.cmax) + tm)) * 1.000001f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.w[k] = g.twist[6 * (size_t)i + k]; r.v[k] = g.twist[6 * (size_t)i + 3 + k]; }
#pragma unroll
    for (int k = 0; k < 6; ++k) r.s[k] = (g.s && in.model == PFC_BRISTLE) ? g.s[6 * (size_t)i + k] : 0.0;
    r.chi = in.chi; r.Ebar = m2.Ebar;  // Ē of mesh_2 only: non_friction.jl:131
    r.mu_s = in.mu_s; r.mu_d = in.mu_d; r.v_c = in.v_c; r.tau = in.tau; r.k_bar = in.k_bar; r.magic = in.magic;
    r.nodes1 = m1.nodes; r.nodes2 = m2.nodes; r.nf1 = m1.nodesf; r.nf2 = m2.nodesf; r.tri = m1.tri; r.tet = m2.tet;
    r.tet1 = m1.tri ? nullptr : m1.tet; r.eps1 = m1.tet_eps; r.eps2 = m2.tet_eps; r.Ebar1 = m1.Ebar;
    r.model = in.model; r.nq = (in.nq == 1) ? 1 : 3;  // quadrature POINTS of rule 1 / rule 2 (quadrature.jl:22,31)
    r.ins = id; r.pad = 0;
    r.pad2[0] = r.pad2[1] = 0;
    if (!finite) atomicOr(g.status, kStNonFinite);
    WorkRec w;
    // pad: bit 0 / bit 1 = node a / node b is a leaf (a one-element mesh: the root is the leaf), so that the
    // depth-first kernel can start a seed without reading the two nodes first
    w.item = i; w.a = 0; w.b = 0; w.pad = (m1.n_node == 1 ? 1 : 0) | (m2.n_node == 1 ? 2 : 0);
    g.frontier0[i] = w;
    *reinterpret_cast<int4 *>(g.icnt + 4 * (size_t)i) = make_int4(0, 0, 0, 0);
    if (i == 0) g.fcount[0] = g.n_items;
}

// After the broadphase of an evaluation with a broadphase pose of its own: x_r1_r2 of the item records becomes the pose's
// (the tet-tet op and the Dual passes read it: find_plane_tet, non_friction.jl:164,174-177).
__global__ void k_repose(int n_items, const double *__restrict__ pose, ItemRec *__restrict__ items) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    const double *p = pose + 24 * (size_t)i + 12;
#pragma unroll
    for (int k = 0; k < 9; ++k) items[i].R12[k] = p[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) items[i].t12[k] = p[9 + k];
}

#include "pfc_bp.h"
#include "pfc_clip.h"
#include "pfc_np.h"
#include "pfc_dual.h"
#include "pfc_br.h"
#include "pfc_fused.h"

}  // namespace pfc

// =================================================================================================================
// host side
// =================================================================================================================
using namespace pfc;

namespace {

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc((void **)&p, n * sizeof(T));
        if (e == hipSuccess) cap = n;
        if (std::getenv("PFC_LOG_ALLOC"))      // diagnostic: match a faulting address with the buffer it lies behind
            std::fprintf(stderr, "pfc alloc %p .. %p (%zu bytes, element %zu)\n", (void *)p, (void *)((char *)p + n * sizeof(T)),
                         n * sizeof(T), sizeof(T));
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct HostMesh {
    int n_pt = 0, n_tri = 0, n_tet = 0, n_node = 0, n_leaf = 0, depth = 0;
    double Ebar = 0.0;
    std::vector<double> xyz, eps;
    std::vector<int> tri, tet;
    std::vector<NodeRec> nodes;
    std::vector<NodeF> nodesf;
    double cmax = 0.0;                   // max |c|_1 over the nodes (MeshDev.cmax)
    NodeRec *d_nodes = nullptr;
    NodeF *d_nodesf = nullptr;
    TriRec *d_tri = nullptr;
    TetRec *d_tet = nullptr;
    double *d_tet_eps = nullptr;
};

enum { EV_START = 0, EV_SETUP, EV_BP, EV_NP, EV_BR, EV_FIN, EV_COUNT };

}  // namespace

struct pfc_multi;
struct pfc_context {
    int device = 0;
    bool finalized = false;
    pfc_multi *multi = nullptr;          // non-null: a multi-device handle (pfc_create_multi); everything below belongs to its shards
    // Broadphase pose of the evaluation being enqueued (device-visible, n_items x 24, only x_r1_r2 is read) or null: set by the
    // pfc_eval_dual*_bp entry points around the value pass (record_eval, enqueue_fused)
    const double *bp_dev = nullptr;
    void *pin_bp = nullptr;              // pinned host copy of the last broadphase pose block (pfc_eval_dual_bp)
    size_t pin_bp_cap = 0;
    int pin_bp_n = 0;                    // items it holds (0: the last host-buffer Dual evaluation had none)
    bool team_owner = false;             // this handle holds its device's team slot (team_acquire)
    int opt_team_fault = -1;             // diagnostic option "team_fault": rank of every team that simulates a timed-out wait
    int opt_dual_fold = 1;               // option "dual_fold": pass B of the Dual evaluation formed inside pass A (tri-tet scenes, batched value pass)
    int opt_fused_f32 = 1;               // option "fused_f32": single-precision SAT filter in the one-launch kernel (A/B knob; same results)
    std::string err;
    hipStream_t stream = nullptr;
    std::vector<HostMesh> meshes;
    std::vector<InsDev> ins;
    MeshDev *d_meshes = nullptr;
    // every mesh's device records (NodeF, NodeRec, TriRec / TetRec, raw eps) are carved out of ONE allocation, the
    // single-precision nodes of all meshes first (a pile of 128 meshes used to make ~600 small hipMallocs; measured neutral
    // for C5's evaluation time, kept for the contiguous hot set and the one free)
    char *mesh_arena = nullptr;
    InsDev *d_ins = nullptr;
    int max_levels = 1;
    int max_leaves = 2;                // largest n_leaf(mesh_1) + n_leaf(mesh_2) over the instructions
    bool any_bristle = false, any_tet_tet = false;
    // options
    int opt_debug = 0, opt_profile = 0, opt_max_levels = 0, opt_bfs_levels = -1, opt_no_filter = 0;
    // work buffers
    DevBuf<ItemRec> items;
    DevBuf<WorkRec> frontier[2], cand;
    DevBuf<int> clip_n, icnt, trac_item;
    DevBuf<double> acc, res, trac_d, rec;   // trac_d: 8 arrays of tcap; rec: moment records of kRecStride doubles
    DevBuf<int> ctr;                   // [0]=ccount [1]=tcount [2..] fcount[levels+2]
    DevBuf<unsigned> status;
    DevBuf<unsigned long long> stamps;   // diagnostic builds
    DevBuf<int> tail;                    // packed status, totals, counters (block 0 of k_final)
    int *h_tail = nullptr;               // pinned host mirror of tail
    const int *tail_host = nullptr;      // set by pfc_eval: the tail is already on its way to this pinned block (with the outputs)
    int *tail_dev = nullptr;             // set by pfc_eval for a small scene: k_final packs straight into pinned host memory
    size_t h_tail_cap = 0;
    void *pin_in = nullptr, *pin_out = nullptr;   // pinned staging of the host-buffer path
    size_t pin_in_cap = 0, pin_out_cap = 0;
    // Inputs of a very small scene written by the host straight into DEVICE memory (fine-grained allocation, reachable through
    // the PCIe BAR when the device has a large one): the one-launch kernel's first two dependent reads -- instruction id, then
    // the instruction's record -- start from HBM instead of from host memory (scripts/micro/bar_probe.hip: launch + two
    // dependent reads + completion word 7.5 -> 6.2 us).  Host stores into it are write-combined: written once per evaluation,
    // never read back; they leave the core with the locked update of the queue's write index that every launch begins with.
    void *bar_in = nullptr;
    void *bar_din = nullptr;                      // the same for the seeds of a small-scene Dual evaluation (up to kBarKeys (item, direction) pairs)
    int bar_state = 0;                            // 0: not probed, 1: in use, -1: no large BAR / allocation failed / PFC_NO_BAR_INPUTS
    void *pin_din = nullptr, *pin_dout = nullptr; // pinned blocks of the small-scene Dual path (partials in / out)
    size_t pin_din_cap = 0, pin_dout_cap = 0;
    long long dual_hint = -1;                     // contributing pairs of the last Dual evaluation (-1: none yet)
    bool pending_dual = false;                    // pfc_eval_dual_device enqueued: pfc_check also checks the speculative polygon capacity
    bool pending_dual_hyb = false;                // ... through the small-scene kernel (value pass + hand-over) and the batched Dual passes
    bool dual_reuse_emit = false;                 // the reusable value pass is a hand-over of the small-scene kernel (pair count: emit_ctr)
    int dual_dev_hyb_skip = 0;                    // evaluations for which pfc_eval_dual_device leaves that path alone after a miss
    // Reuse of a value pass by further Dual evaluations at the same point (the chunks of one Jacobian,
    // src/radau/radau_functions.jl:2-14): set by pfc_check after a Dual evaluation on the device path, cleared by every
    // other evaluation and option change
    bool dual_reuse_ok = false;
    int dual_reuse_n = 0;                         // its item count
    int pin_in_dual_n = 0;                        // pin_in holds the value inputs of a small-scene Dual evaluation of this many items (0: not)
    bool pin_in_dual_ids = false;
    bool small_reuse_ok = false;                  // the same for the one-graph small-scene path (eval_dual_small)
    int small_reuse_n = 0;
    bool small_reuse_ids = false;
    bool hyb_reuse_ok = false;                    // the same for the small-scene hybrid path (fused value kernel + batched Dual passes)
    int hyb_reuse_n = 0;
    bool hyb_reuse_ids = false;
    bool pending_more = false;                    // pfc_eval_dual_device_more enqueued: pfc_check only synchronises
    bool last_dual_reused = false;                // the last Dual evaluation ran on a reused value pass (pfc_last_dual_reused)
    int opt_dual_reuse = 1;
    bool pin_din_valid = false;                   // pin_din / pin_dout still hold the value inputs / outputs of that evaluation
    unsigned long long value_serial = 0;          // bumped by every value pass that overwrites the device lists (record_eval, enqueue_eval, enqueue_fused)
    unsigned long long pin_din_serial = 0;        // value_serial of the evaluation pin_din / pin_dout belong to: the host cache is only valid for THAT value pass
    bool pin_din_ids = false;                     // ... which had ins_ids
    size_t pin_din_nk36 = 0;                      // doubles of seeds between the value block and the ids in pin_din
    std::vector<int> dual_counts_cache;           // its per-item counters
    size_t pending_dpcap = 0;
    int pending_ndir = 0;
    DevBuf<double> dual_zero;                     // zeros standing in for a null d_ds
    unsigned long long epoch = 0;        // bumped whenever a device work buffer is reallocated
    // captured launch sequence (hipGraph) of the last evaluation shape
    hipGraphExec_t gexec[2] = {nullptr, nullptr};   // [0] plain evaluation, [1] with the contributing-pair list (Dual)
    struct GraphKey {
        int n_items, levels, L, debug, bristle, surv;
        const void *p[9];
        void *stream;
        unsigned long long epoch;
    } gkey[2] = {};
    bool ghave[2] = {false, false};
    // value pass + Dual passes of a small scene as ONE graph (eval_dual_small)
    hipGraphExec_t dgexec = nullptr;
    struct DualGraphKey { GraphKey v; int n_dir; size_t bound; const void *din, *dout; } dgkey = {};
    bool dghave = false;
    bool want_surv = false;   // the narrowphase also lists the contributing candidates (pfc_eval_dual)
    int opt_graph = 1;
    size_t fcap = 0, ccap = 0, tcap = 0, rcap = 0;
    // host-pointer path staging
    DevBuf<double> h_pose;               // device mirror of pfc_eval's pinned input block (pose | twist | s | ins_ids)
    // last evaluation
    int last_n_items = 0, last_levels = 0, last_bfs_levels = 0;
    bool pending = false;
    hipStream_t last_stream = nullptr;
    long long stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    DevBuf<double> dual_in, dual_acc, dual_res, dual_out, dual_poly;   // pfc_eval_dual
    DevBuf<int2> dual_pkey;
    DevBuf<int> dual_sel, dual_flag;                        // pairs a chunk has work for; per-item marks (+ the list's counter)
    DevBuf<double> scat_d;                                  // pfc_scatter_generalized
    DevBuf<int> scat_i;
    DevBuf<int> surv;                                       // candidate indices of contributing pairs
    DevBuf<int> rgn;                                        // region counters of the polygon / record lists
    DevBuf<int> poly_item, pcnt, poly_cand;                 // kept polygons (k_narrow -> k_integ, k_fric): keys, count per chunk, candidate index (Dual list)
    DevBuf<double> poly;
    long long last_undecided = 0;      // node pairs the Float32 broadphase settled with the exact Float64 test
    long long last_active = 0;         // items of the last checked evaluation whose root pair overlapped (more than one node test)
    // what the last checked evaluation looked like: a SPARSE pile (at most a quarter of the items in contact: all pairs of a
    // pile of bodies, most of them apart) of mid-sized trees is latency-bound by the descents of its few big pairs, a dense
    // batch by throughput -- the next evaluation of the same shape is laid out accordingly (pile_mode)
    // (the last four shapes: a host that alternates a few batch sizes keeps a picture of each)
    struct ShapeHint { int n = 0; bool sparse = false; } hints[4];
    void note_shape(int n, bool sparse) {
        int k = 0;
        while (k < 3 && hints[k].n != n) ++k;      // the entry of this size, or the oldest
        for (; k > 0; --k) hints[k] = hints[k - 1];
        hints[0].n = n; hints[0].sparse = sparse;
    }
    bool shape_is_sparse(int n) const {
        for (const ShapeHint &e : hints) if (e.n == n) return e.sparse;
        return false;
    }
    long long last_tslots = 0;         // traction slots used by the last evaluation (>= traction points)
    hipEvent_t ev[EV_COUNT] = {};
    bool ev_valid = false;
    // Large batches are evaluated as two concurrent halves: a second set of work buffers on a second stream, so that
    // the broadphase of one half (vector-ALU bound) shares the CUs with the narrowphase of the other (parked on
    // s_waitcnt half of the time).  Measured on the C3 batch: 2.39 -> 2.04 ms per 2 048 poses; four parts are slower.
    pfc_context *twin = nullptr;
    bool is_twin = false;
    // small scenes: one fused kernel, one workgroup per item (pfc_fused.h)
    InsFull *d_insfull = nullptr;
    DevBuf<int> fout;                  // per item 8 ints (status, counts) of the fused kernel (device-buffer entry point)
    int *h_fout = nullptr;             // pinned host mirror
    size_t h_fout_cap = 0;
    int *fout_dev = nullptr;           // set by pfc_eval: the kernel writes its per-item block straight into pinned host memory
    const int *fout_host = nullptr;    //   ... and this is where the host reads it
    int opt_fused = 1;                 // option "fused"
    // option "fixed_order": the candidate list sorted (pfc_sort.hip), an item's run records added in chunk order (k_integ_fixed,
    // k_shift_fixed), the Dual passes' eigen-decomposition on the value pass's K -- batched path only, no graph, no split
    int opt_fixed_order = 0;
    DevBuf<int> det;                          // per item: last record, first chunk, last chunk
    DevBuf<unsigned long long> sort_keys[2];
    DevBuf<int> canon_off, canon_fill, canon_item;      // segments of the candidate list by item (pfc_canon_candidates)
    int sort_tmp_items = 0;
    bool fixed_whole_list = false;            // an item had more candidates than a segment sort takes: the whole list is sorted from then on
    DevBuf<char> sort_tmp;
    size_t sort_tmp_for = 0;                  // (capacity, bits) the temporary storage was sized for
    int sort_tmp_bits = 0;
    int max_elem1 = 1, max_elem2 = 1;         // most elements of a mesh on side 1 / side 2 of an instruction
    DevBuf<double> fx_rec, vfx_rec;           // records of the Dual passes' sums / of the value pass's friction sums (FixedSink)
    DevBuf<int> vfx_head;
    DevBuf<int> fx_head;                      // per key and accumulator block: last record; then the record counter
    long long surv_sorted_serial = -1;        // value pass whose list of contributing pairs has been sorted
    // Slots of the candidate list the sort covers: its capacity for the first evaluation and whenever the list turned out longer
    // than covered (the evaluation is then re-issued), else a power of two above twice the previous evaluation's candidates -- a
    // reference-sized scene sorts 1 024 keys in one launch instead of 65 536 in fifteen.
    size_t sort_cover = 0;                    // 0: the capacity
    size_t sort_cover_used = 0;               // what the pending evaluation covered
    int opt_clip_queue = 1;            // option "clip_queue": clip-only narrowphase of big tri-tet batches in k_clip_queue (survivors queued in the ring); 0: k_narrow<.., 2 / 3>
    int fused_skip = 0;                // evaluations left for which the fused kernel stays off after an item did not fit
    int fused_seq = 0;                 // sequence number of the last fused launch (completion word of the polled path)
    int fu_nw = 1;                     // workgroups per item of the next fused launch (teams: k_fused<.., true>)
    int last_fu_nw = 0, last_team = 0; // of the last fused launch / of the last checked evaluation (pfc_last_team)
    int n_cu = 0;                      // compute units of the device (a team launch keeps every workgroup resident: one per CU)
    int opt_team = 48;                 // option "team": big pairs (more leaves than one workgroup takes) run as teams of up to this many workgroups (0: batched path; <= kTeamMaxWg = 48: s_team and the gather's per-thread granule count are sized from it).  Eight single C3 poses, mean / worst us (an earlier build that allowed 64): 64: 113 / 122, 48: 112 / 118, 32: 113 / 122, 24: 126 / 190, 16: 178 / 249, batched 132 / 136 (scripts/lat_c3_poses.py)
    DevBuf<unsigned long long> team;   // team partial sums: kTeamMaxBlocks x 3 x 2 kTeamSlots granules, zeroed once (tags are launch sequence numbers >= 1)
    DevBuf<int> emit_ctr;              // pair counter of the fused kernel's hand-over to the batched Dual passes
    int *h_emit = nullptr;             // pinned mirror
    unsigned *h_more = nullptr;        // pinned: status word of the Dual passes of pfc_eval_dual_device_more (device word: status.p + 1)
    bool fu_emit = false;              // set by eval_dual_hybrid around enqueue_fused
    int dual_fused_skip = 0;           // Dual evaluations left for which the in-kernel Dual passes stay off (an item had too many polygons)
    const double *fu_dpose = nullptr, *fu_dtwist = nullptr;    // set by eval_dual_fused around enqueue_fused
    double *fu_dwrench = nullptr, *fu_dsdot = nullptr;
    int fu_ndir = 0;
    bool pending_fused = false, last_fused = false;
    int opt_split_min = 1025;          // 0: never split.  (Paired sweeps at the end of round 2: 1 024 poses -- one seed per resident broadphase workgroup -- 0.80 ms unsplit / 0.90 split; 1 100 0.95 / 0.93, 1 280 1.09 / 1.01, 2 048 1.50 / 1.28.)
    int opt_clip_min = 384;            // items per launch from which the narrowphase runs as clip-only kernel + k_integ; 0: never.  (End of round 3, k_clip_queue for every such launch, scripts/sweep_clip_min.py: full-size C3 poses 256: 405 vs 405 us one-kernel / split, 400: 518 vs 502, 511: 598 vs 568; 400 box-on-plane scenes 120 vs 123 -- 384, was 512.)  (First set at 1 024 from scripts/sweep_clip.sh; a paired sweep with the final kernels: 512 poses 0.61 vs 0.63 ms, 768 0.80 vs 0.83, 1 536 as 2 x 768 1.14 vs 1.19, 1 920 1.33 vs 1.40; 384 poses and below are indifferent.)
    int opt_poison = 0;                // diagnostic: fill (re)allocated work lists with 0xFF bytes (item index -1)
    int split_n0 = 0;                  // items in the first half of the pending evaluation (0: not split)
    bool in_split = false;             // this context's launches are one half of a two-half evaluation (set while they are enqueued)
    int last_parts = 1;                // 2 if the last checked evaluation ran as two halves
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join0 = nullptr;
    int twin_queue_fallback = 0;         // make_twin: 0 the two streams run side by side as created; 1 the twin's stream was re-created with another priority because they did not; 2 they still do not
    hipStream_t twin_stream = nullptr;   // created right behind `stream` (the runtime deals streams out to its hardware queues in order of creation), handed to the twin
};

namespace {

int fail(pfc_context *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    return code;
}

#define HIP_TRY(h, expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(h, PFC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int tree_depth(const std::vector<NodeRec> &nodes) {
    // iterative DFS; also validates child indices
    int n = (int)nodes.size(), best = 0;
    std::vector<std::pair<int, int>> st;
    st.push_back({0, 0});
    size_t visited = 0;
    while (!st.empty()) {
        auto [k, d] = st.back();
        st.pop_back();
        if (k < 0 || k >= n || ++visited > (size_t)n) return -1;
        best = d > best ? d : best;
        if (nodes[k].leaf == kInternal) {
            st.push_back({nodes[k].child0, d + 1});
            st.push_back({nodes[k].child1, d + 1});
        }
    }
    return best;
}

int grid_for(size_t n, int block, int max_blocks) {
    size_t b = (n + block - 1) / block;
    if (b < 1) b = 1;
    if (b > (size_t)max_blocks) b = max_blocks;
    return (int)b;
}

// Kept-polygon slots: chunk ch of the candidate list owns slots [ch C, ch C + C) (pfc_np.h, np_chunk): the candidate
// capacity rounded up to a whole chunk.
constexpr int kNpMaxBlocks = 256 * 16;
constexpr int kNpChunkSwitch = 2048;   // a batch is cut into at least this many chunks before the chunks grow beyond a wave round (512: C5 +90 us; 8192: the 2 048-pose step +9 %)
size_t poly_cap(size_t ccap) { return (ccap + kNpChunkBig - 1) / kNpChunkBig * kNpChunkBig; }

// A blocking copy that stays off the legacy stream (hipMemcpy synchronises with it, and the runtime refuses that while ANY
// thread captures a graph: "operation would make the legacy stream depend on a capturing blocking stream" -- a second
// handle being finalized on one host thread while another thread's evaluation records its graph; scripts/soak_threads.py).
hipError_t copy_sync(pfc_context *h, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, h->stream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(h->stream);
}

size_t fixed_fric_cap(size_t ccap, int n_items) { return 2 * (ccap / 64 + 1) + (size_t)n_items + 64; }      // (FixedSink of k_fric_fixed)

hipError_t ensure_work(pfc_context *h, int n_items) {
    hipError_t e;
    const size_t caps0[] = {h->items.cap, h->acc.cap, h->res.cap, h->icnt.cap, h->ctr.cap, h->frontier[0].cap,
                            h->frontier[1].cap, h->cand.cap, h->clip_n.cap, h->trac_item.cap, h->trac_d.cap, h->rec.cap,
                            h->tail.cap, h->poly_item.cap, h->poly.cap, h->pcnt.cap, h->poly_cand.cap};
    if ((e = h->items.ensure(n_items)) != hipSuccess) return e;
    if ((e = h->acc.ensure((size_t)n_items * kAccStride)) != hipSuccess) return e;
    if ((e = h->res.ensure((size_t)n_items * kResStride)) != hipSuccess) return e;
    if ((e = h->icnt.ensure((size_t)n_items * 4)) != hipSuccess) return e;
    {
        const size_t c0 = h->ctr.cap, s0 = h->status.cap;
        if ((e = h->ctr.ensure((size_t)h->max_levels + 12)) != hipSuccess) return e;
        if ((e = h->status.ensure(4)) != hipSuccess) return e;
        // The kernels rely on these being zero on entry and run on non-blocking streams (the handle's, the twin's or
        // the caller's), which are not ordered against the legacy null stream: clear on the handle's stream and wait.
        bool cleared = false;
        if (h->ctr.cap != c0) { if ((e = hipMemsetAsync(h->ctr.p, 0, sizeof(int) * h->ctr.cap, h->stream)) != hipSuccess) return e; cleared = true; }
        if (h->status.cap != s0) { if ((e = hipMemsetAsync(h->status.p, 0, sizeof(unsigned) * h->status.cap, h->stream)) != hipSuccess) return e; cleared = true; }
        const size_t r0 = h->rgn.cap;
        if ((e = h->rgn.ensure((size_t)kRgn * kRgnStride)) != hipSuccess) return e;
        if (h->rgn.cap != r0) { if ((e = hipMemsetAsync(h->rgn.p, 0, sizeof(int) * h->rgn.cap, h->stream)) != hipSuccess) return e; cleared = true; }
        if (cleared && (e = hipStreamSynchronize(h->stream)) != hipSuccess) return e;
    }
    if ((e = h->stamps.ensure(16)) != hipSuccess) return e;
    size_t f = h->fcap ? h->fcap : 1u << 16;
    while (f < (size_t)n_items * 8) f *= 2;
    size_t c = h->ccap ? h->ccap : 1u << 16;
    while (c < (size_t)n_items * 4) c *= 2;
    size_t t = h->tcap ? h->tcap : 1u << 16;
    size_t rc = h->rcap ? h->rcap : 1u << 12;
    while (rc < c / 32 + (size_t)n_items * 2) rc *= 2;   // about one record per wave round and item boundary
    h->fcap = f; h->ccap = c; h->tcap = t; h->rcap = rc;
    if ((e = h->rec.ensure(rc * (h->opt_fixed_order ? kRecStrideFixed : kRecStride))) != hipSuccess) return e;
    if (h->opt_fixed_order) {
        if ((e = h->det.ensure((size_t)n_items * 3)) != hipSuccess) return e;
        if ((e = h->vfx_rec.ensure(fixed_fric_cap(c, n_items) * kSinkStride)) != hipSuccess) return e;
        if ((e = h->vfx_head.ensure((size_t)n_items + 2)) != hipSuccess) return e;
        if ((e = h->sort_keys[0].ensure(c)) != hipSuccess) return e;
        if ((e = h->sort_keys[1].ensure(c)) != hipSuccess) return e;
        int ba, bb;
        const int bits = pfc_sort_key_bits(n_items, h->max_elem1, h->max_elem2, &ba, &bb);
        if ((e = h->canon_off.ensure((size_t)n_items + 1)) != hipSuccess) return e;
        if ((e = h->canon_fill.ensure((size_t)n_items)) != hipSuccess) return e;
        if ((e = h->canon_item.ensure(c)) != hipSuccess) return e;
        if (h->sort_tmp_for != c || h->sort_tmp_bits != bits || h->sort_tmp_items < n_items) {
            size_t bytes = 0;
            if ((e = pfc_sort_temp_bytes(c, bits > 64 ? 64 : bits, &bytes)) != hipSuccess) return e;      // (covers the index sort too)
            if ((e = h->sort_tmp.ensure(bytes ? bytes : 1)) != hipSuccess) return e;
            h->sort_tmp_for = c; h->sort_tmp_bits = bits; h->sort_tmp_items = n_items;
        }
    }
    if ((e = h->frontier[0].ensure(f)) != hipSuccess) return e;
    if ((e = h->frontier[1].ensure(f)) != hipSuccess) return e;
    if ((e = h->cand.ensure(c)) != hipSuccess) return e;
    if ((e = h->clip_n.ensure(c)) != hipSuccess) return e;
    if ((e = h->surv.ensure(c)) != hipSuccess) return e;
    {   // kept polygons: bristle items always (friction pass), every item when the clip-only narrowphase + k_integ run
        const size_t pc = poly_cap(c);
        if ((e = h->poly_item.ensure(pc)) != hipSuccess) return e;
        if ((e = h->poly.ensure(pc * 34)) != hipSuccess) return e;
        if ((e = h->pcnt.ensure(pc / kNpBlock + 1)) != hipSuccess) return e;
        if (h->want_surv && (e = h->poly_cand.ensure(pc)) != hipSuccess) return e;
    }
    if ((e = h->trac_item.ensure(t)) != hipSuccess) return e;
    if ((e = h->trac_d.ensure(t * 8)) != hipSuccess) return e;
    if ((e = h->tail.ensure((size_t)h->max_levels + 40)) != hipSuccess) return e;
    if (h->h_tail_cap < (size_t)h->max_levels + 40) {
        if (h->h_tail) (void)hipHostFree(h->h_tail);
        h->h_tail = nullptr; h->h_tail_cap = 0;
        if ((e = hipHostMalloc((void **)&h->h_tail, sizeof(int) * ((size_t)h->max_levels + 40))) != hipSuccess) return e;
        h->h_tail_cap = (size_t)h->max_levels + 40;
    }
    const size_t caps1[] = {h->items.cap, h->acc.cap, h->res.cap, h->icnt.cap, h->ctr.cap, h->frontier[0].cap,
                            h->frontier[1].cap, h->cand.cap, h->clip_n.cap, h->trac_item.cap, h->trac_d.cap, h->rec.cap,
                            h->tail.cap, h->poly_item.cap, h->poly.cap, h->pcnt.cap, h->poly_cand.cap};
    for (size_t k = 0; k < sizeof caps0 / sizeof caps0[0]; ++k)
        if (caps0[k] != caps1[k]) { ++h->epoch; break; }
    if (h->opt_poison) {    // every evaluation starts from lists full of entries that must never be followed
        if ((e = hipMemsetAsync(h->frontier[0].p, 0xFF, sizeof(WorkRec) * h->frontier[0].cap, h->stream)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(h->frontier[1].p, 0xFF, sizeof(WorkRec) * h->frontier[1].cap, h->stream)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(h->cand.p, 0xFF, sizeof(WorkRec) * h->cand.cap, h->stream)) != hipSuccess) return e;
        if (h->poly_item.p && (e = hipMemsetAsync(h->poly_item.p, 0xFF, sizeof(int) * h->poly_item.cap, h->stream)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return e;
    }
    return hipSuccess;
}

TracSoA trac_view(pfc_context *h) {
    TracSoA t;
    size_t n = h->tcap;
    double *d = h->trac_d.p;
    t.item = h->trac_item.p;
    t.nx = d; t.ny = d + n; t.nz = d + 2 * n; t.rx = d + 3 * n; t.ry = d + 4 * n; t.rz = d + 5 * n;
    t.dA = d + 6 * n; t.p = d + 7 * n;
    return t;
}

// Option "max_levels" caps the number of broadphase levels the counters / seed expansion are laid out for.  Every buffer
// (ctr, tail, h_tail) is sized from the finalized depth h->max_levels, so the effective value never exceeds it; the
// depth-first kernels' stack reserve always uses h->max_levels (a smaller reserve could overrun the LDS stack).
int eff_levels(const pfc_context *h) {
    return (h->opt_max_levels > 0 && h->opt_max_levels < h->max_levels) ? h->opt_max_levels : h->max_levels;
}

constexpr int kDualSelectMin = 512;       // items from which a Dual evaluation first selects the pairs its seeds touch
constexpr int kBpSmallBlockMin = 3072;   // items per launch from which k_bp_dfs32 runs in 128-thread workgroups (paired A/B: 2 048 per launch 2.31 vs 2.36 ms for 256 / 128 threads, 3 072 per launch 3.39 vs 3.30)

int bfs_levels_for(const pfc_context *h, int n_items, int levels) {
    int L = 0;
    if (h->opt_bfs_levels >= 0) {
        L = h->opt_bfs_levels;
    } else {
        // Level-synchronous expansion only buys parallelism for the depth-first kernel (one workgroup per seed, ~1 000
        // resident workgroups); each level is a launch (~8 us).  Measured optimum (scripts/latency.py, bench.py):
        // big trees (a traversal is ~1.5 node tests per leaf) want >= 1 024 seeds, or 4 096 when there are only a few
        // items; small trees are done in a few iterations per seed, so 64 seeds suffice (C4: 200 -> 144 us).  A large
        // batch over small trees is usually a sparse-contact pile where a few pairs carry the work: one level spreads them.
        // Round 2 re-sweep for big trees (scripts/sweep_bfs_small.sh, C3 meshes: 1 pose 139 -> 125 us, 4 poses 167 -> 155,
        // 16 poses 215 -> 202, 64 poses 367 -> 309, 200 poses 519 -> 460): the fewer the items, the more seeds pay.
        const bool big = h->max_leaves >= 8192, mid = h->max_leaves >= 1024;
        // (end of round 2, final kernels: from ~600 items a seed level only costs -- 640 poses 0.71 vs 0.74 ms without /
        // with one level, 768 0.73 vs 0.80, 1 023 0.80 vs 0.92; 512 and below are indifferent or gain)
        // (End of round 3, scripts/sweep_bp_blk_levels.py / sweep_bp_split.py: mid-sized trees want ~256 seeds, not 1 024 --
        // 512 poses of a 1 280-leaf pair 185 vs 213 us with 0 / 1 level -- and a batch of >= 1 024 items over small or mid-sized
        // trees no level at all: the level that used to be forced there, tuned on the sparse pile C5 (340 vs 370 us), cost dense
        // batches 20-30 %: 2 000 poses 381 vs 500 us, 2 500 box-on-plane scenes 343 vs 493.  The pile gets its parallelism from
        // 512-thread workgroups instead, pile_mode.)
        // (a half of a two-half evaluation shares the chip with its twin's seeds: 1 100 poses as 2 x 550, 891 vs 930 us
        // without / with the level its own 550 items would ask for)
        const int n_eff = h->in_split ? 2 * n_items : n_items;
        const double target = big ? (n_eff >= 600 ? 1.0 : (n_eff >= 256 ? 1024.0 : (n_eff >= 128 ? 2048.0 : (n_eff >= 32 ? 16384.0 : 49152.0))))
                                  : (mid ? 256.0 : 64.0);
        double seeds = (double)n_items;
        while (seeds < target && L < 9) { seeds *= 4.0; ++L; }
    }
    return L > levels ? levels : L;
}

// A sparse pile (see hints): >= 1 024 items over small or mid-sized trees of which at most a quarter were in contact the
// last time this shape was evaluated.  Evaluated in ONE launch sequence with 512-thread broadphase workgroups (C5: 318 us
// against 340 as two halves with 256-thread workgroups and a seed level; without the level 370).
bool pile_mode(const pfc_context *h, int n_items) {
    return h->max_leaves < 8192 && n_items >= 1024 && n_items < kBpSmallBlockMin && h->shape_is_sparse(n_items) && !h->is_twin;
}

// Threads per workgroup of the depth-first broadphase kernel.  128: the halves of a big batch (throughput: the finer grain
// shares the CUs with the other half's kernels).  512: launches whose time is the descent of a few big pairs -- 8 ... 1 023 items
// over big trees (16 full-size C3 poses 198 -> 183 us, 128: 386 -> 314, 256: 459 -> 399, 600: 681 -> 662; from 1 024 items
// on 256 threads win again) and sparse piles: an iteration costs ~1.7 us whatever its width, 512 pairs per iteration halve
// the chain.  (1 024 threads: no further gain, C5 105 vs 97 us of broadphase.)  256 otherwise.
int bp_block_for(const pfc_context *h, int n_items) {
    if (n_items >= kBpSmallBlockMin) return 128;
    if (h->in_split) return 256;
    if (h->max_leaves >= 8192 && n_items >= 8 && n_items < 1024) return 512;
    if (pile_mode(h, n_items)) return 512;
    return 256;
}

// The launch sequence of one evaluation on stream st (eagerly, or while st is being captured into a graph).
int record_eval(pfc_context *h, int n_items, const int *d_ins_ids, const double *d_pose, const double *d_twist,
                const double *d_s, double *d_wrench, double *d_sdot, int *d_counts, hipStream_t st, bool prof) {
    h->dual_reuse_ok = false; h->hyb_reuse_ok = false; h->small_reuse_ok = false; h->pending_more = false; h->last_dual_reused = false;
    ++h->value_serial;
    const int levels = eff_levels(h);
    int *ccount = h->ctr.p, *tcount = h->ctr.p + 1, *next_seed = h->ctr.p + 2;   // ctr[3]: total records, filled by k_final
    int *ucount = h->ctr.p + 4, *fcount = h->ctr.p + 6;
    int *pcount = h->ctr.p + ((levels + 9) & ~1);   // after the per-level frontier counts; 8-byte aligned pair
    // counters and status are zero here: k_final of the previous evaluation (or ensure_work after an allocation) left them so
#ifdef PFC_STAMPS
    HIP_TRY(h, hipMemsetAsync(h->stamps.p, 0, sizeof(unsigned long long) * 16, st));
#endif
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[EV_START], st));

    EvalArgs ea;
    ea.n_items = n_items; ea.ins_ids = d_ins_ids; ea.pose = d_pose; ea.twist = d_twist; ea.s = d_s;
    ea.bp_pose = h->bp_dev;
    ea.ins = h->d_ins; ea.meshes = h->d_meshes; ea.n_ins = (int)h->ins.size(); ea.items = h->items.p;
    ea.frontier0 = h->frontier[0].p; ea.fcount = fcount; ea.acc = h->acc.p; ea.icnt = h->icnt.p;
    ea.status = h->status.p;
    hipLaunchKernelGGL(k_setup_items, dim3(grid_for(n_items, 64, 1 << 20)), dim3(64), 0, st, ea);
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[EV_SETUP], st));

    // broadphase: a few level-synchronous expansions to get enough independent seed pairs, then the per-wave
    // depth-first kernel for everything below
    const int L = bfs_levels_for(h, n_items, levels);
    for (int lv = 0; lv < L; ++lv) {
        BpArgs b;
        b.items = h->items.p; b.fin = h->frontier[lv & 1].p; b.fout = h->frontier[(lv + 1) & 1].p;
        b.cand = h->cand.p; b.fcount = fcount; b.ccount = ccount; b.icnt = h->icnt.p; b.status = h->status.p;
        b.level = lv; b.fcap = (int)h->fcap; b.ccap = (int)h->ccap; b.n_items = n_items;
        // upper bound of this level's frontier: n_items * 4^lv, capped by the buffer
        double ub = (double)n_items * std::pow(4.0, (double)(lv < 15 ? lv : 15));
        size_t bound = ub > (double)h->fcap ? h->fcap : (size_t)ub;
        hipLaunchKernelGGL(k_bp_expand, dim3(grid_for(bound, 256, 2048)), dim3(256), 0, st, b);
    }
    {
        double ub = (double)n_items * std::pow(4.0, (double)(L < 15 ? L : 15));
        size_t bound = ub > (double)h->fcap ? h->fcap : (size_t)ub;
        DfsArgs d;
        d.items = h->items.p; d.cand = h->cand.p; d.ccount = ccount; d.ccap = (int)h->ccap; d.icnt = h->icnt.p;
        d.status = h->status.p; d.reserve = 3 * h->max_levels + 3; d.stamps = h->stamps.p; d.n_items = n_items;
        if (h->opt_no_filter) {
            // Float64-only traversal (A/B checks)
            d.seeds = h->frontier[L & 1].p; d.n_seed = fcount + L; d.seed_cap = (int)h->fcap; d.next_seed = next_seed;
            d.no_filter = 1;
            hipLaunchKernelGGL(k_bp_dfs, dim3(grid_for(bound, 1, 256 * 8)), dim3(64), 0, st, d);
        } else {
            Dfs32Args f;
            f.items = h->items.p; f.seeds = h->frontier[L & 1].p; f.n_seed = fcount + L; f.next_seed = next_seed;
            f.seed_cap = (int)h->fcap; f.cand = h->cand.p; f.ccount = ccount; f.ccap = (int)h->ccap;
            f.ucount = ucount; f.icnt = h->icnt.p; f.status = h->status.p; f.stamps = h->stamps.p;
            f.reserve = 3 * h->max_levels + 3; f.n_items = n_items;
            // Workgroups of 128 threads for the big launches (>= 3 072 items, i.e. the halves of a step of >= 6 144): the
            // kernel alone runs as fast either way (2.08 ms), but next to the other half's narrowphase the finer grain
            // shares the CUs better (8 192-pose step 4.53 -> 4.39 ms; 4 096: 2.33 vs 2.38, 2 048: 1.30 vs 1.45 -- a smaller
            // launch needs the 256 pairs per iteration; profiles/r02_sweep_bp_block.txt).  Grid = resident workgroups.
            const int blk = bp_block_for(h, n_items);
            if (blk == 128 && f.reserve <= 128 * 10 - 512)
                hipLaunchKernelGGL((k_bp_dfs32<128>), dim3(grid_for(bound, 1, 256 * 8)), dim3(128), 0, st, f);
            else if (blk == 512)
                hipLaunchKernelGGL((k_bp_dfs32<512>), dim3(grid_for(bound, 1, 256 * 2)), dim3(512), 0, st, f);
            else
                hipLaunchKernelGGL((k_bp_dfs32<kDfsBlock>), dim3(grid_for(bound, 1, 256 * 6)), dim3(kDfsBlock), 0, st, f);
        }
    }
    if (h->bp_dev)      // the broadphase ran on a pose of its own: the item records get the evaluation's x_r1_r2 back
        hipLaunchKernelGGL(k_repose, dim3(grid_for(n_items, 64, 1 << 20)), dim3(64), 0, st, n_items, d_pose, h->items.p);
    if (h->opt_fixed_order) {
        int ba, bb;
        const int bits = pfc_sort_key_bits(n_items, h->max_elem1, h->max_elem2, &ba, &bb);
        if (bits > 64)
            return fail(h, PFC_ERR_BAD_ARG, "option fixed_order: (item, element, element) needs %d key bits for %d items, more than 64", bits, n_items);
        size_t cover = h->sort_cover ? h->sort_cover : h->ccap;
        if (cover > h->ccap) cover = h->ccap;
        h->sort_cover_used = cover;
        // by segments (the per-item counts of the broadphase give the offsets; only the segments are sorted), or -- A/B,
        // PFC_FIXED_GLOBAL_SORT=1 -- by one sort of the covered part of the list: the same order either way
        static const bool global_sort = std::getenv("PFC_FIXED_GLOBAL_SORT") != nullptr;
        if (global_sort || h->fixed_whole_list)
            HIP_TRY(h, pfc_sort_candidates(h->cand.p, ccount, cover, h->sort_keys[0].p, h->sort_keys[1].p, h->sort_tmp.p, h->sort_tmp.cap,
                                           n_items, ba, bb, bits, h->status.p, kStFixedCover, st));
        else
            HIP_TRY(h, pfc_canon_candidates(h->cand.p, ccount, cover, h->icnt.p, n_items, h->sort_keys[0].p, h->sort_keys[1].p, h->canon_off.p,
                                            h->canon_fill.p, h->canon_item.p, bb, h->status.p, kStFixedCover, kStFixedBig, st));
        hipLaunchKernelGGL(k_fixed_init, dim3(grid_for(n_items, 256, 1 << 20)), dim3(256), 0, st, n_items, h->det.p);
    }
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[EV_BP], st));

    NpArgs np;
    np.items = h->items.p; np.cand = h->cand.p; np.ccount = ccount; np.ccap = (int)h->ccap; np.acc = h->acc.p;
    np.icnt = h->icnt.p; np.n_items = n_items; np.clip_n = h->opt_debug ? h->clip_n.p : nullptr; np.trac = trac_view(h);
    np.tcount = tcount; np.tcap = (int)h->tcap; np.status = h->status.p; np.debug = h->opt_debug;
    np.stamps = h->stamps.p;
    np.rec = h->rec.p; np.rgn = h->rgn.p; np.rr_cap = (int)(h->rcap / kRgn);
    np.poly_item = h->poly_item.p; np.poly = h->poly.p; np.pcap = (int)poly_cap(h->ccap);
    np.pcnt = h->pcnt.p; np.chunk_switch = kNpChunkSwitch; np.poly_cand = h->want_surv ? h->poly_cand.p : nullptr;
    np.surv = h->want_surv ? h->surv.p : nullptr; np.scount = pcount + 1;
    const int np_grid = grid_for(h->ccap, kNpBlock, kNpMaxBlocks);
    // (fixed_order: always the clip-only kernel + k_integ_fixed, whose run records carry every sum)
    const int np_mode = h->opt_debug ? 1 : ((h->opt_fixed_order || (h->opt_clip_min > 0 && n_items >= h->opt_clip_min)) ? 2 : 0);
    if (np_mode == 1) {
        if (h->any_tet_tet) hipLaunchKernelGGL((k_narrow<true, 1>), dim3(np_grid), dim3(kNpBlock), 0, st, np);
        else hipLaunchKernelGGL((k_narrow<false, 1>), dim3(np_grid), dim3(kNpBlock), 0, st, np);
    } else if (np_mode == 0) {
        if (h->any_tet_tet) hipLaunchKernelGGL((k_narrow<true, 0>), dim3(np_grid), dim3(kNpBlock), 0, st, np);
        else hipLaunchKernelGGL((k_narrow<false, 0>), dim3(np_grid), dim3(kNpBlock), 0, st, np);
    } else {
        // a half of a two-half evaluation clips on the compacted ring (MODE 3, 12 KiB of LDS per wave: shares the CUs with the
        // other half's kernels), a launch that has the chip to itself on a column per lane (MODE 2)
        // (and a launch that does not quite fill the chip: 768 poses 0.72 vs 0.74 ms; 512 poses the other way, 0.63 vs 0.62)
        // (end of round 3, with the kernel's earlier gather: every clip-only tri-tet launch -- 900 poses 802 -> 767 us, 1 000:
        // 858 -> 824, 520 ... 800 poses equal within 1 %; scripts/sweep_clip_queue_min.py.  Before: from 1 024 items or as a half.)
        if (h->opt_clip_queue && !h->any_tet_tet) {
            // big tri-tet launches: survivors of the trivial reject queued in the ring, clipped 64 at a time (8 192-pose step
            // 4.17 -> 4.13 ms as two halves, 4.87 -> 4.70 ms unsplit, 2 048 poses 1.288 -> 1.278; 768 poses lose, 0.712 -> 0.721:
            // paired A/B, profiles/r03_ab_clip_queue.txt; option value 2 forces it for every clip-only launch: tests)
            hipLaunchKernelGGL(k_clip_queue, dim3(np_grid), dim3(kNpBlock), 0, st, np);
        } else if (h->in_split || (n_items >= 640 && n_items < 1024)) {
            if (h->any_tet_tet) hipLaunchKernelGGL((k_narrow<true, 3>), dim3(np_grid), dim3(kNpBlock), 0, st, np);
            else hipLaunchKernelGGL((k_narrow<false, 3>), dim3(np_grid), dim3(kNpBlock), 0, st, np);
        } else {
            if (h->any_tet_tet) hipLaunchKernelGGL((k_narrow<true, 2>), dim3(np_grid), dim3(kNpBlock), 0, st, np);
            else hipLaunchKernelGGL((k_narrow<false, 2>), dim3(np_grid), dim3(kNpBlock), 0, st, np);
        }
        IntegArgs ig;
        ig.items = h->items.p; ig.n_items = n_items; ig.ccount = ccount; ig.ccap = (int)h->ccap; ig.chunk_switch = kNpChunkSwitch;
        ig.pcnt = h->pcnt.p; ig.poly_item = h->poly_item.p; ig.poly = h->poly.p; ig.pcap = np.pcap;
        ig.poly_cand = np.poly_cand; ig.surv = np.surv; ig.scount = np.scount; ig.acc = h->acc.p; ig.rec = h->rec.p;
        ig.rgn = h->rgn.p; ig.rr_cap = np.rr_cap; ig.icnt = h->icnt.p; ig.status = h->status.p;
        ig.det = h->opt_fixed_order ? h->det.p : nullptr;
        if (h->opt_fixed_order) hipLaunchKernelGGL(k_integ_fixed, dim3(grid_for(h->ccap, 64, kNpMaxBlocks)), dim3(64), 0, st, ig);
        else hipLaunchKernelGGL(k_integ, dim3(grid_for(h->ccap, 64, kNpMaxBlocks)), dim3(64), 0, st, ig);
    }
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[EV_NP], st));

    BrArgs br;
    br.items = h->items.p; br.n_items = n_items; br.acc = h->acc.p; br.res = h->res.p; br.icnt = h->icnt.p;
    br.trac = trac_view(h); br.tcount = tcount; br.tcap = (int)h->tcap; br.wrench = d_wrench; br.sdot = d_sdot;
    br.counts = d_counts;
    br.ctr = h->ctr.p; br.n_ctr = levels + 12; br.status = h->status.p; br.tail = h->tail_dev ? h->tail_dev : h->tail.p;
    br.rgn = h->rgn.p; br.i_pcount = (levels + 9) & ~1;
    if (h->opt_fixed_order && np_mode == 2) {
        FixedArgs fx;
        fx.rec = h->rec.p; fx.det = h->det.p; fx.n_items = n_items; fx.n_slots = (int)(h->rcap / kRgn) * kRgn; fx.items = h->items.p;
        fx.acc = h->acc.p; fx.status = h->status.p;
        hipLaunchKernelGGL(k_shift_fixed, dim3(n_items), dim3(64), 0, st, fx);      // one wave per item
    }
    if (h->any_bristle) {
        ShiftArgs sh;
        sh.rec = h->rec.p; sh.rgn = h->rgn.p; sh.rr_cap = (int)(h->rcap / kRgn); sh.acc = h->acc.p;
        if (!(h->opt_fixed_order && np_mode == 2)) hipLaunchKernelGGL(k_shift, dim3(grid_for(h->rcap, 64, 2048)), dim3(64), 0, st, sh);
        hipLaunchKernelGGL(k_eig, dim3(n_items), dim3(64), 0, st, br);   // one wave per item
        FricArgs fr;
        fr.items = h->items.p; fr.poly_item = h->poly_item.p; fr.poly = h->poly.p; fr.ccount = ccount; fr.ccap = (int)h->ccap;
        fr.chunk_switch = kNpChunkSwitch; fr.pcnt = h->pcnt.p;
        fr.pcap = np.pcap; fr.res = h->res.p; fr.acc = h->acc.p;
        fr.n_items = n_items; fr.status = h->status.p;
        fr.sink = FixedSink{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, nullptr};
        if (h->opt_fixed_order && np_mode == 2) {
            // records of the friction sums: two direct slots per 64-polygon piece (a piece of a batch holds one item, two at a boundary),
            // an overflow area for the pieces of piles that hold more -- at most one further record per item
            const size_t n_pos = h->ccap / 64 + 1, cap = fixed_fric_cap(h->ccap, n_items);      // (allocated by ensure_work)
            HIP_TRY(h, hipMemsetAsync(h->vfx_head.p, 0xFF, sizeof(int) * (size_t)n_items, st));
            HIP_TRY(h, hipMemsetAsync(h->vfx_head.p + n_items, 0, sizeof(int) * 2, st));
            fr.sink = FixedSink{h->vfx_rec.p, h->vfx_head.p + n_items, h->vfx_head.p, 0, 2, (int)n_pos, (int)(2 * n_pos), (int)cap, h->status.p};
        }
        // (grid cap 256 x 32, twice the other narrowphase kernels': paired A/B 4.32 -> 4.22 ms per 8 192-pose step; x 64 and x 128 alike)
        if (fr.sink.rec) {
            hipLaunchKernelGGL(k_fric_fixed, dim3(grid_for(h->ccap, 64, 256 * 32)), dim3(64), 0, st, fr);
            hipLaunchKernelGGL(k_fixed_reduce, dim3(n_items), dim3(64), 0, st, fr.sink, n_items, h->acc.p, kAccStride, kAccFric, 6);
        } else hipLaunchKernelGGL(k_fric, dim3(grid_for(h->ccap, 64, 256 * 32)), dim3(64), 0, st, fr);
    }
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[EV_BR], st));
    hipLaunchKernelGGL(k_final, dim3(grid_for(n_items, 128, 1 << 20)), dim3(128), 0, st, br);
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[EV_FIN], st));
    HIP_TRY(h, hipGetLastError());
    return PFC_OK;
}

// Enqueue one evaluation.  All pointers are device pointers.  The fixed launch sequence (2 memsets + 8..20 small
// kernels) is captured into a hipGraph the first time a shape is seen and replayed afterwards: for the reference's
// own scene sizes (a handful of instructions per calcXd!) launch overhead, not kernel time, is what an evaluation
// costs.  Profiling (HIP events between stages) uses the eager path.
int enqueue_eval(pfc_context *h, int n_items, const int *d_ins_ids, const double *d_pose, const double *d_twist,
                 const double *d_s, double *d_wrench, double *d_sdot, int *d_counts, hipStream_t st) {
    h->dual_reuse_ok = false; h->pending_more = false;     // a new value pass overwrites what a Dual evaluation could reuse
    ++h->value_serial;
    h->hyb_reuse_ok = false; h->small_reuse_ok = false;
    h->last_dual_reused = false;
    HIP_TRY(h, ensure_work(h, n_items));
    const int levels = eff_levels(h);
    const bool prof = h->opt_profile != 0;
    if (prof && !h->ev[0])
        for (int k = 0; k < EV_COUNT; ++k) HIP_TRY(h, hipEventCreate(&h->ev[k]));
    const int L = bfs_levels_for(h, n_items, levels);
    bool use_graph = h->opt_graph && !prof;
#ifdef PFC_STAMPS
    use_graph = false;
#endif
    if (use_graph) {
        pfc_context::GraphKey key = {};
        key.n_items = n_items; key.levels = levels; key.L = L; key.debug = (h->opt_debug ? 1 : 0) | (h->in_split ? 2 : 0) | (bp_block_for(h, n_items) << 4);
        if (h->opt_fixed_order) {      // (the sort's launches depend on the slots it covers: 2^k, k in the key)
            size_t cover = h->sort_cover ? h->sort_cover : h->ccap;
            int lg = 0;
            while (((size_t)1 << lg) < cover && lg < 40) ++lg;
            key.debug |= ((lg + 1) << 16) | (h->fixed_whole_list ? 1 << 24 : 0);
        }
        key.bristle = (h->any_bristle ? 1 : 0) | (h->any_tet_tet ? 2 : 0);
        key.surv = h->want_surv ? 1 : 0;
        key.p[0] = d_ins_ids; key.p[1] = d_pose; key.p[2] = d_twist; key.p[3] = d_s; key.p[4] = d_wrench;
        key.p[5] = d_sdot; key.p[6] = d_counts; key.p[7] = h->tail_dev; key.p[8] = h->bp_dev; key.stream = (void *)st; key.epoch = h->epoch;
        const int gi = key.surv;   // Radau alternates value and Dual evaluations: both graphs stay instantiated
        if (!h->ghave[gi] || std::memcmp(&key, &h->gkey[gi], sizeof key) != 0) {
            if (h->gexec[gi]) { (void)hipGraphExecDestroy(h->gexec[gi]); h->gexec[gi] = nullptr; }
            h->ghave[gi] = false;
            hipGraph_t graph = nullptr;
            HIP_TRY(h, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            int rc = record_eval(h, n_items, d_ins_ids, d_pose, d_twist, d_s, d_wrench, d_sdot, d_counts, st, false);
            hipError_t e = hipStreamEndCapture(st, &graph);
            if (rc != PFC_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            if (e != hipSuccess) return fail(h, PFC_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
            e = hipGraphInstantiate(&h->gexec[gi], graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (e != hipSuccess) return fail(h, PFC_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
            h->gkey[gi] = key; h->ghave[gi] = true;
        }
        HIP_TRY(h, hipGraphLaunch(h->gexec[gi], st));
    } else {
        int rc = record_eval(h, n_items, d_ins_ids, d_pose, d_twist, d_s, d_wrench, d_sdot, d_counts, st, prof);
        if (rc != PFC_OK) return rc;
    }
    h->last_bfs_levels = L;
    h->last_n_items = n_items; h->last_levels = levels; h->pending = true; h->last_stream = st;
    h->ev_valid = prof;
    return PFC_OK;
}

// Synchronise, read counters, grow on overflow.
int check_one(pfc_context *h) {
    if (!h->pending) return PFC_OK;
    const int levels = h->last_levels;
    const size_t n_tail = (size_t)levels + 12 + 12;
    const int *tail = h->h_tail;
    if (h->tail_host) {     // pfc_eval: one D2H copy carries the tail and the outputs
        tail = h->tail_host; h->tail_host = nullptr;
    } else {
        HIP_TRY(h, hipMemcpyAsync(h->h_tail, h->tail.p, sizeof(int) * n_tail, hipMemcpyDeviceToHost, h->last_stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->last_stream));
    h->pending = false;
    const unsigned status = (unsigned)tail[0];
    const unsigned long long *tot = reinterpret_cast<const unsigned long long *>(tail + 4);
    const int *ctr = tail + 12;
    long long fpeak = 0;
    int used_levels = 0;
    for (int lv = 0; lv <= h->last_bfs_levels && lv <= levels; ++lv) {
        if (ctr[6 + lv] > fpeak) fpeak = ctr[6 + lv];
        if (ctr[6 + lv] > 0) used_levels = lv + 1;
    }
    h->stats[1] = ctr[0]; h->last_tslots = ctr[1]; h->stats[4] = used_levels; h->stats[5] = fpeak;
    h->stats[6] = status; h->stats[7] = h->last_n_items;
    if (status & kStBadIns) return fail(h, PFC_ERR_BAD_ARG, "instruction id out of range in ins_ids");
    h->last_undecided = ctr[4];
    // before the overflow branch: also the narrowphase of an overflowing (to be re-issued) evaluation must only see
    // slots that were written
    if (status & kStHole) return fail(h, PFC_ERR_STATE, "internal error: a work-list slot was read before it was written");
    if (status & (kStFrontierOvf | kStCandOvf | kStTracOvf | kStRecOvf)) {
        // VectorCache-style growth (src/obb/vector_cache.jl:13-17): at least double, at least the observed need
        if (status & kStFrontierOvf) { size_t f = h->fcap * 2; while (f < (size_t)fpeak) f *= 2; h->fcap = f; }
        if (status & kStCandOvf) { size_t c = h->ccap * 2; while (c < (size_t)ctr[0]) c *= 2; h->ccap = c; }
        if (status & kStTracOvf) { size_t t = h->tcap * 2; while (t < (size_t)ctr[1]) t *= 2; h->tcap = t; }
        if (status & kStRecOvf) { size_t r = h->rcap * 2; while (r < (size_t)ctr[3]) r *= 2; h->rcap = r; }
        // the lists are indexed by 32-bit integers (counters, slots): an evaluation that needs more is split by the caller
        if (h->fcap > ((size_t)1 << 30) || h->ccap > ((size_t)1 << 30) || h->tcap > ((size_t)1 << 30) || h->rcap > ((size_t)1 << 30))
            return fail(h, PFC_ERR_NOMEM, "work lists beyond 2^30 entries (frontier %zu, candidates %zu, tractions %zu, records %zu): evaluate the batch in parts",
                        h->fcap, h->ccap, h->tcap, h->rcap);
        return fail(h, PFC_ERR_OVERFLOW, "work list overflow (status %u): capacities grown to frontier %zu, candidates %zu, tractions %zu, records %zu",
                    status, h->fcap, h->ccap, h->tcap, h->rcap);
    }
    if (status & kStAbort) return fail(h, PFC_ERR_STATE, "broadphase aborted: iteration guard hit (corrupt tree?)");
    if (status & kStPolyOvf) return fail(h, PFC_ERR_STATE, "internal error: a kept-polygon region overflowed");
    if (h->opt_fixed_order) {
        if (status & kStFixedBig) {      // an item with more candidates than the segment sort takes: sort the whole list, evaluate again
            h->fixed_whole_list = true;
            h->ghave[0] = h->ghave[1] = false;
            return fail(h, PFC_ERR_OVERFLOW, "option fixed_order: an item has more than 4096 candidates: the whole list is sorted from now on, re-issue");
        }
        if (status & kStFixedCover) {      // more candidates than the sort covered: cover the whole list and evaluate again
            h->sort_cover = 0;
            h->ghave[0] = h->ghave[1] = false;
            return fail(h, PFC_ERR_OVERFLOW, "option fixed_order: the candidate list outgrew the part the sort covered (%zu slots): re-issue", h->sort_cover_used);
        }
        size_t want = 1024;
        while (want < 2 * (size_t)ctr[0] + 1024) want *= 2;
        if (want >= h->ccap) want = 0;
        if (want != h->sort_cover && (want == 0 || h->sort_cover == 0 || want > h->sort_cover || 4 * want <= h->sort_cover)) h->sort_cover = want;
    }
    if (status & (kStFixedSpan | kStFixedList))
        return fail(h, PFC_ERR_STATE, "option fixed_order: the candidates of one item span more than %d chunks of 512, or one key has more than %d sum records (status %u): evaluate it without the option",
                    kFixedSpan, kSinkSpan, status);

    if (status & kStNonFinite) return fail(h, PFC_ERR_NONFINITE, "Non-finite vertex likely");
    h->stats[0] = (long long)tot[0]; h->stats[2] = (long long)tot[1]; h->stats[3] = (long long)tot[2];
    h->last_active = (long long)tot[3];
    return PFC_OK;
}

// Synchronise the pending evaluation (both halves of a split one) and merge the counters.
int check_fused(pfc_context *h);
void team_release(pfc_context *h);
int check_eval(pfc_context *h) {
    if (h->pending_fused) return check_fused(h);
    h->last_fused = false;
    team_release(h);
    if (!h->split_n0) {
        h->last_parts = 1;
        const int rc0 = check_one(h);
        if (rc0 == PFC_OK) h->note_shape(h->last_n_items, h->last_active * 4 <= (long long)h->last_n_items);
        return rc0;
    }
    h->last_parts = 2;
    pfc_context *t = h->twin;
    const int rc1 = check_one(h), rc2 = check_one(t);   // both always run: each grows its own work lists on overflow
    h->split_n0 = 0;
    if (rc1 != PFC_OK) return rc1;
    if (rc2 != PFC_OK) { h->err = t->err; return rc2; }
    for (int k = 0; k < 4; ++k) h->stats[k] += t->stats[k];
    if (t->stats[4] > h->stats[4]) h->stats[4] = t->stats[4];
    if (t->stats[5] > h->stats[5]) h->stats[5] = t->stats[5];
    h->stats[6] |= t->stats[6];
    h->stats[7] += t->stats[7];
    h->last_undecided += t->last_undecided;
    h->note_shape((int)h->stats[7], (h->last_active + t->last_active) * 4 <= h->stats[7]);
    return PFC_OK;
}

// ---- fused small-scene path (pfc_fused.h) -------------------------------------------------------------------------
constexpr int kFusedMaxItems = 256;      // one workgroup (one CU) per item
constexpr int kFusedMaxLeaves = 6144;    // n_leaf(mesh_1) + n_leaf(mesh_2): above this one workgroup's descent is the slower one

// Teams of workgroups wait for each other inside a launch, so two team launches that are each only PARTLY resident can keep
// each other's missing workgroups off the chip until the bounded spin (~65 ms, kTeamSpinMax) gives up.  Within one process
// that cannot happen any more: a device has ONE team slot, held by the handle whose team evaluation is in flight (from the
// enqueue to the check); a handle that finds the slot taken evaluates without a team at once (one workgroup per item, or the
// batched path).  Kernels that do not wait for anybody (every other kernel of the library) only delay a team, they finish.
// Across processes the bounded spin stays the safety net.
std::atomic<pfc_context *> g_team_slot[64];
bool team_acquire(pfc_context *h) {
    if (h->team_owner) return true;
    pfc_context *expect = nullptr;
    if (!g_team_slot[h->device & 63].compare_exchange_strong(expect, h)) return false;
    h->team_owner = true;
    return true;
}
void team_release(pfc_context *h) {
    if (!h->team_owner) return;
    h->team_owner = false;
    g_team_slot[h->device & 63].store(nullptr);
}

bool fused_ok(const pfc_context *h, int n_items) {
    return h->opt_fused && !h->opt_fixed_order && h->fused_skip == 0 && !h->opt_debug && !h->opt_profile && !h->want_surv &&
           !h->is_twin && h->d_insfull && n_items <= kFusedMaxItems && h->max_leaves <= kFusedMaxLeaves;
}
// Team size for an evaluation of a few BIG pairs (more leaves than one workgroup takes: BASELINE config 3 as written is one
// 9 680-tet x 5 120-triangle pair): as many workgroups per item as keep every workgroup of the launch resident (one per CU),
// at most kTeamMaxWg; 0: not a team evaluation.
// Small scenes (fused_ok): 1 -- unless the items are few and mid-sized (a 972-tet box on the ground keeps ONE CU busy for
// 52 us, a third of it dealing 1 100 fan triangles out to 256 threads): then a small team, one workgroup per 128 leaves of
// the pair, at most 32 (C2 64 -> 48 us, four reduced C3 poses 106 -> 81 us; a team of 2 loses to its own team sum).
int fused_team(const pfc_context *h, int n_items) {
    const int blocks = h->n_cu < kTeamMaxBlocks ? h->n_cu : kTeamMaxBlocks;      // (a partitioned device has fewer CUs)
    if (fused_ok(h, n_items)) {
        // (end of round 3, scripts/variants/team_cap_run.py: a workgroup per 128 leaves, at most 32 -- a 4 880-leaf pair alone
        // 94 -> 82 us, four of them 132 -> 86, sixteen 148 -> 94; 2 000-leaf pairs 84 -> 79 (x 4), 88 -> 80 (x 16); C2 unchanged)
        int nw = (h->max_leaves + 64) / 128;
        if (nw > 32) nw = 32;
        if (n_items >= 1 && nw > blocks / n_items) nw = blocks / n_items;
        if (nw > h->opt_team) nw = h->opt_team;
        return nw >= 4 ? nw : 1;
    }
    if (!(h->opt_fused && !h->opt_fixed_order && h->opt_team && h->fused_skip == 0 && !h->opt_debug && !h->opt_profile && !h->want_surv && !h->is_twin &&
          h->d_insfull && h->max_leaves > kFusedMaxLeaves && n_items >= 1 && n_items <= kFusedMaxItems))
        return 0;
    int nw = blocks / n_items;
    if (nw > kTeamMaxWg) nw = kTeamMaxWg;
    if (nw > h->opt_team) nw = h->opt_team;
    // A team pays while each of its workgroups has at most ~1 200 leaves of the pair to descend; beyond that the batched path with
    // its 512-thread broadphase workgroups is faster (scripts/team_vs_batched.py, team_rule_sweep.py, profiles/r03_team_rules.txt;
    // full-size C3 poses, 14 800 leaves: 8 in teams of 32 109 us against 172 batched, 16 in teams of 16 166 against 200, 20 in
    // teams of 12 203 against 212, but 24 in teams of 10 283 against 230; a 10 400-leaf pair: 24 in teams of 10 136 against 210;
    // a 7 380-leaf pair: 32 in teams of 8 137 against 188).  Until the end of round 3 the rule was "teams of at least 4", which
    // ran 64 full-size poses in teams of 4: 488 us against 289.
    return (nw >= 4 && h->max_leaves <= 1200 * nw) ? nw : 0;
}

int enqueue_fused(pfc_context *h, int n_items, const int *d_ins_ids, const double *d_pose, const double *d_twist,
                  const double *d_s, double *d_wrench, double *d_sdot, int *d_counts, hipStream_t st) {
    h->dual_reuse_ok = false; h->pending_more = false; h->last_dual_reused = false; h->hyb_reuse_ok = false; h->small_reuse_ok = false;
    ++h->value_serial;
    FuArgs a;
    a.n_items = n_items; a.n_ins = (int)h->ins.size(); a.ins_ids = d_ins_ids; a.pose = d_pose; a.twist = d_twist; a.s = d_s;
    a.ins = h->d_insfull; a.wrench = d_wrench; a.sdot = d_sdot; a.counts = d_counts;
    if (h->fout_dev) {
        a.fout = h->fout_dev;
    } else {
        HIP_TRY(h, h->fout.ensure((size_t)kFusedMaxItems * 8));
        if (!h->h_fout) {
            HIP_TRY(h, hipHostMalloc((void **)&h->h_fout, sizeof(int) * kFusedMaxItems * 8));
            h->h_fout_cap = (size_t)kFusedMaxItems * 8;
        }
        a.fout = h->fout.p;
    }
    if (++h->fused_seq == 0) h->fused_seq = 1;
    a.seq = h->fused_seq;
    if (h->fout_host) {
        // polled path: the completion words live in pinned host memory that several block layouts share (offsets depend on
        // n_items and the number of levels), so a stale small integer of an earlier layout could equal this launch's
        // sequence number -- clear this launch's n words before the kernel can write them (<= 256 host stores)
        volatile int *vf = const_cast<volatile int *>(h->fout_host);
        for (int i = 0; i < n_items; ++i) vf[8 * i + 5] = 0;
        std::atomic_thread_fence(std::memory_order_release);
    }
    a.n_dir = h->fu_ndir; a.d_pose = h->fu_dpose; a.d_twist = h->fu_dtwist; a.d_wrench = h->fu_dwrench; a.d_sdot = h->fu_dsdot;
    a.emit_items = nullptr; a.emit_cand = nullptr; a.emit_surv = nullptr; a.emit_icnt = nullptr; a.emit_ctr = nullptr; a.emit_cap = 0;
    if (h->fu_emit) {
        a.emit_items = h->items.p; a.emit_cand = h->cand.p; a.emit_surv = h->surv.p; a.emit_icnt = h->icnt.p;
        a.emit_ctr = h->emit_ctr.p; a.emit_cap = (int)h->ccap;
    }
    a.stamps = nullptr;
#ifdef PFC_STAMPS
    HIP_TRY(h, h->stamps.ensure(16));
    a.stamps = h->stamps.p;
#endif
    a.bp_pose = h->bp_dev; a.team_fault = h->opt_team_fault; a.f32 = h->opt_fused_f32;
    a.nw = h->fu_nw; a.team = nullptr;
    a.team_seeds = h->max_leaves > kFusedMaxLeaves ? kTeamSeedsBig : kTeamSeeds;
    h->last_fu_nw = a.nw;
    if (a.nw > 1) {
        if (!h->team.p) {
            HIP_TRY(h, h->team.ensure((size_t)kTeamMaxBlocks * 3 * 2 * kTeamSlots));
            HIP_TRY(h, hipMemsetAsync(h->team.p, 0, sizeof(unsigned long long) * h->team.cap, st));
        }
        a.team = h->team.p;
        if (h->any_tet_tet) hipLaunchKernelGGL((k_fused<true, true>), dim3(n_items * a.nw), dim3(kFuBlock), 0, st, a);
        else hipLaunchKernelGGL((k_fused<false, true>), dim3(n_items * a.nw), dim3(kFuBlock), 0, st, a);
    } else
    // a direct launch: cheaper than a graph replay
    if (h->any_tet_tet) hipLaunchKernelGGL((k_fused<true>), dim3(n_items), dim3(kFuBlock), 0, st, a);
    else hipLaunchKernelGGL((k_fused<false>), dim3(n_items), dim3(kFuBlock), 0, st, a);
    HIP_TRY(h, hipGetLastError());
    h->last_n_items = n_items; h->pending = true; h->pending_fused = true; h->last_stream = st; h->ev_valid = false;
    h->last_bfs_levels = 0; h->split_n0 = 0;
    return PFC_OK;
}

int check_fused(pfc_context *h) {
    const int n = h->last_n_items;
    const int *fo = h->fout_host;
    if (!fo) {
        HIP_TRY(h, hipMemcpyAsync(h->h_fout, h->fout.p, sizeof(int) * 8 * (size_t)n, hipMemcpyDeviceToHost, h->last_stream));
        fo = h->h_fout;
        HIP_TRY(h, hipStreamSynchronize(h->last_stream));
    } else {
        // The kernel wrote its results straight into pinned host memory; every workgroup ends with a system-scope
        // release and its completion word.  Polling those words returns ~3 us earlier than hipStreamSynchronize wakes
        // up; after ~200 us without completion (a long evaluation, a stalled queue) the stream is synchronised instead.
        const volatile int *vf = fo;
        const int seq = h->fused_seq;
        bool done = false;
        static const bool no_poll = std::getenv("PFC_FUSED_SYNC") != nullptr;     // A/B knob: synchronise instead of polling
        for (int spin = 0; spin < 200000 && !done && !no_poll; ++spin) {
            done = true;
            for (int i = n - 1; i >= 0; --i)
                if (vf[8 * i + 5] != seq) { done = false; break; }
        }
        if (done) std::atomic_thread_fence(std::memory_order_acquire);
        else {
            if (std::getenv("PFC_LOG_POLL")) std::fprintf(stderr, "pfc: completion poll timed out, synchronising\n");
            HIP_TRY(h, hipStreamSynchronize(h->last_stream));
        }
    }
    h->pending = false; h->pending_fused = false; h->last_parts = 1; h->last_fused = true;
    team_release(h);
    unsigned status = 0;
    long long tot[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        status |= (unsigned)fo[8 * i];
        for (int k = 0; k < 4; ++k) tot[k] += fo[8 * i + 1 + k];
    }
    for (int k = 0; k < 4; ++k) h->stats[k] = tot[k];
    h->stats[4] = 0; h->stats[5] = 0; h->stats[6] = status; h->stats[7] = n;
    h->last_undecided = 0; h->last_tslots = 0;
    if (status & kStBadIns) return fail(h, PFC_ERR_BAD_ARG, "instruction id out of range in ins_ids");
    if (status & kStNonFinite) return fail(h, PFC_ERR_NONFINITE, "Non-finite vertex likely");
    if (status & kStAbort) return fail(h, PFC_ERR_STATE, "broadphase aborted: iteration guard hit (corrupt tree?)");
    if (status & kStFusedOvf) {
        h->fused_skip = 64;     // the batched path takes over (and the re-issue), the fused kernel is tried again later
        return fail(h, PFC_ERR_OVERFLOW, "an item has more candidate pairs than the small-scene kernel holds: re-issue (batched path)");
    }
    if (status & kStFusedDualSkip) {
        h->dual_fused_skip = 64;
        return fail(h, PFC_ERR_OVERFLOW, "an item has more (polygon, direction) pairs than the small-scene kernel's Dual passes hold: re-issue (batched Dual path)");
    }
    return PFC_OK;
}

// Do two streams run side by side?  The halves of a big evaluation only overlap if their streams sit on different hardware
// queues of the runtime, which deals its few queues (four by default) out to streams by its own bookkeeping: with an RCCL
// communicator initialised before pfc_create and no collective issued yet -- bench.py under torch.distributed.run, i.e. every
// multi-GPU run -- both streams of a handle came to sit on ONE queue and the 8 192-pose step took 4.85 instead of 3.89 ms
// (scripts/rccl_queue_probe.py; the kernel trace shows one queue).  Creation order, first-use order: neither is a guarantee.
// So the pair is TESTED once per handle: a one-wave kernel on each stream spins for ~60 us and stamps its start and end with
// the constant-rate clock; on different queues the second starts while the first still spins.
__global__ void k_queue_probe(unsigned long long ticks, unsigned long long *out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[0] = t0; out[1] = wall_clock64(); }
}
bool streams_overlap(pfc_context *h, hipStream_t a, hipStream_t b) {
    if (h->stamps.ensure(16) != hipSuccess) return true;      // cannot test: assume the best
    unsigned long long *d = h->stamps.p;
    unsigned long long v[4] = {0, 0, 0, 0};
    if (hipMemsetAsync(d, 0, sizeof v, a) != hipSuccess || hipStreamSynchronize(a) != hipSuccess) return true;
    hipLaunchKernelGGL(k_queue_probe, dim3(1), dim3(64), 0, a, 6000ull, d);          // 100 MHz clock: 60 us
    hipLaunchKernelGGL(k_queue_probe, dim3(1), dim3(64), 0, b, 6000ull, d + 2);
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return true;
    if (hipMemcpyAsync(v, d, sizeof v, hipMemcpyDeviceToHost, a) != hipSuccess || hipStreamSynchronize(a) != hipSuccess) return true;
    (void)hipMemsetAsync(d, 0, sizeof v, a); (void)hipStreamSynchronize(a);
    return v[2] < v[1] && v[0] < v[3];      // each started before the other ended
}

// the second set of work buffers: shares the (immutable) mesh / instruction records of h
int make_twin(pfc_context *h) {
    if (h->twin) return PFC_OK;
    pfc_context *t = new (std::nothrow) pfc_context();
    if (!t) return fail(h, PFC_ERR_NOMEM, "out of host memory");
    t->device = h->device; t->is_twin = true; t->finalized = true;
    t->ins = h->ins; t->d_meshes = h->d_meshes; t->d_ins = h->d_ins; t->max_levels = h->max_levels;
    t->max_leaves = h->max_leaves; t->max_elem1 = h->max_elem1; t->max_elem2 = h->max_elem2;
    t->any_bristle = h->any_bristle; t->any_tet_tet = h->any_tet_tet; t->opt_split_min = 0;
    t->stream = h->twin_stream; h->twin_stream = nullptr;
    if ((!t->stream && hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess) ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join0, hipEventDisableTiming) != hipSuccess) {
        if (t->stream) (void)hipStreamDestroy(t->stream);
        delete t;
        return fail(h, PFC_ERR_HIP, "could not create the second stream");
    }
    // one queue for both streams: a stream of another priority cannot share a hardware queue with a normal one (its price where
    // the plain pair does overlap: 1 % of the step, which is why it is the fallback and not the rule)
    h->twin_queue_fallback = 0;
    if (!std::getenv("PFC_NO_QUEUE_TEST") && !streams_overlap(h, h->stream, t->stream)) {
        int lo = 0, hi = 0;
        hipStream_t s2 = nullptr;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo &&
            hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi) == hipSuccess) {
            (void)hipStreamDestroy(t->stream);
            t->stream = s2;
            h->twin_queue_fallback = streams_overlap(h, h->stream, t->stream) ? 1 : 2;      // 2: still serial (reported, not fatal)
        } else {
            h->twin_queue_fallback = 2;
        }
    }
    h->twin = t;
    return PFC_OK;
}

}  // namespace

// =================================================================================================================
// C ABI
// =================================================================================================================
// The one-graph Dual path replays kernels with the addresses of the dual_* buffers baked in: a reallocation anywhere
// (also by the two-stage path, which sizes them from the value pass) must invalidate that graph, i.e. bump the epoch
// that is part of its key.
template <class T>
hipError_t ensure_dual(pfc_context *h, DevBuf<T> &b, size_t n) {
    const T *p0 = b.p;
    const hipError_t e = b.ensure(n);
    if (b.p != p0) ++h->epoch;
    return e;
}

#include "pfc_multi.h"

extern "C" {

int pfc_version(void) { return PFC_VERSION; }

int pfc_build_info(void) {
    int f = 0;
#ifdef PFC_STAMPS
    f |= 1;
#endif
    f |= (PFC_EXP & 0xFF) << 8;
#ifdef PFC_VARIANT
    f |= 1 << 16;      // built by scripts/mkvar.sh from patched sources (an A/B variant, not the product)
#endif
    return f;
}

int pfc_create(int device, pfc_handle *out) {
    if (!out) return PFC_ERR_BAD_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return PFC_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return PFC_ERR_HIP;
    pfc_context *h = new (std::nothrow) pfc_context();
    if (!h) return PFC_ERR_NOMEM;
    h->device = device;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return PFC_ERR_HIP; }
    if (hipStreamCreateWithFlags(&h->twin_stream, hipStreamNonBlocking) != hipSuccess) h->twin_stream = nullptr;   // (make_twin creates one then)
    { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) h->n_cu = cu; }
    *out = h;
    return PFC_OK;
}

int pfc_create_multi(const int *devices, int n_devices, pfc_handle *out) {
    if (!out) return PFC_ERR_BAD_ARG;
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return PFC_ERR_BAD_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return PFC_ERR_HIP;
    for (int k = 0; k < n_devices; ++k)
        if (devices[k] < 0 || devices[k] >= n) return PFC_ERR_HIP;
    pfc_context *h = new (std::nothrow) pfc_context();
    pfc_multi *M = new (std::nothrow) pfc_multi();
    if (!h || !M) { delete h; delete M; return PFC_ERR_NOMEM; }
    h->device = devices[0];
    h->multi = M;
    M->dev.assign(devices, devices + n_devices);
    M->stage.resize((size_t)n_devices);
    int rc = PFC_OK;
    for (int k = 0; k < n_devices && rc == PFC_OK; ++k) {
        pfc_handle c = nullptr;
        rc = pfc_create(devices[k], &c);
        if (rc != PFC_OK) break;
        M->shard.push_back(c);
        if (hipEventCreateWithFlags(&M->stage[k].done, hipEventDisableTiming) != hipSuccess) rc = PFC_ERR_HIP;
        // direct peer copies between the first device and this one where the hardware allows it (xGMI); failures -- same device,
        // already enabled, no peer path -- leave the staged copy of hipMemcpyPeerAsync
        if (devices[k] != devices[0]) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[k], devices[0]) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(devices[0], 0);
            (void)hipGetLastError();
        }
    }
    if (rc == PFC_OK) {
        (void)hipSetDevice(devices[0]);
        for (int k = 1; k < n_devices; ++k)
            if (devices[k] != devices[0]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devices[0], devices[k]) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(devices[k], 0);
                (void)hipGetLastError();
            }
        if (hipEventCreateWithFlags(&M->ev_fork, hipEventDisableTiming) != hipSuccess) rc = PFC_ERR_HIP;
    }
    if (rc == PFC_OK) {
        for (int k = 1; k < n_devices; ++k) {
            pfc_multi::Worker *w = new (std::nothrow) pfc_multi::Worker();
            if (!w) { rc = PFC_ERR_NOMEM; break; }
            M->workers.push_back(w);
            w->th = std::thread(multi_worker_loop, w, devices[k]);
        }
    }
    if (rc != PFC_OK) { multi_destroy(h); delete h; return rc; }
    *out = h;
    return PFC_OK;
}

int pfc_last_shards(pfc_handle h) { return h ? (h->multi ? h->multi->n_used : 1) : 0; }

void pfc_destroy(pfc_handle h) {
    if (!h) return;
    if (h->multi) { multi_destroy(h); delete h; return; }
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->pending && h->last_stream) (void)hipStreamSynchronize(h->last_stream);   // an unchecked pfc_eval_device on the caller's stream
    team_release(h);
    if (h->pin_bp) (void)hipHostFree(h->pin_bp);
    if (h->twin) { pfc_destroy(h->twin); h->twin = nullptr; }
    if (h->twin_stream) { (void)hipStreamDestroy(h->twin_stream); h->twin_stream = nullptr; }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->ev_join0) (void)hipEventDestroy(h->ev_join0);
    if (h->is_twin) { h->d_meshes = nullptr; h->d_ins = nullptr; h->d_insfull = nullptr; }   // owned by the parent
    if (h->mesh_arena) (void)hipFree(h->mesh_arena);      // the meshes' d_* pointers point into it
    if (h->d_meshes) (void)hipFree(h->d_meshes);
    if (h->d_ins) (void)hipFree(h->d_ins);
    if (h->d_insfull) (void)hipFree(h->d_insfull);
    if (h->h_fout) (void)hipHostFree(h->h_fout);
    if (h->h_emit) (void)hipHostFree(h->h_emit);
    if (h->h_more) (void)hipHostFree(h->h_more);
    h->emit_ctr.release();
    h->fout.release(); h->team.release();
    h->items.release(); h->frontier[0].release(); h->frontier[1].release(); h->cand.release();
    h->clip_n.release(); h->icnt.release(); h->trac_item.release(); h->acc.release(); h->res.release();
    h->fx_rec.release(); h->fx_head.release(); h->vfx_rec.release(); h->vfx_head.release();
    h->canon_off.release(); h->canon_fill.release(); h->canon_item.release();
    h->det.release(); h->sort_keys[0].release(); h->sort_keys[1].release(); h->sort_tmp.release();
    h->trac_d.release(); h->rec.release(); h->ctr.release(); h->status.release(); h->stamps.release();
    h->h_pose.release();
    for (int k = 0; k < EV_COUNT; ++k)
        if (h->ev[k]) (void)hipEventDestroy(h->ev[k]);
    for (int gi = 0; gi < 2; ++gi)
        if (h->gexec[gi]) (void)hipGraphExecDestroy(h->gexec[gi]);
    if (h->dgexec) (void)hipGraphExecDestroy(h->dgexec);
    if (h->h_tail) (void)hipHostFree(h->h_tail);
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->bar_in) (void)hipFree(h->bar_in);
    if (h->bar_din) (void)hipFree(h->bar_din);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    if (h->pin_din) (void)hipHostFree(h->pin_din);
    if (h->pin_dout) (void)hipHostFree(h->pin_dout);
    h->tail.release();
    h->rgn.release(); h->poly_item.release(); h->pcnt.release(); h->poly_cand.release(); h->poly.release(); h->surv.release(); h->scat_d.release(); h->scat_i.release();
    h->dual_poly.release(); h->dual_pkey.release(); h->dual_sel.release(); h->dual_flag.release();
    h->dual_in.release(); h->dual_acc.release(); h->dual_res.release(); h->dual_out.release(); h->dual_zero.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char *pfc_last_error(pfc_handle h) { return h ? h->err.c_str() : "null handle"; }

int pfc_add_mesh(pfc_handle h, int n_pt, const double *xyz, int n_tri, const int *tri, int n_tet, const int *tet,
                 const double *eps, double Ebar, int n_node, const double *node_c, const double *node_e,
                 const double *node_R, const int *node_child, const int *node_leaf) {
    if (!h) return -PFC_ERR_BAD_ARG;
    if (h->multi) {      // replicated on every shard (the ids agree: same sequence of calls)
        int id0 = -1;
        for (size_t k = 0; k < h->multi->shard.size(); ++k) {
            const int id = pfc_add_mesh(h->multi->shard[k], n_pt, xyz, n_tri, tri, n_tet, tet, eps, Ebar, n_node, node_c, node_e, node_R,
                                        node_child, node_leaf);
            if (id < 0) { h->err = h->multi->shard[k]->err; return id; }
            if (k == 0) id0 = id;
        }
        return id0;
    }
    if (h->finalized) return -fail(h, PFC_ERR_STATE, "pfc_add_mesh after pfc_finalize");
    if (n_pt <= 0 || !xyz || n_node <= 0 || !node_c || !node_e || !node_R || !node_child || !node_leaf)
        return -fail(h, PFC_ERR_BAD_ARG, "pfc_add_mesh: null or empty argument");
    const bool has_tri = tri && n_tri > 0, has_tet = tet && n_tet > 0;
    if (has_tri == has_tet)  // verify_eMesh_ContactProperties: src/mechanism_scenario.jl:301-306
        return -fail(h, PFC_ERR_BAD_ARG, "pfc_add_mesh: exactly one of tri / tet must be given");
    if (has_tet && !eps) return -fail(h, PFC_ERR_BAD_ARG, "pfc_add_mesh: tet mesh without eps");
    HostMesh m;
    m.n_pt = n_pt; m.n_tri = has_tri ? n_tri : 0; m.n_tet = has_tet ? n_tet : 0; m.n_node = n_node; m.Ebar = Ebar;
    const int n_elem = has_tri ? n_tri : n_tet;
    for (int k = 0; k < 3 * n_pt; ++k)
        if (!std::isfinite(xyz[k])) return -fail(h, PFC_ERR_NONFINITE, "pfc_add_mesh: non-finite vertex");
    m.xyz.assign(xyz, xyz + 3 * (size_t)n_pt);
    if (has_tri) {
        for (int k = 0; k < 3 * n_tri; ++k)
            if (tri[k] < 0 || tri[k] >= n_pt) return -fail(h, PFC_ERR_BAD_ARG, "pfc_add_mesh: triangle index out of range");
        m.tri.assign(tri, tri + 3 * (size_t)n_tri);
    } else {
        for (int k = 0; k < 4 * n_tet; ++k)
            if (tet[k] < 0 || tet[k] >= n_pt) return -fail(h, PFC_ERR_BAD_ARG, "pfc_add_mesh: tet index out of range");
        m.tet.assign(tet, tet + 4 * (size_t)n_tet);
        m.eps.assign(eps, eps + n_pt);
        // (0.0 < volume(point[tet[k]])) || error("inverted tetrahedron"): src/geometry/mesh.jl:27-29,
        // volume: src/math_kernel/geometry_kernel.jl:22-38
        for (int k = 0; k < n_tet; ++k) {
            const double *a = xyz + 3 * (size_t)tet[4 * k], *b = xyz + 3 * (size_t)tet[4 * k + 1];
            const double *c = xyz + 3 * (size_t)tet[4 * k + 2], *d = xyz + 3 * (size_t)tet[4 * k + 3];
            double V = (b[0] - a[0]) * (c[1] * d[2] - c[2] * d[1]);
            V = std::fma(b[1] - a[1], c[2] * d[0] - c[0] * d[2], V);
            V = std::fma(b[2] - a[2], c[0] * d[1] - c[1] * d[0], V);
            V = std::fma(c[0] - d[0], a[2] * b[1] - a[1] * b[2], V);
            V = std::fma(c[1] - d[1], a[0] * b[2] - a[2] * b[0], V);
            V = std::fma(c[2] - d[2], a[1] * b[0] - a[0] * b[1], V);
            if (!(0.0 < V * (1.0 / 6.0))) return -fail(h, PFC_ERR_INVERTED_TET, "inverted tetrahedron %d", k);
        }
    }
    m.nodes.resize(n_node);
    int n_leaf = 0;
    for (int k = 0; k < n_node; ++k) {
        NodeRec &r = m.nodes[k];
        for (int j = 0; j < 3; ++j) { r.c[j] = node_c[3 * k + j]; r.e[j] = node_e[3 * k + j]; }
        for (int j = 0; j < 9; ++j) r.R[j] = node_R[9 * k + j];
        r.child0 = node_child[2 * k]; r.child1 = node_child[2 * k + 1]; r.leaf = node_leaf[k]; r.pad = 0.0;
        static const double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        r.aabb = std::memcmp(r.R, I9, sizeof I9) == 0 ? 1 : 0;
        if (r.leaf == kInternal) {
            if (r.child0 <= 0 || r.child0 >= n_node || r.child1 <= 0 || r.child1 >= n_node)
                return -fail(h, PFC_ERR_BAD_ARG, "pfc_add_mesh: node %d has a bad child index", k);
        } else {
            if (r.leaf < 0 || r.leaf >= n_elem) return -fail(h, PFC_ERR_BAD_ARG, "pfc_add_mesh: node %d has a bad leaf id", k);
            ++n_leaf;
        }
    }
    m.n_leaf = n_leaf;
    m.depth = tree_depth(m.nodes);
    if (m.depth < 0) return -fail(h, PFC_ERR_BAD_ARG, "pfc_add_mesh: the node array is not a tree");
    // Device node order: breadth-first (the host passes any order with node 0 = root; node indices never leave the
    // library, leaf ids are element indices).  The two children of a node become neighbours (one 128-byte line of
    // NodeF), and the first N nodes are the top levels of the tree, which the small-scene kernel keeps in LDS.
    {
        std::vector<int> order;          // order[new] = old
        order.reserve(n_node);
        order.push_back(0);
        for (size_t q = 0; q < order.size(); ++q) {
            const NodeRec &r = m.nodes[order[q]];
            if (r.leaf == kInternal) { order.push_back(r.child0); order.push_back(r.child1); }
        }
        if ((int)order.size() != n_node) return -fail(h, PFC_ERR_BAD_ARG, "pfc_add_mesh: the node array is not a tree");
        std::vector<int> inv(n_node);
        for (int k = 0; k < n_node; ++k) inv[order[k]] = k;
        std::vector<NodeRec> bfs(n_node);
        for (int k = 0; k < n_node; ++k) {
            bfs[k] = m.nodes[order[k]];
            if (bfs[k].leaf == kInternal) { bfs[k].child0 = inv[bfs[k].child0]; bfs[k].child1 = inv[bfs[k].child1]; }
        }
        m.nodes.swap(bfs);
    }
    // Device encoding: a child link to a LEAF is stored as ~index (negative), so the broadphase knows from the link
    // alone whether the child needs its rotation R (tight-fitted leaf) and can issue every load of an iteration at once.
    for (NodeRec &r : m.nodes)
        if (r.leaf == kInternal) {
            if (m.nodes[r.child0].leaf != kInternal) r.child0 = ~r.child0;
            if (m.nodes[r.child1].leaf != kInternal) r.child1 = ~r.child1;
        }
    // single-precision mirror (k_bp_dfs32): Float64 centre, Float32 extents, rotation as a unit quaternion
    m.nodesf.resize(n_node);
    for (int k = 0; k < n_node; ++k) {
        const NodeRec &r = m.nodes[k];
        NodeF f;
        std::memset(&f, 0, sizeof f);
        for (int j = 0; j < 3; ++j) { f.c[j] = (float)r.c[j]; f.e[j] = (float)r.e[j]; }
        m.cmax = std::fmax(m.cmax, (std::fabs(r.c[0]) + std::fabs(r.c[1])) + std::fabs(r.c[2]));
        const double *R = r.R;   // column-major: R(i,j) = R[i + 3 j]
        double q[4];
        const double tr = R[0] + R[4] + R[8];
        if (tr > 0.0) {
            const double sq = std::sqrt(tr + 1.0) * 2.0;
            q[0] = 0.25 * sq; q[1] = (R[5] - R[7]) / sq; q[2] = (R[6] - R[2]) / sq; q[3] = (R[1] - R[3]) / sq;
        } else if (R[0] > R[4] && R[0] > R[8]) {
            const double sq = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2.0;
            q[0] = (R[5] - R[7]) / sq; q[1] = 0.25 * sq; q[2] = (R[3] + R[1]) / sq; q[3] = (R[6] + R[2]) / sq;
        } else if (R[4] > R[8]) {
            const double sq = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2.0;
            q[0] = (R[6] - R[2]) / sq; q[1] = (R[3] + R[1]) / sq; q[2] = 0.25 * sq; q[3] = (R[7] + R[5]) / sq;
        } else {
            const double sq = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2.0;
            q[0] = (R[1] - R[3]) / sq; q[1] = (R[6] + R[2]) / sq; q[2] = (R[7] + R[5]) / sq; q[3] = 0.25 * sq;
        }
        const double qn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int j = 0; j < 4; ++j) f.q[j] = (float)(q[j] / qn);
        // host check of what the device's error bound relies on (pfc_bp.h, "Error radius E"): in Float64, the Float32
        // quaternion reproduces R to 4 u per entry and |q|^2 is within 2.25 u of 1 (u = 2^-24)
        {
            const double w = f.q[0], x = f.q[1], y = f.q[2], z = f.q[3];
            const double Rq[9] = {1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w),
                                  2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w),
                                  2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)};
            double worst = 0.0;
            for (int j = 0; j < 9; ++j) worst = std::fmax(worst, std::fabs(Rq[j] - R[j]));
            const double n2 = w * w + x * x + y * y + z * z, u = 5.9604644775390625e-8;
            // a node that fails is never decided in single precision: NaN extent -> S = NaN -> exact test (test_pair_f32)
            if (!(std::isfinite(qn) && worst <= 4.0 * u && std::fabs(n2 - 1.0) <= 2.25 * u)) f.e[0] = std::numeric_limits<float>::quiet_NaN();
        }
        if (r.leaf == kInternal) { f.link0 = r.child0; f.link1 = r.child1; }
        else { f.link0 = r.leaf; f.link1 = -1; }
        m.nodesf[k] = f;
    }
    h->meshes.push_back(std::move(m));
    return (int)h->meshes.size() - 1;
}

int pfc_add_instruction(pfc_handle h, int id_1, int id_2, double chi, int n_quad, int model, const double *params) {
    if (!h) return -PFC_ERR_BAD_ARG;
    if (h->multi) {
        int id0 = -1;
        for (size_t k = 0; k < h->multi->shard.size(); ++k) {
            const int id = pfc_add_instruction(h->multi->shard[k], id_1, id_2, chi, n_quad, model, params);
            if (id < 0) { h->err = h->multi->shard[k]->err; return id; }
            if (k == 0) id0 = id;
        }
        return id0;
    }
    if (h->finalized) return -fail(h, PFC_ERR_STATE, "pfc_add_instruction after pfc_finalize");
    const int nm = (int)h->meshes.size();
    if (id_1 < 0 || id_1 >= nm || id_2 < 0 || id_2 >= nm || !params)
        return -fail(h, PFC_ERR_BAD_ARG, "pfc_add_instruction: bad mesh id");
    if (h->meshes[id_2].n_tet == 0)  // id_2 is always a tet mesh: src/mechanism_scenario.jl:402-416
        return -fail(h, PFC_ERR_BAD_ARG, "pfc_add_instruction: id_2 must be a tet mesh");
    if (n_quad < 1 || n_quad > 2)  // src/mechanism_scenario.jl:45
        return -fail(h, PFC_ERR_BAD_ARG, "only quadrature rules 1 and 2 are currently implemented");
    if (model != PFC_REGULARIZED && model != PFC_BRISTLE) return -fail(h, PFC_ERR_BAD_ARG, "unknown friction model");
    InsDev in;
    std::memset(&in, 0, sizeof in);
    in.m1 = id_1; in.m2 = id_2; in.model = model; in.nq = n_quad; in.chi = chi;
    in.mu_s = params[0]; in.mu_d = params[1];
    if (!(in.mu_d <= in.mu_s))  // determine_μs_μd: src/mechanism_scenario.jl:353-356
        return -fail(h, PFC_ERR_BAD_ARG, "something is wrong: mu_d must be <= mu_s");
    if (model == PFC_REGULARIZED) {
        in.v_c = params[2];
    } else {
        in.tau = params[2]; in.k_bar = params[3]; in.magic = params[4];
        if (!(0.0 < in.mu_d)) return -fail(h, PFC_ERR_BAD_ARG, "mu_d cannot be 0 for bristle friction");
    }
    h->ins.push_back(in);
    return (int)h->ins.size() - 1;
}

int pfc_finalize(pfc_handle h) {
    if (!h) return PFC_ERR_BAD_ARG;
    if (h->multi) {      // every device uploads its replica at the same time
        pfc_multi *M = h->multi;
        std::vector<std::function<int()>> jobs(M->shard.size());
        for (size_t k = 0; k < M->shard.size(); ++k) { pfc_context *c = M->shard[k]; jobs[k] = [c]() { return pfc_finalize(c); }; }
        const int rc = multi_run(h, jobs);
        h->finalized = rc == PFC_OK;
        return rc;
    }
    if (h->finalized) return fail(h, PFC_ERR_STATE, "pfc_finalize called twice");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, h->status.ensure(4));
    HIP_TRY(h, hipMemsetAsync(h->status.p, 0, sizeof(unsigned) * 4, h->stream));   // same stream as k_prep_tet below
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    std::vector<MeshDev> md(h->meshes.size());
    {   // one arena for the records of all meshes, the broadphase's single-precision nodes first (its hot set, contiguous)
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        size_t total = 0;
        for (HostMesh &m : h->meshes) total += up(sizeof(NodeF) * m.nodesf.size());
        for (HostMesh &m : h->meshes)
            total += up(sizeof(NodeRec) * m.nodes.size()) + (m.n_tri ? up(sizeof(TriRec) * m.n_tri)
                                                                      : up(sizeof(TetRec) * m.n_tet) + up(sizeof(double) * 4 * m.n_tet));
        if (total) HIP_TRY(h, hipMalloc((void **)&h->mesh_arena, total));
        size_t off = 0;
        for (HostMesh &m : h->meshes) { m.d_nodesf = (NodeF *)(h->mesh_arena + off); off += up(sizeof(NodeF) * m.nodesf.size()); }
        for (HostMesh &m : h->meshes) {
            m.d_nodes = (NodeRec *)(h->mesh_arena + off); off += up(sizeof(NodeRec) * m.nodes.size());
            if (m.n_tri) { m.d_tri = (TriRec *)(h->mesh_arena + off); off += up(sizeof(TriRec) * m.n_tri); }
            else {
                m.d_tet = (TetRec *)(h->mesh_arena + off); off += up(sizeof(TetRec) * m.n_tet);
                m.d_tet_eps = (double *)(h->mesh_arena + off); off += up(sizeof(double) * 4 * m.n_tet);
            }
        }
    }
    for (size_t k = 0; k < h->meshes.size(); ++k) {
        HostMesh &m = h->meshes[k];
        double *d_xyz = nullptr, *d_eps = nullptr;
        int *d_idx = nullptr;
        HIP_TRY(h, copy_sync(h, m.d_nodes, m.nodes.data(), sizeof(NodeRec) * m.nodes.size(), hipMemcpyHostToDevice));
        HIP_TRY(h, copy_sync(h, m.d_nodesf, m.nodesf.data(), sizeof(NodeF) * m.nodesf.size(), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMalloc((void **)&d_xyz, sizeof(double) * m.xyz.size()));
        HIP_TRY(h, copy_sync(h, d_xyz, m.xyz.data(), sizeof(double) * m.xyz.size(), hipMemcpyHostToDevice));
        if (m.n_tri) {
            HIP_TRY(h, hipMalloc((void **)&d_idx, sizeof(int) * m.tri.size()));
            HIP_TRY(h, copy_sync(h, d_idx, m.tri.data(), sizeof(int) * m.tri.size(), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_prep_tri, dim3((m.n_tri + 127) / 128), dim3(128), 0, h->stream, m.n_tri, d_xyz, d_idx, m.d_tri);
        } else {
            HIP_TRY(h, hipMalloc((void **)&d_idx, sizeof(int) * m.tet.size()));
            HIP_TRY(h, copy_sync(h, d_idx, m.tet.data(), sizeof(int) * m.tet.size(), hipMemcpyHostToDevice));
            HIP_TRY(h, hipMalloc((void **)&d_eps, sizeof(double) * m.eps.size()));
            HIP_TRY(h, copy_sync(h, d_eps, m.eps.data(), sizeof(double) * m.eps.size(), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_prep_tet, dim3((m.n_tet + 127) / 128), dim3(128), 0, h->stream, m.n_tet, d_xyz, d_eps, d_idx,
                               m.d_tet, m.d_tet_eps, h->status.p);
        }
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        (void)hipFree(d_xyz); (void)hipFree(d_idx);
        if (d_eps) (void)hipFree(d_eps);
        md[k].nodes = m.d_nodes; md[k].nodesf = m.d_nodesf; md[k].tri = m.d_tri; md[k].tet = m.d_tet; md[k].tet_eps = m.d_tet_eps; md[k].Ebar = m.Ebar; md[k].cmax = m.cmax;
        md[k].n_tri = m.n_tri; md[k].n_tet = m.n_tet; md[k].n_node = m.n_node; md[k].depth = m.depth;
    }
    unsigned status = 0;
    HIP_TRY(h, copy_sync(h, &status, h->status.p, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (status & kStNonFinite) return fail(h, PFC_ERR_NONFINITE, "singular tetrahedron (non-finite zeta transform)");
    if (!md.empty()) {
        HIP_TRY(h, hipMalloc((void **)&h->d_meshes, sizeof(MeshDev) * md.size()));
        HIP_TRY(h, copy_sync(h, h->d_meshes, md.data(), sizeof(MeshDev) * md.size(), hipMemcpyHostToDevice));
    }
    h->max_levels = 1;
    h->max_leaves = 2;
    h->max_elem1 = h->max_elem2 = 1;
    h->any_bristle = false;
    for (const InsDev &in : h->ins) {
        {   // (key widths of the candidate sort, option "fixed_order": a candidate is (element of mesh_1, tet of mesh_2))
            const HostMesh &m1 = h->meshes[in.m1], &m2 = h->meshes[in.m2];
            const int e1 = m1.n_tri ? m1.n_tri : m1.n_tet;
            if (e1 > h->max_elem1) h->max_elem1 = e1;
            if (m2.n_tet > h->max_elem2) h->max_elem2 = m2.n_tet;
        }
        int lv = h->meshes[in.m1].depth + h->meshes[in.m2].depth + 1;
        if (lv > h->max_levels) h->max_levels = lv;
        const int lf = h->meshes[in.m1].n_leaf + h->meshes[in.m2].n_leaf;
        if (lf > h->max_leaves) h->max_leaves = lf;
        if (in.model == PFC_BRISTLE) h->any_bristle = true;
        if (h->meshes[in.m1].n_tri == 0) h->any_tet_tet = true;
    }
    // the depth-first broadphase keeps 3 * levels + 3 stack slots in reserve (k_bp_dfs)
    if (3 * h->max_levels + 3 > kDfsStack - 128 || 3 * h->max_levels + 3 > kDfsStack32 - 1024)
        return fail(h, PFC_ERR_BAD_ARG, "OBB trees too deep (depth sum %d): rebuild them balanced", h->max_levels - 1);
    if (!h->ins.empty()) {
        HIP_TRY(h, hipMalloc((void **)&h->d_ins, sizeof(InsDev) * h->ins.size()));
        HIP_TRY(h, copy_sync(h, h->d_ins, h->ins.data(), sizeof(InsDev) * h->ins.size(), hipMemcpyHostToDevice));
        // one self-contained record per instruction for the fused small-scene kernel
        std::vector<InsFull> full(h->ins.size());
        for (size_t k = 0; k < h->ins.size(); ++k) {
            const InsDev &in = h->ins[k];
            const HostMesh &m1 = h->meshes[in.m1], &m2 = h->meshes[in.m2];
            InsFull f;
            std::memset(&f, 0, sizeof f);
            f.nodes1 = m1.d_nodes; f.nodes2 = m2.d_nodes; f.nf1 = m1.d_nodesf; f.nf2 = m2.d_nodesf;
            f.tri = m1.d_tri; f.tet = m2.d_tet; f.tet1 = m1.d_tri ? nullptr : m1.d_tet; f.eps1 = m1.d_tet_eps; f.eps2 = m2.d_tet_eps;
            f.chi = in.chi; f.Ebar = m2.Ebar; f.Ebar1 = m1.Ebar; f.mu_s = in.mu_s; f.mu_d = in.mu_d; f.v_c = in.v_c;
            f.tau = in.tau; f.k_bar = in.k_bar; f.magic = in.magic;
            f.model = in.model; f.nq = in.nq; f.n_node1 = m1.n_node; f.n_node2 = m2.n_node;
            f.reserve = 3 * (m1.depth + m2.depth + 1) + 3;
            f.cmax12 = m1.cmax + m2.cmax;
            full[k] = f;
        }
        HIP_TRY(h, hipMalloc((void **)&h->d_insfull, sizeof(InsFull) * full.size()));
        HIP_TRY(h, copy_sync(h, h->d_insfull, full.data(), sizeof(InsFull) * full.size(), hipMemcpyHostToDevice));
    }
    h->finalized = true;
    return PFC_OK;
}

// Argument checks shared by every evaluation entry point (host- or device-buffer, value or Dual): what the reference
// would reject with a MethodError / BoundsError before forceAllElasticIntersections! runs.
static int check_eval_args(pfc_context *h, int n_items, const void *ins_ids, const void *pose, const void *twist,
                           const void *s, const void *wrench, const void *sdot) {
    if (!h->finalized) return fail(h, PFC_ERR_STATE, "pfc_eval before pfc_finalize");
    if (n_items < 0) return fail(h, PFC_ERR_BAD_ARG, "negative n_items");
    if (n_items == 0) return PFC_OK;
    if (h->ins.empty()) return fail(h, PFC_ERR_STATE, "no contact instructions");
    if (!pose || !twist || !wrench || !sdot) return fail(h, PFC_ERR_BAD_ARG, "null buffer");
    if (!ins_ids && n_items > (int)h->ins.size())
        return fail(h, PFC_ERR_BAD_ARG, "n_items exceeds the number of instructions and no ins_ids given");
    if (h->any_bristle && !s) return fail(h, PFC_ERR_BAD_ARG, "bristle instructions need the state buffer s");
    return PFC_OK;
}

int pfc_eval_device(pfc_handle h, int n_items, const int *d_ins_ids, const double *d_pose, const double *d_twist,
                    const double *d_s, double *d_wrench, double *d_sdot, int *d_counts, void *stream) {
    if (!h) return PFC_ERR_BAD_ARG;
    if (h->multi)
        return multi_eval_device(h, n_items, 0, d_ins_ids, d_pose, nullptr, d_twist, d_s, nullptr, nullptr, nullptr, d_wrench, d_sdot,
                                 nullptr, nullptr, d_counts, stream);
    { const int rc = check_eval_args(h, n_items, d_ins_ids, d_pose, d_twist, d_s, d_wrench, d_sdot); if (rc != PFC_OK) return rc; }
    if (n_items == 0) { h->pending = false; h->last_n_items = 0; return PFC_OK; }
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    h->split_n0 = 0;
    h->pending_fused = false;
    h->dual_reuse_ok = false; h->pending_more = false;
    int team = fused_team(h, n_items);
    if (team > 1 && !team_acquire(h)) team = fused_ok(h, n_items) ? 1 : 0;     // another handle's teams are in flight on this device
    if (team) {
        h->fu_nw = team;
        const int rc_t = enqueue_fused(h, n_items, d_ins_ids, d_pose, d_twist, d_s, d_wrench, d_sdot, d_counts, st);
        h->fu_nw = 1;
        return rc_t;
    }
    if (h->fused_skip > 0 && n_items <= kFusedMaxItems) --h->fused_skip;
    const bool split = h->opt_split_min > 0 && n_items >= h->opt_split_min && d_ins_ids && !h->opt_debug && !h->opt_fixed_order &&
                       !h->want_surv && !h->is_twin && !pile_mode(h, n_items);
    if (!split) return enqueue_eval(h, n_items, d_ins_ids, d_pose, d_twist, d_s, d_wrench, d_sdot, d_counts, st);
    int rc = make_twin(h);
    if (rc != PFC_OK) return rc;
    pfc_context *t = h->twin;
    t->opt_profile = h->opt_profile; t->opt_max_levels = h->opt_max_levels; t->opt_bfs_levels = h->opt_bfs_levels;
    t->opt_poison = h->opt_poison;
    if (t->opt_clip_min != h->opt_clip_min) { t->opt_clip_min = h->opt_clip_min; t->ghave[0] = t->ghave[1] = false; }
    if (t->opt_clip_queue != h->opt_clip_queue) { t->opt_clip_queue = h->opt_clip_queue; t->ghave[0] = t->ghave[1] = false; }
    t->opt_graph = h->opt_graph;
    if (t->opt_no_filter != h->opt_no_filter) { t->opt_no_filter = h->opt_no_filter; t->ghave[0] = t->ghave[1] = false; }
    const int n0 = n_items / 2, n1 = n_items - n0;      // (55 / 45 is the same within noise, 60 / 40 and 45 / 55 are slower: round 3)
    // Both halves run on the library's OWN two streams, which start when the caller's stream has reached this point and
    // join it again at the end.  (The first half used to run on the caller's stream.  The halves only overlap if their
    // streams sit on different hardware queues, and a stream the caller created may share one with the twin's: a torch
    // pool stream took the 8 192-pose step from 4.1 to 5.1 ms -- more than the unsplit evaluation; scripts/inflight_probe.py.)
    hipStream_t s0 = h->stream;
    HIP_TRY(h, hipEventRecord(h->ev_fork, st));
    if (s0 != st) HIP_TRY(h, hipStreamWaitEvent(s0, h->ev_fork, 0));
    HIP_TRY(h, hipStreamWaitEvent(t->stream, h->ev_fork, 0));
    h->in_split = true; t->in_split = true;
    rc = enqueue_eval(h, n0, d_ins_ids, d_pose, d_twist, d_s, d_wrench, d_sdot, d_counts, s0);
    h->in_split = false;
    if (rc != PFC_OK) return rc;
    rc = enqueue_eval(t, n1, d_ins_ids + n0, d_pose + 24 * (size_t)n0, d_twist + 6 * (size_t)n0,
                      d_s ? d_s + 6 * (size_t)n0 : nullptr, d_wrench + 6 * (size_t)n0, d_sdot + 6 * (size_t)n0,
                      d_counts ? d_counts + 4 * (size_t)n0 : nullptr, t->stream);
    if (rc != PFC_OK) { h->err = t->err; return rc; }
    HIP_TRY(h, hipEventRecord(h->ev_join, t->stream));
    HIP_TRY(h, hipStreamWaitEvent(st, h->ev_join, 0));
    if (s0 != st) {
        HIP_TRY(h, hipEventRecord(h->ev_join0, s0));
        HIP_TRY(h, hipStreamWaitEvent(st, h->ev_join0, 0));
    }
    h->split_n0 = n0;
    return PFC_OK;
}

int pfc_check(pfc_handle h) {
    if (!h) return PFC_ERR_BAD_ARG;
    if (h->multi) return multi_check(h);
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->pending_more) {       // Dual passes on a value pass that was checked before: nothing to read back
        h->pending_more = false;
        HIP_TRY(h, hipStreamSynchronize(h->last_stream));
        if (h->h_more && (h->h_more[0] & kStHole)) {
            h->dual_reuse_ok = false;
            return fail(h, PFC_ERR_STATE, "internal error: a Dual pass on a reused value pass read a work-list slot out of range");
        }
        return PFC_OK;
    }
    if (h->pending_dual_hyb) {
        h->pending_dual_hyb = false; h->pending_dual = false; h->dual_reuse_ok = false;
        const int rch = check_eval(h);            // the fused kernel's per-item words (copied down, stream synchronised)
        if (rch != PFC_OK) { h->dual_dev_hyb_skip = 64; return rch; }
        if (h->stats[6] & kStCandOvf) {
            h->dual_dev_hyb_skip = 64;
            return fail(h, PFC_ERR_OVERFLOW, "hand-over list of the small-scene kernel overflowed: re-issue (batched path)");
        }
        if ((unsigned)h->h_emit[2] & kStHole) return fail(h, PFC_ERR_STATE, "internal error: a work-list slot was read before it was written");
        const long long pairs_h = h->h_emit[0];
        h->dual_hint = pairs_h;
        const int cpw_h = 64 / h->pending_ndir;
        if ((h->any_bristle || h->opt_fixed_order) && (size_t)((pairs_h + cpw_h - 1) / cpw_h) * 64 + 64 > h->pending_dpcap)
            return fail(h, PFC_ERR_OVERFLOW, "Dual evaluation: %lld contributing pairs exceed the speculative polygon capacity: re-issue", pairs_h);
        h->dual_reuse_ok = true; h->dual_reuse_n = h->last_n_items; h->dual_reuse_emit = true;
        return PFC_OK;
    }
    const bool dual = h->pending_dual;
    h->pending_dual = false;
    h->dual_reuse_ok = false;
    const int rc = check_eval(h);
    if (rc != PFC_OK || !dual) return rc;
    // did the kept Dual polygons of pfc_eval_dual_device fit?  (contributing pairs: the counter next to the polygon total
    // in the packed tail, which check_eval has just brought over)
    const long long pairs = h->h_tail[12 + (((h->last_levels + 9) & ~1) + 1)];
    h->dual_hint = pairs;
    const int cpw = 64 / h->pending_ndir;
    if ((h->any_bristle || h->opt_fixed_order) && (size_t)((pairs + cpw - 1) / cpw) * 64 + 64 > h->pending_dpcap)
        return fail(h, PFC_ERR_OVERFLOW, "Dual evaluation: %lld contributing pairs exceed the speculative polygon capacity: re-issue", pairs);
    h->dual_reuse_ok = true; h->dual_reuse_n = h->last_n_items; h->dual_reuse_emit = false;
    return PFC_OK;
}

// BAR-resident input blocks (pfc_context::bar_in): allocated on first use if the device has a large BAR.
constexpr size_t kBarItems = 4096;     // items whose value inputs go there: every evaluation whose kernels read them in place (kInPlaceItems)
constexpr size_t kBarKeys = 4096;      // (item, direction) pairs whose Dual seeds go there; the Dual kernels then read the seeds and write the partials in place
                                       // (48 box-on-plane scenes x 6 directions: a chunk 96 -> 82 us; 256 scenes 278 -> 250, 100 reduced blob/tool poses 318 -> 288:
                                       // scripts/variants/bar_keys_run.py, dual_in_place_limit_run.py); without a BAR block the in-place limit is 512 pairs
static bool bar_ready(pfc_context *h) {
    if (h->bar_state == 0) {
        h->bar_state = -1;
#if defined(__x86_64__)      // the ordering argument (a locked instruction drains the write-combining buffers) is x86's
        int large_bar = 0;
        if (!std::getenv("PFC_NO_BAR_INPUTS") && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, h->device) == hipSuccess &&
            large_bar && hipExtMallocWithFlags(&h->bar_in, kBarItems * (36 * sizeof(double) + sizeof(int)), hipDeviceMallocFinegrained) == hipSuccess) {
            if (hipExtMallocWithFlags(&h->bar_din, kBarKeys * 36 * sizeof(double), hipDeviceMallocFinegrained) == hipSuccess) h->bar_state = 1;
            else { (void)hipFree(h->bar_in); h->bar_in = nullptr; }
        }
#endif
    }
    return h->bar_state == 1;
}
// After the host has filled a BAR-resident block: drain the core's write-combining buffers, so that the stores are posted
// before the doorbell write of the launch that reads them (PCIe keeps posted writes in order).  The runtime's locked
// queue-index update does the same; this makes the ordering independent of it.
static inline void bar_publish() {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    asm volatile("sfence" ::: "memory");
#endif
}
// The device-visible address of a pinned input block the host has just filled: its copy in BAR-resident device memory if it
// fits there (slot 0: value inputs, slot 1: Dual seeds), else the pinned block itself.  The pinned block stays the host's copy
// (the Dual paths compare the next call's inputs with it).  The caller ends its mirrors with one bar_publish().
static void bar_mirror(pfc_context *h, void **dev, const void *pinned, size_t bytes, int slot) {
    if (!bar_ready(h)) return;
    void *dst = slot == 0 ? h->bar_in : h->bar_din;
    const size_t cap = slot == 0 ? kBarItems * (36 * sizeof(double) + sizeof(int)) : kBarKeys * 36 * sizeof(double);
    if (bytes > cap) return;
    std::memcpy(dst, pinned, bytes);
    *dev = dst;
}

// pinned staging block of at least `bytes` (grown with slack, never shrunk)
static hipError_t ensure_pinned(void **p, size_t *cap, size_t bytes) {
    if (*cap >= bytes) return hipSuccess;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr; *cap = 0;
    hipError_t e = hipHostMalloc(p, bytes * 2);
    if (e == hipSuccess) { *cap = bytes * 2; std::memset(*p, 0, bytes * 2); }   // completion words are polled: never start from stale bytes
    if (std::getenv("PFC_LOG_ALLOC")) std::fprintf(stderr, "pfc pinned %p .. %p\n", *p, (void *)((char *)*p + bytes * 2));
    return e;
}

int pfc_eval(pfc_handle h, int n_items, const int *ins_ids, const double *pose, const double *twist,
             const double *s, double *wrench, double *sdot, int *counts) {
    if (!h) return PFC_ERR_BAD_ARG;
    if (h->multi) return multi_eval(h, n_items, ins_ids, pose, twist, s, wrench, sdot, counts);
    if (!h->finalized) return fail(h, PFC_ERR_STATE, "pfc_eval before pfc_finalize");
    if (n_items < 0) return fail(h, PFC_ERR_BAD_ARG, "negative n_items");
    if (n_items == 0) return PFC_OK;
    if (!pose || !twist || !wrench || !sdot) return fail(h, PFC_ERR_BAD_ARG, "null buffer");
    HIP_TRY(h, hipSetDevice(h->device));
    h->pin_in_dual_n = 0; h->pin_din_valid = false;      // the pinned blocks are about to be overwritten
    const size_t n = (size_t)n_items;
    // one pinned block in (pose | twist | s | ins_ids), one pinned block out (tail | wrench | sdot | counts): two async
    // copies around the launch sequence and a single synchronisation.  The device side of the output block lives
    // behind the packed tail in the same buffer, so that the status / counters and the results come back together.
    const size_t in_d = n * 36, in_bytes = in_d * sizeof(double) + n * sizeof(int);
    const size_t out_d = n * 12, out_bytes = out_d * sizeof(double) + n * 4 * sizeof(int);
    const size_t t0 = (((size_t)h->max_levels + 40) + 3) & ~(size_t)3;      // ints in front of the outputs (16-byte multiple)
    const size_t back_bytes = t0 * sizeof(int) + out_bytes;
    HIP_TRY(h, ensure_pinned(&h->pin_in, &h->pin_in_cap, in_bytes));
    // behind the outputs: the per-item words of the fused small-scene kernel (status, counts: 8 ints per item)
    HIP_TRY(h, ensure_pinned(&h->pin_out, &h->pin_out_cap, back_bytes + (n <= (size_t)kFusedMaxItems ? n * 8 * sizeof(int) : 0)));
    HIP_TRY(h, h->h_pose.ensure(in_d + (n + 1) / 2 + 1));      // device mirror of the input block (doubles)
    {
        const size_t cap0 = h->tail.cap;
        HIP_TRY(h, h->tail.ensure(t0 + out_bytes / sizeof(int) + 4));   // only ever grows; ensure_work asks for less
        if (h->tail.cap != cap0) ++h->epoch;                   // captured graphs hold the old address
    }
    // (every evaluation whose kernels read the inputs in place, i.e. up to 512 items: 64 box-on-plane scenes 52.2 -> 48.7 us, C4's
    // 256 66.5 -> 64.5, 128 full-size poses on the batched path 333 -> 328; scripts/variants/bar_items_run.py)
    const bool bar = n <= kBarItems && !h->want_surv && bar_ready(h);
    double *pi = bar ? (double *)h->bar_in : (double *)h->pin_in;
    std::memcpy(pi, pose, sizeof(double) * n * 24);
    std::memcpy(pi + n * 24, twist, sizeof(double) * n * 6);
    if (s) std::memcpy(pi + n * 30, s, sizeof(double) * n * 6); else std::memset(pi + n * 30, 0, sizeof(double) * n * 6);
    if (ins_ids) std::memcpy(pi + in_d, ins_ids, sizeof(int) * n);
    if (bar) bar_publish();
    hipStream_t st = h->stream;
    // A small scene (what a Radau stage evaluates) pays ~5 us per staging copy, more than the kernels spend on the data:
    // there the kernels read the inputs from, and write the results and the tail to, the pinned blocks directly.
    // (Up to 512 items while the inputs sat in host memory -- a kernel's reads over PCIe stop paying beyond that; with BAR-resident
    // inputs up to 4 096: C5 through host buffers 307 -> 282 us, 1 000 box-on-plane scenes 198 -> 178, 600 full-size poses
    // 635 -> 619, 2 048 unchanged; scripts/variants/zero_copy_limit_run.py.)
    const bool zero_copy = n_items <= (bar ? (int)kBarItems : 512) && !h->want_surv;
    double *di = h->h_pose.p, *dout = reinterpret_cast<double *>(h->tail.p + t0);
    if (zero_copy) {
        void *dpi = h->bar_in, *dpo = nullptr;
        if (!bar) HIP_TRY(h, hipHostGetDevicePointer(&dpi, h->pin_in, 0));
        HIP_TRY(h, hipHostGetDevicePointer(&dpo, h->pin_out, 0));
        di = (double *)dpi;
        dout = reinterpret_cast<double *>((int *)dpo + t0);
    } else {
        HIP_TRY(h, hipMemcpyAsync(di, pi, ins_ids ? in_bytes : in_d * sizeof(double), hipMemcpyHostToDevice, st));
    }
    int rc = PFC_OK;
    for (int attempt = 0; attempt < 40; ++attempt) {
        h->tail_dev = zero_copy ? reinterpret_cast<int *>(dout) - t0 : nullptr;
        if (zero_copy && n <= (size_t)kFusedMaxItems) {
            h->fout_dev = reinterpret_cast<int *>(reinterpret_cast<char *>(dout) + out_bytes);
            h->fout_host = reinterpret_cast<const int *>(reinterpret_cast<const char *>(h->pin_out) + back_bytes);
        }
        rc = pfc_eval_device(h, n_items, ins_ids ? (const int *)(di + in_d) : nullptr, di, di + n * 24,
                             s ? di + n * 30 : nullptr, dout, dout + n * 6, (int *)(dout + out_d), st);
        h->tail_dev = nullptr; h->fout_dev = nullptr;
        if (rc != PFC_OK) { h->fout_host = nullptr; return rc; }
        if (!zero_copy && !h->pending_fused) HIP_TRY(h, hipMemcpyAsync(h->pin_out, h->tail.p, back_bytes, hipMemcpyDeviceToHost, st));
        h->tail_host = (const int *)h->pin_out;
        rc = check_eval(h);
        h->tail_host = nullptr; h->fout_host = nullptr;
        if (rc != PFC_ERR_OVERFLOW) break;
    }
    if (rc != PFC_OK) return rc;
    const double *po = reinterpret_cast<const double *>((const int *)h->pin_out + t0);
    std::memcpy(wrench, po, sizeof(double) * n * 6);
    std::memcpy(sdot, po + n * 6, sizeof(double) * n * 6);
    if (counts) std::memcpy(counts, po + out_d, sizeof(int) * n * 4);
    return PFC_OK;
}

namespace {

// The Dual passes of one evaluation on stream st.  tail: the packed tail of the value pass (device-visible), dp/dt/dsd
// and dw/dsdot: seeds and results (device-visible), n_pairs_bound: upper bound of the contributing pairs used to size
// the kept Dual polygons (the kernels take the actual count from the tail and guard against the capacity).
int launch_dual(pfc_context *h, int n_items, int n_dir, const int *tail, const double *dp, const double *dt,
                const double *dsd, double *dw, double *dsdot, size_t n_pairs_bound, hipStream_t st, size_t *dpcap_out,
                bool acc_cleared = false, const int *pair_count = nullptr, unsigned *status_word = nullptr) {
    const size_t nk = (size_t)n_items * n_dir;
    HIP_TRY(h, ensure_dual(h, h->dual_acc, nk * kDaStride));
    HIP_TRY(h, ensure_dual(h, h->dual_res, nk * kDrStride));
    if (!acc_cleared) HIP_TRY(h, hipMemsetAsync(h->dual_acc.p, 0, sizeof(double) * nk * kDaStride, st));
    DualArgs a;
    a.items = h->items.p; a.cand = h->cand.p; a.ccount = tail + 12; a.ccap = (int)h->ccap;   // packed copy of the counters
    a.surv = h->surv.p; a.scount = tail + 12 + (((h->last_levels + 9) & ~1) + 1);
    if (pair_count) a.ccount = a.scount = pair_count;       // hand-over from the fused small-scene kernel: one list, one count
    a.n_items = n_items; a.n_dir = n_dir; a.d_pose = dp; a.d_twist = dt; a.d_s = dsd; a.icnt = h->icnt.p;
    a.dacc = h->dual_acc.p; a.dres = h->dual_res.p; a.d_wrench = dw; a.d_sdot = dsdot;
    a.status = h->status.p;
    // Hand-over: the passes report into the word behind the pair count (cleared and read back with it).  The status word
    // of the batched value pass is read by its k_final only, so a flag raised here would surface in a later evaluation.
    if (pair_count) a.status = reinterpret_cast<unsigned *>(h->emit_ctr.p) + 2;
    if (status_word) a.status = status_word;       // the passes of a reused value pass: a word pfc_check reads back
    const int cpw = 64 / n_dir;
    const int grid = grid_for((n_pairs_bound + cpw - 1) / cpw, 1, 256 * 16);
    const int kgrid = grid_for(nk, 64, 1 << 20);
    const bool tt = h->any_tet_tet;
    // Dual polygons kept between the passes: a wave of k_narrow_dual owns 64 consecutive slots (no slot counter)
    // (option fixed_order: the bound also sizes the record list of the passes' sums, for every model -- the callers compare the pair count
    // with this capacity after their synchronisation and re-issue the passes when it was exceeded)
    const size_t dpcap = (h->any_bristle || h->opt_fixed_order) ? ((n_pairs_bound + cpw - 1) / cpw) * 64 + 64 : 64;   // 64 slots per group of cpw pairs
    HIP_TRY(h, ensure_dual(h, h->dual_poly, dpcap * kDpFields));
    HIP_TRY(h, ensure_dual(h, h->dual_pkey, dpcap));
    a.dpoly = h->dual_poly.p; a.dpoly_key = h->dual_pkey.p; a.dpcap = (long long)dpcap;
    if (dpcap_out) *dpcap_out = dpcap;
    // Scenes of many items: the passes walk only the contributing pairs of items this chunk seeds (pfc_dual.h, k_dual_select).
    // Two short launches; a small scene's chunk is a handful of waves either way and keeps its few launches.
    // (not while the stream is being captured -- the one-graph path of up to 512 items records these launches, and the
    // lists below may have to be allocated; a captured chunk keeps the per-key skip only)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    // (nor under option fixed_order: the selected list is compacted block by block in the order the blocks come by)
    if (!pair_count && n_items >= kDualSelectMin && cap == hipStreamCaptureStatusNone && !h->opt_fixed_order && std::getenv("PFC_NO_SELECT") == nullptr) {
        HIP_TRY(h, ensure_dual(h, h->dual_sel, h->ccap));
        HIP_TRY(h, ensure_dual(h, h->dual_flag, (size_t)n_items + 1));
        int *selcount = h->dual_flag.p + n_items;
        hipLaunchKernelGGL(k_dual_flags, dim3(n_items), dim3(64), 0, st, a, h->dual_flag.p, selcount);
        // (grid from the list's capacity, not from n_pairs_bound: the first evaluation of a handle has no pair count to go by)
        hipLaunchKernelGGL(k_dual_select, dim3(grid_for(h->ccap, 256 * kSelRounds, 256 * 8)), dim3(256), 0, st, a, h->dual_flag.p,
                           h->dual_sel.p, selcount);
        a.surv = h->dual_sel.p; a.scount = selcount;
    }
    // Pass B folded into pass A wherever the value pass was the batched one (its cop is in h->res; DualArgs::vres) and the scene
    // is tri-tet: one read of the kept polygons less.  (Option "dual_fold", default 1; PFC_NO_DUAL_FOLD=1 for A/B runs.)
    static const bool no_fold = std::getenv("PFC_NO_DUAL_FOLD") != nullptr;
    a.vres = (h->any_bristle && !pair_count && !no_fold && !tt && h->opt_dual_fold) ? h->res.p : nullptr;
    // fixed_order: every key of an item decomposes the K of the VALUE pass (one clamp decision per item and evaluation)
    // (PFC_DUAL_VALUE_K=1: the same without the option, A/B.  It was the default for a day: the same six directions as first and as
    // further chunk of one value pass differ by O(1) on the four flat-patch pairs of config 5 without it and by 1e-2 .. 2e-1 with it,
    // and k_dual_eig saves its Jacobi iteration -- C5 further chunk 355 -> 346 us.  But the value pass forms K by two parallel-axis
    // shifts, the Dual passes sum about the cop directly as the reference does, and on flat patches the value pass's noise
    // eigenvalue lands on the other side of the clamp from the Dual oracle's often enough to fail an oracle comparison of d_sdot
    // in one run of four; the Dual sums' never did in some fifty suite runs.  Both are rounding noise; the default stays with
    // the arithmetic that is closer to the reference's.)
    static const bool value_k = std::getenv("PFC_DUAL_VALUE_K") != nullptr;
    a.vres_k = ((h->opt_fixed_order || value_k) && h->any_bristle && !pair_count) ? h->res.p : nullptr;
    static const bool no_stored_v = std::getenv("PFC_NO_DUAL_STORED_V") != nullptr;
    a.stored_v = no_stored_v ? 0 : 1;
    const bool fx = h->opt_fixed_order && !pair_count;
    a.sink_a = FixedSink{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, nullptr};
    a.sink_b = a.sink_a; a.sink_c = a.sink_a;
    if (fx) {
        // the contributing pairs in candidate order (k_integ_fixed appended them piece by piece as its workgroups came by) ...
        if (h->surv_sorted_serial != h->value_serial) {
            // (as many slots as the candidate sort covered: the contributing pairs are a subset of the candidates)
            HIP_TRY(h, pfc_sort_indices(h->surv.p, a.scount, (h->sort_cover_used && h->sort_cover_used < h->ccap) ? h->sort_cover_used : h->ccap,
                                        reinterpret_cast<unsigned *>(h->sort_keys[0].p),
                                        reinterpret_cast<unsigned *>(h->sort_keys[1].p), h->sort_tmp.p, h->sort_tmp.cap, st));
            h->surv_sorted_serial = h->value_serial;
        }
        // ... and the records of passes A and B: per accumulator block at most one per (wave, item, direction), and since an item's
        // pairs are consecutive in the sorted list the (wave, item) incidences are fewer than waves + items
        const size_t n_grp = (n_pairs_bound + cpw - 1) / cpw;
        // Three blocks (passes A, B, C), each with n_dir direct slots per wave position and a shared overflow area for the waves that
        // straddle items: (items - 1) n_dir records at most per block.
        const size_t n_pos = n_grp + 1, direct = n_pos * (size_t)n_dir;
        const size_t cap = 3 * direct + 3 * (size_t)n_dir * (size_t)n_items + 64;
        if (cap > ((size_t)1 << 30)) return fail(h, PFC_ERR_NOMEM, "option fixed_order: %zu Dual sum records: evaluate the batch in parts", cap);
        HIP_TRY(h, ensure_dual(h, h->fx_rec, cap * kSinkStride));
        HIP_TRY(h, ensure_dual(h, h->fx_head, 3 * nk + 2));
        HIP_TRY(h, hipMemsetAsync(h->fx_head.p, 0xFF, sizeof(int) * 3 * nk, st));
        HIP_TRY(h, hipMemsetAsync(h->fx_head.p + 3 * nk, 0, sizeof(int) * 2, st));
        a.sink_a = FixedSink{h->fx_rec.p, h->fx_head.p + 3 * nk, h->fx_head.p, 0, n_dir, (int)n_pos, (int)(3 * direct), (int)cap, a.status};
        a.sink_b = FixedSink{h->fx_rec.p, h->fx_head.p + 3 * nk, h->fx_head.p + nk, (int)direct, n_dir, (int)n_pos, (int)(3 * direct), (int)cap, a.status};
        a.sink_c = FixedSink{h->fx_rec.p, h->fx_head.p + 3 * nk, h->fx_head.p + 2 * nk, (int)(2 * direct), n_dir, (int)n_pos, (int)(3 * direct), (int)cap, a.status};
        if (!tt) a.vres = h->res.p;      // tri-tet: always the folded pass (one kernel carries both blocks)
    }
    if (fx) {
        if (tt) {
            if (dual_pv_stride(n_dir) == 16) hipLaunchKernelGGL((k_narrow_dual<true, 16, false, true>), dim3(grid), dim3(64), dual_lds_bytes(n_dir), st, a);
            else hipLaunchKernelGGL((k_narrow_dual<true, 64, false, true>), dim3(grid), dim3(64), dual_lds_bytes(n_dir), st, a);
        } else {
            if (dual_pv_stride(n_dir) == 16) hipLaunchKernelGGL((k_narrow_dual<false, 16, true, true>), dim3(grid), dim3(64), dual_lds_bytes(n_dir), st, a);
            else hipLaunchKernelGGL((k_narrow_dual<false, 64, true, true>), dim3(grid), dim3(64), dual_lds_bytes(n_dir), st, a);
        }
        hipLaunchKernelGGL(k_fixed_reduce, dim3((unsigned)nk), dim3(64), 0, st, a.sink_a, (int)nk, h->dual_acc.p, kDaStride, kDaA, 20);
        if (!tt) hipLaunchKernelGGL(k_fixed_reduce, dim3((unsigned)nk), dim3(64), 0, st, a.sink_b, (int)nk, h->dual_acc.p, kDaStride, kDaB, 42);
    } else if (a.vres) {
        if (dual_pv_stride(n_dir) == 16) hipLaunchKernelGGL((k_narrow_dual<false, 16, true>), dim3(grid), dim3(64), dual_lds_bytes(n_dir), st, a);
        else hipLaunchKernelGGL((k_narrow_dual<false, 64, true>), dim3(grid), dim3(64), dual_lds_bytes(n_dir), st, a);
    } else if (dual_pv_stride(n_dir) == 16) {
        if (tt) hipLaunchKernelGGL((k_narrow_dual<true, 16>), dim3(grid), dim3(64), dual_lds_bytes(n_dir), st, a);
        else hipLaunchKernelGGL((k_narrow_dual<false, 16>), dim3(grid), dim3(64), dual_lds_bytes(n_dir), st, a);
    } else {
        if (tt) hipLaunchKernelGGL((k_narrow_dual<true, 64>), dim3(grid), dim3(64), dual_lds_bytes(n_dir), st, a);
        else hipLaunchKernelGGL((k_narrow_dual<false, 64>), dim3(grid), dim3(64), dual_lds_bytes(n_dir), st, a);
    }
    // a few dozen kept polygons only (pencil-scale pair: 116 -> 103 us per chunk): with more, the eightfold number of
    // waves costs more in per-key atomics on the same few rows than the shorter walk saves (single C3 pose: 97 -> 113 us)
    a.tri_split = (h->any_bristle && dpcap * 8 <= 16384 && !fx) ? 1 : 0;
    if (h->any_bristle) {
        const int pgrid = grid_for(dpcap * (a.tri_split ? 8 : 1), 64, 256 * 16);
        if (fx && !a.vres) {
            hipLaunchKernelGGL((k_dual_poly<1, true>), dim3(pgrid), dim3(64), 0, st, a);
            hipLaunchKernelGGL(k_fixed_reduce, dim3((unsigned)nk), dim3(64), 0, st, a.sink_b, (int)nk, h->dual_acc.p, kDaStride, kDaB, 42);
        } else if (!a.vres) hipLaunchKernelGGL((k_dual_poly<1>), dim3(pgrid), dim3(64), 0, st, a);
        hipLaunchKernelGGL(k_dual_eig, dim3((unsigned)nk), dim3(64), 0, st, a);   // one wave per (item, direction)
        if (fx) {
            hipLaunchKernelGGL((k_dual_poly<2, true>), dim3(pgrid), dim3(64), 0, st, a);
            hipLaunchKernelGGL(k_fixed_reduce, dim3((unsigned)nk), dim3(64), 0, st, a.sink_c, (int)nk, h->dual_acc.p, kDaStride, kDaC, 12);
        } else hipLaunchKernelGGL((k_dual_poly<2>), dim3(pgrid), dim3(64), 0, st, a);
    }
    hipLaunchKernelGGL(k_dual_final, dim3(kgrid), dim3(64), 0, st, a);
    HIP_TRY(h, hipGetLastError());
    return PFC_OK;
}

// Smallest scenes, all instructions regularized (test/boxes.jl and the like): value AND Dual passes inside the fused
// small-scene kernel -- one launch, results polled from pinned memory.  Returns PFC_ERR_OVERFLOW when an item does not fit
// (too many candidates / polygons): the caller takes the batched Dual path and this one stays off for a while.
int eval_dual_fused(pfc_context *h, int n_items, int n_dir, const int *ins_ids, const double *pose, const double *twist,
                    const double *s, const double *d_pose, const double *d_twist, double *wrench, double *sdot,
                    double *d_wrench, double *d_sdot, int *counts) {
    h->pin_din_valid = false;    // these pinned blocks are about to be reused
    const size_t n = (size_t)n_items, nk = n * n_dir;
    const size_t in_d = n * 36, in_bytes = in_d * sizeof(double) + n * sizeof(int);
    const size_t out_d = n * 12, out_bytes = out_d * sizeof(double) + n * 4 * sizeof(int) + n * 8 * sizeof(int);
    HIP_TRY(h, ensure_pinned(&h->pin_in, &h->pin_in_cap, in_bytes));
    HIP_TRY(h, ensure_pinned(&h->pin_out, &h->pin_out_cap, out_bytes + 64));
    HIP_TRY(h, ensure_pinned(&h->pin_din, &h->pin_din_cap, sizeof(double) * nk * 30));
    HIP_TRY(h, ensure_pinned(&h->pin_dout, &h->pin_dout_cap, sizeof(double) * nk * 12));
    double *pi = (double *)h->pin_in;
    std::memcpy(pi, pose, sizeof(double) * n * 24);
    std::memcpy(pi + n * 24, twist, sizeof(double) * n * 6);
    if (s) std::memcpy(pi + n * 30, s, sizeof(double) * n * 6); else std::memset(pi + n * 30, 0, sizeof(double) * n * 6);
    if (ins_ids) std::memcpy(pi + in_d, ins_ids, sizeof(int) * n);
    double *pdi = (double *)h->pin_din;
    std::memcpy(pdi, d_pose, sizeof(double) * nk * 24);
    std::memcpy(pdi + nk * 24, d_twist, sizeof(double) * nk * 6);
    void *v_in = nullptr, *v_out = nullptr, *v_din = nullptr, *v_dout = nullptr;
    HIP_TRY(h, hipHostGetDevicePointer(&v_in, h->pin_in, 0));
    HIP_TRY(h, hipHostGetDevicePointer(&v_out, h->pin_out, 0));
    HIP_TRY(h, hipHostGetDevicePointer(&v_din, h->pin_din, 0));
    HIP_TRY(h, hipHostGetDevicePointer(&v_dout, h->pin_dout, 0));
    bar_mirror(h, &v_in, h->pin_in, in_bytes, 0);
    bar_mirror(h, &v_din, h->pin_din, sizeof(double) * nk * 30, 1);
    bar_publish();
    double *di = (double *)v_in, *dout = (double *)v_out, *ddi = (double *)v_din, *ddo = (double *)v_dout;
    h->fout_dev = reinterpret_cast<int *>(dout + out_d) + n * 4;
    h->fout_host = reinterpret_cast<const int *>((const double *)h->pin_out + out_d) + n * 4;
    h->fu_ndir = n_dir; h->fu_dpose = ddi; h->fu_dtwist = ddi + nk * 24; h->fu_dwrench = ddo; h->fu_dsdot = ddo + nk * 6;
    int rc = enqueue_fused(h, n_items, ins_ids ? (const int *)(di + in_d) : nullptr, di, di + n * 24, s ? di + n * 30 : nullptr,
                           dout, dout + n * 6, reinterpret_cast<int *>(dout + out_d), h->stream);
    h->fu_ndir = 0; h->fu_dpose = h->fu_dtwist = nullptr; h->fu_dwrench = h->fu_dsdot = nullptr;
    h->fout_dev = nullptr;
    if (rc == PFC_OK) rc = check_eval(h);
    h->fout_host = nullptr;
    if (rc != PFC_OK) return rc;
    const double *po = (const double *)h->pin_out;
    std::memcpy(wrench, po, sizeof(double) * n * 6);
    std::memcpy(sdot, po + n * 6, sizeof(double) * n * 6);
    if (counts) std::memcpy(counts, po + out_d, sizeof(int) * n * 4);
    std::memcpy(d_wrench, h->pin_dout, sizeof(double) * nk * 6);
    std::memcpy(d_sdot, (const double *)h->pin_dout + nk * 6, sizeof(double) * nk * 6);
    h->pin_in_dual_n = n_items; h->pin_in_dual_ids = ins_ids != nullptr;     // a repeat of this point goes to the hybrid path
    return PFC_OK;
}

// Small scenes the in-kernel Dual passes do not take (bristle items, many polygons per item): the fused kernel evaluates
// the values and hands the item records, counters and the list of candidates with a polygon to the batched Dual passes
// (k_narrow_dual ..., many workgroups), enqueued right behind it; one synchronisation.  Returns PFC_ERR_OVERFLOW when an
// item does not fit the fused kernel or the speculative size of the kept Dual polygons fell short (the caller then takes
// the batched paths).
int eval_dual_hybrid(pfc_context *h, int n_items, int n_dir, const int *ins_ids, const double *pose, const double *twist,
                     const double *s, const double *d_pose, const double *d_twist, const double *d_s, double *wrench,
                     double *sdot, double *d_wrench, double *d_sdot, int *counts) {
    h->pin_din_valid = false;    // these pinned blocks are about to be reused
    const size_t n = (size_t)n_items, nk = n * n_dir;
    const size_t in_d = n * 36, in_bytes = in_d * sizeof(double) + n * sizeof(int);
    const size_t out_d = n * 12, out_bytes = out_d * sizeof(double) + n * 4 * sizeof(int) + n * 8 * sizeof(int);
    {
        const void *p0 = h->pin_in, *q0 = h->pin_out;
        HIP_TRY(h, ensure_pinned(&h->pin_in, &h->pin_in_cap, in_bytes));
        HIP_TRY(h, ensure_pinned(&h->pin_out, &h->pin_out_cap, out_bytes + 64));
        if (h->pin_in != p0 || h->pin_out != q0) h->hyb_reuse_ok = false;      // reallocated: the cached blocks are gone
    }
    HIP_TRY(h, ensure_pinned(&h->pin_din, &h->pin_din_cap, sizeof(double) * nk * 36));
    HIP_TRY(h, ensure_pinned(&h->pin_dout, &h->pin_dout_cap, sizeof(double) * nk * 12));
    HIP_TRY(h, h->emit_ctr.ensure(4));
    if (!h->h_emit) HIP_TRY(h, hipHostMalloc((void **)&h->h_emit, sizeof(int) * 4));
    double *pi = (double *)h->pin_in;
    // the chunks of one Jacobian (pfc_eval_dual's large path has the same test): value inputs bitwise equal to those still
    // in the pinned input block -> the lists and item records the fused kernel handed over are reused, only the Dual passes run
    bool same = h->opt_dual_reuse && h->hyb_reuse_ok && h->hyb_reuse_n == n_items && h->hyb_reuse_ids == (ins_ids != nullptr) &&
                std::memcmp(pi, pose, sizeof(double) * n * 24) == 0 && std::memcmp(pi + n * 24, twist, sizeof(double) * n * 6) == 0;
    if (same && s) same = std::memcmp(pi + n * 30, s, sizeof(double) * n * 6) == 0;
    if (same && !s) {
        const double *z = pi + n * 30;
        for (size_t k = 0; k < n * 6 && same; ++k) same = std::memcmp(z + k, "\0\0\0\0\0\0\0\0", 8) == 0;
    }
    if (same && ins_ids) same = std::memcmp(pi + in_d, ins_ids, sizeof(int) * n) == 0;
    if (!same) {
        h->hyb_reuse_ok = false;
        HIP_TRY(h, ensure_work(h, n_items));      // (not before a reuse: option poison refills the work lists there)
        std::memcpy(pi, pose, sizeof(double) * n * 24);
        std::memcpy(pi + n * 24, twist, sizeof(double) * n * 6);
        if (s) std::memcpy(pi + n * 30, s, sizeof(double) * n * 6); else std::memset(pi + n * 30, 0, sizeof(double) * n * 6);
        if (ins_ids) std::memcpy(pi + in_d, ins_ids, sizeof(int) * n);
    }
    double *pdi = (double *)h->pin_din;
    std::memcpy(pdi, d_pose, sizeof(double) * nk * 24);
    std::memcpy(pdi + nk * 24, d_twist, sizeof(double) * nk * 6);
    if (d_s) std::memcpy(pdi + nk * 30, d_s, sizeof(double) * nk * 6); else std::memset(pdi + nk * 30, 0, sizeof(double) * nk * 6);
    void *v_in = nullptr, *v_out = nullptr, *v_din = nullptr, *v_dout = nullptr;
    HIP_TRY(h, hipHostGetDevicePointer(&v_in, h->pin_in, 0));
    HIP_TRY(h, hipHostGetDevicePointer(&v_out, h->pin_out, 0));
    HIP_TRY(h, hipHostGetDevicePointer(&v_din, h->pin_din, 0));
    HIP_TRY(h, hipHostGetDevicePointer(&v_dout, h->pin_dout, 0));
    bar_mirror(h, &v_in, h->pin_in, in_bytes, 0);
    bar_mirror(h, &v_din, h->pin_din, sizeof(double) * nk * 36, 1);
    bar_publish();
    double *di = (double *)v_in, *dout = (double *)v_out;
    hipStream_t st = h->stream;
    // seeds / results of up to kBarKeys (512 without a BAR block) (item, direction) pairs are read / written in place by the kernels, larger ones staged
    const bool zc = nk <= (h->bar_state == 1 ? (int)kBarKeys : 512);
    if (!zc) {
        HIP_TRY(h, ensure_dual(h, h->dual_in, nk * 36));
        HIP_TRY(h, ensure_dual(h, h->dual_out, nk * 12));
        HIP_TRY(h, hipMemcpyAsync(h->dual_in.p, h->pin_din, sizeof(double) * nk * 36, hipMemcpyHostToDevice, st));
    }
    double *ddi = zc ? (double *)v_din : h->dual_in.p, *ddo = zc ? (double *)v_dout : h->dual_out.p;
    if (same) {
        HIP_TRY(h, ensure_dual(h, h->dual_acc, nk * kDaStride));
        HIP_TRY(h, hipMemsetAsync(h->dual_acc.p, 0, sizeof(double) * nk * kDaStride, st));
        const size_t bound_r = (size_t)(h->dual_hint > 0 ? h->dual_hint : 0) + 64;     // known exactly
        int rcr = launch_dual(h, n_items, n_dir, h->tail.p, ddi, ddi + nk * 24, ddi + nk * 30, ddo, ddo + nk * 6, bound_r, st, nullptr,
                              true, h->emit_ctr.p);
        if (rcr != PFC_OK) return rcr;
        if (!zc) HIP_TRY(h, hipMemcpyAsync(h->pin_dout, ddo, sizeof(double) * nk * 12, hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipStreamSynchronize(st));
        const double *po = (const double *)h->pin_out;
        std::memcpy(wrench, po, sizeof(double) * n * 6);
        std::memcpy(sdot, po + n * 6, sizeof(double) * n * 6);
        if (counts) std::memcpy(counts, po + out_d, sizeof(int) * n * 4);
        std::memcpy(d_wrench, h->pin_dout, sizeof(double) * nk * 6);
        std::memcpy(d_sdot, (const double *)h->pin_dout + nk * 6, sizeof(double) * nk * 6);
        h->last_dual_reused = true;
        return PFC_OK;
    }
    size_t bound = 64;
    while (bound < (size_t)(h->dual_hint > 0 ? h->dual_hint : 0) * 2 + 64) bound *= 2;
    HIP_TRY(h, ensure_dual(h, h->dual_acc, nk * kDaStride));
    HIP_TRY(h, hipMemsetAsync(h->dual_acc.p, 0, sizeof(double) * nk * kDaStride, st));
    HIP_TRY(h, hipMemsetAsync(h->emit_ctr.p, 0, sizeof(int) * 4, st));
    h->fout_dev = reinterpret_cast<int *>(dout + out_d) + n * 4;
    h->fout_host = nullptr;           // this path synchronises (the Dual kernels run behind the fused one)
    h->fu_emit = true;
    int rc = enqueue_fused(h, n_items, ins_ids ? (const int *)(di + in_d) : nullptr, di, di + n * 24, s ? di + n * 30 : nullptr,
                           dout, dout + n * 6, reinterpret_cast<int *>(dout + out_d), st);
    h->fu_emit = false; h->fout_dev = nullptr;
    if (rc != PFC_OK) return rc;
    size_t dpcap = 0;
    h->last_levels = eff_levels(h);
    rc = launch_dual(h, n_items, n_dir, h->tail.p, ddi, ddi + nk * 24, ddi + nk * 30, ddo, ddo + nk * 6, bound, st, &dpcap, true,
                     h->emit_ctr.p);
    if (rc != PFC_OK) return rc;
    if (!zc) HIP_TRY(h, hipMemcpyAsync(h->pin_dout, ddo, sizeof(double) * nk * 12, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(h->h_emit, h->emit_ctr.p, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
    h->fout_host = reinterpret_cast<const int *>((const double *)h->pin_out + out_d) + n * 4;
    // check_fused would poll; here the stream is synchronised (the completion words of the fused kernel are long written)
    HIP_TRY(h, hipStreamSynchronize(st));
    rc = check_eval(h);
    h->fout_host = nullptr;
    if (rc != PFC_OK) return rc;
    if (h->stats[6] & kStCandOvf) return fail(h, PFC_ERR_OVERFLOW, "hand-over list of the small-scene kernel overflowed: batched path");
    if ((unsigned)h->h_emit[2] & kStHole) return fail(h, PFC_ERR_STATE, "internal error: a work-list slot was read before it was written");
    const long long pairs = h->h_emit[0];
    h->dual_hint = pairs;
    const int cpw = 64 / n_dir;
    if ((h->any_bristle || h->opt_fixed_order) && (size_t)((pairs + cpw - 1) / cpw) * 64 + 64 > dpcap) return PFC_ERR_OVERFLOW;
    const double *po = (const double *)h->pin_out;
    std::memcpy(wrench, po, sizeof(double) * n * 6);
    std::memcpy(sdot, po + n * 6, sizeof(double) * n * 6);
    if (counts) std::memcpy(counts, po + out_d, sizeof(int) * n * 4);
    std::memcpy(d_wrench, h->pin_dout, sizeof(double) * nk * 6);
    std::memcpy(d_sdot, (const double *)h->pin_dout + nk * 6, sizeof(double) * nk * 6);
    h->hyb_reuse_ok = true; h->hyb_reuse_n = n_items; h->hyb_reuse_ids = ins_ids != nullptr;
    h->pin_in_dual_n = n_items; h->pin_in_dual_ids = ins_ids != nullptr;
    return PFC_OK;
}

// Small scenes (what Radau evaluates): value pass and Dual passes enqueued back to back, ONE synchronisation, no
// staging copies (the kernels read and write the pinned blocks), the kept Dual polygons sized from the previous Dual
// evaluation's pair count.  Returns PFC_ERR_OVERFLOW when the speculation or a work list fell short: the caller then
// takes the two-stage path, which sizes everything from the value pass.
int eval_dual_small(pfc_context *h, int n_items, int n_dir, const int *ins_ids, const double *pose, const double *twist,
                    const double *s, const double *d_pose, const double *d_twist, const double *d_s, double *wrench,
                    double *sdot, double *d_wrench, double *d_sdot, int *counts) {
    h->pin_din_valid = false;    // these pinned blocks are about to be reused
    h->pin_in_dual_n = 0;
    const size_t n = (size_t)n_items, nk = n * n_dir;
    const size_t in_d = n * 36, in_bytes = in_d * sizeof(double) + n * sizeof(int);
    const size_t out_d = n * 12, out_bytes = out_d * sizeof(double) + n * 4 * sizeof(int);
    const size_t t0 = (((size_t)h->max_levels + 40) + 3) & ~(size_t)3;
    const size_t back_bytes = t0 * sizeof(int) + out_bytes;
    {
        const void *p0 = h->pin_in, *q0 = h->pin_out;
        HIP_TRY(h, ensure_pinned(&h->pin_in, &h->pin_in_cap, in_bytes));
        HIP_TRY(h, ensure_pinned(&h->pin_out, &h->pin_out_cap, back_bytes));
        if (h->pin_in != p0 || h->pin_out != q0) h->small_reuse_ok = false;     // reallocated: the cached blocks are gone
    }
    HIP_TRY(h, ensure_pinned(&h->pin_din, &h->pin_din_cap, sizeof(double) * nk * 36));
    HIP_TRY(h, ensure_pinned(&h->pin_dout, &h->pin_dout_cap, sizeof(double) * nk * 12));
    double *pi = (double *)h->pin_in;
    // the chunks of one Jacobian: value inputs bitwise equal to those still in the pinned input block -> only the Dual passes
    bool same = h->opt_dual_reuse && h->small_reuse_ok && h->small_reuse_n == n_items && h->small_reuse_ids == (ins_ids != nullptr) &&
                std::memcmp(pi, pose, sizeof(double) * n * 24) == 0 && std::memcmp(pi + n * 24, twist, sizeof(double) * n * 6) == 0;
    if (same && s) same = std::memcmp(pi + n * 30, s, sizeof(double) * n * 6) == 0;
    if (same && !s) {
        const double *z = pi + n * 30;
        for (size_t k = 0; k < n * 6 && same; ++k) same = std::memcmp(z + k, "\0\0\0\0\0\0\0\0", 8) == 0;
    }
    if (same && ins_ids) same = std::memcmp(pi + in_d, ins_ids, sizeof(int) * n) == 0;
    if (!same) {
        h->small_reuse_ok = false;
        std::memcpy(pi, pose, sizeof(double) * n * 24);
        std::memcpy(pi + n * 24, twist, sizeof(double) * n * 6);
        if (s) std::memcpy(pi + n * 30, s, sizeof(double) * n * 6); else std::memset(pi + n * 30, 0, sizeof(double) * n * 6);
        if (ins_ids) std::memcpy(pi + in_d, ins_ids, sizeof(int) * n);
    }
    double *pdi = (double *)h->pin_din;
    std::memcpy(pdi, d_pose, sizeof(double) * nk * 24);
    std::memcpy(pdi + nk * 24, d_twist, sizeof(double) * nk * 6);
    if (d_s) std::memcpy(pdi + nk * 30, d_s, sizeof(double) * nk * 6); else std::memset(pdi + nk * 30, 0, sizeof(double) * nk * 6);
    void *v_in = nullptr, *v_out = nullptr, *v_din = nullptr, *v_dout = nullptr;
    HIP_TRY(h, hipHostGetDevicePointer(&v_in, h->pin_in, 0));
    HIP_TRY(h, hipHostGetDevicePointer(&v_out, h->pin_out, 0));
    HIP_TRY(h, hipHostGetDevicePointer(&v_din, h->pin_din, 0));
    HIP_TRY(h, hipHostGetDevicePointer(&v_dout, h->pin_dout, 0));
    bar_mirror(h, &v_in, h->pin_in, in_bytes, 0);
    bar_mirror(h, &v_din, h->pin_din, sizeof(double) * nk * 36, 1);
    bar_publish();
    double *di = (double *)v_in, *dout = reinterpret_cast<double *>((int *)v_out + t0);
    hipStream_t st = h->stream;
    // One captured graph: accumulator fill, the value pass, the Dual passes.  (Launched eagerly behind the replayed value
    // graph, the Dual kernels started 9 us late.)  The capacity of the kept Dual polygons is a power of two above twice
    // the previous pair count, so that the graph survives the small changes from one evaluation to the next.
    size_t bound = 64;
    while (bound < (size_t)h->dual_hint * 2 + 64) bound *= 2;
    const int cpw = 64 / n_dir;
    if (!same) HIP_TRY(h, ensure_work(h, n_items));      // (not before a reuse: option poison refills the work lists there)
    HIP_TRY(h, ensure_dual(h, h->dual_acc, nk * kDaStride));
    HIP_TRY(h, ensure_dual(h, h->dual_res, nk * kDrStride));
    const size_t dpcap = (h->any_bristle || h->opt_fixed_order) ? ((bound + cpw - 1) / cpw) * 64 + 64 : 64;
    HIP_TRY(h, ensure_dual(h, h->dual_poly, dpcap * kDpFields));
    HIP_TRY(h, ensure_dual(h, h->dual_pkey, dpcap));
    const int levels = eff_levels(h);
    const int L = bfs_levels_for(h, n_items, levels);
    // seeds / results of up to 512 (item, direction) pairs are read / written in place by the kernels; larger ones are
    // copied by memcpy nodes of the graph (reading 288 B per pair over PCIe from inside k_narrow_dual stops paying)
    const bool zc_dual = nk <= (h->bar_state == 1 ? (int)kBarKeys : 512);
    if (!zc_dual) {
        HIP_TRY(h, ensure_dual(h, h->dual_in, nk * 36));
        HIP_TRY(h, ensure_dual(h, h->dual_out, nk * 12));
    }
    double *ddi = zc_dual ? (double *)v_din : h->dual_in.p, *ddo = zc_dual ? (double *)v_dout : h->dual_out.p;
    if (same) {
        // eager launches of the Dual passes on the value pass the last graph replay left (lists, item records, the packed tail
        // in the pinned output block); the pair count is known, so there is no speculation to check
        HIP_TRY(h, hipMemsetAsync(h->dual_acc.p, 0, sizeof(double) * nk * kDaStride, st));
        if (!zc_dual) HIP_TRY(h, hipMemcpyAsync(ddi, h->pin_din, sizeof(double) * nk * 36, hipMemcpyHostToDevice, st));
        h->last_levels = levels;
        const size_t bound_r = (size_t)(h->dual_hint > 0 ? h->dual_hint : 0) + 64;
        int rcr = launch_dual(h, n_items, n_dir, (const int *)v_out, ddi, ddi + nk * 24, ddi + nk * 30, ddo, ddo + nk * 6, bound_r, st,
                              nullptr, true);
        if (rcr != PFC_OK) return rcr;
        if (!zc_dual) HIP_TRY(h, hipMemcpyAsync(h->pin_dout, ddo, sizeof(double) * nk * 12, hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipStreamSynchronize(st));
        const double *po = reinterpret_cast<const double *>((const int *)h->pin_out + t0);
        std::memcpy(wrench, po, sizeof(double) * n * 6);
        std::memcpy(sdot, po + n * 6, sizeof(double) * n * 6);
        if (counts) std::memcpy(counts, po + out_d, sizeof(int) * n * 4);
        std::memcpy(d_wrench, h->pin_dout, sizeof(double) * nk * 6);
        std::memcpy(d_sdot, (const double *)h->pin_dout + nk * 6, sizeof(double) * nk * 6);
        h->last_dual_reused = true;
        return PFC_OK;
    }
    const int *d_ins = ins_ids ? (const int *)(di + in_d) : nullptr;
    const double *d_sv = s ? di + n * 30 : nullptr;
    pfc_context::DualGraphKey key = {};
    key.v.n_items = n_items; key.v.levels = levels; key.v.L = L; key.v.debug = 0;
    key.v.bristle = (h->any_bristle ? 1 : 0) | (h->any_tet_tet ? 2 : 0); key.v.surv = 1;
    key.v.p[0] = d_ins; key.v.p[1] = di; key.v.p[2] = di + n * 24; key.v.p[3] = d_sv; key.v.p[4] = dout;
    key.v.p[5] = dout + n * 6; key.v.p[6] = dout + out_d; key.v.p[7] = v_out; key.v.p[8] = h->bp_dev; key.v.stream = (void *)st; key.v.epoch = h->epoch;
    key.n_dir = n_dir; key.bound = bound; key.din = ddi; key.dout = ddo;
    h->want_surv = true;
    h->tail_dev = (int *)v_out;
    h->last_levels = levels;      // launch_dual locates the pair counter in the tail by it
    int rc = PFC_OK;
    if (!h->dghave || std::memcmp(&key, &h->dgkey, sizeof key) != 0) {
        if (h->dgexec) { (void)hipGraphExecDestroy(h->dgexec); h->dgexec = nullptr; }
        h->dghave = false;
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            e = hipMemsetAsync(h->dual_acc.p, 0, sizeof(double) * nk * kDaStride, st);
            if (!zc_dual && e == hipSuccess) e = hipMemcpyAsync(ddi, h->pin_din, sizeof(double) * nk * 36, hipMemcpyHostToDevice, st);
            rc = record_eval(h, n_items, d_ins, di, di + n * 24, d_sv, dout, dout + n * 6, (int *)(dout + out_d), st, false);
            if (rc == PFC_OK)
                rc = launch_dual(h, n_items, n_dir, (const int *)v_out, ddi, ddi + nk * 24, ddi + nk * 30, ddo, ddo + nk * 6, bound,
                                 st, nullptr, true);
            if (!zc_dual && rc == PFC_OK && e == hipSuccess)
                e = hipMemcpyAsync(h->pin_dout, ddo, sizeof(double) * nk * 12, hipMemcpyDeviceToHost, st);
            const hipError_t e2 = hipStreamEndCapture(st, &graph);
            if (e == hipSuccess) e = e2;
        }
        if (rc == PFC_OK && e == hipSuccess) e = hipGraphInstantiate(&h->dgexec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        h->want_surv = false; h->tail_dev = nullptr;
        if (rc != PFC_OK) return rc;
        if (e != hipSuccess) return fail(h, PFC_ERR_HIP, "Dual graph capture failed: %s", hipGetErrorString(e));
        h->dgkey = key; h->dghave = true;
    }
    h->want_surv = false; h->tail_dev = nullptr;
    h->dual_reuse_ok = false; h->hyb_reuse_ok = false; h->small_reuse_ok = false; h->last_dual_reused = false;   // the replay overwrites the device state
    HIP_TRY(h, hipGraphLaunch(h->dgexec, st));
    h->last_bfs_levels = L; h->last_n_items = n_items; h->pending = true; h->last_stream = st; h->ev_valid = false;
    h->split_n0 = 0;
    h->tail_host = (const int *)h->pin_out;
    rc = check_eval(h);                 // the one synchronisation; grows the work lists on overflow
    h->tail_host = nullptr;
    if (rc != PFC_OK) return rc;
    // did the kept Dual polygons fit?  (contributing pairs: the counter next to the polygon total in the packed tail)
    const int *tail = (const int *)h->pin_out;
    const long long pairs = tail[12 + (((h->last_levels + 9) & ~1) + 1)];
    h->dual_hint = pairs;
    if ((h->any_bristle || h->opt_fixed_order) && (size_t)((pairs + cpw - 1) / cpw) * 64 + 64 > dpcap) return PFC_ERR_OVERFLOW;
    const double *po = reinterpret_cast<const double *>((const int *)h->pin_out + t0);
    std::memcpy(wrench, po, sizeof(double) * n * 6);
    std::memcpy(sdot, po + n * 6, sizeof(double) * n * 6);
    if (counts) std::memcpy(counts, po + out_d, sizeof(int) * n * 4);
    std::memcpy(d_wrench, h->pin_dout, sizeof(double) * nk * 6);
    std::memcpy(d_sdot, (const double *)h->pin_dout + nk * 6, sizeof(double) * nk * 6);
    h->small_reuse_ok = true; h->small_reuse_n = n_items; h->small_reuse_ids = ins_ids != nullptr;
    return PFC_OK;
}
}  // namespace

int pfc_eval_dual_device_bp(pfc_handle h, int n_items, int n_dir, const int *d_ins_ids, const double *d_pose, const double *d_bp_pose,
                            const double *d_twist, const double *d_s, const double *d_dpose, const double *d_dtwist,
                            const double *d_ds, double *d_wrench, double *d_sdot, double *d_dwrench, double *d_dsdot,
                            int *d_counts, void *stream) {
    if (!h) return PFC_ERR_BAD_ARG;
    if (h->multi)
        return multi_eval_device(h, n_items, n_dir, d_ins_ids, d_pose, d_bp_pose, d_twist, d_s, d_dpose, d_dtwist, d_ds, d_wrench, d_sdot,
                                 d_dwrench, d_dsdot, d_counts, stream);
    h->bp_dev = d_bp_pose;
    const int rc = pfc_eval_dual_device(h, n_items, n_dir, d_ins_ids, d_pose, d_twist, d_s, d_dpose, d_dtwist, d_ds, d_wrench, d_sdot,
                                        d_dwrench, d_dsdot, d_counts, stream);
    h->bp_dev = nullptr;
    return rc;
}

int pfc_eval_dual_device(pfc_handle h, int n_items, int n_dir, const int *d_ins_ids, const double *d_pose,
                         const double *d_twist, const double *d_s, const double *d_dpose, const double *d_dtwist,
                         const double *d_ds, double *d_wrench, double *d_sdot, double *d_dwrench, double *d_dsdot,
                         int *d_counts, void *stream) {
    if (!h) return PFC_ERR_BAD_ARG;
    if (h->multi)
        return multi_eval_device(h, n_items, n_dir, d_ins_ids, d_pose, nullptr, d_twist, d_s, d_dpose, d_dtwist, d_ds, d_wrench, d_sdot,
                                 d_dwrench, d_dsdot, d_counts, stream);
    if (n_dir < 1 || n_dir > 16) return fail(h, PFC_ERR_BAD_ARG, "pfc_eval_dual_device: n_dir must be in 1..16");
    { const int rc = check_eval_args(h, n_items, d_ins_ids, d_pose, d_twist, d_s, d_wrench, d_sdot); if (rc != PFC_OK) return rc; }
    h->pending_dual = false; h->dual_reuse_ok = false; h->pending_more = false;
    if (n_items == 0) { h->pending = false; h->last_n_items = 0; return PFC_OK; }
    if (!d_dpose || !d_dtwist || !d_dwrench || !d_dsdot) return fail(h, PFC_ERR_BAD_ARG, "pfc_eval_dual_device: null buffer");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    const size_t nk = (size_t)n_items * n_dir;
    // Kept Dual polygons are sized WITHOUT reading the value pass back: twice the contributing pairs of the previous Dual
    // evaluation (a power of two, so that the buffers settle), a guess the first time.  pfc_check compares the actual
    // count with the capacity after the one synchronisation and asks for a re-issue if it fell short.
    size_t bound = 4096;
    while (bound < (size_t)(h->dual_hint > 0 ? h->dual_hint : 0) * 2 + 64 || bound < (size_t)n_items * 8) bound *= 2;
    HIP_TRY(h, ensure_dual(h, h->dual_acc, nk * kDaStride));
    HIP_TRY(h, ensure_dual(h, h->dual_res, nk * kDrStride));
    if (!d_ds) {
        const size_t c0 = h->dual_zero.cap;
        HIP_TRY(h, h->dual_zero.ensure(nk * 6));
        if (h->dual_zero.cap != c0) HIP_TRY(h, hipMemsetAsync(h->dual_zero.p, 0, sizeof(double) * h->dual_zero.cap, st));
        d_ds = h->dual_zero.p;
    }
    HIP_TRY(h, hipMemsetAsync(h->dual_acc.p, 0, sizeof(double) * nk * kDaStride, st));
    h->split_n0 = 0; h->pending_fused = false; h->pending_dual_hyb = false;
    // Small scenes: the value pass in the one-workgroup-per-item kernel, which hands item records and lists to the batched
    // Dual passes (what pfc_eval_dual does for host buffers, eval_dual_hybrid).  A miss (an item that does not fit, a
    // hand-over list or a speculation that fell short) is reported by pfc_check as PFC_ERR_OVERFLOW and the re-issue takes
    // the batched value pass below.
    if (fused_ok(h, n_items) && nk <= 4096 && (h->dual_hint >= 0 || !h->any_bristle) && std::getenv("PFC_NO_HYBRID") == nullptr) {
        if (h->dual_dev_hyb_skip > 0) {
            --h->dual_dev_hyb_skip;
        } else {
            HIP_TRY(h, ensure_work(h, n_items));
            HIP_TRY(h, h->emit_ctr.ensure(4));
            if (!h->h_emit) HIP_TRY(h, hipHostMalloc((void **)&h->h_emit, sizeof(int) * 4));
            size_t bh = 64;
            while (bh < (size_t)(h->dual_hint > 0 ? h->dual_hint : 0) * 2 + 64) bh *= 2;
            HIP_TRY(h, hipMemsetAsync(h->emit_ctr.p, 0, sizeof(int) * 4, st));
            h->fu_emit = true; h->fout_dev = nullptr; h->fout_host = nullptr;
            int rcf = enqueue_fused(h, n_items, d_ins_ids, d_pose, d_twist, d_s, d_wrench, d_sdot, d_counts, st);
            h->fu_emit = false;
            if (rcf != PFC_OK) return rcf;
            h->last_levels = eff_levels(h);
            size_t dpcap_h = 0;
            rcf = launch_dual(h, n_items, n_dir, h->tail.p, d_dpose, d_dtwist, d_ds, d_dwrench, d_dsdot, bh, st, &dpcap_h, true, h->emit_ctr.p);
            if (rcf != PFC_OK) return rcf;
            HIP_TRY(h, hipMemcpyAsync(h->h_emit, h->emit_ctr.p, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
            h->pending_dual_hyb = true; h->pending_dpcap = dpcap_h; h->pending_ndir = n_dir;
            return PFC_OK;
        }
    }
    h->want_surv = true;         // the value pass also lists the contributing candidates; batched path, one part
    int rc = enqueue_eval(h, n_items, d_ins_ids, d_pose, d_twist, d_s, d_wrench, d_sdot, d_counts, st);
    h->want_surv = false;
    if (rc != PFC_OK) return rc;
    size_t dpcap = 0;
    rc = launch_dual(h, n_items, n_dir, h->tail.p, d_dpose, d_dtwist, d_ds, d_dwrench, d_dsdot, bound, st, &dpcap, true);
    if (rc != PFC_OK) return rc;
    h->pending_dual = true; h->pending_dpcap = dpcap; h->pending_ndir = n_dir;
    return PFC_OK;
}

int pfc_eval_dual_device_more(pfc_handle h, int n_dir, const double *d_dpose, const double *d_dtwist, const double *d_ds,
                              double *d_dwrench, double *d_dsdot, void *stream) {
    if (!h) return PFC_ERR_BAD_ARG;
    if (h->multi) return multi_eval_dual_device_more(h, n_dir, d_dpose, d_dtwist, d_ds, d_dwrench, d_dsdot, stream);
    if (n_dir < 1 || n_dir > 16) return fail(h, PFC_ERR_BAD_ARG, "pfc_eval_dual_device_more: n_dir must be in 1..16");
    if (!h->dual_reuse_ok)
        return fail(h, PFC_ERR_STATE, "pfc_eval_dual_device_more: no checked pfc_eval_dual_device evaluation on this handle to extend");
    if (!d_dpose || !d_dtwist || !d_dwrench || !d_dsdot) return fail(h, PFC_ERR_BAD_ARG, "pfc_eval_dual_device_more: null buffer");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    const int n_items = h->dual_reuse_n;
    const size_t nk = (size_t)n_items * n_dir;
    HIP_TRY(h, ensure_dual(h, h->dual_acc, nk * kDaStride));
    HIP_TRY(h, ensure_dual(h, h->dual_res, nk * kDrStride));
    if (!d_ds) {
        const size_t c0 = h->dual_zero.cap;
        HIP_TRY(h, h->dual_zero.ensure(nk * 6));
        if (h->dual_zero.cap != c0) HIP_TRY(h, hipMemsetAsync(h->dual_zero.p, 0, sizeof(double) * h->dual_zero.cap, st));
        d_ds = h->dual_zero.p;
    }
    HIP_TRY(h, hipMemsetAsync(h->dual_acc.p, 0, sizeof(double) * nk * kDaStride, st));
    // the contributing pairs are known exactly (dual_hint, from the value pass pfc_check has seen): no speculation
    const size_t bound = (size_t)(h->dual_hint > 0 ? h->dual_hint : 0) + 64;
    // The passes report (kStHole: a list entry out of range, capacity guards) into a word of their own, cleared here and read
    // back by pfc_check in front of its synchronisation: neither the value pass's status word (read by ITS k_final only) nor
    // the hand-over block (read with the pair count of a first chunk only) is looked at again on this path.
    if (!h->h_more) HIP_TRY(h, hipHostMalloc((void **)&h->h_more, sizeof(unsigned) * 4));
    unsigned *more_status = h->status.p + 1;
    HIP_TRY(h, hipMemsetAsync(more_status, 0, sizeof(unsigned), st));
    const int rc = launch_dual(h, n_items, n_dir, h->tail.p, d_dpose, d_dtwist, d_ds, d_dwrench, d_dsdot, bound, st, nullptr, true,
                               h->dual_reuse_emit ? h->emit_ctr.p : nullptr, more_status);
    if (rc != PFC_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->h_more, more_status, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    h->pending_more = true; h->last_stream = st; h->last_dual_reused = true;
    return PFC_OK;
}

// The Dual evaluation with the broadphase pose of m.float's state (calcTriTetIntersections!, non_friction.jl:94-101): the block is
// kept in pinned host memory, which the value pass reads in place (96 bytes per item, once).
int pfc_eval_dual_bp(pfc_handle h, int n_items, int n_dir, const int *ins_ids, const double *pose, const double *bp_pose,
                     const double *twist, const double *s, const double *d_pose, const double *d_twist, const double *d_s,
                     double *wrench, double *sdot, double *d_wrench, double *d_sdot, int *counts) {
    if (!h) return PFC_ERR_BAD_ARG;
    if (h->multi)
        return multi_eval_dual(h, n_items, n_dir, ins_ids, pose, bp_pose, twist, s, d_pose, d_twist, d_s, wrench, sdot, d_wrench, d_sdot, counts);
    if (!bp_pose || n_items <= 0)
        return pfc_eval_dual(h, n_items, n_dir, ins_ids, pose, twist, s, d_pose, d_twist, d_s, wrench, sdot, d_wrench, d_sdot, counts);
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t bytes = sizeof(double) * 24 * (size_t)n_items;
    {
        const void *p0 = h->pin_bp;
        HIP_TRY(h, ensure_pinned(&h->pin_bp, &h->pin_bp_cap, bytes));
        if (h->pin_bp != p0) h->pin_bp_n = -1;
    }
    if (h->pin_bp_n != n_items || std::memcmp(h->pin_bp, bp_pose, bytes) != 0) {
        // another broadphase pose: nothing a previous Dual evaluation left can be reused (the candidate lists differ)
        h->dual_reuse_ok = false; h->hyb_reuse_ok = false; h->small_reuse_ok = false; h->pin_in_dual_n = 0; h->pin_din_valid = false;
        std::memcpy(h->pin_bp, bp_pose, bytes);
        h->pin_bp_n = n_items;
    }
    void *dev = nullptr;
    HIP_TRY(h, hipHostGetDevicePointer(&dev, h->pin_bp, 0));
    h->bp_dev = (const double *)dev;
    const int rc = pfc_eval_dual(h, n_items, n_dir, ins_ids, pose, twist, s, d_pose, d_twist, d_s, wrench, sdot, d_wrench, d_sdot, counts);
    h->bp_dev = nullptr;
    return rc;
}

int pfc_eval_dual(pfc_handle h, int n_items, int n_dir, const int *ins_ids, const double *pose, const double *twist,
                  const double *s, const double *d_pose, const double *d_twist, const double *d_s, double *wrench,
                  double *sdot, double *d_wrench, double *d_sdot, int *counts) {
    if (!h) return PFC_ERR_BAD_ARG;
    if (h->multi)
        return multi_eval_dual(h, n_items, n_dir, ins_ids, pose, nullptr, twist, s, d_pose, d_twist, d_s, wrench, sdot, d_wrench, d_sdot, counts);
    if (!h->bp_dev && h->pin_bp_n != 0) {
        // the previous host-buffer Dual evaluation culled with a pose of its own: its lists are not this evaluation's
        h->dual_reuse_ok = false; h->hyb_reuse_ok = false; h->small_reuse_ok = false; h->pin_in_dual_n = 0; h->pin_din_valid = false;
        h->pin_bp_n = 0;
    }
    if (n_dir < 1 || n_dir > 16) return fail(h, PFC_ERR_BAD_ARG, "pfc_eval_dual: n_dir must be in 1..16");
    if (n_items > 0 && (!d_pose || !d_twist || !d_wrench || !d_sdot))
        return fail(h, PFC_ERR_BAD_ARG, "pfc_eval_dual: null buffer");
    // the same argument checks in front of BOTH paths (the one-graph path used to skip them)
    { const int rc = check_eval_args(h, n_items, ins_ids, pose, twist, s, wrench, sdot); if (rc != PFC_OK) return rc; }
    // A repeat of the previous small-scene Dual evaluation's point (the next chunk of a Jacobian): leave the all-in-one
    // kernel, whose Dual passes keep nothing, for the hybrid path, which hands lists over that the chunks after it reuse.
    bool repeat = false;
    if (n_items > 0 && h->opt_dual_reuse && h->pin_in_dual_n == n_items && h->pin_in && h->pin_in_dual_ids == (ins_ids != nullptr)) {
        const double *pi = (const double *)h->pin_in;
        const size_t n = (size_t)n_items;
        repeat = std::memcmp(pi, pose, sizeof(double) * n * 24) == 0 && std::memcmp(pi + n * 24, twist, sizeof(double) * n * 6) == 0 &&
                 (!s || std::memcmp(pi + n * 30, s, sizeof(double) * n * 6) == 0) &&
                 (!ins_ids || std::memcmp(pi + n * 36, ins_ids, sizeof(int) * n) == 0);
    }
    if (!repeat) h->pin_in_dual_n = 0;
    if (n_items > 0 && !h->any_bristle && fused_ok(h, n_items) && !repeat) {
        if (h->dual_fused_skip > 0) {
            --h->dual_fused_skip;
        } else {
            HIP_TRY(h, hipSetDevice(h->device));
            const int rc_f = eval_dual_fused(h, n_items, n_dir, ins_ids, pose, twist, s, d_pose, d_twist, wrench, sdot, d_wrench,
                                             d_sdot, counts);
            if (rc_f != PFC_ERR_OVERFLOW) return rc_f;       // else: an item did not fit -> the batched Dual paths below
        }
    }
    if (n_items > 0 && fused_ok(h, n_items) && (h->dual_hint >= 0 || !h->any_bristle) && (size_t)n_items * n_dir <= 4096 &&
        std::getenv("PFC_NO_HYBRID") == nullptr) {
        HIP_TRY(h, hipSetDevice(h->device));
        const int rc_h = eval_dual_hybrid(h, n_items, n_dir, ins_ids, pose, twist, s, d_pose, d_twist, d_s, wrench, sdot, d_wrench,
                                          d_sdot, counts);
        if (rc_h != PFC_ERR_OVERFLOW) return rc_h;           // else: batched paths below (lists grown / speculation short)
        if (h->dual_dev_hyb_skip < 1) h->dual_dev_hyb_skip = 1;   // (not the same sequence again inside pfc_eval_dual_device)
    }
    if (h->finalized && n_items > 0 && n_items <= 512 && (size_t)n_items * n_dir <= 4096 && h->dual_hint >= 0 && pose && twist &&
        wrench && sdot && !h->opt_debug && !h->opt_fixed_order && !(h->opt_split_min > 0 && n_items >= h->opt_split_min)) {
        HIP_TRY(h, hipSetDevice(h->device));
        const int rc_small = eval_dual_small(h, n_items, n_dir, ins_ids, pose, twist, s, d_pose, d_twist, d_s, wrench, sdot,
                                             d_wrench, d_sdot, counts);
        if (rc_small != PFC_ERR_OVERFLOW) return rc_small;    // else: lists grown / speculation short -> two-stage path
    }
    if (n_items == 0) return PFC_OK;
    const size_t nk = (size_t)n_items * n_dir;
    hipStream_t st = h->stream;
    if (!h->opt_debug && std::getenv("PFC_DUAL_TWO_STAGE") == nullptr) {
        // One synchronisation: inputs up in one pinned block, pfc_eval_dual_device (value pass + Dual passes back to back,
        // kept Dual polygons sized from the previous evaluation), results down in one pinned block, pfc_check.  A work
        // list that overflowed or a speculation that fell short re-issues the evaluation (buffers have grown).
        HIP_TRY(h, hipSetDevice(h->device));
        const size_t n = (size_t)n_items;
        const size_t in_d = n * 36 + nk * 36, in_bytes = in_d * sizeof(double) + n * sizeof(int);
        const size_t out_d = n * 12 + nk * 12, out_bytes = out_d * sizeof(double) + n * 4 * sizeof(int);
        {
            const void *p0 = h->pin_din, *q0 = h->pin_dout;
            HIP_TRY(h, ensure_pinned(&h->pin_din, &h->pin_din_cap, in_bytes));
            HIP_TRY(h, ensure_pinned(&h->pin_dout, &h->pin_dout_cap, out_bytes));
            if (h->pin_din != p0 || h->pin_dout != q0) h->pin_din_valid = false;      // reallocated: the cached blocks are gone
        }
        HIP_TRY(h, ensure_dual(h, h->dual_in, in_d + (n + 1) / 2 + 1));
        HIP_TRY(h, ensure_dual(h, h->dual_out, out_d + (n * 4 + 1) / 2 + 1));
        double *pi = (double *)h->pin_din;
        double *pd = pi + n * 36;
        double *di = h->dual_in.p, *dd = di + n * 36, *dout = h->dual_out.p, *ddo = dout + n * 12;
        // Further seed directions at the point of the previous Dual evaluation (the chunks of one Jacobian: Radau calls
        // the path ceil(NX / N_chunk) times with the same values and different partials, src/radau/radau_functions.jl:2-14):
        // if the value inputs equal, bit for bit, those still sitting in the pinned input block, the value pass on the
        // device is reused -- candidates, contributing pairs, per-item results -- and only the Dual passes run.
        static const double kZero6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        // (the device lists must be those of the evaluation the pinned block belongs to: a pfc_eval_dual_device + pfc_check of
        // the caller's own in between leaves dual_reuse_ok set for ITS point -- the serial tells the two apart)
        bool same = h->opt_dual_reuse && h->dual_reuse_ok && h->pin_din_valid && h->pin_din_serial == h->value_serial && h->dual_reuse_n == n_items &&
                    h->pin_din_ids == (ins_ids != nullptr) && h->dual_counts_cache.size() == n * 4 &&
                    std::memcmp(pi, pose, sizeof(double) * n * 24) == 0 && std::memcmp(pi + n * 24, twist, sizeof(double) * n * 6) == 0;
        if (same && s) same = std::memcmp(pi + n * 30, s, sizeof(double) * n * 6) == 0;
        if (same && !s)
            for (size_t k = 0; k < n && same; ++k) same = std::memcmp(pi + n * 30 + 6 * k, kZero6, sizeof kZero6) == 0;
        if (same && ins_ids) same = std::memcmp(pi + n * 36 + h->pin_din_nk36, ins_ids, sizeof(int) * n) == 0;
        if (same) {
            // the seeds go behind the value block as before (the ids of the cached evaluation are not needed again)
            std::memcpy(pd, d_pose, sizeof(double) * nk * 24);
            std::memcpy(pd + nk * 24, d_twist, sizeof(double) * nk * 6);
            if (d_s) std::memcpy(pd + nk * 30, d_s, sizeof(double) * nk * 6); else std::memset(pd + nk * 30, 0, sizeof(double) * nk * 6);
            if (ins_ids) std::memcpy(pd + nk * 36, ins_ids, sizeof(int) * n);      // keep the block's layout: ids behind the seeds
            h->pin_din_nk36 = nk * 36;
            HIP_TRY(h, hipMemcpyAsync(dd, pd, sizeof(double) * nk * 36, hipMemcpyHostToDevice, st));
            int rc3 = pfc_eval_dual_device_more(h, n_dir, dd, dd + nk * 24, dd + nk * 30, ddo, ddo + nk * 6, st);
            if (rc3 != PFC_OK) return rc3;
            double *pdo = (double *)h->pin_dout + n * 12;
            HIP_TRY(h, hipMemcpyAsync(pdo, ddo, sizeof(double) * nk * 12, hipMemcpyDeviceToHost, st));
            rc3 = pfc_check(h);
            if (rc3 != PFC_OK) return rc3;
            const double *po = (const double *)h->pin_dout;
            std::memcpy(wrench, po, sizeof(double) * n * 6);
            std::memcpy(sdot, po + n * 6, sizeof(double) * n * 6);
            std::memcpy(d_wrench, pdo, sizeof(double) * nk * 6);
            std::memcpy(d_sdot, pdo + nk * 6, sizeof(double) * nk * 6);
            if (counts) std::memcpy(counts, h->dual_counts_cache.data(), sizeof(int) * n * 4);
            return PFC_OK;
        }
        h->pin_din_valid = false;
        std::memcpy(pi, pose, sizeof(double) * n * 24);
        std::memcpy(pi + n * 24, twist, sizeof(double) * n * 6);
        if (s) std::memcpy(pi + n * 30, s, sizeof(double) * n * 6); else std::memset(pi + n * 30, 0, sizeof(double) * n * 6);
        std::memcpy(pd, d_pose, sizeof(double) * nk * 24);
        std::memcpy(pd + nk * 24, d_twist, sizeof(double) * nk * 6);
        if (d_s) std::memcpy(pd + nk * 30, d_s, sizeof(double) * nk * 6); else std::memset(pd + nk * 30, 0, sizeof(double) * nk * 6);
        if (ins_ids) std::memcpy(pi + in_d, ins_ids, sizeof(int) * n);
        HIP_TRY(h, hipMemcpyAsync(di, pi, ins_ids ? in_bytes : in_d * sizeof(double), hipMemcpyHostToDevice, st));
        int rc2 = PFC_OK;
        for (int attempt = 0; attempt < 40; ++attempt) {
            rc2 = pfc_eval_dual_device(h, n_items, n_dir, ins_ids ? (const int *)(di + in_d) : nullptr, di, di + n * 24,
                                       s ? di + n * 30 : nullptr, dd, dd + nk * 24, dd + nk * 30, dout, dout + n * 6,
                                       ddo, ddo + nk * 6, (int *)(dout + out_d), st);
            if (rc2 != PFC_OK) return rc2;
            HIP_TRY(h, hipMemcpyAsync(h->pin_dout, dout, out_bytes, hipMemcpyDeviceToHost, st));
            rc2 = pfc_check(h);
            if (rc2 != PFC_ERR_OVERFLOW) break;
        }
        if (rc2 != PFC_OK) return rc2;
        const double *po = (const double *)h->pin_dout;
        std::memcpy(wrench, po, sizeof(double) * n * 6);
        std::memcpy(sdot, po + n * 6, sizeof(double) * n * 6);
        std::memcpy(d_wrench, po + n * 12, sizeof(double) * nk * 6);
        std::memcpy(d_sdot, po + n * 12 + nk * 6, sizeof(double) * nk * 6);
        if (counts) std::memcpy(counts, po + out_d, sizeof(int) * n * 4);
        // what a following call at the same point needs: the counters (the pinned blocks keep the rest)
        h->dual_counts_cache.assign(reinterpret_cast<const int *>(po + out_d), reinterpret_cast<const int *>(po + out_d) + n * 4);
        h->pin_din_valid = true; h->pin_din_ids = ins_ids != nullptr; h->pin_din_nk36 = nk * 36; h->pin_din_serial = h->value_serial;
        return PFC_OK;
    }
    // two-stage path (debug option, or PFC_DUAL_TWO_STAGE set for A/B runs): values, candidate list and per-item counters
    // by the ordinary evaluation (the broadphase ignores partials, src/contact_algorithms_non_friction.jl:95), read the
    // contributing-pair count, then the Dual passes
    h->want_surv = true;
    int rc = pfc_eval(h, n_items, ins_ids, pose, twist, s, wrench, sdot, counts);
    h->want_surv = false;
    if (rc != PFC_OK) return rc;
    HIP_TRY(h, ensure_dual(h, h->dual_in, nk * 36));
    HIP_TRY(h, ensure_dual(h, h->dual_acc, nk * kDaStride));
    HIP_TRY(h, ensure_dual(h, h->dual_res, nk * kDrStride));
    HIP_TRY(h, ensure_dual(h, h->dual_out, nk * 12));
    double *dp = h->dual_in.p, *dt = dp + nk * 24, *dsd = dt + nk * 6;
    // one pinned block up (d_pose | d_twist | d_s), one down (d_wrench | d_sdot), as in pfc_eval (whose staging
    // buffers are free again at this point)
    const size_t in_bytes = sizeof(double) * nk * 36, out_bytes = sizeof(double) * nk * 12;
    HIP_TRY(h, ensure_pinned(&h->pin_in, &h->pin_in_cap, in_bytes));
    HIP_TRY(h, ensure_pinned(&h->pin_out, &h->pin_out_cap, out_bytes));
    {
        double *pi = (double *)h->pin_in;
        std::memcpy(pi, d_pose, sizeof(double) * nk * 24);
        std::memcpy(pi + nk * 24, d_twist, sizeof(double) * nk * 6);
        if (d_s) std::memcpy(pi + nk * 30, d_s, sizeof(double) * nk * 6);
        else std::memset(pi + nk * 30, 0, sizeof(double) * nk * 6);
        HIP_TRY(h, hipMemcpyAsync(dp, pi, in_bytes, hipMemcpyHostToDevice, st));
    }
    rc = launch_dual(h, n_items, n_dir, h->tail.p, dp, dt, dsd, h->dual_out.p, h->dual_out.p + nk * 6, (size_t)h->stats[2], st, nullptr);
    if (rc != PFC_OK) return rc;
    h->dual_hint = h->stats[2];
    HIP_TRY(h, hipMemcpyAsync(h->pin_out, h->dual_out.p, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    std::memcpy(d_wrench, h->pin_out, sizeof(double) * nk * 6);
    std::memcpy(d_sdot, (const double *)h->pin_out + nk * 6, sizeof(double) * nk * 6);
    return PFC_OK;
}

int pfc_set_option(pfc_handle h, const char *name, long long value) {
    if (!h || !name) return PFC_ERR_BAD_ARG;
    if (h->multi) {
        if (!std::strcmp(name, "multi_min")) { h->multi->opt_min_items = value < 1 ? 1 : (int)value; h->multi->part_n = 0; return PFC_OK; }
        for (pfc_context *c : h->multi->shard) {
            const int rc = pfc_set_option(c, name, value);
            if (rc != PFC_OK) { h->err = c->err; return rc; }
        }
        return PFC_OK;
    }
    h->dual_reuse_ok = false; h->hyb_reuse_ok = false; h->small_reuse_ok = false;
    if (!std::strcmp(name, "debug")) h->opt_debug = value != 0;
    else if (!std::strcmp(name, "profile")) h->opt_profile = value != 0;
    else if (!std::strcmp(name, "max_levels")) {
        if (value < 0 || value > (1 << 20) || (h->finalized && value > h->max_levels))
            return fail(h, PFC_ERR_BAD_ARG, "max_levels must be 0 (automatic) or 1..%d (depth of the finalized trees + 1)", h->max_levels);
        h->opt_max_levels = (int)value; h->ghave[0] = h->ghave[1] = false; h->dghave = false;
    }
    else if (!std::strcmp(name, "bfs_levels")) h->opt_bfs_levels = (int)value;
    else if (!std::strcmp(name, "graph")) h->opt_graph = value != 0;
    else if (!std::strcmp(name, "split_min")) h->opt_split_min = (int)value;
    else if (!std::strcmp(name, "dual_reuse")) h->opt_dual_reuse = (int)value;
    else if (!std::strcmp(name, "clip_min")) { h->opt_clip_min = (int)value; h->ghave[0] = h->ghave[1] = false; h->dghave = false; }
    else if (!std::strcmp(name, "clip_queue")) { h->opt_clip_queue = (int)value; h->ghave[0] = h->ghave[1] = false; h->dghave = false; }
    else if (!std::strcmp(name, "poison")) h->opt_poison = value != 0;
    else if (!std::strcmp(name, "fused")) { h->opt_fused = value != 0; h->fused_skip = 0; }
    else if (!std::strcmp(name, "team")) h->opt_team = value < 0 ? 0 : (value > kTeamMaxWg ? kTeamMaxWg : (int)value);
    else if (!std::strcmp(name, "team_fault")) h->opt_team_fault = (int)value;
    else if (!std::strcmp(name, "fused_f32")) h->opt_fused_f32 = value != 0;
    else if (!std::strcmp(name, "dual_fold")) { h->opt_dual_fold = value != 0; h->dghave = false; }
    else if (!std::strcmp(name, "fixed_order")) {
        // the batched path only while it is on, whatever "fused" / "team" / "split_min" / "graph" say (fused_ok, fused_team, the split
        // rule and use_graph look at it): no one-launch kernel, no two-half split, eager launches (the sort is library code)
        const int on = value != 0;
        if (on != h->opt_fixed_order) {      // (setting it to what it is changes nothing: a kept value pass stays reusable)
            h->opt_fixed_order = on; h->fused_skip = 0;
            h->ghave[0] = h->ghave[1] = false; h->dghave = false;
            h->dual_reuse_ok = false; h->hyb_reuse_ok = false; h->small_reuse_ok = false;
            h->rec.release();      // (the records change their stride)
        }
    }
    else if (!std::strcmp(name, "no_filter")) { h->opt_no_filter = (int)value; h->ghave[0] = h->ghave[1] = false; h->dghave = false; }
    else return fail(h, PFC_ERR_BAD_ARG, "unknown option %s", name);
    return PFC_OK;
}

int pfc_get_stats(pfc_handle h, long long *out8) {
    if (!h || !out8) return PFC_ERR_BAD_ARG;
    if (h->multi) { for (int k = 0; k < 8; ++k) out8[k] = h->multi->stats[k]; return PFC_OK; }
    for (int k = 0; k < 8; ++k) out8[k] = h->stats[k];
    return PFC_OK;
}

int pfc_get_stage_ms(pfc_handle h, float *out6) {
    if (!h || !out6) return PFC_ERR_BAD_ARG;
    if (h->multi) {      // the first shard's stages (every shard runs the same sequence on its range)
        const int rc = pfc_get_stage_ms(h->multi->shard[0], out6);
        if (rc != PFC_OK) h->err = h->multi->shard[0]->err;
        return rc;
    }
    if (!h->ev_valid) return fail(h, PFC_ERR_STATE, "profile option was off for the last evaluation");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipEventSynchronize(h->ev[EV_FIN]));
    for (int k = 0; k < 5; ++k) HIP_TRY(h, hipEventElapsedTime(&out6[k], h->ev[k], h->ev[k + 1]));
    HIP_TRY(h, hipEventElapsedTime(&out6[5], h->ev[EV_START], h->ev[EV_FIN]));
    if (h->last_parts == 2 && h->twin && h->twin->ev_valid) {
        // two concurrent halves: mean duration of a stage over the two half-launches, total = the longer half
        pfc_context *t = h->twin;
        float b[6];
        HIP_TRY(h, hipEventSynchronize(t->ev[EV_FIN]));
        for (int k = 0; k < 5; ++k) HIP_TRY(h, hipEventElapsedTime(&b[k], t->ev[k], t->ev[k + 1]));
        HIP_TRY(h, hipEventElapsedTime(&b[5], t->ev[EV_START], t->ev[EV_FIN]));
        for (int k = 0; k < 5; ++k) out6[k] = 0.5f * (out6[k] + b[k]);
        if (b[5] > out6[5]) out6[5] = b[5];
    }
    return PFC_OK;
}

int pfc_last_parts(pfc_handle h) { return h ? (h->multi ? pfc_last_parts(h->multi->shard[0]) : (h->last_fused ? 0 : h->last_parts)) : 0; }
int pfc_last_team(pfc_handle h) { return h ? (h->multi ? pfc_last_team(h->multi->shard[0]) : (h->last_fused ? h->last_fu_nw : 0)) : 0; }
int pfc_last_dual_reused(pfc_handle h) {
    if (!h) return 0;
    if (h->multi) {      // 1 only if every shard that took part reused its value pass
        for (int k = 0; k < h->multi->n_used; ++k)
            if (h->multi->bound[k + 1] > h->multi->bound[k] && !h->multi->shard[k]->last_dual_reused) return 0;
        return 1;
    }
    return h->last_dual_reused ? 1 : 0;
}

int pfc_debug_pairs(pfc_handle h, int item, int *pairs, int *clip_n, int cap) {
    if (!h) return -PFC_ERR_BAD_ARG;
    if (h->multi) {
        int local = 0;
        pfc_context *c = multi_locate(h, item, &local);
        if (!c) return -fail(h, PFC_ERR_BAD_ARG, "bad item");
        const int r = pfc_debug_pairs(c, local, pairs, clip_n, cap);
        if (r < 0) h->err = c->err;
        return r;
    }
    if (!h->opt_debug) return -fail(h, PFC_ERR_STATE, "debug option is off");
    if (h->pending) { int rc = check_eval(h); if (rc) return -rc; }
    if (item < 0 || item >= h->last_n_items) return -fail(h, PFC_ERR_BAD_ARG, "bad item");
    size_t nc = (size_t)h->stats[1];
    std::vector<WorkRec> c(nc);
    std::vector<int> cn(nc);
    if (nc) {
        if (copy_sync(h, c.data(), h->cand.p, sizeof(WorkRec) * nc, hipMemcpyDeviceToHost) != hipSuccess ||
            copy_sync(h, cn.data(), h->clip_n.p, sizeof(int) * nc, hipMemcpyDeviceToHost) != hipSuccess)
            return -fail(h, PFC_ERR_HIP, "copy failed");
    }
    int n = 0;
    for (size_t k = 0; k < nc; ++k)
        if (c[k].item == item) {
            if (n < cap) {
                if (pairs) { pairs[2 * n] = c[k].a; pairs[2 * n + 1] = c[k].b; }
                if (clip_n) clip_n[n] = cn[k];
            }
            ++n;
        }
    return n;
}

int pfc_debug_tractions(pfc_handle h, int item, double *buf, int cap) {
    if (!h) return -PFC_ERR_BAD_ARG;
    if (h->multi) {
        int local = 0;
        pfc_context *c = multi_locate(h, item, &local);
        if (!c) return -fail(h, PFC_ERR_BAD_ARG, "bad item");
        const int r = pfc_debug_tractions(c, local, buf, cap);
        if (r < 0) h->err = c->err;
        return r;
    }
    if (!h->opt_debug) return -fail(h, PFC_ERR_STATE, "debug option is off");
    if (h->pending) { int rc = check_eval(h); if (rc) return -rc; }
    if (item < 0 || item >= h->last_n_items) return -fail(h, PFC_ERR_BAD_ARG, "bad item");
    size_t nt = (size_t)h->last_tslots;
    std::vector<int> ti(nt);
    std::vector<double> td(nt * 8);
    if (nt) {
        if (copy_sync(h, ti.data(), h->trac_item.p, sizeof(int) * nt, hipMemcpyDeviceToHost) != hipSuccess)
            return -fail(h, PFC_ERR_HIP, "copy failed");
        for (int a = 0; a < 8; ++a)
            if (copy_sync(h, td.data() + a * nt, h->trac_d.p + a * h->tcap, sizeof(double) * nt, hipMemcpyDeviceToHost) != hipSuccess)
                return -fail(h, PFC_ERR_HIP, "copy failed");
    }
    int n = 0;
    for (size_t k = 0; k < nt; ++k)
        if (ti[k] == item) {
            if (n < cap && buf)
                for (int a = 0; a < 8; ++a) buf[8 * (size_t)n + a] = td[a * nt + k];
            ++n;
        }
    return n;
}

int pfc_debug_stiffness(pfc_handle h, int item, double *K36, double *Kis36, double *Sinv6, double *cop3) {
    if (!h) return -PFC_ERR_BAD_ARG;
    if (h->multi) {
        int local = 0;
        pfc_context *c = multi_locate(h, item, &local);
        if (!c) return -fail(h, PFC_ERR_BAD_ARG, "bad item");
        const int r = pfc_debug_stiffness(c, local, K36, Kis36, Sinv6, cop3);
        if (r < 0) h->err = c->err;
        return r;
    }
    if (h->pending) { int rc = check_eval(h); if (rc) return -rc; }
    if (h->last_fused)      // the fused small-scene kernel keeps K in LDS only
        return -fail(h, PFC_ERR_STATE, "pfc_debug_stiffness: set option debug (or fused = 0) before the evaluation");
    if (item < 0 || item >= h->last_n_items) return -fail(h, PFC_ERR_BAD_ARG, "bad item");
    std::vector<double> r(kResStride);
    int ic[4];
    if (copy_sync(h, r.data(), h->res.p + (size_t)item * kResStride, sizeof(double) * kResStride, hipMemcpyDeviceToHost) != hipSuccess ||
        copy_sync(h, ic, h->icnt.p + 4 * (size_t)item, sizeof ic, hipMemcpyDeviceToHost) != hipSuccess)
        return -fail(h, PFC_ERR_HIP, "copy failed");
    if (ic[3] == 0) return 0;
    if (K36) std::memcpy(K36, r.data() + kResK, sizeof(double) * 36);
    if (Kis36) std::memcpy(Kis36, r.data() + kResKis, sizeof(double) * 36);
    if (Sinv6) std::memcpy(Sinv6, r.data() + kResSinv, sizeof(double) * 6);
    if (cop3) std::memcpy(cop3, r.data() + kResCop, sizeof(double) * 3);
    return 1;
}

int pfc_scatter_generalized(pfc_handle h, int n_items, const double *wrench, const double *x_w_r2, const int *body_1,
                            const int *body_2, const int *scene, int n_scene, int n_body, int nv, const double *jac,
                            double *f_out) {
    if (h && h->multi) {      // a few microseconds of work: the first device does it
        const int rc = pfc_scatter_generalized(h->multi->shard[0], n_items, wrench, x_w_r2, body_1, body_2, scene, n_scene, n_body, nv, jac, f_out);
        if (rc != PFC_OK) h->err = h->multi->shard[0]->err;
        return rc;
    }
    if (!h || n_items < 0 || nv <= 0 || n_scene <= 0 || n_body < 0 || !f_out)
        return fail(h, PFC_ERR_BAD_ARG, "pfc_scatter_generalized: bad argument");
    if (n_items > 0 && (!wrench || !x_w_r2 || !body_1 || !body_2 || (n_body > 0 && !jac)))
        return fail(h, PFC_ERR_BAD_ARG, "pfc_scatter_generalized: null buffer");
    for (int i = 0; i < n_items; ++i) {
        if (body_1[i] >= n_body || body_2[i] >= n_body || (scene && (scene[i] < 0 || scene[i] >= n_scene)))
            return fail(h, PFC_ERR_BAD_ARG, "pfc_scatter_generalized: body / scene id out of range (item %d)", i);
    }
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t n = (size_t)n_items, nf = (size_t)n_scene * nv, nj = (size_t)n_body * 6 * nv;
    // work buffers of the handle (grown on demand, reused by later calls): [f | wrench | x_w_r2 | jac] and the ids
    HIP_TRY(h, h->scat_d.ensure(nf + n * 18 + nj + 1));
    HIP_TRY(h, h->scat_i.ensure(n * 3 + 1));
    double *df = h->scat_d.p, *dw = df + nf, *dx = dw + n * 6, *dj = dx + n * 12;
    int *db = h->scat_i.p;
    hipStream_t st = h->stream;
    HIP_TRY(h, hipMemsetAsync(df, 0, sizeof(double) * nf, st));
    if (n) {
        HIP_TRY(h, hipMemcpyAsync(dw, wrench, sizeof(double) * n * 6, hipMemcpyHostToDevice, st));
        HIP_TRY(h, hipMemcpyAsync(dx, x_w_r2, sizeof(double) * n * 12, hipMemcpyHostToDevice, st));
        if (nj) HIP_TRY(h, hipMemcpyAsync(dj, jac, sizeof(double) * nj, hipMemcpyHostToDevice, st));
        HIP_TRY(h, hipMemcpyAsync(db, body_1, sizeof(int) * n, hipMemcpyHostToDevice, st));
        HIP_TRY(h, hipMemcpyAsync(db + n, body_2, sizeof(int) * n, hipMemcpyHostToDevice, st));
        if (scene) HIP_TRY(h, hipMemcpyAsync(db + 2 * n, scene, sizeof(int) * n, hipMemcpyHostToDevice, st));
        ScatterArgs a;
        a.n_items = n_items; a.nv = nv; a.wrench = dw; a.x_w_r2 = dx; a.body_1 = db; a.body_2 = db + n;
        a.scene = scene ? db + 2 * n : nullptr; a.jac = dj; a.f = df;
        const long long tot = (long long)n_items * nv;
        hipLaunchKernelGGL(k_scatter, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, a);
        HIP_TRY(h, hipGetLastError());
    }
    HIP_TRY(h, hipMemcpyAsync(f_out, df, sizeof(double) * nf, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return PFC_OK;
}

int pfc_scatter_generalized_device(pfc_handle h, int n_items, const double *d_wrench, const double *d_x_w_r2, const int *d_body_1,
                                   const int *d_body_2, const int *d_scene, int n_scene, int nv, const double *d_jac, double *d_f,
                                   int accumulate, void *stream) {
    if (h && h->multi) {      // the wrenches of a multi-device evaluation end up on the first device: so does this
        const int rc = pfc_scatter_generalized_device(h->multi->shard[0], n_items, d_wrench, d_x_w_r2, d_body_1, d_body_2, d_scene, n_scene, nv,
                                                      d_jac, d_f, accumulate, stream);
        if (rc != PFC_OK) h->err = h->multi->shard[0]->err;
        return rc;
    }
    if (!h || n_items < 0 || nv <= 0 || n_scene <= 0 || !d_f)
        return fail(h, PFC_ERR_BAD_ARG, "pfc_scatter_generalized_device: bad argument");
    if (n_items > 0 && (!d_wrench || !d_x_w_r2 || !d_body_1 || !d_body_2 || !d_jac))
        return fail(h, PFC_ERR_BAD_ARG, "pfc_scatter_generalized_device: null buffer");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    if (!accumulate) HIP_TRY(h, hipMemsetAsync(d_f, 0, sizeof(double) * (size_t)n_scene * nv, st));
    if (n_items > 0) {
        ScatterArgs a;
        a.n_items = n_items; a.nv = nv; a.wrench = d_wrench; a.x_w_r2 = d_x_w_r2; a.body_1 = d_body_1; a.body_2 = d_body_2;
        a.scene = d_scene; a.jac = d_jac; a.f = d_f;
        const long long tot = (long long)n_items * nv;
        hipLaunchKernelGGL(k_scatter, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, a);
        HIP_TRY(h, hipGetLastError());
    }
    return PFC_OK;
}

int pfc_debug_stamps(pfc_handle h, long long *out16) {
    if (!h || !out16) return PFC_ERR_BAD_ARG;
    if (h->multi) return pfc_debug_stamps(h->multi->shard[0], out16);
    if (h->pending) { int rc = check_eval(h); if (rc) return rc; }
    unsigned long long v[16] = {0};
    if (h->stamps.p) HIP_TRY(h, copy_sync(h, v, h->stamps.p, sizeof v, hipMemcpyDeviceToHost));
    for (int k = 0; k < 16; ++k) out16[k] = (long long)v[k];
    out16[7] = h->last_undecided;
#ifndef PFC_STAMPS
    out16[6] = h->twin_queue_fallback;      // make_twin's finding about the two streams (statistics view of the product build)
#endif
    return PFC_OK;
}

int pfc_selftest_math(pfc_handle h, int n, const double *x, const double *y, double *out3n) {
    if (!h || n <= 0 || !x || !y || !out3n) return PFC_ERR_BAD_ARG;
    if (h->multi) return pfc_selftest_math(h->multi->shard[0], n, x, y, out3n);
    HIP_TRY(h, hipSetDevice(h->device));
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    HIP_TRY(h, hipMalloc((void **)&dx, sizeof(double) * n));
    HIP_TRY(h, hipMalloc((void **)&dy, sizeof(double) * n));
    HIP_TRY(h, hipMalloc((void **)&dout, sizeof(double) * 3 * (size_t)n));
    HIP_TRY(h, copy_sync(h, dx, x, sizeof(double) * n, hipMemcpyHostToDevice));
    HIP_TRY(h, copy_sync(h, dy, y, sizeof(double) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_selftest, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, dx, dy, dout);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, copy_sync(h, out3n, dout, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dout);
    return PFC_OK;
}

}  // extern "C"
