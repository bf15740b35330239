// pfc_hip.hip — kernels and C ABI of libpfc_hip (MI355X / gfx950).  See include/pfc.h and DESIGN.md.
//
// Evaluation pipeline (one stream, no host synchronisation between stages):
//   k_setup_items   per item: ItemRec from (instruction, pose, twist, s); seeds the broadphase frontier with the
//                   (root, root) node pair; clears the per-item accumulators
//   k_bp_expand     level-synchronous dual-tree descent over ALL items at once: one lane per (item, node_a,
//                   node_b) frontier entry, 15-axis SAT, children / leaf pairs appended with wave-aggregated
//                   (ballot + mbcnt prefix) atomics                      [tree_tree_intersect, tree_types.jl:88]
//   k_narrow        one lane per candidate (triangle, tet) pair: gather 96 B + 256 B records, transform to tet
//                   coordinates, trivial reject in registers, Sutherland-Hodgman clip with the polygon staged in
//                   LDS ([slot][coord][lane] layout: conflict-free per-lane dynamic indexing), fan quadrature,
//                   pressure; regularized friction fused; bristle items accumulate the patch moments
//   k_eig           the bristle model: cop + patch stiffness from the origin moments, 6x6 symmetric eigen solve
//   k_narrow<1,..>  bristle friction pass: clip + quadrature recomputed, calc_spatial_bristle_force integrated
//   k_final         per item: wrench, sdot, counts
#include "pfc_kernels.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pfc.h"

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

__global__ void k_setup_items(EvalArgs g) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n_items) return;
    int id = g.ins_ids ? g.ins_ids[i] : i;
    ItemRec r;
    if (id < 0 || id >= g.n_ins) {
        atomicOr(g.status, kStBadIns);
        id = 0;
    }
    const InsDev in = g.ins[id];
    const MeshDev m1 = g.meshes[in.m1], m2 = g.meshes[in.m2];
    const double *p = g.pose + 24 * (size_t)i;
#pragma unroll
    for (int k = 0; k < 9; ++k) { r.R21[k] = p[k]; r.R12[k] = p[12 + k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r.t21[k] = p[9 + k]; r.t12[k] = p[21 + k];
        r.w[k] = g.twist[6 * (size_t)i + k]; r.v[k] = g.twist[6 * (size_t)i + 3 + k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) r.s[k] = (g.s && in.model == PFC_BRISTLE) ? g.s[6 * (size_t)i + k] : 0.0;
    r.chi = in.chi; r.Ebar = m2.Ebar;  // Ē of mesh_2 only: non_friction.jl:131
    r.mu_s = in.mu_s; r.mu_d = in.mu_d; r.v_c = in.v_c; r.tau = in.tau; r.k_bar = in.k_bar; r.magic = in.magic;
    r.nodes1 = m1.nodes; r.nodes2 = m2.nodes; r.nf1 = m1.nodesf; r.nf2 = m2.nodesf; r.tri = m1.tri; r.tet = m2.tet;
    r.tet1 = m1.tri ? nullptr : m1.tet; r.eps1 = m1.tet_eps; r.eps2 = m2.tet_eps; r.Ebar1 = m1.Ebar;
    r.model = in.model; r.nq = (in.nq == 1) ? 1 : 3;  // quadrature POINTS of rule 1 / rule 2 (quadrature.jl:22,31)
    r.ins = id; r.pad = 0;
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 24; ++k) finite &= (__builtin_fabs(p[k]) <= 1.79769313486231570815e308);
    if (!finite) atomicOr(g.status, kStNonFinite);
    g.items[i] = r;
    WorkRec w;
    w.item = i; w.a = 0; w.b = 0; w.pad = 0;
    g.frontier0[i] = w;
    for (int k = 0; k < kAccStride; ++k) g.acc[(size_t)i * kAccStride + k] = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) g.icnt[4 * (size_t)i + k] = 0;
    if (i == 0) g.fcount[0] = g.n_items;
}

// =================================================================================================================
// broadphase: one level of the simultaneous descent (src/obb/tree_types.jl:88-111)
// =================================================================================================================
struct BpArgs {
    const ItemRec *items;
    const WorkRec *fin;
    WorkRec *fout;
    WorkRec *cand;
    int *fcount;     // fcount[level] = size of fin, fcount[level + 1] accumulates the size of fout
    int *ccount;     // candidate counter
    int *icnt;
    unsigned *status;
    int level, fcap, ccap;
};

__device__ __forceinline__ void count_per_item(int *icnt, int item, int slot, bool listed, bool flag, int n = 1) {
    // per-item integer counter: one atomic per run of equal items in the wave
    if (__ballot(flag) == 0) return;
    const Seg sg = seg_setup(listed ? item : -1);
    const int t = seg_sum(flag ? n : 0, sg);
    if (sg.tail && sg.valid && t != 0) atomicAdd(&icnt[4 * (size_t)item + slot], t);
}

__global__ void __launch_bounds__(256) k_bp_expand(BpArgs g) {
    int n_in = g.fcount[g.level];
    if (n_in > g.fcap) n_in = g.fcap;  // the previous level overflowed (flagged there); never read past the buffer
    const int stride = gridDim.x * blockDim.x;
    const int lane = lane_id();
    // every lane of a wave runs the same number of iterations so the wave-level ballots are well defined
    const int n_round = (n_in + stride - 1) / stride;
    for (int rd = 0; rd < n_round; ++rd) {
        int idx = rd * stride + blockIdx.x * blockDim.x + threadIdx.x;
        bool active = idx < n_in;
        WorkRec w;
        w.item = 0; w.a = 0; w.b = 0; w.pad = 0;
        bool hit = false, la = false, lb = false;
        int ca0 = 0, ca1 = 0, cb0 = 0, cb1 = 0, leaf_a = 0, leaf_b = 0;
        if (active) {
            w = g.fin[idx];
            const ItemRec *it = g.items + w.item;
            const NodeRec a = it->nodes1[w.a];
            const NodeRec b = it->nodes2[w.b];
            hit = bb_bb_intersect(a, b, it->R12, it->t12);
            la = a.leaf != kInternal; lb = b.leaf != kInternal;
            ca0 = node_index(a.child0); ca1 = node_index(a.child1); cb0 = node_index(b.child0); cb1 = node_index(b.child1);
            leaf_a = a.leaf; leaf_b = b.leaf;
        }
        count_per_item(g.icnt, w.item, 0, active, active);
        bool is_cand = hit && la && lb;
        bool two = hit && (la != lb);
        bool four = hit && !la && !lb;
        unsigned long long mc = __ballot(is_cand), m2 = __ballot(two), m4 = __ballot(four);
        // candidates
        if (mc) {
            int base = 0;
            if (lane == 0) base = atomicAdd(g.ccount, __builtin_popcountll(mc));
            base = __shfl(base, 0, 64);
            if (is_cand) {
                int pos = base + prefix_count(mc);
                if (pos < g.ccap) {
                    WorkRec c;
                    c.item = w.item; c.a = leaf_a; c.b = leaf_b; c.pad = 0;
                    g.cand[pos] = c;
                } else {
                    atomicOr(g.status, kStCandOvf);
                }
            }
            count_per_item(g.icnt, w.item, 1, active, is_cand);
        }
        // children
        if (m2 | m4) {
            int tot = 2 * __builtin_popcountll(m2) + 4 * __builtin_popcountll(m4);
            int base = 0;
            if (lane == 0) base = atomicAdd(&g.fcount[g.level + 1], tot);
            base = __shfl(base, 0, 64);
            int pos = base + 2 * prefix_count(m2) + 4 * prefix_count(m4);
            int nout = two ? 2 : (four ? 4 : 0);
            if (nout) {
                if (pos + nout <= g.fcap) {
                    WorkRec c;
                    c.item = w.item; c.pad = 0;
                    if (two) {
                        if (la) {  // leaf_1: descend tree_2 (:97-98)
                            c.a = w.a; c.b = cb0; g.fout[pos] = c;
                            c.b = cb1; g.fout[pos + 1] = c;
                        } else {   // leaf_2: descend tree_1 (:101-103)
                            c.b = w.b; c.a = ca0; g.fout[pos] = c;
                            c.a = ca1; g.fout[pos + 1] = c;
                        }
                    } else {       // (1.1,2.1) (1.2,2.1) (1.1,2.2) (1.2,2.2) (:104-107)
                        c.a = ca0; c.b = cb0; g.fout[pos] = c;
                        c.a = ca1; c.b = cb0; g.fout[pos + 1] = c;
                        c.a = ca0; c.b = cb1; g.fout[pos + 2] = c;
                        c.a = ca1; c.b = cb1; g.fout[pos + 3] = c;
                    }
                } else {
                    atomicOr(g.status, kStFrontierOvf);
                }
            }
        }
    }
}

// =================================================================================================================
// broadphase, deep part: one wave per seed node pair, cooperative depth-first descent with the work stack in LDS.
// Same node-pair tests as the recursion of tree_tree_intersect (src/obb/tree_types.jl:88-111), 64 at a time: each
// iteration pops up to 64 node pairs from the top of the stack (one per lane), runs the SAT, and pushes the 2 or 4
// child pairs / stages the leaf-leaf candidates with ballot + mbcnt prefix sums.  No global frontier, no global
// atomics per test: a seed's candidates leave in runs of up to kDfsOut records (one atomic per flush), which also
// keeps the candidate list grouped by item for the reductions downstream.
// =================================================================================================================
// In-kernel phase stamps (diagnostic builds only: -DPFC_STAMPS).  s_memtime ticks = shader cycles; the sums go to a
// buffer of their own that no kernel reads (MI355X guide: 'In-kernel stamps').
#ifdef PFC_STAMPS
#define STAMP(t)                                                                 \
    do {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                       \
    } while (0)
#else
#define STAMP(t) do { } while (0)
#endif

constexpr int kDfsStack = 1024;  // node pairs per wave (8 KiB)
constexpr int kDfsOut = 320;     // staged candidates per wave (2.5 KiB)

struct DfsArgs {
    const ItemRec *items;
    const WorkRec *seeds;
    const int *n_seed;   // device counter
    int *next_seed;      // device counter (zeroed per evaluation): dynamic seed queue head
    int seed_cap;
    WorkRec *cand;
    int *ccount;
    int ccap;
    int *icnt;
    unsigned *status;
    int reserve;         // 3 * (max remaining depth) + 3 slots kept free for the pure depth-first mode
    unsigned long long *stamps;  // diagnostic builds: [8..12] cycles in pop+load / SAT / push+flush, iterations, lanes
    int no_filter;       // 1: skip the FP32 filter (every pair runs the Float64 test)
};

// one ticket per wave from a device-wide counter: lane 0 takes it, the wave reads it back as a scalar
__device__ __forceinline__ int next_ticket(int *ctr) {
    int t = 0;
    if (lane_id() == 0) t = atomicAdd(ctr, 1);
    return __builtin_amdgcn_readfirstlane(t);
}

// =================================================================================================================
// broadphase main kernel: the same wave-cooperative depth-first descent as k_bp_dfs, in single precision on one
// 64-byte NodeF line per node.  Measured (in-kernel stamps): the Float64 kernel is latency-bound at 2 waves per SIMD
// (246 VGPRs, two 144-byte scattered records per lane and iteration), not ALU-bound; the Float32 kernel needs a
// third of the registers and a third of the cache-line requests, and is exact in the following sense.
//
// For a node pair it forms v = R_a_b c_b + (t_a_b - c_a) in Float64 (the centre offset, of the order of the box
// sizes), everything else in Float32: R_a, R_b from unit quaternions (|err| <= 8 u per entry, u = 2^-24, checked on
// the host when the quaternion is made), T = R_a' R_a_b, t = R_a' v, R = T R_b, then the 15 axes
// d = |T.L| - (r_a + r_b).  The error of R is below 72 u, so |d_float - d_reference| < 192 u S with
// S = |v|_1 + sum e_a + sum e_b (internal-internal pairs carry no quaternion error: 16 u S).  d > E proves
// separation, d < -E on all 15 axes proves overlap; an undecided pair (~1e-5 of its margin scale) is NOT decided by
// this test: it is parked in LDS and settled at the top of the next iteration by the exact Float64 test.  The
// candidate set and the node-test counts therefore equal the reference's bit for bit (tests/test_gpu_*.py).
// =================================================================================================================
struct Dfs32Args {
    const ItemRec *items;
    const WorkRec *seeds;
    const int *n_seed;
    int *next_seed;
    int seed_cap;
    WorkRec *cand;
    int *ccount;
    int ccap;
    int *ucount;         // statistics: node pairs settled by the exact Float64 test
    int *icnt;
    unsigned *status;
    int reserve;
};

__device__ __forceinline__ NodeF load_nodef(const NodeF *n) {
    // four 16-byte loads of one 64-byte line
    const int4 *p = reinterpret_cast<const int4 *>(n);
    union { int4 v[4]; NodeF f; } u;
    u.v[0] = p[0]; u.v[1] = p[1]; u.v[2] = p[2]; u.v[3] = p[3];
    return u.f;
}

// one node pair of k_bp_dfs32: returns 0 = separated, 1 = overlapping, 2 = undecided
__device__ __forceinline__ int test_pair_f32(const NodeF &a, const NodeF &b, bool any_leaf, const double *R12,
                                             const float *R12f, const double *t12) {
    // centre offset in Float64, then everything in Float32
    float v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        v[i] = (float)(((R12[i] * b.c[0] + R12[i + 3] * b.c[1]) + R12[i + 6] * b.c[2]) + (t12[i] - a.c[i]));
    float Ra[9], Rb[9], T[9], R[9], t[3];
    quat_to_R(a.q, Ra);
    quat_to_R(b.q, Rb);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float r0 = Ra[3 * i], r1 = Ra[3 * i + 1], r2 = Ra[3 * i + 2];   // row i of R_a' = column i of R_a
#pragma unroll
        for (int j = 0; j < 3; ++j)
            T[i + 3 * j] = __builtin_fmaf(r2, R12f[3 * j + 2], __builtin_fmaf(r1, R12f[3 * j + 1], r0 * R12f[3 * j]));
        t[i] = __builtin_fmaf(r2, v[2], __builtin_fmaf(r1, v[1], r0 * v[0]));
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            R[i + 3 * j] = __builtin_fmaf(T[i + 6], Rb[3 * j + 2], __builtin_fmaf(T[i + 3], Rb[3 * j + 1], T[i] * Rb[3 * j]));
    const float S = ((__builtin_fabsf(v[0]) + __builtin_fabsf(v[1])) + __builtin_fabsf(v[2])) +
                    ((a.e[0] + a.e[1]) + a.e[2]) + ((b.e[0] + b.e[1]) + b.e[2]);
    const float E = (any_leaf ? 1.15e-5f : 9.6e-7f) * S;   // 192 u, 16 u
    int verdict = sat15_f32_core(a.e, b.e, t, R, E);
    if (a.exact_only | b.exact_only) verdict = 2;
    return verdict;
}

constexpr int kDfsBlock = 256;
constexpr int kDfsWaves = kDfsBlock / 64;
constexpr int kDfsStack32 = 2560;   // node pairs per workgroup (20 KiB)
constexpr int kDfsOut32 = 1280;     // staged candidates per workgroup (10 KiB)

// The exact Float64 BB_BB_intersect (general composition, src/obb/bb_intersection.jl:2-74) of the node pairs the
// Float32 test leaves undecided (~2e-5 of all node tests), evaluated COOPERATIVELY: 16 lanes per pair.  Lanes 0..8
// form one entry each of R_a' R_a_b, then of R_tot = (R_a' R_a_b) R_b (lanes 9..11 the translation), through LDS, and
// lanes 0..14 test one of the 15 axes each.  Every entry / axis is the same expression, in the same order, as in
// bb_compose() / sat15(), so the boolean is the reference's bit for bit, and the per-lane register need is a few
// dozen instead of the ~220 of the one-lane-per-pair Float64 test (which cost the kernel a third of its occupancy
// when inlined, and as a no-inline call needed scratch).  xs: 33 doubles per 16-lane group.
__device__ __forceinline__ void exact_pairs_coop(const ItemRec *it, const double *pose, const int2 *und_l, int n_def,
                                                 double *xs, int *und_v, int tid) {
    const int grp = tid >> 4, sub = tid & 15;
    double *T = xs + grp * 33, *tt = T + 9, *R = T + 12, *aR = T + 21, *t = T + 30;
    for (int c0 = 0; c0 < n_def; c0 += kDfsBlock / 16) {
        const int j = c0 + grp;
        const bool valid = j < n_def;
        int2 e = make_int2(0, 0);
        if (valid) e = und_l[j];
        const NodeRec *na = it->nodes1 + node_index(e.x), *nb = it->nodes2 + node_index(e.y);
        if (valid && sub < 9) {
            const int i = sub % 3, jj = sub / 3;
            const double r0 = na->R[3 * i], r1 = na->R[3 * i + 1], r2 = na->R[3 * i + 2];
            T[i + 3 * jj] = (r0 * pose[3 * jj] + r1 * pose[3 * jj + 1]) + r2 * pose[3 * jj + 2];
        } else if (valid && sub < 12) {
            const int i = sub - 9;
            const double r0 = na->R[3 * i], r1 = na->R[3 * i + 1], r2 = na->R[3 * i + 2];
            const double nt = ((-r0) * na->c[0] + (-r1) * na->c[1]) + (-r2) * na->c[2];
            tt[i] = ((r0 * pose[9] + r1 * pose[10]) + r2 * pose[11]) + nt;
        }
        __syncthreads();
        if (valid && sub < 9) {
            const int i = sub % 3, jj = sub / 3;
            const double r = (T[i] * nb->R[3 * jj] + T[i + 3] * nb->R[3 * jj + 1]) + T[i + 6] * nb->R[3 * jj + 2];
            R[i + 3 * jj] = r;
            aR[i + 3 * jj] = __builtin_fabs(r) + 1.0e-14;
        } else if (valid && sub < 12) {
            const int i = sub - 9;
            t[i] = ((T[i] * nb->c[0] + T[i + 3] * nb->c[1]) + T[i + 6] * nb->c[2]) + tt[i];
        }
        __syncthreads();
        bool sep = false;
        if (valid && sub < 15) {
            const double ea[3] = {na->e[0], na->e[1], na->e[2]}, eb[3] = {nb->e[0], nb->e[1], nb->e[2]};
#define R_(i, j) R[(i) + 3 * (j)]
#define AR_(i, j) aR[(i) + 3 * (j)]
            if (sub < 3) {          // face test 1/2 (:29-32)
                const int i = sub;
                const double rb = (AR_(i, 0) * eb[0] + AR_(i, 1) * eb[1]) + AR_(i, 2) * eb[2];
                sep = (ea[i] + rb) < __builtin_fabs(t[i]);
            } else if (sub < 6) {   // face test 2/2 (:35-38)
                const int jj = sub - 3;
                const double tl = __builtin_fabs((R_(0, jj) * t[0] + R_(1, jj) * t[1]) + R_(2, jj) * t[2]);
                const double ra = (AR_(0, jj) * ea[0] + AR_(1, jj) * ea[1]) + AR_(2, jj) * ea[2];
                sep = (ra + eb[jj]) < tl;
            } else {                // cross tests (:56-72): row m of the cross block, column jj
                const int m = (sub - 6) / 3, jj = (sub - 6) % 3;
                const int u = (m + 1) % 3, v = (m + 2) % 3;
                const int p100 = jj == 0 ? 1 : 0, p221 = jj == 2 ? 1 : 2;
                const double tl = __builtin_fabs(t[v] * R_(u, jj) - t[u] * R_(v, jj));
                // sat15 writes the two products of ra with the lower axis index first; a + b == b + a exactly
                const double ra = ea[u] * AR_(v, jj) + ea[v] * AR_(u, jj);
                const double rb = eb[p100] * AR_(m, p221) + eb[p221] * AR_(m, p100);
                sep = (ra + rb) < tl;
            }
#undef R_
#undef AR_
        }
        const unsigned long long ms = __ballot(sep);
        if (valid && sub == 0) und_v[j] = ((ms >> ((tid & 63) & ~15)) & 0xFFFFull) ? 0 : 1;
        __syncthreads();
    }
}

// One WORKGROUP (4 waves) per seed, one shared LDS stack: a seed of the 2 048-pose C3 batch is ~45 k node tests, i.e.
// ~700 dependent iterations for a single wave -- that serial chain, not ALU or memory, bounded the one-wave-per-seed
// version (every variant of its inner loop ran 2.0 ms).  Four waves pop 256 pairs per iteration from the same stack.

// one ticket per workgroup: thread 0 takes it, LDS broadcast between two barriers
__device__ __forceinline__ int next_ticket_block(int *ctr, int *slot) {
    if (threadIdx.x == 0) *slot = atomicAdd(ctr, 1);
    __syncthreads();
    const int t = *slot;
    __syncthreads();
    return t;
}

__device__ __forceinline__ void flush_candidates(const Dfs32Args &g, const int2 *ob, int n_out, int item, int tid,
                                                 int *s_base) {
    // all threads of the workgroup call this (n_out is uniform)
    if (tid == 0) *s_base = atomicAdd(g.ccount, n_out);
    __syncthreads();
    const int base = *s_base;
    if (base + n_out <= g.ccap) {
        for (int j = tid; j < n_out; j += kDfsBlock) {
            WorkRec c;
            c.item = item; c.a = ob[j].x; c.b = ob[j].y; c.pad = 0;
            g.cand[base + j] = c;
        }
    } else if (tid == 0) {
        atomicOr(g.status, kStCandOvf);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(kDfsBlock, 4) k_bp_dfs32(Dfs32Args g) {
    __shared__ int2 stk[kDfsStack32];
    __shared__ int2 ob[kDfsOut32];
    __shared__ int s_cnt[kDfsWaves][2];   // per wave: candidates, pushed pairs of the current iteration
    __shared__ int2 und_l[kDfsBlock];     // node pairs the Float32 test left undecided in the last iteration
    __shared__ int und_v[kDfsBlock];      // their exact verdicts
    __shared__ double xs[(kDfsBlock / 16) * 33];
    __shared__ int s_seed, s_base, s_def;
    __shared__ double s_pose[12];         // R_a_b (9, column-major), t_a_b (3) of the current seed's item
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int n_seed = *g.n_seed;
    if (n_seed > g.seed_cap) n_seed = g.seed_cap;
    // one ticket per workgroup (same loop shape as k_bp_dfs: condition in the for header, no break)
    for (int sd = next_ticket_block(g.next_seed, &s_seed); sd < n_seed; sd = next_ticket_block(g.next_seed, &s_seed)) {
        const WorkRec s = g.seeds[__builtin_amdgcn_readfirstlane(sd)];
        const int item = __builtin_amdgcn_readfirstlane(s.item);   // uniform: scalar loads of the pose below
        const ItemRec *it = g.items + item;
        // the item's pose lives in LDS (broadcast reads inside the iteration) rather than in 33 registers that would
        // stay live across the call of the exact test
        if (tid < 9) s_pose[tid] = it->R12[tid];
        else if (tid < 12) s_pose[tid] = it->t12[tid - 9];
        const NodeF *n1 = it->nf1, *n2 = it->nf2;
        int sp = 1, n_out = 0, n_test = 0, n_cand = 0, n_def = 0, n_und = 0;
        if (tid == 0) {
            // stack entries hold node links: ~index (negative) for a leaf, index for an internal node
            const int sa = (it->nodes1[s.a].leaf != kInternal) ? ~s.a : s.a;
            const int sb = (it->nodes2[s.b].leaf != kInternal) ? ~s.b : s.b;
            stk[0] = make_int2(sa, sb);
            s_def = 0;
        }
        __syncthreads();
        // every workgroup must reach its exit: the iteration guard stops a corrupt (cyclic) tree from spinning forever
        for (int guard = 0; (sp > 0 || n_def > 0) && guard < (1 << 22); ++guard) {
            // Either settle the pairs the previous iteration left undecided (exact Float64 test; their children still
            // have the room that iteration reserved for them), or pop up to 256 pairs, but never more than the stack can
            // take back as children (4 per pair).  n_def is uniform over the workgroup.
            const bool settle = n_def > 0;
            int pw = (kDfsStack32 - g.reserve - sp) / 3;
            int p = sp < kDfsBlock ? sp : kDfsBlock;
            if (pw < 1) pw = 1;
            if (p > pw) p = pw;
            if (settle) {
                p = n_def;
                exact_pairs_coop(it, s_pose, und_l, n_def, xs, und_v, tid);   // ends with a barrier
            }
            const bool act = tid < p;
            int2 e = make_int2(0, 0);
            if (act) e = settle ? und_l[tid] : stk[sp - 1 - tid];
            __syncthreads();
            if (!settle) { sp -= p; n_test += p; } else { n_und += p; }
            int verdict = 0, a0 = 0, a1 = 0, b0 = 0, b1 = 0;
            if (settle && act) verdict = und_v[tid];
            const bool la = act && e.x < 0, lb = act && e.y < 0;
            if (act) {
                const NodeF a = load_nodef(n1 + node_index(e.x));
                const NodeF b = load_nodef(n2 + node_index(e.y));
                a0 = a.link0; a1 = a.link1; b0 = b.link0; b1 = b.link1;
                if (!settle) {
                    double R12[9], t12[3];
                    float R12f[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k) { R12[k] = s_pose[k]; R12f[k] = (float)R12[k]; }
#pragma unroll
                    for (int k = 0; k < 3; ++k) t12[k] = s_pose[9 + k];
                    verdict = test_pair_f32(a, b, la || lb, R12, R12f, t12);
                }
            }
            // an undecided pair is parked for the next iteration (verdict 2 only comes from the Float32 test)
            if (verdict == 2) und_l[atomicAdd(&s_def, 1)] = e;
            const bool hit = verdict == 1;
            const bool is_cand = hit && la && lb;
            const bool two = hit && (la != lb);
            const bool four = hit && !la && !lb;
            const unsigned long long mc = __ballot(is_cand), m2 = __ballot(two), m4 = __ballot(four);
            if (lane == 0) {
                s_cnt[wave][0] = __builtin_popcountll(mc);
                s_cnt[wave][1] = 2 * __builtin_popcountll(m2) + 4 * __builtin_popcountll(m4);
            }
            __syncthreads();
            n_def = s_def;   // read by everyone between this barrier and the next; reset after the next
            int c_off = 0, p_off = 0, c_tot = 0, p_tot = 0;
#pragma unroll
            for (int w = 0; w < kDfsWaves; ++w) {
                const int c = s_cnt[w][0], q = s_cnt[w][1];
                if (w < wave) { c_off += c; p_off += q; }
                c_tot += c; p_tot += q;
            }
            if (is_cand) ob[n_out + c_off + prefix_count(mc)] = make_int2(a0, b0);   // leaf: link0 = element index
            if (two | four) {
                const int pos = sp + p_off + 2 * prefix_count(m2) + 4 * prefix_count(m4);
                if (two) {
                    if (la) {  // leaf_1: descend tree_2 (:97-98)
                        stk[pos] = make_int2(e.x, b0); stk[pos + 1] = make_int2(e.x, b1);
                    } else {   // leaf_2: descend tree_1 (:101-103)
                        stk[pos] = make_int2(a0, e.y); stk[pos + 1] = make_int2(a1, e.y);
                    }
                } else {       // (1.1,2.1) (1.2,2.1) (1.1,2.2) (1.2,2.2) (:104-107)
                    stk[pos] = make_int2(a0, b0); stk[pos + 1] = make_int2(a1, b0);
                    stk[pos + 2] = make_int2(a0, b1); stk[pos + 3] = make_int2(a1, b1);
                }
            }
            n_out += c_tot;
            sp += p_tot;
            __syncthreads();
            if (tid == 0) s_def = 0;   // ordered before the next iteration's parking by its first barrier
            if (n_out > kDfsOut32 - kDfsBlock || (sp == 0 && n_def == 0 && n_out > 0)) {
                flush_candidates(g, ob, n_out, item, tid, &s_base);
                n_cand += n_out;
                n_out = 0;
            }
        }
        if (tid == 0) {
            if (sp > 0 || n_def > 0) atomicOr(g.status, kStAbort);
            atomicAdd(&g.icnt[4 * (size_t)item], n_test);
            if (n_cand) atomicAdd(&g.icnt[4 * (size_t)item + 1], n_cand);
            if (n_und) atomicAdd(g.ucount, n_und);   // statistics
        }
    }
}

// first 64 bytes of a NodeRec (c, e, links, flags) as four 16-byte loads
struct NodeHead {
    double c[3], e[3];
    int child0, child1, leaf, aabb;
};
__device__ __forceinline__ NodeHead load_head(const NodeRec *n) {
    NodeHead h;
    const double2 *p = reinterpret_cast<const double2 *>(n);
    const double2 a = p[0], b = p[1], c = p[2];
    const int4 l = reinterpret_cast<const int4 *>(n)[3];
    h.c[0] = a.x; h.c[1] = a.y; h.c[2] = b.x; h.e[0] = b.y; h.e[1] = c.x; h.e[2] = c.y;
    h.child0 = l.x; h.child1 = l.y; h.leaf = l.z; h.aabb = l.w;
    return h;
}

__global__ void __launch_bounds__(64) k_bp_dfs(DfsArgs g) {
    __shared__ int2 stk[kDfsStack];
    __shared__ int2 ob[kDfsOut];
    const int lane = threadIdx.x;
    int n_seed = *g.n_seed;
    if (n_seed > g.seed_cap) n_seed = g.seed_cap;
    // dynamic seed queue: seeds differ in work by orders of magnitude (most of a contact lives in one subtree)
    for (int sd = next_ticket(g.next_seed); sd < n_seed; sd = next_ticket(g.next_seed)) {
        const WorkRec s = g.seeds[sd];
        const int item = __builtin_amdgcn_readfirstlane(s.item);
        const ItemRec *it = g.items + item;
        double R12[9], aR12[9], t12[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) { R12[k] = it->R12[k]; aR12[k] = __builtin_fabs(R12[k]) + 1.0e-14; }
#pragma unroll
        for (int k = 0; k < 3; ++k) t12[k] = it->t12[k];
        const NodeRec *n1 = it->nodes1, *n2 = it->nodes2;
        int sp = 1, n_out = 0, n_test = 0, n_cand = 0;
        // stack entries hold node links: ~index (negative) for a leaf, index for an internal node
        if (lane == 0) {
            const int sa = (n1[s.a].leaf != kInternal) ? ~s.a : s.a;
            const int sb = (n2[s.b].leaf != kInternal) ? ~s.b : s.b;
            stk[0] = make_int2(sa, sb);
        }
        __syncthreads();
#ifdef PFC_STAMPS
        unsigned long long c_load = 0, c_sat = 0, c_push = 0, c_iter = 0, c_lanes = 0;
#endif
        // every wave must reach its exit: the iteration guard stops a corrupt (cyclic) tree from spinning forever
        for (int guard = 0; sp > 0 && guard < (1 << 22); ++guard) {
            unsigned long long u0 = 0, u1 = 0, u2 = 0, u3 = 0;
            (void)u0; (void)u1; (void)u2; (void)u3;
            STAMP(u0);
            // wide mode while there is room for 4 children per popped pair above the depth-first reserve
            int pw = (kDfsStack - g.reserve - sp) / 3;
            int p = sp < 64 ? sp : 64;
            if (pw < 1) pw = 1;
            if (p > pw) p = pw;
            const bool act = lane < p;
            int2 e = make_int2(0, 0);
            if (act) e = stk[sp - 1 - lane];
            __syncthreads();
            sp -= p;
            n_test += p;
            const bool la = act && e.x < 0, lb = act && e.y < 0;
            const int ia = node_index(e.x), ib = node_index(e.y);
            // The path is chosen per wave, never per lane: popped pairs sit at similar depths, so a wave is usually
            // all internal-internal (axis-aligned shortcut: R_tot = R_a_b, 128 bytes per pair) or reaches the
            // tight-fitted leaves together (general composition; exact for identity rotations too).  Leaf-ness comes
            // with the link, so all loads of the iteration are issued before the first use.
            const bool general = __any(la || lb);
            NodeHead a, b;
            a.leaf = kInternal; b.leaf = kInternal; a.child0 = a.child1 = b.child0 = b.child1 = 0; a.aabb = b.aabb = 1;
#pragma unroll
            for (int k = 0; k < 3; ++k) { a.c[k] = a.e[k] = b.c[k] = b.e[k] = 0.0; }
            double Ra[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0}, Rb[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0};
            if (act) {
                a = load_head(n1 + ia);
                b = load_head(n2 + ib);
                if (general) {
                    if (la) {
#pragma unroll
                        for (int k = 0; k < 9; ++k) Ra[k] = n1[ia].R[k];
                    }
                    if (lb) {
#pragma unroll
                        for (int k = 0; k < 9; ++k) Rb[k] = n2[ib].R[k];
                    }
                }
            }
#ifdef PFC_STAMPS
            { double keep = a.c[0] + b.c[0] + Ra[4] + Rb[4]; asm volatile("" ::"v"(keep)); }   // the loads have landed
#endif
            STAMP(u1);
            // The Float64 composition gives R_tot and t; the 15 axes are decided by the single-precision filter
            // (sat15_f32) and only undecided pairs (within ~1e-6 of touching) run the Float64 test.
            double Rt[9], aRt[9], tt3[3];
            if (!general) {
#pragma unroll
                for (int k = 0; k < 9; ++k) { Rt[k] = R12[k]; aRt[k] = aR12[k]; }
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    tt3[i] = ((R12[i] * b.c[0] + R12[i + 3] * b.c[1]) + R12[i + 6] * b.c[2]) + (t12[i] - a.c[i]);
            } else {
                NodeRec fa, fb;
#pragma unroll
                for (int k = 0; k < 3; ++k) { fa.c[k] = a.c[k]; fa.e[k] = a.e[k]; fb.c[k] = b.c[k]; fb.e[k] = b.e[k]; }
#pragma unroll
                for (int k = 0; k < 9; ++k) { fa.R[k] = Ra[k]; fb.R[k] = Rb[k]; }
                bb_compose(fa, fb, R12, t12, Rt, aRt, tt3);
            }
            bool hit = false;
            int verdict = 0;
            if (act) verdict = (g.no_filter & 1) ? 2 : sat15_f32(a.e, b.e, tt3, Rt);
            hit = verdict == 1;
            if (verdict == 2) hit = sat15(a.e, b.e, tt3, Rt, aRt);
#ifdef PFC_STAMPS
            {
                const unsigned long long mu = __ballot(verdict == 2);
                if (lane == 0 && mu && g.stamps) { atomicAdd(&g.stamps[13], (unsigned long long)__builtin_popcountll(mu)); atomicAdd(&g.stamps[14], 1ull); }
                if (lane == 0 && general && g.stamps) atomicAdd(&g.stamps[15], 1ull);
            }
#endif
            const int ca0 = a.child0, ca1 = a.child1, cb0 = b.child0, cb1 = b.child1;   // links (sign = leaf)
            const int leaf_a = a.leaf, leaf_b = b.leaf;
            const bool is_cand = hit && la && lb;
            const bool two = hit && (la != lb);
            const bool four = hit && !la && !lb;
            const unsigned long long mc = __ballot(is_cand), m2 = __ballot(two), m4 = __ballot(four);
            STAMP(u2);
            if (is_cand) ob[n_out + prefix_count(mc)] = make_int2(leaf_a, leaf_b);
            n_out += __builtin_popcountll(mc);
            if (two | four) {
                int pos = sp + 2 * prefix_count(m2) + 4 * prefix_count(m4);
                if (two) {
                    if (la) {  // leaf_1: descend tree_2 (:97-98)
                        stk[pos] = make_int2(e.x, cb0); stk[pos + 1] = make_int2(e.x, cb1);
                    } else {   // leaf_2: descend tree_1 (:101-103)
                        stk[pos] = make_int2(ca0, e.y); stk[pos + 1] = make_int2(ca1, e.y);
                    }
                } else {       // (1.1,2.1) (1.2,2.1) (1.1,2.2) (1.2,2.2) (:104-107)
                    stk[pos] = make_int2(ca0, cb0); stk[pos + 1] = make_int2(ca1, cb0);
                    stk[pos + 2] = make_int2(ca0, cb1); stk[pos + 3] = make_int2(ca1, cb1);
                }
            }
            sp += 2 * __builtin_popcountll(m2) + 4 * __builtin_popcountll(m4);
            __syncthreads();
            if (n_out > kDfsOut - 64 || (sp == 0 && n_out > 0)) {
                int base = 0;
                if (lane == 0) base = atomicAdd(g.ccount, n_out);
                base = __shfl(base, 0, 64);
                if (base + n_out <= g.ccap) {
                    for (int j = lane; j < n_out; j += 64) {
                        WorkRec c;
                        c.item = item; c.a = ob[j].x; c.b = ob[j].y; c.pad = 0;
                        g.cand[base + j] = c;
                    }
                } else if (lane == 0) {
                    atomicOr(g.status, kStCandOvf);
                }
                n_cand += n_out;
                n_out = 0;
                __syncthreads();
            }
#ifdef PFC_STAMPS
            STAMP(u3);
            c_load += u1 - u0; c_sat += u2 - u1; c_push += u3 - u2; c_iter += 1; c_lanes += p;
#endif
        }
#ifdef PFC_STAMPS
        if (lane == 0 && g.stamps) {
            atomicAdd(&g.stamps[8], c_load); atomicAdd(&g.stamps[9], c_sat); atomicAdd(&g.stamps[10], c_push);
            atomicAdd(&g.stamps[11], c_iter); atomicAdd(&g.stamps[12], c_lanes);
        }
#endif
        if (lane == 0) {
            if (sp > 0) atomicOr(g.status, kStAbort);
            if (n_test) atomicAdd(&g.icnt[4 * (size_t)item], n_test);
            if (n_cand) atomicAdd(&g.icnt[4 * (size_t)item + 1], n_cand);
        }
    }
}

// =================================================================================================================
// narrowphase
// =================================================================================================================
struct TracSoA {
    int *item;
    double *nx, *ny, *nz, *rx, *ry, *rz, *dA, *p;
};

struct NpArgs {
    const ItemRec *items;
    const WorkRec *cand;
    const int *ccount;
    int ccap;
    double *acc;
    double *rec;       // moment records (bristle)
    int *rcount;
    int rcap;
    int *icnt;
    int *clip_n;     // per candidate, or null
    // clipped polygons of bristle items, kept for the friction pass (k_fric): SoA [field][slot], slot < pcap
    int *poly_item;  // item | n_poly << 28
    double *poly;    // 34 fields: n̂ 3, centroid 3, ϵ_r² 4, vertices 8 x 3 (frame r²)
    int *pcount;
    int pcap;
    int *surv;       // candidate indices of the pairs that contributed traction points (work list of the Dual passes)
    int *scount;
    TracSoA trac;
    int *tcount;
    int tcap;
    unsigned *status;
    int debug;       // materialise traction points for every item
    unsigned long long *stamps;  // diagnostic builds: [0..5] cycles in gather / clip / reserve / integrate / reduce, rounds
};

constexpr int kNpBlock = 64;  // one wave per block: 16 KiB of LDS polygon staging per wave

// weightPoly (src/math_kernel/utility.jl:21-26) on 4-vectors held in LDS slots
// polygon ring in LDS: 8 physical slots x 4 coords per lane, [slot][coord][lane] layout (conflict-free per-lane
// dynamic indexing); logical vertex k of a lane lives in physical slot (rbase + k) & 7
#define PR(k, c) poly[((((rbase) + (k)) & 7) * 4 + (c)) * kNpBlock + lane]

__device__ __forceinline__ double readlane_f64(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
                            __builtin_amdgcn_readlane(__double2loint(v), src));
}

// Per-item accumulation of N per-lane partial sums.  Segmented scan per value, then the N totals of each run are
// transposed onto lanes 0..N-1 (readlane from the run's tail) and leave as ONE wave-wide FP64 atomic instruction
// on N consecutive accumulator slots: single-lane atomics are issue-bound (one wave instruction per ~50 ns per CU,
// MI355X guide 'Global float atomics'), a 37-lane one costs the same as a 1-lane one.
template <int N>
__device__ __forceinline__ void accumulate_items(double *acc, int item, bool listed, bool any, const double *v, int n0,
                                                 int stride = kAccStride) {
    // listed: the lane holds a work-list entry (its item keys the run even if it contributes nothing, so empty
    // polygons do not chop an item's run into pieces); any: the lane has a contribution
    static_assert(N <= 64, "one value per lane");
    if (__ballot(any) == 0) return;
    const Seg sg = seg_setup(listed ? item : -1);
    double tot[N];
#pragma unroll
    for (int k = 0; k < N; ++k) tot[k] = seg_sum(any ? v[k] : 0.0, sg);
    unsigned long long tails = __ballot(sg.tail && sg.valid);
    const int lane = lane_id();
    while (tails) {
        const int t = __builtin_ctzll(tails);
        tails &= tails - 1;
        const int item_t = __builtin_amdgcn_readlane(item, t);
        double mine = 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const double x = readlane_f64(tot[k], t);
            if (lane == k) mine = x;
        }
        if (lane < N && mine != 0.0) unsafeAtomicAdd(&acc[(size_t)item_t * stride + n0 + lane], mine);
    }
}

// Sums of N per-lane values over a whole wave through LDS: lane `lane` writes column `lane` of N rows (row stride 65
// doubles: conflict-free both ways), lanes lane0 .. lane0+N-1 then add up one row each.  N + ~130 instructions per wave
// instead of ~30 N for N segmented DPP scans; used when all work items of the wave belong to one item (97 % of the
// waves of the C3 batch).
// LDS ordering inside ONE wave (the block is a single wave): the LDS unit serves a wave's instructions in order, so a
// compiler-level fence is all that is needed.  __syncthreads() would also drain vmcnt, i.e. wait for every outstanding
// global store and atomic of the wave.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int N>
__device__ __forceinline__ double lds_row_sums(double *buf, const double *v, bool any, int lane, int lane0) {
#pragma unroll
    for (int k = 0; k < N; ++k) buf[k * 65 + lane] = any ? v[k] : 0.0;
    wave_lds_sync();
    double t = 0.0;
    const int row = lane - lane0;
    if (row >= 0 && row < N) {
        const double *r = buf + row * 65;
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll 4
        for (int j = 0; j < 64; j += 4) { t0 += r[j]; t1 += r[j + 1]; t2 += r[j + 2]; t3 += r[j + 3]; }
        t = (t0 + t1) + (t2 + t3);
    }
    wave_lds_sync();
    return t;
}

// Everything up to the per-item sums (regularized friction fused; bristle: normal wrench + patch moments).  For bristle
// items the clipped polygon of every contributing pair is kept (34 doubles, SoA by compacted slot: every store
// instruction of a wave writes consecutive doubles) so that the friction pass after k_eig (k_fric) re-integrates the
// bit-identical traction points without gathering and clipping again.  Materialising the TractionCache itself was
// measured at 2.7x the whole clip + quadrature (9 scattered 8-byte stores per point, ~12 points per polygon); it is only
// kept in debug mode (pfc_debug_tractions).
//
// TT: the scenario contains tet-tet instructions (non_friction.jl:166-194); compiled out otherwise so that the common
// tri-tet-only scenario does not pay the registers of the plane / tet intersection.
template <bool TT>
__global__ void __launch_bounds__(kNpBlock) k_narrow(NpArgs g) {
    __shared__ double poly[8 * 4 * kNpBlock];
    const int lane = threadIdx.x;
    int n_c = *g.ccount;
    if (n_c > g.ccap) n_c = g.ccap;
    const int stride = gridDim.x * kNpBlock;
    const int n_round = (n_c + stride - 1) / stride;
    for (int rd = 0; rd < n_round; ++rd) {
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
        (void)t0; (void)t1; (void)t2; (void)t3; (void)t4; (void)t5;
        STAMP(t0);
        const int idx = rd * stride + blockIdx.x * kNpBlock + lane;
        const bool active = idx < n_c;
        WorkRec cw;
        cw.item = 0; cw.a = 0; cw.b = 0; cw.pad = 0;
        if (active) cw = g.cand[idx];
        const ItemRec *it = g.items + cw.item;
        const TetRec *tp = it->tet + cw.b;
        const int nq = it->nq;
        const bool reg = it->model == PFC_REGULARIZED;
        const bool materialise = active && g.debug;
        const bool work = active;
        int n_poly = 0, rbase = 0;
        V3 nh = mk3(0.0, 0.0, 0.0);
        // ==== phase 1 (divergent): gather, transform to tet coordinates, clip ========================================
        if (work) {
            double R21[9], t21[3];
#pragma unroll
            for (int k = 0; k < 9; ++k) R21[k] = it->R21[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) t21[k] = it->t21[k];
            double z[4][4];      // input polygon (3 or 4 vertices) in the coordinates of tet 2
            int n_in = 0;
            V3 nh_in = mk3(0.0, 0.0, 0.0);
            double Z[16];        // x_ζ2_r2
#pragma unroll
            for (int k = 0; k < 16; ++k) Z[k] = tp->xzr[k];
            if (!TT || it->tet1 == nullptr) {
                // ---- tri-tet op (non_friction.jl:196-215) -----------------------------------------------------------
                const TriRec tr = it->tri[cw.a];
                // x_ζ2_r1 = x_ζ2_r2 * x_r2_r1.mat (:204); last row of x_r2_r1.mat is (0 0 0 1)
                double X[16];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        X[i + 4 * j] = (Z[i] * R21[3 * j] + Z[i + 4] * R21[3 * j + 1]) + Z[i + 8] * R21[3 * j + 2];
                    X[i + 12] = ((Z[i] * t21[0] + Z[i + 4] * t21[1]) + Z[i + 8] * t21[2]) + Z[i + 12];
                }
                // v_k = x_ζ2_r1 * onePad(vert_k) (:205-207)
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        z[k][i] = ((X[i] * tr.v[3 * k] + X[i + 4] * tr.v[3 * k + 1]) + X[i + 8] * tr.v[3 * k + 2]) + X[i + 12];
#pragma unroll
                for (int i = 0; i < 4; ++i) z[3][i] = 0.0;
                n_in = 3;
                // n̂2 = R(x_r2_r1) * n̂_r1 (:211-212)
                nh_in = mk3((R21[0] * tr.n[0] + R21[3] * tr.n[1]) + R21[6] * tr.n[2],
                            (R21[1] * tr.n[0] + R21[4] * tr.n[1]) + R21[7] * tr.n[2],
                            (R21[2] * tr.n[0] + R21[5] * tr.n[1]) + R21[8] * tr.n[2]);
            } else {
                // ---- tet-tet op (non_friction.jl:166-194) -----------------------------------------------------------
                const TetRec *t1 = it->tet1 + cw.a;
                double plane[4];
                {
                    // ϵ_plane_r2 = (Ē2 ϵ2) x_ζ2_r2 - (Ē1 ϵ1) (x_ζ1_r1 x_r1_r2)   (find_plane_tet :164, :174-177)
                    double R12[9], t12[3], Z1[16], X1[16];
#pragma unroll
                    for (int k = 0; k < 9; ++k) R12[k] = it->R12[k];
#pragma unroll
                    for (int k = 0; k < 3; ++k) t12[k] = it->t12[k];
#pragma unroll
                    for (int k = 0; k < 16; ++k) Z1[k] = t1->xzr[k];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            X1[i + 4 * j] = (Z1[i] * R12[3 * j] + Z1[i + 4] * R12[3 * j + 1]) + Z1[i + 8] * R12[3 * j + 2];
                        X1[i + 12] = ((Z1[i] * t12[0] + Z1[i + 4] * t12[1]) + Z1[i + 8] * t12[2]) + Z1[i + 12];
                    }
                    double Ee1[4], Ee2[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        Ee1[j] = it->Ebar1 * it->eps1[4 * (size_t)cw.a + j];
                        Ee2[j] = it->Ebar * it->eps2[4 * (size_t)cw.b + j];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const double p1 = ((Ee1[0] * X1[4 * j] + Ee1[1] * X1[4 * j + 1]) + Ee1[2] * X1[4 * j + 2]) + Ee1[3] * X1[4 * j + 3];
                        const double p2 = ((Ee2[0] * Z[4 * j] + Ee2[1] * Z[4 * j + 1]) + Ee2[2] * Z[4 * j + 2]) + Ee2[3] * Z[4 * j + 3];
                        plane[j] = p2 - p1;
                    }
                }
                // x_r2_ζ1 = x_r2_r1.mat * x_r1_ζ1: the vertices of tet 1 in frame r2 (:180); proj = plane * tet (:19)
                V3 P[4];
                double proj[4];
                int n_neg = 0, n_pos = 0;
                unsigned posm = 0, negm = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double vx = t1->xrz[3 * j], vy = t1->xrz[3 * j + 1], vz = t1->xrz[3 * j + 2];
                    P[j] = mk3(((R21[0] * vx + R21[3] * vy) + R21[6] * vz) + t21[0],
                               ((R21[1] * vx + R21[4] * vy) + R21[7] * vz) + t21[1],
                               ((R21[2] * vx + R21[5] * vy) + R21[8] * vz) + t21[2]);
                    proj[j] = ((plane[0] * P[j].x + plane[1] * P[j].y) + plane[2] * P[j].z) + plane[3];
                    if (proj[j] < 0.0) { ++n_neg; negm |= 1u << j; }
                    if (0.0 < proj[j]) { ++n_pos; posm |= 1u << j; }
                }
                // clip_plane_tet (plane_tet_intersection.jl:9-106).  weightPoly(v[i1], v[i2], proj[i1], proj[i2]) does
                // not depend on the order of (i1, i2) bit for bit, so one edge function serves every case.
                V3 q[4];
                q[0] = q[1] = q[2] = q[3] = mk3(0.0, 0.0, 0.0);
                int n_q = 0;
#define PW_(i1, i2) (P[i2] * (proj[i1] / (proj[i1] - proj[i2])) - P[i1] * (proj[i2] / (proj[i1] - proj[i2])))
                if (n_pos != 0 && n_neg != 0) {
                    int lone = -1;
                    if (n_pos == 1) lone = __builtin_ctz(posm);
                    else if (n_neg == 1) lone = __builtin_ctz(negm);
                    if (lone >= 0) {
                        V3 a, b, c;   // :52-79
                        if (lone == 0) { a = PW_(1, 0); b = PW_(3, 0); c = PW_(2, 0); }
                        else if (lone == 1) { a = PW_(0, 1); b = PW_(2, 1); c = PW_(3, 1); }
                        else if (lone == 2) { a = PW_(0, 2); b = PW_(3, 2); c = PW_(1, 2); }
                        else { a = PW_(0, 3); b = PW_(1, 3); c = PW_(2, 3); }
                        double pl = (lone == 0) ? proj[0] : (lone == 1) ? proj[1] : (lone == 2) ? proj[2] : proj[3];
                        n_q = 3;
                        if (0.0 < pl) { q[0] = a; q[1] = b; q[2] = c; } else { q[0] = c; q[1] = b; q[2] = a; }
                    } else {
                        V3 a, b, c, d;   // :81-106
                        const bool p0 = (posm & 1u) != 0, p1 = (posm & 2u) != 0, p2 = (posm & 4u) != 0;
                        if (p0 == p1) { a = PW_(1, 2); b = PW_(1, 3); c = PW_(0, 3); d = PW_(0, 2); }
                        else if (p0 == p2) { a = PW_(0, 1); b = PW_(0, 3); c = PW_(2, 3); d = PW_(2, 1); }
                        else { a = PW_(0, 2); b = PW_(0, 1); c = PW_(3, 1); d = PW_(3, 2); }
                        n_q = 4;
                        if (0.0 < proj[0]) { q[0] = a; q[1] = b; q[2] = c; q[3] = d; }
                        else { q[0] = d; q[1] = c; q[2] = b; q[3] = a; }
                    }
                }
#undef PW_
                // poly_ζ2 = one_pad_then_mul(x_ζ2_r2, poly_r2), then zero_small_coordinates (:184-187)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const double v = ((Z[i] * q[k].x + Z[i + 4] * q[k].y) + Z[i + 8] * q[k].z) + Z[i + 12];
                        z[k][i] = v * ((1.0e-14 < __builtin_fabs(v)) ? 1.0 : 0.0);
                    }
                n_in = n_q;
                nh_in = normalize(mk3(plane[0], plane[1], plane[2]));   // :190
            }
            bool finite = true;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) finite &= (k >= n_in) || (__builtin_fabs(z[k][i]) <= 1.79769313486231570815e308);
            if (!finite) atomicOr(g.status, kStNonFinite);
            // Trivial reject: if every vertex is non-positive on some plane the clip is empty.  Bit-exact shortcut:
            // every clipped vertex is c1*p2 - c2*p1 with c1 >= 0 >= c2 (static_clip.jl:197-201), whose sign on that
            // plane is exact, so Sutherland-Hodgman returns the empty polygon at that plane (:44).
            bool reject = !finite || n_in < 3;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                reject |= (z[0][i] <= 0.0) && (z[1][i] <= 0.0) && (z[2][i] <= 0.0) && (n_in < 4 || z[3][i] <= 0.0);
            if (!reject) {
                STAMP(t1);
                // ---- clip_in_tet_coordinates (static_clip.jl:7-23,34-201), polygon ring in LDS, clipped in place ---
                int n = n_in;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < n_in) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) PR(k, i) = z[k][i];
                    }
                bool err = false;
                for (int i = 0; i < 4 && n > 0; ++i) {
                    unsigned nonpos = 0, nonneg = 0;
                    for (int k = 0; k < n; ++k) {
                        double sv = PR(k, i);
                        nonpos |= (unsigned)(sv <= 0.0) << k;
                        nonneg |= (unsigned)(0.0 <= sv) << k;
                    }
                    const unsigned full = (1u << n) - 1u;
                    if (nonpos == full) { n = 0; break; }       // :44
                    if (nonneg == full) continue;               // :45-46
                    // first k with is_non_pos[k] && !is_non_pos[k+1] (cyclic) (:48-50)
                    unsigned nxt = ((nonpos >> 1) | ((nonpos & 1u) << (n - 1))) & full;
                    unsigned cand_start = nonpos & ~nxt & full;
                    if (cand_start == 0) { err = true; n = 0; break; }  // "Non-finite vertex likely" (:52)
                    const int st = __builtin_ctz(cand_start);
                    // cut_clip (:135-195): drop trailing vertices while z_{m-1} is non-positive
                    int m = n;
                    while (m > 3) {
                        int k2 = st + m - 2; if (k2 >= n) k2 -= n;
                        if ((nonpos >> k2) & 1u) --m; else break;
                    }
                    int k1 = st + 1; if (k1 >= n) k1 -= n;
                    int kl = st + m - 1; if (kl >= n) kl -= n;   // z_m (last)
                    int kp = st + m - 2; if (kp >= n) kp -= n;   // z_{m-1}
                    // inside test of the last vertex: 0 < z for arity 3..5 (:140,150,162), 0 <= z for 6..7 (:176,188)
                    const bool inside = (m <= 5) ? (((nonpos >> kl) & 1u) == 0) : (((nonneg >> kl) & 1u) != 0);
                    // z_start = clip_node(z1, z2); z_end = clip_node(z1, z_m) or clip_node(z_m, z_{m-1}); both are
                    // formed in registers before the ring is touched
                    double zs[4], ze[4];
                    {
                        double w1 = PR(st, i), w2 = PR(k1, i);
                        double sw = w1 - w2, c1 = w1 / sw, c2 = w2 / sw;
#pragma unroll
                        for (int c = 0; c < 4; ++c) zs[c] = c1 * PR(k1, c) - c2 * PR(st, c);
                    }
                    {
                        const int kn = inside ? st : kl, kq = inside ? kl : kp;
                        double w1 = PR(kn, i), w2 = PR(kq, i);
                        double sw = w1 - w2, c1 = w1 / sw, c2 = w2 / sw;
#pragma unroll
                        for (int c = 0; c < 4; ++c) ze[c] = c1 * PR(kq, c) - c2 * PR(kn, c);
                    }
                    const int ncopy = inside ? (m - 1) : (m - 2);   // z2 .. z_m  or  z2 .. z_{m-1} stay in the polygon
                    // In place: the new polygon starts at old logical st.  Kept vertices st+1 .. n-1 do not move;
                    // kept vertices that wrapped around (old logical 0 .. ) move up by n slots, in increasing order
                    // (a destination is either a free slot or the source of an earlier move).
                    for (int q = n - st - 1; q < ncopy; ++q) {
                        const int src = st + 1 + q - n, dst = st + 1 + q;
#pragma unroll
                        for (int c = 0; c < 4; ++c) { const double t = PR(src, c); PR(dst, c) = t; }
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) { PR(st, c) = zs[c]; PR(st + ncopy + 1, c) = ze[c]; }
                    rbase = (rbase + st) & 7;
                    n = ncopy + 2;
                    if (m == 7) break;  // the 7-vertex method returns the polygon directly (:185-195)
                }
                if (err) atomicOr(g.status, kStNonFinite);
                n_poly = n;
                if (n >= 3) nh = nh_in;
            }
        }
        if (g.clip_n && active) g.clip_n[idx] = n_poly;
        STAMP(t2);
        // ==== phase 2 (wave-uniform): reserve a contiguous run of traction slots for the whole wave ================
        // A lane with an n-gon owns n * nq consecutive slots, so the traction points of a wave (and, because the
        // candidate list is grouped by item, of an item) are contiguous: the later per-point passes then reduce
        // wave-uniformly with one atomic per wave instead of one per lane.
        const int slots = (materialise && n_poly >= 3) ? n_poly * nq : 0;
        int tbase = 0;
        if (g.debug) {
            int incl = seg_incl_scan(slots);
            const int tot = __shfl(incl, 63, 64);
            int base = 0;
            if (tot > 0) {
                if (lane == 0) base = atomicAdd(g.tcount, tot);
                base = __shfl(base, 0, 64);
            }
            tbase = base + incl - slots;
        }
        STAMP(t3);
        // ==== phase 3 (divergent): integrate_over_polygon_patch! (non_friction.jl:217-234) ============================
        double sum[10], wr1[3], wrr[6];   // wr1, wrr: first / second moments of w about the polygon centroid
#pragma unroll
        for (int k = 0; k < 10; ++k) sum[k] = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) wrr[k] = 0.0;
        wr1[0] = wr1[1] = wr1[2] = 0.0;
        V3 cen = mk3(0.0, 0.0, 0.0);
        int n_trac_lane = 0;
        if (n_poly >= 3) {
            const int n = n_poly;
            // poly_r2 = mul_then_un_pad(x_r2_ζ2, poly_ζ2) (poly_eight.jl:83-98), converted in place (x, y, z)
            {
                double V[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) V[k] = tp->xrz[k];
                for (int k = 0; k < n; ++k) {
                    const double z0 = PR(k, 0), z1 = PR(k, 1), z2 = PR(k, 2), z3 = PR(k, 3);
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        PR(k, c) = ((V[c] * z0 + V[c + 3] * z1) + V[c + 6] * z2) + V[c + 9] * z3;
                }
            }
            // centroid(poly_r2, n̂2) (poly_eight.jl:35-52)
            {
                V3 a = mk3(PR(0, 0), PR(0, 1), PR(0, 2));
                V3 cc = mk3(PR(1, 0), PR(1, 1), PR(1, 2));
                double cum_sum = 0.0;
                V3 cum_prod = mk3(0.0, 0.0, 0.0);
                for (int k = 2; k < n; ++k) {
                    V3 b = cc;
                    cc = mk3(PR(k, 0), PR(k, 1), PR(k, 2));
                    double ar = triangle_area(a, b, cc, nh);
                    cum_prod = cum_prod + ((a + b) + cc) * (1.0 / 3.0) * ar;
                    cum_sum += ar;
                }
                cen = (cum_sum == 0.0) ? a : cum_prod / cum_sum;
            }
            const double er0 = tp->epsr[0], er1 = tp->epsr[1], er2 = tp->epsr[2], er3 = tp->epsr[3];
            const V3 w = ld3(it->w), vl = ld3(it->v);
            const double chi = it->chi, Ebar = it->Ebar;
            const double v_c = it->v_c, mu_s = it->mu_s, mu_d = it->mu_d;
            const bool store = materialise && (tbase + slots <= g.tcap);
            if (materialise && !store) atomicOr(g.status, kStTracOvf);
            int tpos = tbase;
            V3 v2 = mk3(PR(n - 1, 0), PR(n - 1, 1), PR(n - 1, 2));
            for (int k = 0; k < n; ++k) {
                V3 v1 = v2;
                v2 = mk3(PR(k, 0), PR(k, 1), PR(k, 2));
                double area = triangle_area(v1, v2, cen, nh);
                if (!(0.0 < area)) continue;  // :232
                for (int q = 0; q < nq; ++q) {
                    // TriTetQuadRule rules 1 and 2, literal decimals of src/clip/quadrature.jl:24-39
                    double q0, q1, q2, qw;
                    if (nq == 1) {
                        q0 = q1 = q2 = 0.33333333333333331483; qw = 1.0;
                    } else {
                        const double qa = 0.16666666666666674068, qb = 0.66666666666666651864;
                        q0 = (q == 1) ? qb : qa; q1 = (q == 0) ? qb : qa; q2 = (q == 2) ? qb : qa;
                        qw = 0.33333333333333331483;
                    }
                    // fillTractionCacheInnerLoop! (:251-265)
                    V3 r = mk3((v1.x * q0 + v2.x * q1) + cen.x * q2, (v1.y * q0 + v2.y * q1) + cen.y * q2,
                               (v1.z * q0 + v2.z * q1) + cen.z * q2);
                    double eq = __builtin_fma(er0, r.x, er3);
                    eq = __builtin_fma(er1, r.y, eq);
                    eq = __builtin_fma(er2, r.z, eq);
                    V3 rdot = vl + cross(w, r);
                    double ee = -dot(mk3(er0, er1, er2), rdot);
                    double damp = fmax(0.0, 1.0 + chi * ee);
                    double p = eq * Ebar * damp;
                    double dA = qw * area;
                    if (!(0.0 < p)) continue;  // :245
                    ++n_trac_lane;
                    double p_dA = p * dA;
                    if (store) {
                        g.trac.item[tpos] = cw.item;
                        g.trac.nx[tpos] = nh.x; g.trac.ny[tpos] = nh.y; g.trac.nz[tpos] = nh.z;
                        g.trac.rx[tpos] = r.x; g.trac.ry[tpos] = r.y; g.trac.rz[tpos] = r.z;
                        g.trac.dA[tpos] = dA; g.trac.p[tpos] = p;
                        ++tpos;
                    }
                    if (reg) {
                        // yes_contact!(::Regularized) (friction.jl:50-72) fused
                        V3 vt = vec_sub_vec_proj(rdot, nh);
                        double m2 = dot(vt, vt);
                        V3 T;
                        if (m2 < v_c * v_c) {
                            T = (vt * (-mu_s)) / v_c;
                        } else {
                            double mg = __builtin_sqrt(m2);
                            double mu = clamped_piecewise(mg, 2 * v_c, 3 * v_c, mu_s, mu_d);
                            T = (vt * (-mu)) / mg;
                        }
                        const V3 tk = nh * p_dA + T * p_dA;
                        const V3 ta = cross(r, tk);
                        sum[0] += ta.x; sum[1] += ta.y; sum[2] += ta.z;
                        sum[3] += tk.x; sum[4] += tk.y; sum[5] += tk.z;
                    } else {
                        // normal_wrench_cop (normal.jl:17-34) fused: pass 1 of the bristle model.  The traction of a
                        // point is n̂ w (w = p dA) with n̂ constant over the polygon, so only W = sum w and the moments
                        // of w about the polygon centroid are accumulated per point; sum w r, the force n̂ W and the
                        // torque (sum w r) x n̂ follow after the loop (a quarter of the loop's instructions).
                        sum[6] += p_dA;
                        const V3 rc = r - cen;
                        const double wx = p_dA * rc.x, wy = p_dA * rc.y, wz = p_dA * rc.z;
                        wr1[0] += wx; wr1[1] += wy; wr1[2] += wz;
                        wrr[0] += wx * rc.x; wrr[1] += wx * rc.y; wrr[2] += wx * rc.z;
                        wrr[3] += wy * rc.y; wrr[4] += wy * rc.z; wrr[5] += wz * rc.z;
                    }
                }
            }
            if (!reg) {
                const double W = sum[6];
                const V3 Sr = mk3(wr1[0] + W * cen.x, wr1[1] + W * cen.y, wr1[2] + W * cen.z);   // sum w r
                const V3 ta = cross(Sr, nh);
                sum[0] = ta.x; sum[1] = ta.y; sum[2] = ta.z;
                sum[3] = nh.x * W; sum[4] = nh.y * W; sum[5] = nh.z * W;
                sum[7] = Sr.x; sum[8] = Sr.y; sum[9] = Sr.z;
            }
            if (store)  // unused slots of this lane's run (area <= 0 or p <= 0 points)
                for (; tpos < tbase + slots; ++tpos) g.trac.item[tpos] = -1;
        }
        STAMP(t4);
        // ==== phase 4 (wave-uniform): per-item reductions =============================================================
        const bool contributed = work && n_trac_lane > 0;
        {
            // ---- compacted slots for (a) the polygons of contributing bristle pairs, kept for k_fric, and (b) when
            // pfc_eval_dual asked for it, the candidate indices of all contributing pairs.  ONE 64-bit atomic per wave
            // reserves both (pcount in the low word, scount in the high word): a second single-address atomic per wave
            // cost 0.6 ms on the C3 batch.
            const bool keep = contributed && !reg;
            const bool list = contributed && g.surv != nullptr;
            if (__any(keep || list)) {
                const unsigned long long km = __ballot(keep), sm = __ballot(list);
                unsigned long long base2 = 0;
                if (lane == 0) {
                    if (g.surv == nullptr)
                        base2 = (unsigned)atomicAdd(g.pcount, __popcll(km));
                    else
                        base2 = atomicAdd(reinterpret_cast<unsigned long long *>(g.pcount),
                                          ((unsigned long long)__popcll(sm) << 32) | (unsigned long long)__popcll(km));
                }
                const int base = __builtin_amdgcn_readfirstlane((int)(base2 & 0xFFFFFFFFull));
                const int sbase = __builtin_amdgcn_readfirstlane((int)(base2 >> 32));
                const unsigned long long below = (1ull << lane) - 1ull;
                if (list) g.surv[sbase + __popcll(sm & below)] = idx;   // <= ccap entries
                const int slot = base + __popcll(km & below);
                if (keep && slot < g.pcap) {      // pcap >= ccap: cannot overflow
                    const size_t P = (size_t)g.pcap;
                    double *o = g.poly + slot;
                    // streaming stores: 0.5 GB per C3 batch must not evict the mesh records from the XCD's 4 MiB L2
#define NT_(p, v) __builtin_nontemporal_store((v), (p))
                    NT_(&g.poly_item[slot], (int)((unsigned)cw.item | ((unsigned)n_poly << 28)));
                    NT_(o, nh.x); NT_(o + P, nh.y); NT_(o + 2 * P, nh.z);
                    NT_(o + 3 * P, cen.x); NT_(o + 4 * P, cen.y); NT_(o + 5 * P, cen.z);
                    NT_(o + 6 * P, tp->epsr[0]); NT_(o + 7 * P, tp->epsr[1]); NT_(o + 8 * P, tp->epsr[2]);
                    NT_(o + 9 * P, tp->epsr[3]);
                    for (int k = 0; k < n_poly; ++k) {
                        NT_(o + (10 + 3 * k) * P, PR(k, 0)); NT_(o + (11 + 3 * k) * P, PR(k, 1));
                        NT_(o + (12 + 3 * k) * P, PR(k, 2));
                    }
#undef NT_
                }
            }
            // ---- the ten per-item sums.  Single-item wave (the rule: an item has ~30 waves of candidates): LDS transpose;
            // otherwise segmented scans keyed by item.  The polygon ring is free from here on (its last reader was the
            // polygon store above).
            const unsigned long long am = __ballot(active);
            const int item_first = __builtin_amdgcn_readlane(cw.item, am ? __builtin_ctzll(am) : 0);
            const bool single = __all(!active || cw.item == item_first);
            double t10 = 0.0;   // single: lane k < 10 holds total k
            if (single) {
                if (__any(contributed)) {
                    t10 = lds_row_sums<10>(poly, sum, contributed, lane, 0);
                    if (lane < 10 && t10 != 0.0) unsafeAtomicAdd(&g.acc[(size_t)item_first * kAccStride + lane], t10);
                }
            } else {
                accumulate_items<10>(g.acc, cw.item, active, contributed, sum, 0);
            }
            if (__any(contributed && !reg)) {
                // ---- patch-stiffness moments of the bristle model, one record per run of an item in this wave ----
                const bool cb = contributed && !reg;
                const Seg sg = seg_setup(active ? cw.item : -1);
                // the run's own pressure centroid c_w = sum w r / sum w, broadcast from the run's tail
                double Wt, cx, cy, cz;
                if (single) {
                    Wt = readlane_f64(t10, 6); cx = readlane_f64(t10, 7); cy = readlane_f64(t10, 8); cz = readlane_f64(t10, 9);
                } else {
                    Wt = seg_sum(cb ? sum[6] : 0.0, sg);
                    cx = seg_sum(cb ? sum[7] : 0.0, sg); cy = seg_sum(cb ? sum[8] : 0.0, sg);
                    cz = seg_sum(cb ? sum[9] : 0.0, sg);
                    Wt = __shfl(Wt, sg.tail_lane, 64);
                    cx = __shfl(cx, sg.tail_lane, 64); cy = __shfl(cy, sg.tail_lane, 64); cz = __shfl(cz, sg.tail_lane, 64);
                }
                const double iW = (Wt > 0.0) ? 1.0 / Wt : 0.0;
                const V3 cwv = mk3(cx * iW, cy * iW, cz * iW);
                // lane moments: polygon centroid -> c_w (parallel axis; |d| is at most the patch size)
                const double W = sum[6];
                const V3 d = cen - cwv;
                const V3 m1 = mk3(wr1[0] + W * d.x, wr1[1] + W * d.y, wr1[2] + W * d.z);   // sum w (r - c_w)
                double q[6];                                                              // sum w (r-c_w)(r-c_w)'
                q[0] = wrr[0] + 2.0 * wr1[0] * d.x + W * d.x * d.x;
                q[1] = wrr[1] + wr1[0] * d.y + wr1[1] * d.x + W * d.x * d.y;
                q[2] = wrr[2] + wr1[0] * d.z + wr1[2] * d.x + W * d.x * d.z;
                q[3] = wrr[3] + 2.0 * wr1[1] * d.y + W * d.y * d.y;
                q[4] = wrr[4] + wr1[1] * d.z + wr1[2] * d.y + W * d.y * d.z;
                q[5] = wrr[5] + 2.0 * wr1[2] * d.z + W * d.z * d.z;
                // n̂ is constant over a lane's polygon: sum w n n' = W n n', sum w (x x n) n' = (m1 x n) n',
                // sum w (x x n)(x x n)' = [n]x Q [n]x'
                double v[27];
                v[0] = W * nh.x * nh.x; v[1] = W * nh.x * nh.y; v[2] = W * nh.x * nh.z;
                v[3] = W * nh.y * nh.y; v[4] = W * nh.y * nh.z; v[5] = W * nh.z * nh.z;
                const V3 an = cross(m1, nh);
                v[6] = an.x * nh.x; v[7] = an.y * nh.x; v[8] = an.z * nh.x;
                v[9] = an.x * nh.y; v[10] = an.y * nh.y; v[11] = an.z * nh.y;
                v[12] = an.x * nh.z; v[13] = an.y * nh.z; v[14] = an.z * nh.z;
                {
                    const V3 c0 = mk3(q[0], q[1], q[2]), c1 = mk3(q[1], q[3], q[4]), c2 = mk3(q[2], q[4], q[5]);
                    const V3 m0 = cross(nh, c0), m1c = cross(nh, c1), m2 = cross(nh, c2);       // M = [n]x Q
                    // Saa = M [n]x': row i of Saa = n x (row i of M)
                    const V3 r0 = cross(nh, mk3(m0.x, m1c.x, m2.x)), r1 = cross(nh, mk3(m0.y, m1c.y, m2.y));
                    const V3 r2 = cross(nh, mk3(m0.z, m1c.z, m2.z));
                    v[15] = r0.x; v[16] = r0.y; v[17] = r0.z; v[18] = r1.y; v[19] = r1.z; v[20] = r2.z;
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) v[21 + k] = q[k];
                if (single) {
                    // rows on lanes 5..31: the record is item, W, c_w, 27 moments
                    double mine = lds_row_sums<27>(poly, v, cb, lane, 5);
                    if (Wt > 0.0) {
                        if (lane == 0) mine = (double)item_first;
                        if (lane == 1) mine = Wt;
                        if (lane == 2) mine = cwv.x;
                        if (lane == 3) mine = cwv.y;
                        if (lane == 4) mine = cwv.z;
                        int slot = 0;
                        if (lane == 0) slot = atomicAdd(g.rcount, 1);
                        slot = __builtin_amdgcn_readfirstlane(slot);
                        if (slot < g.rcap) {
                            if (lane < kRecStride) g.rec[(size_t)slot * kRecStride + lane] = mine;
                        } else if (lane == 0) {
                            atomicOr(g.status, kStRecOvf);
                        }
                    }
                }
                double tot[27];
                unsigned long long tails = 0;
                if (!single) {
#pragma unroll
                    for (int k = 0; k < 27; ++k) tot[k] = seg_sum(cb ? v[k] : 0.0, sg);
                    tails = __ballot(sg.tail && sg.valid && Wt > 0.0);
                } else {
#pragma unroll
                    for (int k = 0; k < 27; ++k) tot[k] = 0.0;
                }
                while (tails) {
                    const int t = __builtin_ctzll(tails);
                    tails &= tails - 1;
                    // lanes 0..31 assemble the record: item, W, c_w, 27 moments
                    double mine = 0.0;
                    if (lane == 0) mine = (double)__builtin_amdgcn_readlane(cw.item, t);
                    { const double x = readlane_f64(Wt, t); if (lane == 1) mine = x; }
                    { const double x = readlane_f64(cwv.x, t); if (lane == 2) mine = x; }
                    { const double x = readlane_f64(cwv.y, t); if (lane == 3) mine = x; }
                    { const double x = readlane_f64(cwv.z, t); if (lane == 4) mine = x; }
#pragma unroll
                    for (int k = 0; k < 27; ++k) {
                        const double x = readlane_f64(tot[k], t);
                        if (lane == 5 + k) mine = x;
                    }
                    int slot = 0;
                    if (lane == 0) slot = atomicAdd(g.rcount, 1);
                    slot = __builtin_amdgcn_readfirstlane(slot);
                    if (slot < g.rcap) {
                        if (lane < kRecStride) g.rec[(size_t)slot * kRecStride + lane] = mine;
                    } else if (lane == 0) {
                        atomicOr(g.status, kStRecOvf);
                    }
                }
            }
            count_per_item(g.icnt, cw.item, 2, active, active && n_poly >= 3);
            count_per_item(g.icnt, cw.item, 3, active, contributed, n_trac_lane);
        }
#ifdef PFC_STAMPS
        STAMP(t5);
        if (lane == 0 && g.stamps) {
            // t1 is only stamped when lane 0's wave entered the clip; fold gather+clip when it was not
            if (t1 == 0) t1 = t2;
            atomicAdd(&g.stamps[0], t1 - t0); atomicAdd(&g.stamps[1], t2 - t1); atomicAdd(&g.stamps[2], t3 - t2);
            atomicAdd(&g.stamps[3], t4 - t3); atomicAdd(&g.stamps[4], t5 - t4); atomicAdd(&g.stamps[5], 1ull);
        }
#endif
    }
}
#undef PR

// Bristle friction pass (after k_eig): calc_spatial_bristle_force (friction.jl:171-201) + traction(::Bristle) (:32-48)
// over the polygons k_narrow kept.  One lane per kept polygon, every load is a coalesced read of consecutive slots;
// the fan / quadrature arithmetic is the one of k_narrow, so the traction points are bit-identical.
struct FricArgs {
    const ItemRec *items;
    const int *poly_item;
    const double *poly;
    const int *pcount;
    int pcap;
    const double *res;
    double *acc;
};
__global__ void __launch_bounds__(64) k_fric(FricArgs g) {
    const int lane = threadIdx.x;
    int n_p = *g.pcount;
    if (n_p > g.pcap) n_p = g.pcap;
    const size_t P = (size_t)g.pcap;
    const int stride = gridDim.x * 64;
    const int n_round = (n_p + stride - 1) / stride;
    for (int rd = 0; rd < n_round; ++rd) {
        const int idx = rd * stride + blockIdx.x * 64 + lane;
        const bool active = idx < n_p;
        double sum[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) sum[k] = 0.0;
        int item = 0;
        bool contributed = false;
        if (active) {
            const unsigned pk = (unsigned)g.poly_item[idx];
            item = (int)(pk & 0x0FFFFFFFu);
            const int n = (int)(pk >> 28);
            const ItemRec *it = g.items + item;
            const double *o = g.poly + idx;
            const V3 nh = mk3(o[0], o[P], o[2 * P]);
            const V3 cen = mk3(o[3 * P], o[4 * P], o[5 * P]);
            const double er0 = o[6 * P], er1 = o[7 * P], er2 = o[8 * P], er3 = o[9 * P];
            const int nq = it->nq;
            const V3 w = ld3(it->w), vl = ld3(it->v);
            const double chi = it->chi, Ebar = it->Ebar, mu_s = it->mu_s, mu_d = it->mu_d;
            const double tau = it->tau, k_bar = it->k_bar;
            const double *res = g.res + (size_t)item * kResStride;
            const V3 cop = ld3(res + kResCop), Da = ld3(res + kResDelta), Dl = ld3(res + kResDelta + 3);
            V3 v2 = mk3(o[(10 + 3 * (n - 1)) * P], o[(11 + 3 * (n - 1)) * P], o[(12 + 3 * (n - 1)) * P]);
            for (int k = 0; k < n; ++k) {
                const V3 v1 = v2;
                v2 = mk3(o[(10 + 3 * k) * P], o[(11 + 3 * k) * P], o[(12 + 3 * k) * P]);
                const double area = triangle_area(v1, v2, cen, nh);
                if (!(0.0 < area)) continue;
                for (int q = 0; q < nq; ++q) {
                    double q0, q1, q2, qw;
                    if (nq == 1) {
                        q0 = q1 = q2 = 0.33333333333333331483; qw = 1.0;
                    } else {
                        const double qa = 0.16666666666666674068, qb = 0.66666666666666651864;
                        q0 = (q == 1) ? qb : qa; q1 = (q == 0) ? qb : qa; q2 = (q == 2) ? qb : qa;
                        qw = 0.33333333333333331483;
                    }
                    const V3 r = mk3((v1.x * q0 + v2.x * q1) + cen.x * q2, (v1.y * q0 + v2.y * q1) + cen.y * q2,
                                     (v1.z * q0 + v2.z * q1) + cen.z * q2);
                    double eq = __builtin_fma(er0, r.x, er3);
                    eq = __builtin_fma(er1, r.y, eq);
                    eq = __builtin_fma(er2, r.z, eq);
                    const V3 rdot = vl + cross(w, r);
                    const double ee = -dot(mk3(er0, er1, er2), rdot);
                    const double damp = fmax(0.0, 1.0 + chi * ee);
                    const double p = eq * Ebar * damp;
                    const double dA = qw * area;
                    if (!(0.0 < p)) continue;
                    contributed = true;
                    const double p_dA = p * dA;
                    const V3 x = r - cop;
                    const V3 del = Dl + cross(Da, x);
                    V3 Ts = (del + rdot * tau) * (-k_bar);
                    Ts = vec_sub_vec_proj(Ts, nh);
                    const double m2 = dot(Ts, Ts);
                    V3 T;
                    if (m2 < mu_s * mu_s) {
                        T = Ts;
                    } else {
                        const double mg = __builtin_sqrt(m2);
                        const double mu = clamped_piecewise(mg, 2 * mu_s, 3 * mu_s, mu_s, mu_d);
                        T = (Ts * mu) / mg;
                    }
                    const V3 Tc = T * p_dA;
                    const V3 ta = cross(x, Tc);
                    sum[0] += ta.x; sum[1] += ta.y; sum[2] += ta.z;
                    sum[3] += Tc.x; sum[4] += Tc.y; sum[5] += Tc.z;
                }
            }
        }
        accumulate_items<6>(g.acc, item, active, contributed, sum, kAccFric);
    }
}

#include "pfc_dual.h"

// =================================================================================================================
// bristle model: cop, patch stiffness, 6x6 eigen, friction pass, finalisation
// =================================================================================================================
struct BrArgs {
    const ItemRec *items;
    int n_items;
    double *acc;
    double *res;
    const int *icnt;
    TracSoA trac;
    const int *tcount;
    int tcap;
    double *wrench, *sdot;
    int *counts;
};

// Moves every moment record from its run centroid c_w to the item's cop and adds it to the item accumulators.
// With d = c_w - cop and sum w (r - c_w) = 0 by construction of c_w:
//   Snn' = Snn            San' = San + [d]x Snn            Srr' = Srr + W d d'
//   Saa' = Saa + San [d]x' + [d]x San' + [d]x Snn [d]x'
// One lane computes one record, the block transposes through LDS so that each record leaves as ONE 27-lane atomic.
struct ShiftArgs {
    const double *rec;
    const int *rcount;
    int rcap;
    double *acc;
};
__global__ void __launch_bounds__(64) k_shift(ShiftArgs g) {
    __shared__ double out[64 * 28];
    __shared__ int items[64];
    int n_r = *g.rcount;
    if (n_r > g.rcap) n_r = g.rcap;
    const int lane = threadIdx.x;
    for (int base = blockIdx.x * 64; base < n_r; base += gridDim.x * 64) {
        const int i = base + lane;
        if (i < n_r) {
            const double *r = g.rec + (size_t)i * kRecStride;
            const int item = (int)r[0];
            const double W = r[1];
            const double *a = g.acc + (size_t)item * kAccStride;
            const double S = a[kAccIp];
            const double d[3] = {r[2] - a[kAccIpc] / S, r[3] - a[kAccIpc + 1] / S, r[4] - a[kAccIpc + 2] / S};
            const int s6[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
            double Snn[9], San[9], Saa[9], Srr[9];
            const double dx[9] = {0.0, d[2], -d[1], -d[2], 0.0, d[0], d[1], -d[0], 0.0};   // [d]x column-major
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                Snn[k] = r[5 + s6[k]]; San[k] = r[11 + k]; Saa[k] = r[20 + s6[k]]; Srr[k] = r[26 + s6[k]];
            }
            double dS[9], Sd[9], dSd[9];   // [d]x Snn,  San [d]x',  [d]x Snn [d]x'
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int ii = 0; ii < 3; ++ii) {
                    double x = 0.0, y = 0.0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) { x += dx[ii + 3 * k] * Snn[k + 3 * j]; y += San[ii + 3 * k] * dx[j + 3 * k]; }
                    dS[ii + 3 * j] = x; Sd[ii + 3 * j] = y;
                }
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int ii = 0; ii < 3; ++ii) {
                    double x = 0.0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) x += dS[ii + 3 * k] * dx[j + 3 * k];
                    dSd[ii + 3 * j] = x;
                }
            double *o = out + lane * 28;
            const int u6[6] = {0, 3, 6, 4, 7, 8};   // xx xy xz yy yz zz in a column-major 3x3
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = Snn[u6[k]];
#pragma unroll
            for (int k = 0; k < 9; ++k) o[6 + k] = San[k] + dS[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int ii = u6[k] % 3, j = u6[k] / 3;
                o[15 + k] = Saa[u6[k]] + Sd[ii + 3 * j] + Sd[j + 3 * ii] + dSd[u6[k]];
                o[21 + k] = Srr[u6[k]] + W * d[ii] * d[j];
            }
            items[lane] = item;
        }
        __syncthreads();
        const int n_here = (n_r - base < 64) ? (n_r - base) : 64;
        for (int q = 0; q < n_here; ++q) {
            if (lane < 27) {
                const double x = out[q * 28 + lane];
                if (x != 0.0) unsafeAtomicAdd(&g.acc[(size_t)items[q] * kAccStride + kAccSnn + lane], x);
            }
        }
        __syncthreads();
    }
}

// Jacobi eigen-solver for a symmetric 6x6 (stands in for LAPACK eigen!(Hermitian), friction.jl:88).  One thread per
// item, so the kernel's duration is the length of the serial dependency chain: the sweep uses the round-robin
// ordering (5 rounds of 3 index-disjoint pairs).  The three rotations of a round read disjoint entries of A, so their
// angle computations (the sqrt / divide chains) are independent and overlap; every index is a compile-time constant
// after unrolling, so A and V live in registers (runtime-indexed arrays would go to scratch).
__device__ __forceinline__ void jacobi_angle(double app, double aqq, double apq, double &cs, double &sn) {
    // apq == 0: identity rotation
    const double theta = (aqq - app) / (2.0 * apq);
    double t = (theta >= 0 ? 1.0 : -1.0) / (__builtin_fabs(theta) + __builtin_sqrt(theta * theta + 1.0));
    if (apq == 0.0) t = 0.0;
    cs = 1.0 / __builtin_sqrt(t * t + 1.0);
    sn = t * cs;
}
template <int P, int Q>
__device__ __forceinline__ void jacobi_apply(double *A, double *V, double cs, double sn) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double akp = A[k + 6 * P], akq = A[k + 6 * Q];
        A[k + 6 * P] = cs * akp - sn * akq; A[k + 6 * Q] = sn * akp + cs * akq;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double apk = A[P + 6 * k], aqk = A[Q + 6 * k];
        A[P + 6 * k] = cs * apk - sn * aqk; A[Q + 6 * k] = sn * apk + cs * aqk;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double vkp = V[k + 6 * P], vkq = V[k + 6 * Q];
        V[k + 6 * P] = cs * vkp - sn * vkq; V[k + 6 * Q] = sn * vkp + cs * vkq;
    }
}
template <int P0, int Q0, int P1, int Q1, int P2, int Q2>
__device__ __forceinline__ void jacobi_round(double *A, double *V) {
    double c0, s0, c1, s1, c2, s2;
    jacobi_angle(A[7 * P0], A[7 * Q0], A[P0 + 6 * Q0], c0, s0);
    jacobi_angle(A[7 * P1], A[7 * Q1], A[P1 + 6 * Q1], c1, s1);
    jacobi_angle(A[7 * P2], A[7 * Q2], A[P2 + 6 * Q2], c2, s2);
    jacobi_apply<P0, Q0>(A, V, c0, s0);
    jacobi_apply<P1, Q1>(A, V, c1, s1);
    jacobi_apply<P2, Q2>(A, V, c2, s2);
}
__device__ __forceinline__ void jacobi6(double *A, double *V, double *w) {
#pragma unroll
    for (int i = 0; i < 36; ++i) V[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) V[7 * i] = 1.0;
    double off_prev = 1.79769313486231570815e308;
    for (int sweep = 0; sweep < 40; ++sweep) {
        // Converged when every off-diagonal entry is below the rounding floor of the matrix (eps * largest diagonal)
        // or negligible against its own two diagonal entries; also stop once (after 4 sweeps) a sweep no longer
        // halves the off-diagonal mass (nothing but rounding noise is left to annihilate).  Waiting for an absolute 1e-17
        // would spin through all sweeps: entries coupled to the large eigenvalues never get below eps * |A|.
        double off = 0.0, dmax = 0.0;
        bool done = true;
#pragma unroll
        for (int i = 0; i < 6; ++i) dmax = fmax(dmax, __builtin_fabs(A[7 * i]));
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = i + 1; j < 6; ++j) {
                const double a = __builtin_fabs(A[i + 6 * j]);
                off += a * a;
                done &= a <= 2.3e-16 * dmax || a * a <= 1e-30 * __builtin_fabs(A[7 * i] * A[7 * j]);
            }
        if (done || (sweep >= 4 && !(off < 0.5 * off_prev))) break;
        off_prev = off;
        jacobi_round<0, 5, 1, 4, 2, 3>(A, V);
        jacobi_round<0, 4, 3, 5, 1, 2>(A, V);
        jacobi_round<0, 3, 2, 4, 1, 5>(A, V);
        jacobi_round<0, 2, 1, 3, 4, 5>(A, V);
        jacobi_round<0, 1, 2, 5, 3, 4>(A, V);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) w[i] = A[7 * i];
}

// decompose_K! / calc_K̄_sqrt_inv / Δ² (friction.jl:85-132): one thread per bristle item in contact
__global__ void __launch_bounds__(64) k_eig(BrArgs g) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n_items) return;
    const ItemRec *it = g.items + i;
    if (it->model != PFC_BRISTLE || g.icnt[4 * (size_t)i + 3] == 0) return;
    const double *a = g.acc + (size_t)i * kAccStride;
    double *r = g.res + (size_t)i * kResStride;
    // cop = sum w r / sum w (normal.jl:33)
    const double S = a[kAccIp];
    const double c[3] = {a[kAccIpc] / S, a[kAccIpc + 1] / S, a[kAccIpc + 2] / S};
    r[kResCop] = c[0]; r[kResCop + 1] = c[1]; r[kResCop + 2] = c[2];
    // calc_patch_spatial_stiffness! (friction.jl:147-169) from the moments about the cop (x = r - cop):
    //   K22 = S I - sum w n n'      K12 = -sum w (x x n) n'   (sum w [x]x = 0 about the cop)
    //   K11 = -(sum w x x' - tr(.) I + sum w (x x n)(x x n)')
    const int s6[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};  // symmetric 3x3 from 6 unique
    double Snn[9], San[9], Saa[9], Srr[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        Snn[k] = a[kAccSnn + s6[k]]; Saa[k] = a[kAccSaa + s6[k]]; San[k] = a[kAccSan + k]; Srr[k] = a[kAccSrr + s6[k]];
    }
    const double trC = Srr[0] + Srr[4] + Srr[8];
    double K[36];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int ii = 0; ii < 3; ++ii) {
            const double I = (ii == j) ? 1.0 : 0.0;
            const double k11 = -(Srr[ii + 3 * j] - trC * I + Saa[ii + 3 * j]);
            const double k12 = -San[ii + 3 * j];
            const double k22 = S * I - Snn[ii + 3 * j];
            K[ii + 6 * j] = k11;
            K[ii + 6 * (j + 3)] = k12;
            K[(j + 3) + 6 * ii] = k12;
            K[(ii + 3) + 6 * (j + 3)] = k22;
        }
#pragma unroll
    for (int k = 0; k < 36; ++k) { K[k] *= it->k_bar; r[kResK + k] = K[k]; }
    double t1 = (K[0] + K[7]) + K[14], t2 = (K[21] + K[28]) + K[35];
    double s1 = 1.0 / __builtin_sqrt(t1), s2 = 1.0 / __builtin_sqrt(t2);
    double Sinv[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) { Sinv[k] = s1 * it->magic; Sinv[k + 3] = s2; }
    double Kb[36], V[36], sig[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int ii = 0; ii < 6; ++ii) {
            double kij = (ii <= j) ? K[ii + 6 * j] : K[j + 6 * ii];
            Kb[ii + 6 * j] = (Sinv[ii] * kij) * Sinv[j];
        }
    jacobi6(Kb, V, sig);
    double mx = sig[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) mx = fmax(mx, sig[k]);
#pragma unroll
    for (int k = 0; k < 6; ++k) sig[k] = 1.0 / __builtin_sqrt(fmax(sig[k], mx * 1.0e-16));
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int ii = 0; ii < 6; ++ii) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) acc += (V[ii + 6 * k] * sig[k]) * V[j + 6 * k];
            r[kResKis + ii + 6 * j] = acc;
        }
#pragma unroll
    for (int ii = 0; ii < 6; ++ii) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc += r[kResKis + ii + 6 * k] * it->s[k];
        r[kResDelta + ii] = Sinv[ii] * acc;
        r[kResSinv + ii] = Sinv[ii];
    }
}

// yes_contact! / no_contact! epilogue (friction.jl:76-81,119-143; non_friction.jl:77-83)
__global__ void k_final(BrArgs g) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n_items) return;
    const ItemRec *it = g.items + i;
    const double *a = g.acc + (size_t)i * kAccStride;
    const double *r = g.res + (size_t)i * kResStride;
    double *w = g.wrench + 6 * (size_t)i, *sd = g.sdot + 6 * (size_t)i;
    const bool contact = g.icnt[4 * (size_t)i + 3] > 0;
    if (g.counts)
        for (int k = 0; k < 4; ++k) g.counts[4 * (size_t)i + k] = g.icnt[4 * (size_t)i + k];
    for (int k = 0; k < 6; ++k) { w[k] = 0.0; sd[k] = 0.0; }
    if (it->model == PFC_REGULARIZED) {
        if (contact)
            for (int k = 0; k < 6; ++k) w[k] = a[kAccWrench + k];
        return;
    }
    const double tau_inv = 1.0 / it->tau;
    if (!contact) {
        for (int k = 0; k < 6; ++k) sd[k] = -tau_inv * it->s[k];
        return;
    }
    V3 fang = ld3(a + kAccFric), flin = ld3(a + kAccFric + 3), cop = ld3(r + kResCop);
    V3 fang2 = fang + cross(cop, flin);
    w[0] = a[kAccWrench] + fang2.x; w[1] = a[kAccWrench + 1] + fang2.y; w[2] = a[kAccWrench + 2] + fang2.z;
    w[3] = a[kAccWrench + 3] + flin.x; w[4] = a[kAccWrench + 4] + flin.y; w[5] = a[kAccWrench + 5] + flin.z;
    double sw[6];
    for (int k = 0; k < 6; ++k) sw[k] = r[kResSinv + k] * a[kAccFric + k];
    for (int ii = 0; ii < 6; ++ii) {
        double acc = 0.0;
        for (int k = 0; k < 6; ++k) acc += r[kResKis + ii + 6 * k] * sw[k];
        sd[ii] = -tau_inv * (acc + it->s[ii]);
    }
}

// Gathers everything the host needs to judge an evaluation into one small block (one D2H copy instead of five):
// tail[0..3] status words, tail[4..11] totals {node tests, non-empty pairs, traction points, 0} as 64-bit,
// tail[12..] the counter block (candidates, traction slots, seed ticket, records, frontier sizes per level).
// Also leaves the counters and the status word zeroed for the next evaluation (two memset nodes less per launch
// sequence: what a small scene pays is launches, not kernels); the packed copy in `tail` is what later readers use.
__global__ void __launch_bounds__(256) k_pack(int n_items, const int *icnt, int *ctr, int n_ctr, unsigned *status,
                                               int *tail) {
    __shared__ unsigned long long tot[3];
    if (threadIdx.x < 3) tot[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned long long a = 0, b = 0, c = 0;
    for (int i = threadIdx.x; i < n_items; i += blockDim.x) {
        a += (unsigned)icnt[4 * (size_t)i]; b += (unsigned)icnt[4 * (size_t)i + 2]; c += (unsigned)icnt[4 * (size_t)i + 3];
    }
    atomicAdd(&tot[0], a); atomicAdd(&tot[1], b); atomicAdd(&tot[2], c);
    __syncthreads();
    if (threadIdx.x < 4) { tail[threadIdx.x] = (int)status[threadIdx.x]; status[threadIdx.x] = 0u; }
    if (threadIdx.x < 3) reinterpret_cast<unsigned long long *>(tail + 4)[threadIdx.x] = tot[threadIdx.x];
    if (threadIdx.x == 3) reinterpret_cast<unsigned long long *>(tail + 4)[3] = 0ull;
    for (int k = threadIdx.x; k < n_ctr; k += blockDim.x) { tail[12 + k] = ctr[k]; ctr[k] = 0; }
}

// addGeneralizedForcesThirdLaw! (non_friction.jl:267-286): per item, the wrench on body 2 (frame r2) goes to the
// world frame (RigidBodyDynamics transform(wrench, x_rw_r2): lin = R lin, ang = R ang + t x lin) and is projected
// on the geometric Jacobians: f += J_2' w - J_1' w (torque!: tau_j = J_ang[:,j].ang + J_lin[:,j].lin).
// One thread per (item, velocity coordinate); bodies without a Jacobian (root / no mesh path) have id < 0.
struct ScatterArgs {
    int n_items, nv;
    const double *wrench;   // n_items x 6 (device, as written by the evaluation)
    const double *x_w_r2;   // n_items x 12: R (9, column-major), t (3)
    const int *body_1, *body_2, *scene;
    const double *jac;      // n_body x 6 x nv: rows 0..2 angular, 3..5 linear, column-major (6 x nv)
    double *f;              // n_scene x nv
};
__global__ void k_scatter(ScatterArgs g) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)g.n_items * g.nv) return;
    const int i = (int)(tid / g.nv), j = (int)(tid % g.nv);
    const double *w = g.wrench + 6 * (size_t)i;
    const double *x = g.x_w_r2 + 12 * (size_t)i;
    const V3 ang = ld3(w), lin = ld3(w + 3);
    const V3 lw = mk3((x[0] * lin.x + x[3] * lin.y) + x[6] * lin.z, (x[1] * lin.x + x[4] * lin.y) + x[7] * lin.z,
                      (x[2] * lin.x + x[5] * lin.y) + x[8] * lin.z);
    const V3 aw = mk3((x[0] * ang.x + x[3] * ang.y) + x[6] * ang.z, (x[1] * ang.x + x[4] * ang.y) + x[7] * ang.z,
                      (x[2] * ang.x + x[5] * ang.y) + x[8] * ang.z) + cross(ld3(x + 9), lw);
    double tau = 0.0;
    const int b2 = g.body_2[i], b1 = g.body_1[i];
    if (b2 >= 0) {
        const double *J = g.jac + ((size_t)b2 * g.nv + j) * 6;
        tau += dot(ld3(J), aw) + dot(ld3(J + 3), lw);
    }
    if (b1 >= 0) {
        const double *J = g.jac + ((size_t)b1 * g.nv + j) * 6;
        tau -= dot(ld3(J), aw) + dot(ld3(J + 3), lw);
    }
    const int sc = g.scene ? g.scene[i] : 0;
    if (tau != 0.0) unsafeAtomicAdd(&g.f[(size_t)sc * g.nv + j], tau);
}

__global__ void k_selftest(int n, const double *x, const double *y, double *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = x[i] / y[i];
    out[n + i] = __builtin_sqrt(__builtin_fabs(x[i]));
    out[2 * n + i] = __builtin_fma(x[i], y[i], x[i]);
}

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
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct HostMesh {
    int n_pt = 0, n_tri = 0, n_tet = 0, n_node = 0, depth = 0;
    double Ebar = 0.0;
    std::vector<double> xyz, eps;
    std::vector<int> tri, tet;
    std::vector<NodeRec> nodes;
    std::vector<NodeF> nodesf;
    NodeRec *d_nodes = nullptr;
    NodeF *d_nodesf = nullptr;
    TriRec *d_tri = nullptr;
    TetRec *d_tet = nullptr;
    double *d_tet_eps = nullptr;
};

enum { EV_START = 0, EV_SETUP, EV_BP, EV_NP, EV_BR, EV_FIN, EV_COUNT };

}  // namespace

struct pfc_context {
    int device = 0;
    bool finalized = false;
    std::string err;
    hipStream_t stream = nullptr;
    std::vector<HostMesh> meshes;
    std::vector<InsDev> ins;
    MeshDev *d_meshes = nullptr;
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
    DevBuf<int> tail;                    // k_pack output (status, totals, counters)
    int *h_tail = nullptr;               // pinned host mirror of tail
    size_t h_tail_cap = 0;
    void *pin_in = nullptr, *pin_out = nullptr;   // pinned staging of the host-buffer path
    size_t pin_in_cap = 0, pin_out_cap = 0;
    unsigned long long epoch = 0;        // bumped whenever a device work buffer is reallocated
    // captured launch sequence (hipGraph) of the last evaluation shape
    hipGraphExec_t gexec[2] = {nullptr, nullptr};   // [0] plain evaluation, [1] with the contributing-pair list (Dual)
    struct GraphKey {
        int n_items, levels, L, debug, bristle, surv;
        const void *p[7];
        void *stream;
        unsigned long long epoch;
    } gkey[2] = {};
    bool ghave[2] = {false, false};
    bool want_surv = false;   // the narrowphase also lists the contributing candidates (pfc_eval_dual)
    int opt_graph = 1;
    size_t fcap = 0, ccap = 0, tcap = 0, rcap = 0;
    // host-pointer path staging
    DevBuf<double> h_pose, h_twist, h_s, h_wrench, h_sdot;
    DevBuf<int> h_ins, h_counts;
    // last evaluation
    int last_n_items = 0, last_levels = 0, last_bfs_levels = 0;
    bool pending = false;
    hipStream_t last_stream = nullptr;
    long long stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    DevBuf<double> dual_in, dual_acc, dual_res, dual_out, dual_poly;   // pfc_eval_dual
    DevBuf<int2> dual_pkey;
    DevBuf<int> dual_cnt;
    DevBuf<int> surv;                                       // candidate indices of contributing pairs
    DevBuf<int> poly_item;                                  // kept polygons of bristle pairs (k_narrow -> k_fric)
    DevBuf<double> poly;
    long long last_undecided = 0;      // node pairs the Float32 broadphase settled with the exact Float64 test
    long long last_tslots = 0;         // traction slots used by the last evaluation (>= traction points)
    hipEvent_t ev[EV_COUNT] = {};
    bool ev_valid = false;
    // Large batches are evaluated as two concurrent halves: a second set of work buffers on a second stream, so that
    // the broadphase of one half (vector-ALU bound) shares the CUs with the narrowphase of the other (parked on
    // s_waitcnt half of the time).  Measured on the C3 batch: 2.39 -> 2.04 ms per 2 048 poses; four parts are slower.
    pfc_context *twin = nullptr;
    bool is_twin = false;
    int opt_split_min = 1024;          // 0: never split
    int split_n0 = 0;                  // items in the first half of the pending evaluation (0: not split)
    int last_parts = 1;                // 2 if the last checked evaluation ran as two halves
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
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

hipError_t ensure_work(pfc_context *h, int n_items) {
    hipError_t e;
    const size_t caps0[] = {h->items.cap, h->acc.cap, h->res.cap, h->icnt.cap, h->ctr.cap, h->frontier[0].cap,
                            h->frontier[1].cap, h->cand.cap, h->clip_n.cap, h->trac_item.cap, h->trac_d.cap, h->rec.cap,
                            h->tail.cap};
    if ((e = h->items.ensure(n_items)) != hipSuccess) return e;
    if ((e = h->acc.ensure((size_t)n_items * kAccStride)) != hipSuccess) return e;
    if ((e = h->res.ensure((size_t)n_items * kResStride)) != hipSuccess) return e;
    if ((e = h->icnt.ensure((size_t)n_items * 4)) != hipSuccess) return e;
    {
        const size_t c0 = h->ctr.cap, s0 = h->status.cap;
        if ((e = h->ctr.ensure((size_t)h->max_levels + 12)) != hipSuccess) return e;
        if ((e = h->status.ensure(4)) != hipSuccess) return e;
        if (h->ctr.cap != c0 && (e = hipMemset(h->ctr.p, 0, sizeof(int) * h->ctr.cap)) != hipSuccess) return e;
        if (h->status.cap != s0 && (e = hipMemset(h->status.p, 0, sizeof(unsigned) * h->status.cap)) != hipSuccess) return e;
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
    if ((e = h->rec.ensure(rc * kRecStride)) != hipSuccess) return e;
    if ((e = h->frontier[0].ensure(f)) != hipSuccess) return e;
    if ((e = h->frontier[1].ensure(f)) != hipSuccess) return e;
    if ((e = h->cand.ensure(c)) != hipSuccess) return e;
    if ((e = h->clip_n.ensure(c)) != hipSuccess) return e;
    if ((e = h->surv.ensure(c)) != hipSuccess) return e;
    if (h->any_bristle) {
        if ((e = h->poly_item.ensure(c)) != hipSuccess) return e;
        if ((e = h->poly.ensure(c * 34)) != hipSuccess) return e;
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
                            h->tail.cap};
    for (size_t k = 0; k < sizeof caps0 / sizeof caps0[0]; ++k)
        if (caps0[k] != caps1[k]) { ++h->epoch; break; }
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

// Enqueue one evaluation.  All pointers are device pointers.
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
        const bool big = h->max_leaves >= 8192, mid = h->max_leaves >= 1024;
        const double target = big ? (n_items >= 256 ? 1024.0 : 4096.0) : (mid ? 1024.0 : 64.0);
        double seeds = (double)n_items;
        while (seeds < target && L < 8) { seeds *= 4.0; ++L; }
        if (L == 0 && !big && n_items >= 1024) L = 1;
    }
    return L > levels ? levels : L;
}

// The launch sequence of one evaluation on stream st (eagerly, or while st is being captured into a graph).
int record_eval(pfc_context *h, int n_items, const int *d_ins_ids, const double *d_pose, const double *d_twist,
                const double *d_s, double *d_wrench, double *d_sdot, int *d_counts, hipStream_t st, bool prof) {
    const int levels = h->opt_max_levels > 0 ? h->opt_max_levels : h->max_levels;
    int *ccount = h->ctr.p, *tcount = h->ctr.p + 1, *next_seed = h->ctr.p + 2, *rcount = h->ctr.p + 3;
    int *ucount = h->ctr.p + 4, *fcount = h->ctr.p + 6;
    int *pcount = h->ctr.p + ((levels + 9) & ~1);   // after the per-level frontier counts; 8-byte aligned pair
    // counters and status are zero here: k_pack of the previous evaluation (or ensure_work after an allocation) left them so
#ifdef PFC_STAMPS
    HIP_TRY(h, hipMemsetAsync(h->stamps.p, 0, sizeof(unsigned long long) * 16, st));
#endif
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[EV_START], st));

    EvalArgs ea;
    ea.n_items = n_items; ea.ins_ids = d_ins_ids; ea.pose = d_pose; ea.twist = d_twist; ea.s = d_s;
    ea.ins = h->d_ins; ea.meshes = h->d_meshes; ea.n_ins = (int)h->ins.size(); ea.items = h->items.p;
    ea.frontier0 = h->frontier[0].p; ea.fcount = fcount; ea.acc = h->acc.p; ea.icnt = h->icnt.p;
    ea.status = h->status.p;
    hipLaunchKernelGGL(k_setup_items, dim3(grid_for(n_items, 128, 1 << 20)), dim3(128), 0, st, ea);
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[EV_SETUP], st));

    // broadphase: a few level-synchronous expansions to get enough independent seed pairs, then the per-wave
    // depth-first kernel for everything below
    const int L = bfs_levels_for(h, n_items, levels);
    for (int lv = 0; lv < L; ++lv) {
        BpArgs b;
        b.items = h->items.p; b.fin = h->frontier[lv & 1].p; b.fout = h->frontier[(lv + 1) & 1].p;
        b.cand = h->cand.p; b.fcount = fcount; b.ccount = ccount; b.icnt = h->icnt.p; b.status = h->status.p;
        b.level = lv; b.fcap = (int)h->fcap; b.ccap = (int)h->ccap;
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
        d.status = h->status.p; d.reserve = 3 * levels + 3; d.stamps = h->stamps.p;
        if (h->opt_no_filter) {
            // Float64-only traversal (A/B checks)
            d.seeds = h->frontier[L & 1].p; d.n_seed = fcount + L; d.seed_cap = (int)h->fcap; d.next_seed = next_seed;
            d.no_filter = 1;
            hipLaunchKernelGGL(k_bp_dfs, dim3(grid_for(bound, 1, 256 * 8)), dim3(64), 0, st, d);
        } else {
            Dfs32Args f;
            f.items = h->items.p; f.seeds = h->frontier[L & 1].p; f.n_seed = fcount + L; f.next_seed = next_seed;
            f.seed_cap = (int)h->fcap; f.cand = h->cand.p; f.ccount = ccount; f.ccap = (int)h->ccap;
            f.ucount = ucount; f.icnt = h->icnt.p; f.status = h->status.p;
            f.reserve = 3 * levels + 3;
            hipLaunchKernelGGL(k_bp_dfs32, dim3(grid_for(bound, 1, 256 * 6)), dim3(kDfsBlock), 0, st, f);
        }
    }
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[EV_BP], st));

    NpArgs np;
    np.items = h->items.p; np.cand = h->cand.p; np.ccount = ccount; np.ccap = (int)h->ccap; np.acc = h->acc.p;
    np.icnt = h->icnt.p; np.clip_n = h->opt_debug ? h->clip_n.p : nullptr; np.trac = trac_view(h);
    np.tcount = tcount; np.tcap = (int)h->tcap; np.status = h->status.p; np.debug = h->opt_debug;
    np.stamps = h->stamps.p;
    np.rec = h->rec.p; np.rcount = rcount; np.rcap = (int)h->rcap;
    np.poly_item = h->poly_item.p; np.poly = h->poly.p; np.pcount = pcount; np.pcap = (int)h->ccap;
    np.surv = h->want_surv ? h->surv.p : nullptr; np.scount = pcount + 1;
    const int np_grid = grid_for(h->ccap, kNpBlock, 256 * 16);
    if (h->any_tet_tet) hipLaunchKernelGGL((k_narrow<true>), dim3(np_grid), dim3(kNpBlock), 0, st, np);
    else hipLaunchKernelGGL((k_narrow<false>), dim3(np_grid), dim3(kNpBlock), 0, st, np);
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[EV_NP], st));

    BrArgs br;
    br.items = h->items.p; br.n_items = n_items; br.acc = h->acc.p; br.res = h->res.p; br.icnt = h->icnt.p;
    br.trac = trac_view(h); br.tcount = tcount; br.tcap = (int)h->tcap; br.wrench = d_wrench; br.sdot = d_sdot;
    br.counts = d_counts;
    if (h->any_bristle) {
        ShiftArgs sh;
        sh.rec = h->rec.p; sh.rcount = rcount; sh.rcap = (int)h->rcap; sh.acc = h->acc.p;
        hipLaunchKernelGGL(k_shift, dim3(grid_for(h->rcap, 64, 2048)), dim3(64), 0, st, sh);
        hipLaunchKernelGGL(k_eig, dim3(grid_for(n_items, 64, 1 << 20)), dim3(64), 0, st, br);
        FricArgs fr;
        fr.items = h->items.p; fr.poly_item = h->poly_item.p; fr.poly = h->poly.p; fr.pcount = pcount;
        fr.pcap = (int)h->ccap; fr.res = h->res.p; fr.acc = h->acc.p;
        hipLaunchKernelGGL(k_fric, dim3(grid_for(h->ccap, 64, 256 * 16)), dim3(64), 0, st, fr);
    }
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[EV_BR], st));
    hipLaunchKernelGGL(k_final, dim3(grid_for(n_items, 128, 1 << 20)), dim3(128), 0, st, br);
    if (prof) HIP_TRY(h, hipEventRecord(h->ev[EV_FIN], st));
    hipLaunchKernelGGL(k_pack, dim3(1), dim3(256), 0, st, n_items, h->icnt.p, h->ctr.p, levels + 12, h->status.p, h->tail.p);
    HIP_TRY(h, hipGetLastError());
    return PFC_OK;
}

// Enqueue one evaluation.  All pointers are device pointers.  The fixed launch sequence (2 memsets + 8..20 small
// kernels) is captured into a hipGraph the first time a shape is seen and replayed afterwards: for the reference's
// own scene sizes (a handful of instructions per calcXd!) launch overhead, not kernel time, is what an evaluation
// costs.  Profiling (HIP events between stages) uses the eager path.
int enqueue_eval(pfc_context *h, int n_items, const int *d_ins_ids, const double *d_pose, const double *d_twist,
                 const double *d_s, double *d_wrench, double *d_sdot, int *d_counts, hipStream_t st) {
    HIP_TRY(h, ensure_work(h, n_items));
    const int levels = h->opt_max_levels > 0 ? h->opt_max_levels : h->max_levels;
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
        key.n_items = n_items; key.levels = levels; key.L = L; key.debug = h->opt_debug;
        key.bristle = (h->any_bristle ? 1 : 0) | (h->any_tet_tet ? 2 : 0);
        key.surv = h->want_surv ? 1 : 0;
        key.p[0] = d_ins_ids; key.p[1] = d_pose; key.p[2] = d_twist; key.p[3] = d_s; key.p[4] = d_wrench;
        key.p[5] = d_sdot; key.p[6] = d_counts; key.stream = (void *)st; key.epoch = h->epoch;
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
    HIP_TRY(h, hipMemcpyAsync(h->h_tail, h->tail.p, sizeof(int) * n_tail, hipMemcpyDeviceToHost, h->last_stream));
    HIP_TRY(h, hipStreamSynchronize(h->last_stream));
    h->pending = false;
    const int *tail = h->h_tail;
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
    if (status & (kStFrontierOvf | kStCandOvf | kStTracOvf | kStRecOvf)) {
        // VectorCache-style growth (src/obb/vector_cache.jl:13-17): at least double, at least the observed need
        if (status & kStFrontierOvf) { size_t f = h->fcap * 2; while (f < (size_t)fpeak) f *= 2; h->fcap = f; }
        if (status & kStCandOvf) { size_t c = h->ccap * 2; while (c < (size_t)ctr[0]) c *= 2; h->ccap = c; }
        if (status & kStTracOvf) { size_t t = h->tcap * 2; while (t < (size_t)ctr[1]) t *= 2; h->tcap = t; }
        if (status & kStRecOvf) { size_t r = h->rcap * 2; while (r < (size_t)ctr[3]) r *= 2; h->rcap = r; }
        return fail(h, PFC_ERR_OVERFLOW, "work list overflow (status %u): capacities grown to frontier %zu, candidates %zu, tractions %zu, records %zu",
                    status, h->fcap, h->ccap, h->tcap, h->rcap);
    }
    if (status & kStAbort) return fail(h, PFC_ERR_STATE, "broadphase aborted: iteration guard hit (corrupt tree?)");
    if (status & kStNonFinite) return fail(h, PFC_ERR_NONFINITE, "Non-finite vertex likely");
    h->stats[0] = (long long)tot[0]; h->stats[2] = (long long)tot[1]; h->stats[3] = (long long)tot[2];
    return PFC_OK;
}

// Synchronise the pending evaluation (both halves of a split one) and merge the counters.
int check_eval(pfc_context *h) {
    if (!h->split_n0) { h->last_parts = 1; return check_one(h); }
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
    return PFC_OK;
}

// the second set of work buffers: shares the (immutable) mesh / instruction records of h
int make_twin(pfc_context *h) {
    if (h->twin) return PFC_OK;
    pfc_context *t = new (std::nothrow) pfc_context();
    if (!t) return fail(h, PFC_ERR_NOMEM, "out of host memory");
    t->device = h->device; t->is_twin = true; t->finalized = true;
    t->ins = h->ins; t->d_meshes = h->d_meshes; t->d_ins = h->d_ins; t->max_levels = h->max_levels;
    t->max_leaves = h->max_leaves;
    t->any_bristle = h->any_bristle; t->any_tet_tet = h->any_tet_tet; t->opt_split_min = 0;
    if (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
        delete t;
        return fail(h, PFC_ERR_HIP, "could not create the second stream");
    }
    h->twin = t;
    return PFC_OK;
}

}  // namespace

// =================================================================================================================
// C ABI
// =================================================================================================================
extern "C" {

int pfc_version(void) { return PFC_VERSION; }

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
    *out = h;
    return PFC_OK;
}

void pfc_destroy(pfc_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->twin) { pfc_destroy(h->twin); h->twin = nullptr; }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->is_twin) { h->d_meshes = nullptr; h->d_ins = nullptr; }   // owned by the parent
    for (auto &m : h->meshes) {
        if (m.d_nodes) (void)hipFree(m.d_nodes);
        if (m.d_nodesf) (void)hipFree(m.d_nodesf);
        if (m.d_tri) (void)hipFree(m.d_tri);
        if (m.d_tet) (void)hipFree(m.d_tet);
        if (m.d_tet_eps) (void)hipFree(m.d_tet_eps);
    }
    if (h->d_meshes) (void)hipFree(h->d_meshes);
    if (h->d_ins) (void)hipFree(h->d_ins);
    h->items.release(); h->frontier[0].release(); h->frontier[1].release(); h->cand.release();
    h->clip_n.release(); h->icnt.release(); h->trac_item.release(); h->acc.release(); h->res.release();
    h->trac_d.release(); h->rec.release(); h->ctr.release(); h->status.release(); h->stamps.release();
    h->h_pose.release(); h->h_twist.release(); h->h_s.release(); h->h_wrench.release(); h->h_sdot.release();
    h->h_ins.release(); h->h_counts.release();
    for (int k = 0; k < EV_COUNT; ++k)
        if (h->ev[k]) (void)hipEventDestroy(h->ev[k]);
    for (int gi = 0; gi < 2; ++gi)
        if (h->gexec[gi]) (void)hipGraphExecDestroy(h->gexec[gi]);
    if (h->h_tail) (void)hipHostFree(h->h_tail);
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    h->tail.release();
    h->poly_item.release(); h->poly.release(); h->surv.release();
    h->dual_poly.release(); h->dual_pkey.release(); h->dual_cnt.release();
    h->dual_in.release(); h->dual_acc.release(); h->dual_res.release(); h->dual_out.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

const char *pfc_last_error(pfc_handle h) { return h ? h->err.c_str() : "null handle"; }

int pfc_add_mesh(pfc_handle h, int n_pt, const double *xyz, int n_tri, const int *tri, int n_tet, const int *tet,
                 const double *eps, double Ebar, int n_node, const double *node_c, const double *node_e,
                 const double *node_R, const int *node_child, const int *node_leaf) {
    if (!h) return -PFC_ERR_BAD_ARG;
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
    m.depth = tree_depth(m.nodes);
    if (m.depth < 0) return -fail(h, PFC_ERR_BAD_ARG, "pfc_add_mesh: the node array is not a tree");
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
        for (int j = 0; j < 3; ++j) { f.c[j] = r.c[j]; f.e[j] = (float)r.e[j]; }
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
        // host check of the bound the device relies on: the Float32 quaternion must reproduce R to 8 u per entry
        {
            const double w = f.q[0], x = f.q[1], y = f.q[2], z = f.q[3];
            const double Rq[9] = {1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w),
                                  2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w),
                                  2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)};
            double worst = 0.0;
            for (int j = 0; j < 9; ++j) worst = std::fmax(worst, std::fabs(Rq[j] - R[j]));
            f.exact_only = (std::isfinite(qn) && worst <= 4.0 * 5.9604644775390625e-8) ? 0 : 1;
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
    if (h->finalized) return fail(h, PFC_ERR_STATE, "pfc_finalize called twice");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, h->status.ensure(4));
    HIP_TRY(h, hipMemset(h->status.p, 0, sizeof(unsigned) * 4));
    std::vector<MeshDev> md(h->meshes.size());
    for (size_t k = 0; k < h->meshes.size(); ++k) {
        HostMesh &m = h->meshes[k];
        double *d_xyz = nullptr, *d_eps = nullptr;
        int *d_idx = nullptr;
        HIP_TRY(h, hipMalloc((void **)&m.d_nodes, sizeof(NodeRec) * m.nodes.size()));
        HIP_TRY(h, hipMemcpy(m.d_nodes, m.nodes.data(), sizeof(NodeRec) * m.nodes.size(), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMalloc((void **)&m.d_nodesf, sizeof(NodeF) * m.nodesf.size()));
        HIP_TRY(h, hipMemcpy(m.d_nodesf, m.nodesf.data(), sizeof(NodeF) * m.nodesf.size(), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMalloc((void **)&d_xyz, sizeof(double) * m.xyz.size()));
        HIP_TRY(h, hipMemcpy(d_xyz, m.xyz.data(), sizeof(double) * m.xyz.size(), hipMemcpyHostToDevice));
        if (m.n_tri) {
            HIP_TRY(h, hipMalloc((void **)&d_idx, sizeof(int) * m.tri.size()));
            HIP_TRY(h, hipMemcpy(d_idx, m.tri.data(), sizeof(int) * m.tri.size(), hipMemcpyHostToDevice));
            HIP_TRY(h, hipMalloc((void **)&m.d_tri, sizeof(TriRec) * m.n_tri));
            hipLaunchKernelGGL(k_prep_tri, dim3((m.n_tri + 127) / 128), dim3(128), 0, h->stream, m.n_tri, d_xyz, d_idx, m.d_tri);
        } else {
            HIP_TRY(h, hipMalloc((void **)&d_idx, sizeof(int) * m.tet.size()));
            HIP_TRY(h, hipMemcpy(d_idx, m.tet.data(), sizeof(int) * m.tet.size(), hipMemcpyHostToDevice));
            HIP_TRY(h, hipMalloc((void **)&d_eps, sizeof(double) * m.eps.size()));
            HIP_TRY(h, hipMemcpy(d_eps, m.eps.data(), sizeof(double) * m.eps.size(), hipMemcpyHostToDevice));
            HIP_TRY(h, hipMalloc((void **)&m.d_tet, sizeof(TetRec) * m.n_tet));
            HIP_TRY(h, hipMalloc((void **)&m.d_tet_eps, sizeof(double) * 4 * m.n_tet));
            hipLaunchKernelGGL(k_prep_tet, dim3((m.n_tet + 127) / 128), dim3(128), 0, h->stream, m.n_tet, d_xyz, d_eps, d_idx,
                               m.d_tet, m.d_tet_eps, h->status.p);
        }
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        (void)hipFree(d_xyz); (void)hipFree(d_idx);
        if (d_eps) (void)hipFree(d_eps);
        md[k].nodes = m.d_nodes; md[k].nodesf = m.d_nodesf; md[k].tri = m.d_tri; md[k].tet = m.d_tet; md[k].tet_eps = m.d_tet_eps; md[k].Ebar = m.Ebar;
        md[k].n_tri = m.n_tri; md[k].n_tet = m.n_tet; md[k].n_node = m.n_node; md[k].depth = m.depth;
    }
    unsigned status = 0;
    HIP_TRY(h, hipMemcpy(&status, h->status.p, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (status & kStNonFinite) return fail(h, PFC_ERR_NONFINITE, "singular tetrahedron (non-finite zeta transform)");
    if (!md.empty()) {
        HIP_TRY(h, hipMalloc((void **)&h->d_meshes, sizeof(MeshDev) * md.size()));
        HIP_TRY(h, hipMemcpy(h->d_meshes, md.data(), sizeof(MeshDev) * md.size(), hipMemcpyHostToDevice));
    }
    h->max_levels = 1;
    h->max_leaves = 2;
    h->any_bristle = false;
    for (const InsDev &in : h->ins) {
        int lv = h->meshes[in.m1].depth + h->meshes[in.m2].depth + 1;
        if (lv > h->max_levels) h->max_levels = lv;
        const int lf = (h->meshes[in.m1].n_node + 1) / 2 + (h->meshes[in.m2].n_node + 1) / 2;
        if (lf > h->max_leaves) h->max_leaves = lf;
        if (in.model == PFC_BRISTLE) h->any_bristle = true;
        if (h->meshes[in.m1].n_tri == 0) h->any_tet_tet = true;
    }
    // the depth-first broadphase keeps 3 * levels + 3 stack slots in reserve (k_bp_dfs)
    if (3 * h->max_levels + 3 > kDfsStack - 128 || 3 * h->max_levels + 3 > kDfsStack32 - 1024)
        return fail(h, PFC_ERR_BAD_ARG, "OBB trees too deep (depth sum %d): rebuild them balanced", h->max_levels - 1);
    if (!h->ins.empty()) {
        HIP_TRY(h, hipMalloc((void **)&h->d_ins, sizeof(InsDev) * h->ins.size()));
        HIP_TRY(h, hipMemcpy(h->d_ins, h->ins.data(), sizeof(InsDev) * h->ins.size(), hipMemcpyHostToDevice));
    }
    h->finalized = true;
    return PFC_OK;
}

int pfc_eval_device(pfc_handle h, int n_items, const int *d_ins_ids, const double *d_pose, const double *d_twist,
                    const double *d_s, double *d_wrench, double *d_sdot, int *d_counts, void *stream) {
    if (!h) return PFC_ERR_BAD_ARG;
    if (!h->finalized) return fail(h, PFC_ERR_STATE, "pfc_eval before pfc_finalize");
    if (n_items < 0) return fail(h, PFC_ERR_BAD_ARG, "negative n_items");
    if (n_items == 0) { h->pending = false; h->last_n_items = 0; return PFC_OK; }
    if (h->ins.empty()) return fail(h, PFC_ERR_STATE, "no contact instructions");
    if (!d_pose || !d_twist || !d_wrench || !d_sdot) return fail(h, PFC_ERR_BAD_ARG, "null buffer");
    if (!d_ins_ids && n_items > (int)h->ins.size())
        return fail(h, PFC_ERR_BAD_ARG, "n_items exceeds the number of instructions and no ins_ids given");
    if (h->any_bristle && !d_s) return fail(h, PFC_ERR_BAD_ARG, "bristle instructions need the state buffer s");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    h->split_n0 = 0;
    const bool split = h->opt_split_min > 0 && n_items >= h->opt_split_min && d_ins_ids && !h->opt_debug &&
                       !h->want_surv && !h->is_twin;
    if (!split) return enqueue_eval(h, n_items, d_ins_ids, d_pose, d_twist, d_s, d_wrench, d_sdot, d_counts, st);
    int rc = make_twin(h);
    if (rc != PFC_OK) return rc;
    pfc_context *t = h->twin;
    t->opt_profile = h->opt_profile; t->opt_max_levels = h->opt_max_levels; t->opt_bfs_levels = h->opt_bfs_levels;
    t->opt_graph = h->opt_graph;
    if (t->opt_no_filter != h->opt_no_filter) { t->opt_no_filter = h->opt_no_filter; t->ghave[0] = t->ghave[1] = false; }
    const int n0 = n_items / 2, n1 = n_items - n0;
    // the second half starts when the caller's stream has reached this point and joins it again at the end
    HIP_TRY(h, hipEventRecord(h->ev_fork, st));
    HIP_TRY(h, hipStreamWaitEvent(t->stream, h->ev_fork, 0));
    rc = enqueue_eval(h, n0, d_ins_ids, d_pose, d_twist, d_s, d_wrench, d_sdot, d_counts, st);
    if (rc != PFC_OK) return rc;
    rc = enqueue_eval(t, n1, d_ins_ids + n0, d_pose + 24 * (size_t)n0, d_twist + 6 * (size_t)n0,
                      d_s ? d_s + 6 * (size_t)n0 : nullptr, d_wrench + 6 * (size_t)n0, d_sdot + 6 * (size_t)n0,
                      d_counts ? d_counts + 4 * (size_t)n0 : nullptr, t->stream);
    if (rc != PFC_OK) { h->err = t->err; return rc; }
    HIP_TRY(h, hipEventRecord(h->ev_join, t->stream));
    HIP_TRY(h, hipStreamWaitEvent(st, h->ev_join, 0));
    h->split_n0 = n0;
    return PFC_OK;
}

int pfc_check(pfc_handle h) {
    if (!h) return PFC_ERR_BAD_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    return check_eval(h);
}

int pfc_eval(pfc_handle h, int n_items, const int *ins_ids, const double *pose, const double *twist,
             const double *s, double *wrench, double *sdot, int *counts) {
    if (!h) return PFC_ERR_BAD_ARG;
    if (!h->finalized) return fail(h, PFC_ERR_STATE, "pfc_eval before pfc_finalize");
    if (n_items < 0) return fail(h, PFC_ERR_BAD_ARG, "negative n_items");
    if (n_items == 0) return PFC_OK;
    if (!pose || !twist || !wrench || !sdot) return fail(h, PFC_ERR_BAD_ARG, "null buffer");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t n = (size_t)n_items;
    // one pinned block in (pose | twist | s | ins_ids), one pinned block out (wrench | sdot | counts): two async
    // copies around the launch sequence and a single synchronisation
    const size_t in_d = n * 36, in_bytes = in_d * sizeof(double) + n * sizeof(int);
    const size_t out_d = n * 12, out_bytes = out_d * sizeof(double) + n * 4 * sizeof(int);
    if (h->pin_in_cap < in_bytes) {
        if (h->pin_in) (void)hipHostFree(h->pin_in);
        h->pin_in = nullptr; h->pin_in_cap = 0;
        HIP_TRY(h, hipHostMalloc(&h->pin_in, in_bytes * 2));
        h->pin_in_cap = in_bytes * 2;
    }
    if (h->pin_out_cap < out_bytes) {
        if (h->pin_out) (void)hipHostFree(h->pin_out);
        h->pin_out = nullptr; h->pin_out_cap = 0;
        HIP_TRY(h, hipHostMalloc(&h->pin_out, out_bytes * 2));
        h->pin_out_cap = out_bytes * 2;
    }
    HIP_TRY(h, h->h_pose.ensure(in_d + (n + 1) / 2 + 1));      // device mirror of the input block (doubles)
    HIP_TRY(h, h->h_wrench.ensure(out_d + 2 * n + 1));         // device mirror of the output block
    double *pi = (double *)h->pin_in;
    std::memcpy(pi, pose, sizeof(double) * n * 24);
    std::memcpy(pi + n * 24, twist, sizeof(double) * n * 6);
    if (s) std::memcpy(pi + n * 30, s, sizeof(double) * n * 6); else std::memset(pi + n * 30, 0, sizeof(double) * n * 6);
    if (ins_ids) std::memcpy(pi + in_d, ins_ids, sizeof(int) * n);
    hipStream_t st = h->stream;
    double *di = h->h_pose.p, *dout = h->h_wrench.p;
    HIP_TRY(h, hipMemcpyAsync(di, pi, ins_ids ? in_bytes : in_d * sizeof(double), hipMemcpyHostToDevice, st));
    int rc = PFC_OK;
    for (int attempt = 0; attempt < 40; ++attempt) {
        rc = pfc_eval_device(h, n_items, ins_ids ? (const int *)(di + in_d) : nullptr, di, di + n * 24,
                             s ? di + n * 30 : nullptr, dout, dout + n * 6, (int *)(dout + out_d), st);
        if (rc != PFC_OK) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->pin_out, dout, out_bytes, hipMemcpyDeviceToHost, st));
        rc = check_eval(h);
        if (rc != PFC_ERR_OVERFLOW) break;
    }
    if (rc != PFC_OK) return rc;
    const double *po = (const double *)h->pin_out;
    std::memcpy(wrench, po, sizeof(double) * n * 6);
    std::memcpy(sdot, po + n * 6, sizeof(double) * n * 6);
    if (counts) std::memcpy(counts, po + out_d, sizeof(int) * n * 4);
    return PFC_OK;
}

int pfc_eval_dual(pfc_handle h, int n_items, int n_dir, const int *ins_ids, const double *pose, const double *twist,
                  const double *s, const double *d_pose, const double *d_twist, const double *d_s, double *wrench,
                  double *sdot, double *d_wrench, double *d_sdot, int *counts) {
    if (!h) return PFC_ERR_BAD_ARG;
    if (n_dir < 1 || n_dir > 16) return fail(h, PFC_ERR_BAD_ARG, "pfc_eval_dual: n_dir must be in 1..16");
    if (n_items > 0 && (!d_pose || !d_twist || !d_wrench || !d_sdot))
        return fail(h, PFC_ERR_BAD_ARG, "pfc_eval_dual: null buffer");
    // values, candidate list and per-item counters: the ordinary evaluation (the broadphase ignores partials,
    // src/contact_algorithms_non_friction.jl:95)
    h->want_surv = true;
    int rc = pfc_eval(h, n_items, ins_ids, pose, twist, s, wrench, sdot, counts);
    h->want_surv = false;
    if (rc != PFC_OK || n_items == 0) return rc;
    const size_t nk = (size_t)n_items * n_dir;
    hipStream_t st = h->stream;
    HIP_TRY(h, h->dual_in.ensure(nk * 36));
    HIP_TRY(h, h->dual_acc.ensure(nk * kDaStride));
    HIP_TRY(h, h->dual_res.ensure(nk * kDrStride));
    HIP_TRY(h, h->dual_out.ensure(nk * 12));
    double *dp = h->dual_in.p, *dt = dp + nk * 24, *dsd = dt + nk * 6;
    HIP_TRY(h, hipMemcpyAsync(dp, d_pose, sizeof(double) * nk * 24, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(dt, d_twist, sizeof(double) * nk * 6, hipMemcpyHostToDevice, st));
    if (d_s) HIP_TRY(h, hipMemcpyAsync(dsd, d_s, sizeof(double) * nk * 6, hipMemcpyHostToDevice, st));
    else HIP_TRY(h, hipMemsetAsync(dsd, 0, sizeof(double) * nk * 6, st));
    HIP_TRY(h, hipMemsetAsync(h->dual_acc.p, 0, sizeof(double) * nk * kDaStride, st));
    DualArgs a;
    a.items = h->items.p; a.cand = h->cand.p; a.ccount = h->tail.p + 12; a.ccap = (int)h->ccap;   // packed copy of the counters
    a.surv = h->surv.p; a.scount = h->tail.p + 12 + (((h->last_levels + 9) & ~1) + 1);
    a.n_items = n_items; a.n_dir = n_dir; a.d_pose = dp; a.d_twist = dt; a.d_s = dsd; a.icnt = h->icnt.p;
    a.dacc = h->dual_acc.p; a.dres = h->dual_res.p; a.d_wrench = h->dual_out.p; a.d_sdot = h->dual_out.p + nk * 6;
    a.status = h->status.p;
    const int cpw = 64 / n_dir;
    const size_t n_cand = (size_t)h->stats[2];   // contributing pairs <= pairs with a non-empty polygon
    const int grid = grid_for((n_cand + cpw - 1) / cpw, 1, 256 * 16);
    const int kgrid = grid_for(nk, 64, 1 << 20);
    const bool tt = h->any_tet_tet;
    // Dual polygons kept between the passes: at most (pairs with a non-empty polygon) x n_dir slots
    const size_t dpcap = h->any_bristle ? (size_t)h->stats[2] * n_dir + 64 : 64;
    HIP_TRY(h, h->dual_poly.ensure(dpcap * kDpFields));
    HIP_TRY(h, h->dual_pkey.ensure(dpcap));
    HIP_TRY(h, h->dual_cnt.ensure(4));
    HIP_TRY(h, hipMemsetAsync(h->dual_cnt.p, 0, sizeof(int) * 4, st));
    a.dpoly = h->dual_poly.p; a.dpoly_key = h->dual_pkey.p; a.dpcount = h->dual_cnt.p; a.dpcap = (long long)dpcap;
    if (tt) hipLaunchKernelGGL((k_narrow_dual<true>), dim3(grid), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((k_narrow_dual<false>), dim3(grid), dim3(64), 0, st, a);
    if (h->any_bristle) {
        const int pgrid = grid_for(dpcap, 64, 256 * 16);
        hipLaunchKernelGGL((k_dual_poly<1>), dim3(pgrid), dim3(64), 0, st, a);
        hipLaunchKernelGGL(k_dual_eig, dim3(kgrid), dim3(64), 0, st, a);
        hipLaunchKernelGGL((k_dual_poly<2>), dim3(pgrid), dim3(64), 0, st, a);
    }
    hipLaunchKernelGGL(k_dual_final, dim3(kgrid), dim3(64), 0, st, a);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(d_wrench, a.d_wrench, sizeof(double) * nk * 6, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(d_sdot, a.d_sdot, sizeof(double) * nk * 6, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return PFC_OK;
}

int pfc_set_option(pfc_handle h, const char *name, long long value) {
    if (!h || !name) return PFC_ERR_BAD_ARG;
    if (!std::strcmp(name, "debug")) h->opt_debug = value != 0;
    else if (!std::strcmp(name, "profile")) h->opt_profile = value != 0;
    else if (!std::strcmp(name, "max_levels")) h->opt_max_levels = (int)value;
    else if (!std::strcmp(name, "bfs_levels")) h->opt_bfs_levels = (int)value;
    else if (!std::strcmp(name, "graph")) h->opt_graph = value != 0;
    else if (!std::strcmp(name, "split_min")) h->opt_split_min = (int)value;
    else if (!std::strcmp(name, "no_filter")) { h->opt_no_filter = (int)value; h->ghave[0] = h->ghave[1] = false; }
    else return fail(h, PFC_ERR_BAD_ARG, "unknown option %s", name);
    return PFC_OK;
}

int pfc_get_stats(pfc_handle h, long long *out8) {
    if (!h || !out8) return PFC_ERR_BAD_ARG;
    for (int k = 0; k < 8; ++k) out8[k] = h->stats[k];
    return PFC_OK;
}

int pfc_get_stage_ms(pfc_handle h, float *out6) {
    if (!h || !out6) return PFC_ERR_BAD_ARG;
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

int pfc_last_parts(pfc_handle h) { return h ? h->last_parts : 0; }

int pfc_debug_pairs(pfc_handle h, int item, int *pairs, int *clip_n, int cap) {
    if (!h) return -PFC_ERR_BAD_ARG;
    if (!h->opt_debug) return -fail(h, PFC_ERR_STATE, "debug option is off");
    if (h->pending) { int rc = check_eval(h); if (rc) return -rc; }
    if (item < 0 || item >= h->last_n_items) return -fail(h, PFC_ERR_BAD_ARG, "bad item");
    size_t nc = (size_t)h->stats[1];
    std::vector<WorkRec> c(nc);
    std::vector<int> cn(nc);
    if (nc) {
        if (hipMemcpy(c.data(), h->cand.p, sizeof(WorkRec) * nc, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(cn.data(), h->clip_n.p, sizeof(int) * nc, hipMemcpyDeviceToHost) != hipSuccess)
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
    if (!h->opt_debug) return -fail(h, PFC_ERR_STATE, "debug option is off");
    if (h->pending) { int rc = check_eval(h); if (rc) return -rc; }
    if (item < 0 || item >= h->last_n_items) return -fail(h, PFC_ERR_BAD_ARG, "bad item");
    size_t nt = (size_t)h->last_tslots;
    std::vector<int> ti(nt);
    std::vector<double> td(nt * 8);
    if (nt) {
        if (hipMemcpy(ti.data(), h->trac_item.p, sizeof(int) * nt, hipMemcpyDeviceToHost) != hipSuccess)
            return -fail(h, PFC_ERR_HIP, "copy failed");
        for (int a = 0; a < 8; ++a)
            if (hipMemcpy(td.data() + a * nt, h->trac_d.p + a * h->tcap, sizeof(double) * nt, hipMemcpyDeviceToHost) != hipSuccess)
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
    if (h->pending) { int rc = check_eval(h); if (rc) return -rc; }
    if (item < 0 || item >= h->last_n_items) return -fail(h, PFC_ERR_BAD_ARG, "bad item");
    std::vector<double> r(kResStride);
    int ic[4];
    if (hipMemcpy(r.data(), h->res.p + (size_t)item * kResStride, sizeof(double) * kResStride, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(ic, h->icnt.p + 4 * (size_t)item, sizeof ic, hipMemcpyDeviceToHost) != hipSuccess)
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
    double *dw = nullptr, *dx = nullptr, *dj = nullptr, *df = nullptr;
    int *db = nullptr;
    HIP_TRY(h, hipMalloc((void **)&df, sizeof(double) * nf));
    HIP_TRY(h, hipMemsetAsync(df, 0, sizeof(double) * nf, h->stream));
    if (n) {
        HIP_TRY(h, hipMalloc((void **)&dw, sizeof(double) * n * 6));
        HIP_TRY(h, hipMalloc((void **)&dx, sizeof(double) * n * 12));
        HIP_TRY(h, hipMalloc((void **)&dj, sizeof(double) * (nj ? nj : 1)));
        HIP_TRY(h, hipMalloc((void **)&db, sizeof(int) * n * 3));
        HIP_TRY(h, hipMemcpyAsync(dw, wrench, sizeof(double) * n * 6, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(dx, x_w_r2, sizeof(double) * n * 12, hipMemcpyHostToDevice, h->stream));
        if (nj) HIP_TRY(h, hipMemcpyAsync(dj, jac, sizeof(double) * nj, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(db, body_1, sizeof(int) * n, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(db + n, body_2, sizeof(int) * n, hipMemcpyHostToDevice, h->stream));
        if (scene) HIP_TRY(h, hipMemcpyAsync(db + 2 * n, scene, sizeof(int) * n, hipMemcpyHostToDevice, h->stream));
        ScatterArgs a;
        a.n_items = n_items; a.nv = nv; a.wrench = dw; a.x_w_r2 = dx; a.body_1 = db; a.body_2 = db + n;
        a.scene = scene ? db + 2 * n : nullptr; a.jac = dj; a.f = df;
        const long long tot = (long long)n_items * nv;
        hipLaunchKernelGGL(k_scatter, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, a);
        HIP_TRY(h, hipGetLastError());
    }
    HIP_TRY(h, hipMemcpyAsync(f_out, df, sizeof(double) * nf, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    (void)hipFree(df);
    if (dw) (void)hipFree(dw);
    if (dx) (void)hipFree(dx);
    if (dj) (void)hipFree(dj);
    if (db) (void)hipFree(db);
    return PFC_OK;
}

int pfc_debug_stamps(pfc_handle h, long long *out16) {
    if (!h || !out16) return PFC_ERR_BAD_ARG;
    if (h->pending) { int rc = check_eval(h); if (rc) return rc; }
    unsigned long long v[16] = {0};
    if (h->stamps.p) HIP_TRY(h, hipMemcpy(v, h->stamps.p, sizeof v, hipMemcpyDeviceToHost));
    for (int k = 0; k < 16; ++k) out16[k] = (long long)v[k];
    out16[7] = h->last_undecided;
    return PFC_OK;
}

int pfc_selftest_math(pfc_handle h, int n, const double *x, const double *y, double *out3n) {
    if (!h || n <= 0 || !x || !y || !out3n) return PFC_ERR_BAD_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    HIP_TRY(h, hipMalloc((void **)&dx, sizeof(double) * n));
    HIP_TRY(h, hipMalloc((void **)&dy, sizeof(double) * n));
    HIP_TRY(h, hipMalloc((void **)&dout, sizeof(double) * 3 * (size_t)n));
    HIP_TRY(h, hipMemcpy(dx, x, sizeof(double) * n, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(dy, y, sizeof(double) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_selftest, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, dx, dy, dout);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(out3n, dout, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dout);
    return PFC_OK;
}

}  // extern "C"
