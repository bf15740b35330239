// pfc_bp.h -- broadphase kernels: level-synchronous seed expansion, the single-precision workgroup depth-first kernel with its cooperative exact test, and the all-Float64 descent kept for A/B checks.  Included by pfc_hip.hip inside namespace pfc (device code only).
#pragma once

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
    int n_items;
};

__device__ __forceinline__ void count_per_item(int *icnt, int item, int slot, bool listed, bool flag, int n = 1) {
    // per-item integer counter: one atomic per run of equal items in the wave
    if (__ballot(flag) == 0) return;
    const Seg sg = seg_setup(listed ? item : -1);
    const int t = seg_sum(flag ? n : 0, sg);
    if (sg.tail && sg.valid && t != 0) atomicAdd(&icnt[4 * (size_t)item + slot], t);
}

// a whole NodeRec through the global address space (nine 16-byte global_load instead of flat_load)
__device__ __forceinline__ NodeRec load_node(const NodeRec *n) {
    union U { vec4i v[9]; NodeRec r; __device__ U() {} } u;
    const gvec4i *p = (const gvec4i *)n;
#pragma unroll
    for (int k = 0; k < 9; ++k) u.v[k] = p[k];
    return u.r;
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
        int fa0 = 0, fa1 = 0, fb0 = 0, fb1 = 0;   // WorkRec.pad flags of the children: bit 0 / 1 = a / b is a leaf (negative link)
        if (active) {
            w = g.fin[idx];
            if ((unsigned)w.item >= (unsigned)g.n_items) {     // never follows an index that cannot be an item
                atomicOr(g.status, kStHole);
                w.item = 0; w.a = 0; w.b = 0; w.pad = 0;
                active = false;
            }
        }
        if (active) {
            const ItemRec *it = g.items + w.item;
            const NodeRec a = load_node(it->nodes1 + w.a);
            const NodeRec b = load_node(it->nodes2 + w.b);
            hit = bb_bb_intersect(a, b, it->R12, it->t12);
            la = a.leaf != kInternal; lb = b.leaf != kInternal;
            ca0 = node_index(a.child0); ca1 = node_index(a.child1); cb0 = node_index(b.child0); cb1 = node_index(b.child1);
            fa0 = a.child0 < 0 ? 1 : 0; fa1 = a.child1 < 0 ? 1 : 0; fb0 = b.child0 < 0 ? 2 : 0; fb1 = b.child1 < 0 ? 2 : 0;
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
                // entries are stored one by one up to the capacity: a skipped slot below it would be read by the next
                // level as whatever the buffer held before (wild item / node indices)
                WorkRec c[4];
                for (int k = 0; k < 4; ++k) { c[k].item = w.item; c[k].a = w.a; c[k].b = w.b; c[k].pad = 0; }
                if (two) {
                    if (la) {  // leaf_1: descend tree_2 (:97-98)
                        c[0].b = cb0; c[0].pad = 1 | fb0;
                        c[1].b = cb1; c[1].pad = 1 | fb1;
                    } else {   // leaf_2: descend tree_1 (:101-103)
                        c[0].a = ca0; c[0].pad = fa0 | 2;
                        c[1].a = ca1; c[1].pad = fa1 | 2;
                    }
                } else {       // (1.1,2.1) (1.2,2.1) (1.1,2.2) (1.2,2.2) (:104-107)
                    c[0].a = ca0; c[0].b = cb0; c[0].pad = fa0 | fb0;
                    c[1].a = ca1; c[1].b = cb0; c[1].pad = fa1 | fb0;
                    c[2].a = ca0; c[2].b = cb1; c[2].pad = fa0 | fb1;
                    c[3].a = ca1; c[3].b = cb1; c[3].pad = fa1 | fb1;
                }
                for (int k = 0; k < 4; ++k)
                    if (k < nout && pos + k < g.fcap) g.fout[pos + k] = c[k];
                if (pos + nout > g.fcap) atomicOr(g.status, kStFrontierOvf);
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
    int n_items;
};

// one ticket per wave from a device-wide counter: lane 0 takes it, the wave reads it back as a scalar
__device__ __forceinline__ int next_ticket(int *ctr) {
    int t = 0;
    if (lane_id() == 0) t = atomicAdd(ctr, 1);
    return __builtin_amdgcn_readfirstlane(t);
}

// =================================================================================================================
// broadphase main kernel: the same wave-cooperative depth-first descent as k_bp_dfs, in single precision on one
// 48-byte NodeF record per node.  Measured (in-kernel stamps): the Float64 kernel is latency-bound at 2 waves per SIMD
// (246 VGPRs, two 144-byte scattered records per lane and iteration), not ALU-bound; the Float32 kernel needs a
// third of the registers and a third of the cache-line requests, and is exact in the following sense.
//
// For a node pair it forms the centre offset v = R_a_b c_b + (t_a_b - c_a) (of the order of the box sizes) and everything
// else in Float32, rotations on unit quaternions: q = conj(q_a) (x) q_a_b (x) q_b (q_a, q_b from the node
// records, q_a_b formed once per item from its pose (k_setup_items); each is checked in Float64 to reproduce its matrix to 4 u per
// entry, u = 2^-24), R = matrix of q, t = R_a' v by the rotation formula, then the 15 axes d = |T.L| - (r_a + r_b).
// The error of R is below 142 u, so |d_float - d_reference| < 320 u S + eabs with S = |v|_1 + sum e_a + sum e_b
// (internal-internal pairs carry identity quaternions and only the pose's error: 24 u S) and eabs = 24 u x the size of
// the scene, the price of forming v from Float32 centres.  d > E proves
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
    unsigned long long *stamps;   // diagnostic builds: [8..12] cycles in pop+load / test / barrier+prefix / push, iterations
    int n_items;
};

__device__ __forceinline__ NodeF load_nodef(const NodeF *n) {
    // three 16-byte loads of one 48-byte record
    const gvec4i *p = (const gvec4i *)n;
    union { vec4i v[3]; NodeF f; } u;
    u.v[0] = p[0]; u.v[1] = p[1]; u.v[2] = p[2];
    return u.f;
}

// Error radius E of the single-precision test (why 24 u and 320 u; u = 2^-24, round to nearest, |fl(x op y) - (x op y)|
// <= u |x op y|, a fused multiply-add rounds once; S = |v|_1 + sum e_a + sum e_b is formed from the SAME rounded inputs
// the axes use).  The reference evaluates, per axis L, d_ref = |T.L| - (r_a + r_b) in Float64 with abs_R = |R| + 1e-14
// (src/obb/bb_intersection.jl:10,29-72); its own roundings (~1e-16 S) and the 1e-14 (<= 2e-7 u per unit of extent) are
// far below one u S and are covered by the slack left below.
//   Notation.  R(q) is the polynomial map of quat_to_R (1 - 2(y^2 + z^2), 2(xy - wz), ...); it satisfies
//   R(p (x) q) = R(p) R(q) for UNIT quaternions and R(s q) = I + s^2 (R(q) - I) for a scaled one.
//   (0) Centre offset (round 3; before, v_i = fl32 of the Float64 offset, |dv_i| <= u |v_i|, which (3) still budgets).  v_i is
//       formed from fl32(R_a_b), fl32(c_b), fl32(t_a_b), fl32(c_a) as a difference and three fused multiply-adds.  Rounded
//       inputs: u (|t_i| + |c_a,i|) + 2 u sum_j |R_ij| |c_b,j|; the four operations round partial sums bounded by
//       B_i = |t_i| + |c_a,i| + sum_j |R_ij| |c_b,j|: together |dv_i| <= 5.1 u B_i, and B_i <= Babs = cmax_a + cmax_b + max_i
//       |t_a_b,i| (cmax = max over the mesh's nodes of |c|_1 >= |c|_2 >= a row of R_a_b against c_b).  This error is ABSOLUTE:
//       it does not shrink with the boxes.  t = R_a' v is a rotation of v: a perturbation dv changes every axis value |T.L| by
//       at most |dv|_2 |L|_2 <= sqrt 3 x 5.1 u Babs x (1 + 1e-5) < 8.9 u Babs (face axes: unit L; edge axes: (R_uj, R_vj) is
//       part of a unit column), and S (formed from the computed v) by at most 3 x 5.1 u Babs, i.e. k S by < 1e-4 u Babs.
//       The radius is E = k S + eabs with eabs = 24 u Babs (k_setup_items: ItemRec.bp_eabs, rounded up): 2.7 x the need.  A
//       scene whose meshes sit far from their frame origins relative to the box sizes only sends more pairs to the exact test.
//   Inputs.  e = fl32(extent): relative u.  Every quaternion used
//   here (q_a, q_b: pfc_add_mesh; q_a_b: pose_quat) has passed, IN FLOAT64, the check |R(q_f) - R_ref|_max <= 4 u and
//   | |q_f|^2 - 1 | <= 2.25 u; a node / pose that fails is exact_only and never decided here.
//   (1) Representation.  With q^ = q_f / |q_f|: |R(q^) - R(q_f)| = |1/|q_f|^2 - 1| |R(q_f) - I| <= 2.26 u x 2, so
//       |R(q^) - R_ref| <= rho = 8.6 u for each of the three rotations.  For unit quaternions
//       R(conj q_a^ (x) q_ab^ (x) q_b^) = R(q_a^)' R(q_ab^) R(q_b^) exactly; against R_a' R_a_b R_b the three error matrices
//       contribute (sqrt 3 + 3 + sqrt 3) rho = 55.6 u (rows / columns of a rotation have 1-norm <= sqrt 3).  The exact
//       product Q of the three STORED quaternions has |Q|^2 within 6.8 u of 1: |R(Q) - R(Q^)| <= 13.6 u.  Together 69.2 u.
//   (2) Arithmetic.  quat_mul evaluates each component as a four-term dot product with four roundings of partial sums
//       bounded by |a| |b| <= 1 + 3 u: |dp_i| <= 4 u after the first product; the second product carries it with a factor
//       |q_b|_1 <= 2 and adds its own 4 u: the computed q is Q + e with |e_i| <= 12 u.  R is quadratic in q: a diagonal
//       entry moves by 4 (|y| |e_y| + |z| |e_z|) <= 48 sqrt 2 u = 68 u, an off-diagonal one by 2 (|x| + |y| + |w| + |z|) 12 u
//       <= 48 u (second order: < 1e-3 u).  quat_to_R itself: <= 5 u (diagonal: two products of magnitude <= 2, their sum,
//       the difference from 1).
//       |dR| <= 69.2 + 68 + 5 < 143 u per entry.
//   (3) t = R(q_a_f)' v by quat_rot_inv, exact in real arithmetic (polynomial identity, no unit assumption): 4 u |v|_1 from the
//       representation, u |v_i| from the input; c = u x v - w v: three roundings of partial sums <= |v|_2: 3 u |v|_1;
//       d = u x c (|c|_2 <= |v|_2): sqrt 2 x 3 u + 2 u = 6.3 u |v|_1; t = v + 2 d: 12.6 u + u.  |dt_i| <= 19 u |v|_1, and
//       |t|_2 <= |v|_2 (1 + 1e-6).
//   (4) Axes, leaf involved.  Face axis of B, |sum_i R_ij t_i| - (sum_i |R_ij| e_a,i + e_b,j), is the worst: dR against
//       |t|_1 <= sqrt 3 |v|_1: 248 u |v|_1; dt against a column of R: 33 u |v|_1; three fused roundings 5 u |v|_1; r_a: (143 +
//       1 + 3) u sum e_a; e_b,j: u; sum and difference 2 u S: below 290 u S.  Edge-edge axes: dR (|t_u| + |t_v|) <= 143 sqrt 2
//       u |v|_1 = 203 u, dt 27 u, roundings 3 u; r_a + r_b: 146 u sum e; + 2 u S: below 236 u S.  Face axis of A: below 150 u S.
//       E = 320 u S holds with 10 % to spare.
//   (5) Axes, both boxes axis aligned (identity quaternions; every internal x internal pair): the two quaternion products
//       return q_a_b(float) exactly (factors 1 and 0), quat_rot_inv returns v exactly, so |dR| <= 4 u + 5 u and t = v.
//       Face axis of A: u |t_i| + u e_a,i + (9 + 1 + 3) u sum e_b + 2 u S <= 15 u S; face axis of B: (9 + 1 + 3) u |v|_1 +
//       13 u sum e_a + u e_b,j + 2 u S <= 15 u S; edge-edge: (9 + 1) u on each product + 2 u of fused rounding, + 2 u S:
//       <= 14 u S.  E = 24 u S.
// Hence |d_float - d_ref| < E = k S + eabs on every axis: d_float > E proves the reference separates on that axis, d_float < -E on all
// 15 axes proves it does not separate on any; everything else is settled by the exact Float64 test.  The constants in the
// code are 1.44e-6 (>= 24 u = 1.4305e-6) and 1.92e-5 (>= 320 u = 1.9073e-5).  tests/test_gpu_broadphase.py probes touching
// configurations at 1 -/+ {1e-3, 1e-6, 1e-9, 1e-12} against the oracle for both box kinds; tests/test_bp_error_bound.py
// samples a NumPy Float32 emulation of this function against these bounds (worst observed: |dR| 13 u, |d_float - d_ref| 8 u S).
// one node pair of k_bp_dfs32: returns 0 = separated, 1 = overlapping, 2 = undecided
__device__ __forceinline__ int test_pair_f32(const NodeF &a, const NodeF &b, bool any_leaf, const float *R12,
                                             const float *q12, const float *t12, float eabs) {
    // centre offset v = R_a_b c_b + (t_a_b - c_a) in Float32 on Float32 inputs: |dv_i| <= 5.1 u (|c_a|_1 + |c_b|_2 + |t_a_b,i|),
    // an ABSOLUTE error that the radius carries as eabs (Error radius E, (0))
    float v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        v[i] = __builtin_fmaf(R12[i + 6], b.c[2], __builtin_fmaf(R12[i + 3], b.c[1], __builtin_fmaf(R12[i], b.c[0], t12[i] - a.c[i])));
    // R = R_a' R_a_b R_b as ONE rotation: q = conj(q_a) (x) q_a_b (x) q_b, then its matrix; t = R_a' v by the quaternion
    // rotation formula: 32 + 24 + 18 instructions instead of 2 x 24 (two matrices) + 27 + 9 + 27 (two 3x3 products)
    float p[4], q[4], R[9], t[3];
    quat_mul<true>(a.q, q12, p);
    quat_mul<false>(p, b.q, q);
    quat_to_R(q, R);
    quat_rot_inv(a.q, v, t);
    const float S = ((__builtin_fabsf(v[0]) + __builtin_fabsf(v[1])) + __builtin_fabsf(v[2])) +
                    ((a.e[0] + a.e[1]) + a.e[2]) + ((b.e[0] + b.e[1]) + b.e[2]);
    const float E = __builtin_fmaf(any_leaf ? 1.92e-5f : 1.44e-6f, S, eabs);   // 320 u S, 24 u S; + 24 u x the scene's size
    int verdict = sat15_f32_core(a.e, b.e, t, R, E);
    // huge or non-finite inputs are not for this filter (sat15_f32_core): !(S < 1e18) also catches NaN -- which is also how
    // a node whose quaternion failed its check announces itself (e[0] = NaN, pfc_add_mesh)
    if (!(S < 1.0e18f)) verdict = 2;
    return verdict;
}

constexpr int kDfsBlock = 256;
constexpr int kDfsStack32 = 2560;   // node pairs per workgroup (20 KiB)
constexpr int kDfsOut32 = 1280;     // staged candidates per workgroup (10 KiB)

// The exact Float64 BB_BB_intersect (general composition, src/obb/bb_intersection.jl:2-74) of the node pairs the
// Float32 test leaves undecided (~2e-5 of all node tests), evaluated COOPERATIVELY: 16 lanes per pair.  Lanes 0..8
// form one entry each of R_a' R_a_b, then of R_tot = (R_a' R_a_b) R_b (lanes 9..11 the translation), through LDS, and
// lanes 0..14 test one of the 15 axes each.  Every entry / axis is the same expression, in the same order, as in
// bb_compose() / sat15(), so the boolean is the reference's bit for bit, and the per-lane register need is a few
// dozen instead of the ~220 of the one-lane-per-pair Float64 test (which cost the kernel a third of its occupancy
// when inlined, and as a no-inline call needed scratch).  xs: 33 doubles per 16-lane group.
template <int BLK>
__device__ __forceinline__ void exact_pairs_coop(const NodeRec *nodes1, const NodeRec *nodes2, const double *pose,
                                                 const int2 *und_l, int n_def, double *xs, int *und_v, int tid) {
    const int grp = tid >> 4, sub = tid & 15;
    double *T = xs + grp * 33, *tt = T + 9, *R = T + 12, *aR = T + 21, *t = T + 30;
    for (int c0 = 0; c0 < n_def; c0 += BLK / 16) {
        const int j = c0 + grp;
        const bool valid = j < n_def;
        int2 e = make_int2(0, 0);
        if (valid) e = und_l[j];
        const GNodeRec *na = (const GNodeRec *)(nodes1 + node_index(e.x)), *nb = (const GNodeRec *)(nodes2 + node_index(e.y));
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
            // (the extents are read from the node records by a lane-dependent index: as private arrays `ea[u]` they lived in
            // scratch memory -- 32 bytes per lane of the library's largest kernel, for a path 5e-4 of the node tests take)
#define EA_(i) (na->e[(i)])
#define EB_(i) (nb->e[(i)])
#define R_(i, j) R[(i) + 3 * (j)]
#define AR_(i, j) aR[(i) + 3 * (j)]
            if (sub < 3) {          // face test 1/2 (:29-32)
                const int i = sub;
                const double rb = (AR_(i, 0) * EB_(0) + AR_(i, 1) * EB_(1)) + AR_(i, 2) * EB_(2);
                sep = (EA_(i) + rb) < __builtin_fabs(t[i]);
            } else if (sub < 6) {   // face test 2/2 (:35-38)
                const int jj = sub - 3;
                const double tl = __builtin_fabs((R_(0, jj) * t[0] + R_(1, jj) * t[1]) + R_(2, jj) * t[2]);
                const double ra = (AR_(0, jj) * EA_(0) + AR_(1, jj) * EA_(1)) + AR_(2, jj) * EA_(2);
                sep = (ra + EB_(jj)) < tl;
            } else {                // cross tests (:56-72): row m of the cross block, column jj
                const int m = (sub - 6) / 3, jj = (sub - 6) % 3;
                const int u = (m + 1) % 3, v = (m + 2) % 3;
                const int p100 = jj == 0 ? 1 : 0, p221 = jj == 2 ? 1 : 2;
                const double tl = __builtin_fabs(t[v] * R_(u, jj) - t[u] * R_(v, jj));
                // sat15 writes the two products of ra with the lower axis index first; a + b == b + a exactly
                const double ra = EA_(u) * AR_(v, jj) + EA_(v) * AR_(u, jj);
                const double rb = EB_(p100) * AR_(m, p221) + EB_(p221) * AR_(m, p100);
                sep = (ra + rb) < tl;
            }
#undef EA_
#undef EB_
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

// Seed tickets of the workgroup kernel: workgroup b starts with seed b (no atomic), further seeds are handed out by
// the counter, offset by the grid size -- and not asked for at all when the grid already covers every seed (a small
// scene: each returning atomic is ~2 us of a ~19 us kernel).  One ticket per workgroup: thread 0 takes it, LDS
// broadcast between two barriers; the early return is uniform over the workgroup.
__device__ __forceinline__ int next_ticket_block(int *ctr, int *slot, int n_seed) {
    if ((int)gridDim.x >= n_seed) return n_seed;
    if (threadIdx.x == 0) *slot = (int)gridDim.x + atomicAdd(ctr, 1);
    __syncthreads();
    const int t = *slot;
    __syncthreads();
    return t;
}

template <int BLK>
__device__ __forceinline__ void flush_candidates(const Dfs32Args &g, const int2 *ob, int n_out, int item, int tid,
                                                 int *s_base) {
    // all threads of the workgroup call this (n_out is uniform)
    if (tid == 0) *s_base = atomicAdd(g.ccount, n_out);
    __syncthreads();
    const int base = *s_base;
    // On overflow the part of the run that still fits IS written: the narrowphase of an overflowing evaluation runs
    // over the first ccap slots (its results are discarded, the evaluation is re-issued with a longer list), and a
    // slot skipped here would hand it whatever the freshly allocated buffer held -- wild item / element indices.
    for (int j = tid; j < n_out; j += BLK) {
        if (base + j < g.ccap) {
            WorkRec c;
            c.item = item; c.a = ob[j].x; c.b = ob[j].y; c.pad = 0;
            g.cand[base + j] = c;
        }
    }
    if (tid == 0 && base + n_out > g.ccap) atomicOr(g.status, kStCandOvf);
    __syncthreads();
}

template <int BLK>
__global__ void __launch_bounds__(BLK, 4) k_bp_dfs32(Dfs32Args g) {
    constexpr int kWaves = BLK / 64, kStack = BLK * 10, kOut = BLK * 5;
    __shared__ int2 stk[kStack];
    __shared__ int2 ob[kOut];
    __shared__ int s_cnt[kWaves][2];      // per wave: candidates, pushed pairs of the current iteration
    __shared__ int2 und_l[BLK];           // node pairs the Float32 test left undecided in the last iteration
    __shared__ int und_v[BLK];            // their exact verdicts
    __shared__ double xs[(BLK / 16) * 33];
    __shared__ int s_seed, s_base, s_def[2];   // s_def: parked-pair counters, alternating by iteration parity
    __shared__ double s_pose[12];         // R_a_b (9, column-major), t_a_b (3) of the current seed's item
    __shared__ float s_q12[4];            // unit quaternion of R_a_b (Float32), from the item record
    __shared__ float s_posef[13];         // fl32 of R_a_b, t_a_b (the centre offset of the single-precision test) and bp_eabs
    __shared__ int s_pose_exact;          // the pose failed pose_quat's check: every test of the seed is settled exactly
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int n_seed = *g.n_seed;
    if (n_seed > g.seed_cap) n_seed = g.seed_cap;
    // the first seed record is fetched together with the seed count (blockIdx.x < grid <= seed_cap: in bounds)
    // (as a 16-byte vector: a WorkRec selected by `cond ? s_first : g.seeds[sd]` was given a stack slot -- one dead 16-byte store,
    // but 32 bytes of scratch per lane in the dispatch of the library's largest kernel)
    const vec4i s_first = ((const gvec4i *)g.seeds)[blockIdx.x];
    // one ticket per workgroup (same loop shape as k_bp_dfs: condition in the for header, no break)
    for (int sd = blockIdx.x; sd < n_seed; sd = next_ticket_block(g.next_seed, &s_seed, n_seed)) {
        vec4i s_v = s_first;
        if (sd != (int)blockIdx.x) s_v = ((const gvec4i *)g.seeds)[__builtin_amdgcn_readfirstlane(sd)];
        WorkRec s;
        s.item = s_v.x; s.a = s_v.y; s.b = s_v.z; s.pad = s_v.w;
        if ((unsigned)s.item >= (unsigned)g.n_items) {   // uniform over the workgroup: an unwritten seed slot is skipped, not followed
            if (tid == 0) atomicOr(g.status, kStHole);
            continue;
        }
        const int item = __builtin_amdgcn_readfirstlane(s.item);   // uniform: scalar loads of the pose below
        const ItemRec *it = g.items + item;
        // the item's pose lives in LDS (broadcast reads inside the iteration) rather than in 33 registers that would
        // stay live across the call of the exact test
        if (tid < 9) { const double x = it->R12[tid]; s_pose[tid] = x; s_posef[tid] = (float)x; }
        else if (tid < 12) { const double x = it->t12[tid - 9]; s_pose[tid] = x; s_posef[tid] = (float)x; }
        else if (tid < 16) s_q12[tid - 12] = it->q12[tid - 12];     // formed and checked once per item (k_setup_items, pose_quat)
        else if (tid == 16) s_pose_exact = it->pose_exact;
        else if (tid == 17) s_posef[12] = it->bp_eabs;
        const NodeF *n1 = it->nf1, *n2 = it->nf2;
        int sp = 1, n_out = 0, n_test = 0, n_cand = 0, n_def = 0, n_und = 0;
        if (tid == 0) {
            // stack entries hold node links: ~index (negative) for a leaf, index for an internal node
            const int sa = (s.pad & 1) ? ~s.a : s.a;     // leaf flags travel with the seed (k_setup_items, k_bp_expand)
            const int sb = (s.pad & 2) ? ~s.b : s.b;
            stk[0] = make_int2(sa, sb);
            s_def[0] = s_def[1] = 0;
        }
        __syncthreads();
#ifdef PFC_STAMPS
        unsigned long long c_a = 0, c_b = 0, c_c = 0, c_d = 0, c_it = 0, c_p = 0, c_w1 = 0, c_w2 = 0;   // c_w*: inside the two barriers, every wave
#endif
        // every workgroup must reach its exit: the iteration guard stops a corrupt (cyclic) tree from spinning forever
        int par = 0;   // parity of the iteration: selects the parking counter
        for (int guard = 0; (sp > 0 || n_def > 0) && guard < (1 << 22); ++guard, par ^= 1) {
            // Either settle the pairs the previous iteration left undecided (exact Float64 test; their children still
            // have the room that iteration reserved for them), or pop up to 256 pairs, but never more than the stack can
            // take back as children (4 per pair).  n_def is uniform over the workgroup.
            unsigned long long u0 = 0, u1 = 0, u2 = 0, u3 = 0, u4 = 0; (void)u0; (void)u1; (void)u2; (void)u3; (void)u4;
            STAMP(u0);
            const bool settle = n_def > 0;
            int pw = (kStack - g.reserve - sp) / 3;
            int p = sp < BLK ? sp : BLK;
            if (pw < 1) pw = 1;
            if (p > pw) p = pw;
            if (settle) {
                p = n_def;
                exact_pairs_coop<BLK>(it->nodes1, it->nodes2, s_pose, und_l, n_def, xs, und_v, tid);   // ends with a barrier
            }
            const bool act = tid < p;
            int2 e = make_int2(0, 0);
            if (act) e = settle ? und_l[tid] : stk[sp - 1 - tid];
            // no barrier here: the stack and the parking list are only written after the barrier below the test, by
            // which every thread has done this read
            if (!settle) { sp -= p; n_test += p; } else { n_und += p; }
            int verdict = 0, a0 = 0, a1 = 0, b0 = 0, b1 = 0;
            if (settle && act) verdict = und_v[tid];
            const bool la = act && e.x < 0, lb = act && e.y < 0;
            if (act) {
                const NodeF a = load_nodef(n1 + node_index(e.x));
                const NodeF b = load_nodef(n2 + node_index(e.y));
                a0 = a.link0; a1 = a.link1; b0 = b.link0; b1 = b.link1;
#ifdef PFC_STAMPS
                { float keep = a.e[0] + b.e[0] + a.q[3] + b.q[3]; asm volatile("" ::"v"(keep)); }   // the loads have landed
                STAMP(u1);
#endif
                if (!settle) {
                    float R12[9], t12[3], q12[4];
#pragma unroll
                    for (int k = 0; k < 9; ++k) R12[k] = s_posef[k];
#pragma unroll
                    for (int k = 0; k < 3; ++k) t12[k] = s_posef[9 + k];
#pragma unroll
                    for (int k = 0; k < 4; ++k) q12[k] = s_q12[k];
                    // Issue priority: the ~250 instructions of the test run at the default level, everything else of an
                    // iteration (ballots, prefix, push, pop, the next node loads: a few dozen instructions between two
                    // barriers and two memory round trips) one level above it.  In-kernel stamps had shown those short
                    // phases taking as long as the test itself because they queue behind the tests of the other three
                    // workgroups on the SIMD; with the raised level a wave reaches its barrier / gets its loads out
                    // first (exclusive 2.06 -> 1.95 ms for the 8 192-pose batch, step 4.40 -> 4.27 ms; levels 1, 2, 3
                    // alike).  The same in the clip kernel and k_integ changed nothing.
                    __builtin_amdgcn_s_setprio(0);
                    verdict = test_pair_f32(a, b, la || lb, R12, q12, t12, s_posef[12]);
                    __builtin_amdgcn_s_setprio(1);
                    if (s_pose_exact) verdict = 2;
                }
            }
            STAMP(u2);
            // an undecided pair is parked for the next iteration (verdict 2 only comes from the Float32 test)
            if (verdict == 2) und_l[atomicAdd(&s_def[par], 1)] = e;
            const bool hit = verdict == 1;
            const bool is_cand = hit && la && lb;
            const bool two = hit && (la != lb);
            const bool four = hit && !la && !lb;
            const unsigned long long mc = __ballot(is_cand), m2 = __ballot(two), m4 = __ballot(four);
            if (lane == 0) {
                s_cnt[wave][0] = __builtin_popcountll(mc);
                s_cnt[wave][1] = 2 * __builtin_popcountll(m2) + 4 * __builtin_popcountll(m4);
            }
#ifdef PFC_STAMPS
            const unsigned long long w1a = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();
#ifdef PFC_STAMPS
            c_w1 += __builtin_amdgcn_s_memtime() - w1a;
#endif
            STAMP(u3);
            n_def = s_def[par];                     // read by everyone between this barrier and the next
            if (tid == 0) s_def[par ^ 1] = 0;       // the next iteration's counter: its last readers passed a barrier ago
            int c_off = 0, p_off = 0, c_tot = 0, p_tot = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) {
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
#ifdef PFC_STAMPS
            const unsigned long long w2a = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();
#ifdef PFC_STAMPS
            c_w2 += __builtin_amdgcn_s_memtime() - w2a;
#endif
            if (n_out > kOut - BLK || (sp == 0 && n_def == 0 && n_out > 0)) {
                flush_candidates<BLK>(g, ob, n_out, item, tid, &s_base);
                n_cand += n_out;
                n_out = 0;
            }
#ifdef PFC_STAMPS
            STAMP(u4);
            if (!settle) {   // summed in registers: an atomic per iteration would itself dominate the timing
                c_a += u1 - u0; c_b += u2 - u1; c_c += u3 - u2; c_d += u4 - u3; c_it += 1; c_p += (unsigned long long)p;
            }
#endif
        }
#ifdef PFC_STAMPS
        if (lane == 0 && g.stamps) { atomicAdd(&g.stamps[0], c_w1); atomicAdd(&g.stamps[1], c_w2); atomicAdd(&g.stamps[2], c_it); }
        if (tid == 0 && g.stamps) {
            atomicAdd(&g.stamps[8], c_a); atomicAdd(&g.stamps[9], c_b); atomicAdd(&g.stamps[10], c_c);
            atomicAdd(&g.stamps[13], c_d); atomicAdd(&g.stamps[11], c_it); atomicAdd(&g.stamps[12], c_p);
        }
#endif
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
    union { vec4i v[4]; NodeHead h; } u;
    const gvec4i *p = (const gvec4i *)n;
    u.v[0] = p[0]; u.v[1] = p[1]; u.v[2] = p[2]; u.v[3] = p[3];
    return u.h;
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
        if ((unsigned)s.item >= (unsigned)g.n_items) {
            if (lane == 0) atomicOr(g.status, kStHole);
            continue;
        }
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
            const int sa = (((const GNodeRec *)n1)[s.a].leaf != kInternal) ? ~s.a : s.a;
            const int sb = (((const GNodeRec *)n2)[s.b].leaf != kInternal) ? ~s.b : s.b;
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
                        for (int k = 0; k < 9; ++k) Ra[k] = ((const GNodeRec *)n1)[ia].R[k];
                    }
                    if (lb) {
#pragma unroll
                        for (int k = 0; k < 9; ++k) Rb[k] = ((const GNodeRec *)n2)[ib].R[k];
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
                for (int j = lane; j < n_out; j += 64) {      // partial runs are written on overflow (see flush_candidates)
                    if (base + j < g.ccap) {
                        WorkRec c;
                        c.item = item; c.a = ob[j].x; c.b = ob[j].y; c.pad = 0;
                        g.cand[base + j] = c;
                    }
                }
                if (lane == 0 && base + n_out > g.ccap) atomicOr(g.status, kStCandOvf);
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

