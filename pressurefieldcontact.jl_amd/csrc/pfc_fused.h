// pfc_fused.h -- small scenes: the whole evaluation of ONE item by ONE workgroup in ONE launch.  Included by pfc_hip.hip
// inside namespace pfc (device code only).
//
// What Radau actually evaluates (src/radau/radau_functions.jl:2-14,64-70 on scenes like test/boxes.jl: a handful of
// instructions, <= a few hundred candidate pairs each) is latency, not throughput: the batched pipeline is six to
// fourteen dependent kernels, each paying a launch boundary and its own chain of memory round trips, for ~200 ops.
// Here one 256-thread workgroup per (instruction, pose) item does everything force_single_elastic_intersection!
// (src/contact_algorithms_non_friction.jl:70-84) does, without leaving the CU:
//   0. item record (instruction parameters, mesh pointers, pose, twist, s) -> LDS: two dependent loads
//   1. broadphase: small trees whole, large ones by the heads of their first kFuNodes nodes (the device node order is
//      breadth-first, so these are the top levels), are copied to LDS once; the workgroup depth-first descent of
//      k_bp_dfs32 (same node tests, 256 per iteration) then runs from LDS with the exact Float64 test per pair;
//      candidates stay in an LDS list
//   2. clip: one thread per candidate (rounds of 256): gather, tet coordinates, trivial reject, Sutherland-Hodgman in
//      the LDS polygon ring, Cartesian polygon + centroid -- the expressions of k_narrow, bit for bit
//   3. integrate: the fan triangles of all polygons of the round are dealt out one per thread (the batched kernel walks
//      a polygon's <= 8 triangles x 3 points serially in one lane: ~17 k cycles; here a thread does 3 points), sums
//      stay in registers, one block reduction at the end
//   4. bristle items: cop -> second pass (patch stiffness moments about the cop, the reference's own second pass,
//      src/contact_algorithms_friction.jl:147-169) -> 6x6 eigen on wave 0 -> third pass (friction) -> epilogue
// Outputs: wrench, sdot, counts and a per-item status word; no global counters, no atomics, nothing to zero.
// An item that does not fit (candidates > kFuCand) reports kStFusedOvf; the host then takes the batched path.
#pragma once

constexpr int kFuBlock = 256;
constexpr int kFuWaves = kFuBlock / 64;
constexpr int kFuCand = 4096;      // candidate pairs per item held in LDS (32 KiB)
constexpr int kFuNodes = 256;      // node heads per tree cached in LDS (2 x 16 KiB) ...
constexpr int kFuFull = (kFuNodes * 4) / 9;   // ... or the whole tree (144-byte NodeRec) if it has at most this many nodes
constexpr int kFuStack = 4096;     // node pairs (32 KiB: with the two node caches exactly the 64 KiB the polygon ring needs anyway)
constexpr unsigned kStFusedOvf = 512u;   // item does not fit the fused kernel's LDS lists: use the batched path
constexpr unsigned kStFusedDualSkip = 1024u;   // too many (polygon, direction) pairs for the in-kernel Dual passes: batched Dual path
constexpr int kFuDualLanes = 128;        // (polygon, direction) lanes per Dual clip round: value + partial ring = 512 B per lane
constexpr int kFuDualMax = 512;          // (polygon, direction) pairs an item may have for the in-kernel Dual passes

// Everything an item needs from its instruction, in one 256-byte record (built at pfc_finalize): one load instead of
// the chain InsDev -> MeshDev x 2 of the batched path.
struct alignas(16) InsFull {
    const NodeRec *nodes1, *nodes2;
    const NodeF *nf1, *nf2;
    const TriRec *tri;       // mesh_1 triangles, or null (tet-tet)
    const TetRec *tet;       // mesh_2 tets
    const TetRec *tet1;      // mesh_1 tets (tet-tet) or null
    const double *eps1, *eps2;
    double chi, Ebar, Ebar1, mu_s, mu_d, v_c, tau, k_bar, magic;
    int model, nq, n_node1, n_node2, reserve, pad0;
    double cmax12;           // cmax(mesh_1) + cmax(mesh_2): the absolute part of the single-precision test's error radius (ItemRec.bp_eabs)
    int pad[8];
};
static_assert(sizeof(InsFull) == 208 || sizeof(InsFull) == 256 || sizeof(InsFull) % 16 == 0, "InsFull layout");

struct FuArgs {
    int n_items, n_ins;
    const int *ins_ids;      // may be null
    const double *pose, *twist, *s;
    const InsFull *ins;
    double *wrench, *sdot;
    int *counts;             // may be null
    int *fout;               // per item 8 ints: status, counts[4], seq (written LAST, behind a system-scope fence), 0, 0
    int seq;                 // evaluation sequence number (never 0): the host may poll fout[8 i + 5] instead of synchronising
    // Dual evaluation (pfc_eval_dual of a small all-regularized scene): n_dir > 0
    int n_dir;
    const double *d_pose, *d_twist;     // (item, dir) x 24 / x 6 partials of the inputs
    double *d_wrench, *d_sdot;          // OUT (item, dir) x 6
    // Hand-over to the batched Dual passes (k_narrow_dual ...) for scenes the in-kernel Dual passes do not take (bristle
    // items, many polygons): the item record, the per-item counters and the list of candidates that gave a polygon, in
    // the formats of the batched value pass.  emit_ctr[0] (zero on entry) ends as the number of listed pairs.
    ItemRec *emit_items;
    WorkRec *emit_cand;
    int *emit_surv, *emit_icnt, *emit_ctr;
    int emit_cap;
    unsigned long long *stamps;   // diagnostic builds (-DPFC_STAMPS): block 0 leaves wall-clock stamps (10 ns ticks) of its phases
    // Teams (k_fused<.., true>, round 3): nw workgroups per item -- blockIdx.x = item * nw + rank.  team: per (workgroup,
    // phase) 2 kTeamSlots 8-byte granules {32 data bits, 32-bit tag = seq} of partial sums (team_sum).
    int nw;
    int team_seeds;     // the descent of a team is shared out once a level holds this many pairs per workgroup (kTeamSeeds / kTeamSeedsBig)
    unsigned long long *team;
    int team_fault;     // diagnostic (option "team_fault", default -1): this rank behaves as if its wait for the team after pass 0 had
                        // timed out while everybody else saw every granule -- stale totals, kStFusedOvf in ITS status word only
    // Broadphase pose of a Dual evaluation (pfc_eval_dual_bp): the reference culls with m.float's transforms whatever the
    // state of the Dual scenario (calcTriTetIntersections!, src/contact_algorithms_non_friction.jl:94-101); null: the pose itself
    const double *bp_pose;
    int f32;            // option "fused_f32" (default 1): single-precision SAT filter in front of the exact test (0: Float64 only)
};
constexpr int kTeamSlots = 48;           // doubles a workgroup publishes per phase (first phase: 10 sums, 4 counters, status, 27 moments, their reference point)
constexpr int kTeamMaxWg = 48;           // workgroups per item at most
constexpr int kTeamMaxBlocks = 256;      // item * nw + rank < this (one workgroup per CU: every workgroup of a launch is resident)
constexpr int kTeamSeedsBig = 32;        // the same for pairs too big for one workgroup: 12 full-size C3 poses 170 -> 144 us, 8: 114 -> 108, 16 poses of a 7 380-leaf pair 101 -> 92 (mid-sized pairs lose with it: four 4 880-leaf pairs 82 -> 93 us)
constexpr int kTeamSeeds = 16;           // the descent is shared out once a level holds this many pairs per workgroup ...
constexpr int kTeamShareMax = 768;       // ... or this many in all: its children (four per pair at most) still fit the stack above it
constexpr int kTeamSpinMax = 1 << 15;    // bounded wait for the team (a poll is ~2 us: ~65 ms): if a workgroup never arrives -- teams of several handles
                                         // launched at once can each be PARTLY resident on a full chip -- the item reports kStFusedOvf and the host re-issues
                                         // the evaluation on the batched path (the one-launch kernel then stays off for 64 evaluations of that handle)

#ifdef PFC_STAMPS
#define FSTAMP(k)                                                                                   \
    do {                                                                                            \
        if (blockIdx.x == 0 && threadIdx.x == 0 && g.stamps) g.stamps[k] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define FSTAMP(k) do { } while (0)
#endif

struct FuItem {
    InsFull ins;
    double pose[24];         // R21 9, t21 3, R12 9, t12 3
    double twist[6];         // w 3, v 3
    double s[6];
};

// block totals of N per-thread values: wave totals on DPP, then the four wave totals meet in LDS; every thread returns
// the same N totals in out[] (same summation order in every thread).  red: kFuWaves x 32 doubles.
template <int N>
__device__ __forceinline__ void block_partials(const double *v, double *red, int tid) {
    static_assert(N <= 32, "red holds 32 values per wave");
    const int lane = tid & 63, wave = tid >> 6;
    __syncthreads();                       // red may still be read by the previous reduction
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double t = wave_total(v[k], lane);
        if (lane == 0) red[wave * 32 + k] = t;
    }
    __syncthreads();
}
template <int N>
__device__ __forceinline__ void block_totals(const double *v, double *out, double *red, int tid) {
    block_partials<N>(v, red, tid);
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = ((red[k] + red[32 + k]) + red[64 + k]) + red[96 + k];
}

struct FuBp {                 // broadphase view of the scratch region (aliases the polygon ring)
    int2 stk[kFuStack];
    vec4i na[kFuNodes * 4], nb[kFuNodes * 4];   // per tree 16 KiB: kFuNodes heads (4 x 16 B) or <= kFuFull whole NodeRec (9 x 16 B)
};
union FuScratch {
    FuBp bp;
    double ring[8 * 4 * kFuBlock];   // polygon ring: [slot][coord][thread]
    __device__ FuScratch() {}
};

// polygon ring of thread `t`: logical vertex k lives in physical slot (rb + k) & 7
#define FR(t, rb, k, c) ring[(((((rb) + (k)) & 7) * 4 + (c)) * kFuBlock) + (t)]
// after the conversion to Cartesian coordinates the fourth coordinate of every slot is free: slots 0..5 hold n̂, centroid
#define FX(t, s) ring[((((s) * 4) + 3) * kFuBlock) + (t)]

// Team sum of n <= kTeamSlots values, one per thread tid < n (`mine`); returns the team's total of slot tid to thread tid,
// summed in rank order by every workgroup alike (all workgroups of a team continue with bit-identical totals).  Slot
// or_slot is combined by bitwise OR of its integer value instead (status words).
// The exchange is made of self-validating 8-byte GRANULES (MI355X guide, "Persistent kernels: synchronisation and hand-off
// price list", handoff-1to1 / R2): a double leaves as two granules {32 bits of data, 32-bit tag = the launch's sequence
// number}, each written by ONE sc1 (agent-scope, write-through) 8-byte store and polled with sc1 8-byte loads until its tag
// is this launch's.  No flag, no counter, no fence, nothing to zero between evaluations (the sequence number never
// repeats within 2^32 launches; the buffer starts zeroed and sequence numbers start at 1): one store and one load round
// trip per team sum.  A first version -- sc1 partials, a returning atomic add on an arrival counter, a poll of the counter,
// then the loads -- cost 6-7 us per team sum; this one 3-4.
__device__ __forceinline__ void team_gather(const FuArgs &g, int item, int nw, int phase, int n, double mine, int tid,
                                            double *s_team, int *s_flag, unsigned &status) {
    const unsigned long long tag = (unsigned long long)(unsigned)g.seq << 32;
    unsigned long long *my = g.team + ((size_t)blockIdx.x * 3 + phase) * (2 * kTeamSlots);
    if (tid < n) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(mine);
        __hip_atomic_store(my + 2 * tid, tag | (bits & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(my + 2 * tid + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) *s_flag = 0;
    __syncthreads();
    // thread -> (rank w2, slot t): every granule of a thread is requested before the first is looked at
    const unsigned long long *base = g.team + ((size_t)(item * nw) * 3 + phase) * (2 * kTeamSlots);
    constexpr int kPer = kTeamMaxWg * kTeamSlots / kFuBlock;
    static_assert(kTeamMaxWg * kTeamSlots % kFuBlock == 0, "every granule of a full team is requested by exactly one thread");
    unsigned long long lo[kPer], hi[kPer];
    bool need[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int idx = tid + k * kFuBlock, w2 = idx / kTeamSlots, t = idx % kTeamSlots;
        need[k] = w2 < nw && t < n;
        lo[k] = hi[k] = 0ull;
    }
    bool pending = true;
    for (int spins = 0; pending && spins < kTeamSpinMax; ++spins) {
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int idx = tid + k * kFuBlock, w2 = idx / kTeamSlots, t = idx % kTeamSlots;
            if (need[k]) {
                const unsigned long long *q = base + (size_t)w2 * 3 * (2 * kTeamSlots) + 2 * t;
                lo[k] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hi[k] = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        pending = false;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            if (need[k]) {
                if ((lo[k] >> 32 << 32) == tag && (hi[k] >> 32 << 32) == tag) need[k] = false;
                else pending = true;
            }
        }
        if (pending) __builtin_amdgcn_s_sleep(1);
    }
    if (pending) *s_flag = 1;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int idx = tid + k * kFuBlock;
        if (idx < nw * kTeamSlots)
            s_team[idx] = __longlong_as_double((long long)(((hi[k] & 0xFFFFFFFFull) << 32) | (lo[k] & 0xFFFFFFFFull)));
    }
    if (phase == 0 && g.team_fault >= 0 && (int)blockIdx.x - item * nw == g.team_fault) {      // uniform; diagnostic option only
        // wrong but finite totals, so that the rank goes on into the friction pass as a rank with stale granules would
        if (tid == 0) *s_flag = 1;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int idx = tid + k * kFuBlock;
            if (idx < nw * kTeamSlots) s_team[idx] *= 0.5;
        }
    }
    __syncthreads();
    if (*s_flag) status |= kStFusedOvf;    // a team-mate never arrived: the host re-issues on the batched path (uniform over the workgroup)
}
// the rank-order total of slot tid (tid < n) over the nw rows of s_team; slot or_slot by bitwise OR of its integer value
__device__ __forceinline__ double team_total(const double *s_team, int nw, int n, int or_slot, int tid) {
    double acc = 0.0;
    if (tid < n) {
        for (int w2 = 0; w2 < nw; ++w2) {
            const double x = s_team[w2 * kTeamSlots + tid];
            if (tid == or_slot) acc = (double)((unsigned)acc | (unsigned)x); else acc += x;
        }
    }
    return acc;
}
__device__ __forceinline__ double team_sum(const FuArgs &g, int item, int nw, int phase, int n, double mine, int or_slot, int tid,
                                           double *s_team, int *s_flag, unsigned &status) {
    team_gather(g, item, nw, phase, n, mine, tid, s_team, s_flag, status);
    const double acc = team_total(s_team, nw, n, or_slot, tid);
    __syncthreads();        // s_team is free again
    return acc;
}

// TT: the scenario has tet-tet instructions (as k_narrow<TT>).  MW: a TEAM of g.nw workgroups per item (one big pair --
// BASELINE config 3 as written is a single 9 680-tet x 5 120-triangle pair: a lone workgroup would need 110 broadphase
// iterations and its candidates do not fit LDS).  Every workgroup of a team runs the top of the descent redundantly --
// the traversal is deterministic, so all hold the same stack -- until it holds kTeamSeeds pairs per workgroup; rank r
// keeps the pairs j = r (mod nw) and descends them alone (no shared queue, no communication in the broadphase); each
// workgroup clips and integrates the candidates IT found; the three per-pass sums of the bristle model (normal wrench +
// cop, patch stiffness, friction) meet in team_sum, every workgroup forms cop / K / its eigen-decomposition from the same
// totals, rank 0 writes the item's outputs.
template <bool TT, bool MW = false>
__global__ void __launch_bounds__(kFuBlock) k_fused(FuArgs g) {
    __shared__ FuScratch S;
    __shared__ int2 cand[kFuCand];
    __shared__ unsigned fan[8 * kFuBlock];      // (thread that owns the polygon) | (fan triangle) << 16
    __shared__ double s_epsr[4][kFuBlock];      // ϵ_r² of the lane's tet
    __shared__ int s_np[kFuBlock], s_rb[kFuBlock];
    __shared__ double red[kFuWaves * 32];
    __shared__ int s_cnt[kFuWaves][4], s_scan[kFuWaves];
    __shared__ FuItem I;
    __shared__ double s_acc[kAccStride], s_res[kResStride];
    __shared__ EigScratch E;
    __shared__ double s_aR12[9];
    __shared__ double s_bp[12];                 // R_a_b (9, column-major), t_a_b (3) of the broadphase
    __shared__ float s_posef[13], s_q12[4];     // single-precision filter: fl32 of R_a_b, t_a_b, the error radius' absolute part; R_a_b as a quaternion
    __shared__ int s_pose_exact;                // the pose is not a proper rotation (pose_quat): every test of the item is the exact one
    __shared__ int s_plist[kFuCand];            // candidates that gave a polygon (Dual passes)
    __shared__ int s_npoly;
    __shared__ double s_dacc[16][6];            // per direction: partials of the wrench
    __shared__ ItemRec s_it;                    // what dual_integrate reads of an item
    double *ring = S.ring;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nw = MW ? g.nw : 1;
    const int item = MW ? (int)blockIdx.x / nw : (int)blockIdx.x;
    const int wr = MW ? (int)blockIdx.x - item * nw : 0;      // rank in the team
    __shared__ double s_team[MW ? kTeamMaxWg * kTeamSlots : kTeamSlots];
    __shared__ double s_tres[kTeamSlots];
    __shared__ int s_tflag;
    unsigned status = 0;

    FSTAMP(0);
#ifdef PFC_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0 && g.stamps) g.stamps[10] = __builtin_amdgcn_s_memtime();
#endif
    // ==== 0. the item ==============================================================================================
    int id = g.ins_ids ? g.ins_ids[item] : item;
    if (id < 0 || id >= g.n_ins) { status |= kStBadIns; id = 0; }
    {
        constexpr int nw = (int)(sizeof(InsFull) / sizeof(int));
        static_assert(nw <= 64, "InsFull is loaded by the first wave");
        if (tid < nw) reinterpret_cast<int *>(&I.ins)[tid] = reinterpret_cast<const int *>(g.ins + id)[tid];
        if (tid >= 64 && tid < 88) {
            // I.pose: x_r2_r1 of the pose, then the x_r1_r2 the BROADPHASE culls with -- the pose's own, or (pfc_eval_dual_bp) the
            // one of m.float's state; in that case the pose's own x_r1_r2 (read by the tet-tet op and the hand-over only) is s_bp
            const double x = ((g.bp_pose && tid >= 76) ? g.bp_pose : g.pose)[24 * (size_t)item + (tid - 64)];
            I.pose[tid - 64] = x;
            if (!(__builtin_fabs(x) <= 1.79769313486231570815e308)) status |= kStNonFinite;
        }
        if (tid >= 96 && tid < 102) I.twist[tid - 96] = g.twist[6 * (size_t)item + (tid - 96)];
        if (tid >= 128 && tid < 134) I.s[tid - 128] = g.s ? g.s[6 * (size_t)item + (tid - 128)] : 0.0;
        if (g.bp_pose && tid >= 160 && tid < 172) {
            const double x = g.pose[24 * (size_t)item + 12 + (tid - 160)];
            s_bp[tid - 160] = x;
            if (!(__builtin_fabs(x) <= 1.79769313486231570815e308)) status |= kStNonFinite;
        }
    }
    if (tid == 0) s_npoly = 0;
    __syncthreads();
    const bool reg = I.ins.model == PFC_REGULARIZED;
    const int nq = (I.ins.nq == 1) ? 1 : 3;
    // a non-finite pose is reported ("Non-finite vertex likely", static_clip.jl:52), not traversed (uniform)
    bool pose_ok = true;
#pragma unroll
    for (int k = 0; k < 24; ++k) pose_ok &= (__builtin_fabs(I.pose[k]) <= 1.79769313486231570815e308);
    if (g.bp_pose) {
#pragma unroll
        for (int k = 0; k < 12; ++k) pose_ok &= (__builtin_fabs(s_bp[k]) <= 1.79769313486231570815e308);
    }

    FSTAMP(1);
    // ==== 1. broadphase (tree_tree_intersect, src/obb/tree_types.jl:88-111) ===========================================
    // Every node pair is decided by the reference's own Float64 BB_BB_intersect (src/obb/bb_intersection.jl:2-74) in the
    // iteration that pops it: axis-aligned shortcut for two merged boxes, general composition when a tight-fitted leaf
    // is involved.  (The batched kernel's Float32 filter + parked exact settle buys occupancy, which one workgroup per
    // CU does not need, and costs an extra iteration whenever pairs come back undecided -- every pair of an axis-aligned
    // scene like a box resting on a plane, whose parallel-edge cross axes are exactly degenerate.)
    int n_cand = 0, n_test = 0;
    if (pose_ok) {
        FuBp &B = S.bp;
        // node cache: the whole tree as NodeRec if it fits its 16 KiB area, else the 64-byte heads (c, e, links, flags)
        // of the first kFuNodes nodes (breadth-first order: the top levels)
        const int nn1 = I.ins.n_node1, nn2 = I.ins.n_node2;
        const bool full1 = nn1 <= kFuFull, full2 = nn2 <= kFuFull;
        const int nc1 = nn1 < kFuNodes ? nn1 : kFuNodes, nc2 = nn2 < kFuNodes ? nn2 : kFuNodes;   // nodes held in LDS
        if (full1) { for (int k = tid; k < nn1 * 9; k += kFuBlock) B.na[k] = ((const gvec4i *)I.ins.nodes1)[k]; }
        else { for (int k = tid; k < nc1 * 4; k += kFuBlock) B.na[k] = ((const gvec4i *)I.ins.nodes1)[(k >> 2) * 9 + (k & 3)]; }
        if (full2) { for (int k = tid; k < nn2 * 9; k += kFuBlock) B.nb[k] = ((const gvec4i *)I.ins.nodes2)[k]; }
        else { for (int k = tid; k < nc2 * 4; k += kFuBlock) B.nb[k] = ((const gvec4i *)I.ins.nodes2)[(k >> 2) * 9 + (k & 3)]; }
        if (tid == 0) {
            // stack entries hold node links: ~index (negative) for a leaf, index for an internal node
            B.stk[0] = make_int2(nn1 == 1 ? ~0 : 0, nn2 == 1 ? ~0 : 0);
        }
        const double *s_pose = I.pose + 12;          // R_a_b (9, column-major), t_a_b (3) of the broadphase (update_TT_Cache!, tree_types.jl:43-50)
        if (tid < 9) s_aR12[tid] = __builtin_fabs(s_pose[tid]) + 1.0e-14;    // abs_R of an all-identity pair (:10)
        if (tid < 12) s_posef[tid] = (float)s_pose[tid];
        if (tid == 192) {
            // what k_setup_items forms per item for the batched broadphase (pose_quat; "Error radius E", (0), pfc_bp.h)
            double R12[9];
            float q[4];
#pragma unroll
            for (int k = 0; k < 9; ++k) R12[k] = s_pose[k];
            // A frame axis of body 2 parallel to one of body 1 (|R_ij| = 1: a box resting on a plane, a yaw-only pose): the cross
            // axis of those two edges is exactly degenerate for every pair of merged boxes, the filter can never prove "no
            // separation" on it and every overlapping pair would come back undecided -- such an item starts with the filter off
            // (test/boxes.jl: one wasted single-precision test and the quaternion below, 1 us of a 31 us evaluation).
            bool aligned = false;
#pragma unroll
            for (int k = 0; k < 9; ++k) aligned |= __builtin_fabs(R12[k]) > 1.0 - 1.0e-12;
            q[0] = 1.0f; q[1] = q[2] = q[3] = 0.0f;
            s_pose_exact = (aligned || !pose_quat(R12, q)) ? 1 : 0;
            s_q12[0] = q[0]; s_q12[1] = q[1]; s_q12[2] = q[2]; s_q12[3] = q[3];
            const double tm = fmax(fmax(__builtin_fabs(s_pose[9]), __builtin_fabs(s_pose[10])), __builtin_fabs(s_pose[11]));
            s_posef[12] = (float)(1.4306e-6 * (I.ins.cmax12 + tm)) * 1.000001f;
        }
        __syncthreads();
        FSTAMP(2);
        const int reserve = I.ins.reserve;
        int sp = 1;
        bool ovf = false;
        // a node: head from LDS or global; R only where the link says leaf (tight-fitted box) or the head says "not
        // axis aligned" (a host-supplied tree may hold such internal boxes).  Explicit branches: a select between an LDS
        // and a global load would issue both.
#define FU_FETCH(u, LDSARR, FULL, NC, GLOB, IDX, LEAF)                                              \
    do {                                                                                            \
        const int idx_ = (IDX);                                                                     \
        const gvec4i *gp_ = (const gvec4i *)((GLOB) + idx_);                                        \
        const bool in_lds_ = idx_ < (NC);                                                           \
        const int base_ = (FULL) ? idx_ * 9 : idx_ * 4;                                             \
        if (in_lds_) { u.v[0] = LDSARR[base_]; u.v[1] = LDSARR[base_ + 1]; u.v[2] = LDSARR[base_ + 2]; u.v[3] = LDSARR[base_ + 3]; } \
        else { u.v[0] = gp_[0]; u.v[1] = gp_[1]; u.v[2] = gp_[2]; u.v[3] = gp_[3]; }               \
        if ((LEAF) || !u.r.aabb) {                                                                  \
            if (in_lds_ && (FULL)) { u.v[4] = LDSARR[base_ + 4]; u.v[5] = LDSARR[base_ + 5]; u.v[6] = LDSARR[base_ + 6]; u.v[7] = LDSARR[base_ + 7]; u.v[8] = LDSARR[base_ + 8]; } \
            else { u.v[4] = gp_[4]; u.v[5] = gp_[5]; u.v[6] = gp_[6]; u.v[7] = gp_[7]; u.v[8] = gp_[8]; } \
        } else {                                                                                    \
            u.r.R[0] = 1.0; u.r.R[1] = 0.0; u.r.R[2] = 0.0; u.r.R[3] = 0.0; u.r.R[4] = 1.0; u.r.R[5] = 0.0; \
            u.r.R[6] = 0.0; u.r.R[7] = 0.0; u.r.R[8] = 1.0;                                         \
        }                                                                                           \
    } while (0)
        union NodeU { vec4i v[9]; NodeRec r; __device__ NodeU() {} };
        // the single-precision view of a node (NodeF, pfc_kernels.h): converted from the cached head of an axis-aligned box (its
        // quaternion is the identity; (float) of the Float64 centre / extent is what pfc_add_mesh stored), else loaded
#define FU_FETCHF(f, LDSARR, FULL, NC, GLOBF, IDX)                                                  \
    do {                                                                                            \
        const int idx_ = (IDX);                                                                     \
        bool done_ = false;                                                                         \
        if (idx_ < (NC)) {                                                                          \
            const int base_ = (FULL) ? idx_ * 9 : idx_ * 4;                                         \
            const vec4i h3_ = LDSARR[base_ + 3];                                                    \
            if (h3_.w) {                                                                            \
                union { vec4i v[3]; double d[6]; } u_;                                              \
                u_.v[0] = LDSARR[base_]; u_.v[1] = LDSARR[base_ + 1]; u_.v[2] = LDSARR[base_ + 2];  \
                f.c[0] = (float)u_.d[0]; f.c[1] = (float)u_.d[1]; f.c[2] = (float)u_.d[2];          \
                f.e[0] = (float)u_.d[3]; f.e[1] = (float)u_.d[4]; f.e[2] = (float)u_.d[5];          \
                f.q[0] = 1.0f; f.q[1] = 0.0f; f.q[2] = 0.0f; f.q[3] = 0.0f;                         \
                f.link0 = h3_.z == kInternal ? h3_.x : h3_.z;                                       \
                f.link1 = h3_.z == kInternal ? h3_.y : -1;                                          \
                done_ = true;                                                                       \
            }                                                                                       \
        }                                                                                           \
        if (!done_) f = load_nodef((GLOBF) + idx_);                                                 \
    } while (0)
#ifdef PFC_STAMPS
        unsigned long long cy[4] = {0, 0, 0, 0}, it_n = 0;
#endif
        // Teams: the top of the descent runs LEVEL BY LEVEL, identically in every workgroup of the team (the traversal is
        // deterministic): the pairs of the current level sit in stk[0, hi), their children are pushed above them (from hi0 =
        // the level's size on), and when the level is used up the next one moves down.  Once a level holds enough pairs for
        // every rank (or the next might not fit the stack), rank wr keeps the pairs j = wr (mod nw) and goes on depth first,
        // alone.  (Shared out from the depth-first stack at 128 pairs, a single C3 pose had nearly all its work under two or
        // three of them: 270 us, slower than the batched path.)
        bool bfs = MW;
        // Single-precision filter in front of the exact test (the batched kernel's test_pair_f32: a third of the exact test's
        // cycles at one wave per SIMD).  A pair it leaves undecided is settled by the exact test IN THE SAME ITERATION (no parking:
        // one workgroup has nothing else to run meanwhile), so node tests, candidates and their order are unchanged.  Scenes whose
        // boxes are axis aligned to each other -- a box resting on a plane: every parallel-edge cross axis is exactly degenerate,
        // the filter cannot prove "no separation" on it -- would pay for both tests on every pair: once more than a quarter of an
        // iteration's live pairs come back undecided the workgroup stays with the exact test for the rest of the item (uniform
        // over the workgroup and, the traversal being deterministic, over a team).
        bool f32_on = g.f32 != 0 && s_pose_exact == 0;
        int lo = 0, hi = MW ? 1 : 0, hi0 = hi;      // the level: stk[lo, hi); its children from hi0 on
        const int t_share = nw * g.team_seeds < kTeamShareMax ? nw * g.team_seeds : kTeamShareMax;
        for (int guard = 0; (bfs ? hi > lo : sp > 0) && guard < (1 << 22); ++guard) {
            unsigned long long u0 = 0, u1 = 0, u2 = 0, u3 = 0, u4 = 0; (void)u0; (void)u1; (void)u2; (void)u3; (void)u4;
            STAMP(u0);
            // Lookahead.  An iteration costs ~4 000 cycles whatever the number of busy lanes (one Float64 test per lane, a
            // single wave per SIMD), and the top of a descent has 1, 2, 4 ... pairs: with few pairs on the stack the
            // children (and grandchildren) of a popped pair are tested in the SAME iteration by neighbouring lanes -- 8
            // (32) lanes per popped pair: lane 0 the pair, lanes 1..4 its child pairs, lanes 5..20 theirs.  A lane's
            // result counts only if every ancestor pair it descends from overlaps (the reference tests a pair iff its
            // parent pair intersects, tree_types.jl:88-111), so node tests and candidates are the reference's.
            const int room = kFuStack - reserve - sp;
            const int avail = bfs ? hi - lo : sp;      // pairs that may be popped now
            const int top = bfs ? hi : sp;            // ... from here downwards
            int LA = 1, p;
            if (avail <= 8 && room >= 64 * avail) { LA = 3; p = avail; }
            else if (avail <= 32 && room >= 16 * avail) { LA = 2; p = avail; }
            else {
                int pw = room / 3;
                p = avail < kFuBlock ? avail : kFuBlock;
                if (pw < 1) pw = 1;
                if (p > pw) p = pw;
            }
            const int gl = LA == 3 ? 32 : (LA == 2 ? 8 : 1);       // lanes per popped pair
            const int grp = tid / gl, r = tid - grp * gl;
            int lvl = 0, c1 = 0, c2 = 0;
            bool act = grp < p;
            if (r >= 1 && r <= 4) { lvl = 1; c1 = r - 1; }
            else if (LA == 3 && r >= 5 && r <= 20) { lvl = 2; c1 = (r - 5) >> 2; c2 = (r - 5) & 3; }
            else if (r >= 5) act = false;
            int2 e = make_int2(0, 0);
            if (act) e = B.stk[top - 1 - grp];
            // walk down to the lane's own pair: child order (1.1,2.1) (1.2,2.1) (1.1,2.2) (1.2,2.2) (:104-107), or the two
            // children of the internal node when the other one is a leaf (:97-103)
#define FU_LINKS(LDSARR, FULL, NC, GLOB, IDX) ((IDX) < (NC) ? LDSARR[((FULL) ? (IDX) * 9 : (IDX) * 4) + 3] : ((const gvec4i *)((GLOB) + (IDX)))[3])
            for (int step = 0; step < LA - 1; ++step) {
                if (act && step < lvl) {
                    const int c = step == 0 ? c1 : c2;
                    const bool xa = e.x < 0, xb = e.y < 0;
                    if ((xa && xb) || ((xa || xb) && c >= 2)) {
                        act = false;
                    } else {
                        if (!xa) {
                            const int ia = e.x;
                            vec4i lk;
                            if (ia < nc1) lk = B.na[(full1 ? ia * 9 : ia * 4) + 3]; else lk = ((const gvec4i *)(I.ins.nodes1 + ia))[3];
                            const int sel = xb ? c : (c & 1);
                            e.x = sel ? lk.y : lk.x;
                        }
                        if (!xb) {
                            const int ib = e.y;
                            vec4i lk;
                            if (ib < nc2) lk = B.nb[(full2 ? ib * 9 : ib * 4) + 3]; else lk = ((const gvec4i *)(I.ins.nodes2 + ib))[3];
                            const int sel = xa ? c : (c >> 1);
                            e.y = sel ? lk.y : lk.x;
                        }
                    }
                }
            }
#undef FU_LINKS
            if (bfs) hi -= p; else sp -= p;      // (level mode: the children go above the level, not into the popped slots)
            bool hit = false;
            int a0 = 0, a1 = 0, b0 = 0, b1 = 0, la_id = 0, lb_id = 0;
            const bool la = act && e.x < 0, lb = act && e.y < 0;
            bool need_exact = act, und = false;
#ifndef PFC_FUSED_F32
#define PFC_FUSED_F32 1      // 0: compile the single-precision filter out (A/B builds)
#endif
            // (restricted to waves that hold tight-fitted leaf boxes -- merged boxes have the exact test's axis-aligned shortcut --
            // the filter gained nothing: single full-size pose 93.3 us against 81.6 with the filter in every wave)
            const bool wave_f32 = PFC_FUSED_F32 && f32_on;
            if (wave_f32 && act) {
                // 48-byte single-precision nodes: a merged (axis-aligned) box held in LDS is converted from its cached head, every
                // other node comes from the NodeF array (three 16-byte loads instead of the four to nine of a NodeRec)
                NodeF fa, fb;
                FU_FETCHF(fa, B.na, full1, nc1, I.ins.nf1, node_index(e.x));
                FU_FETCHF(fb, B.nb, full2, nc2, I.ins.nf2, node_index(e.y));
#ifdef PFC_STAMPS
                { float keep = fa.e[0] + fb.e[0] + fa.q[3] + fb.q[3]; asm volatile("" ::"v"(keep)); }
                STAMP(u1);
#endif
                float R12f[9], t12f[3], q12f[4];
#pragma unroll
                for (int k = 0; k < 9; ++k) R12f[k] = s_posef[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) t12f[k] = s_posef[9 + k];
#pragma unroll
                for (int k = 0; k < 4; ++k) q12f[k] = s_q12[k];
                const int verdict = test_pair_f32(fa, fb, la || lb, R12f, q12f, t12f, s_posef[12]);
                a0 = fa.link0; a1 = fa.link1; b0 = fb.link0; b1 = fb.link1; la_id = fa.link0; lb_id = fb.link0;
                und = verdict == 2;
                need_exact = und;
                hit = verdict == 1;
            }
            if (__any(need_exact)) {       // (per wave: a wave whose pairs were all decided skips the exact test altogether)
                NodeU ua, ub;
                bool both_aabb = true;
                if (need_exact) {
                    FU_FETCH(ua, B.na, full1, nc1, I.ins.nodes1, node_index(e.x), la);
                    FU_FETCH(ub, B.nb, full2, nc2, I.ins.nodes2, node_index(e.y), lb);
                    const NodeRec &a = ua.r, &b = ub.r;
                    a0 = a.child0; a1 = a.child1; b0 = b.child0; b1 = b.child1; la_id = a.leaf; lb_id = b.leaf;
#ifdef PFC_STAMPS
                    { double keep = a.c[0] + b.c[0] + a.R[4] + b.R[4]; asm volatile("" ::"v"(keep)); }   // the loads have landed
                    if (!f32_on) STAMP(u1);
#endif
                    both_aabb = a.aabb && b.aabb;
                }
                // The form of the test is chosen per wave, never per lane (a lane-level branch would make a mixed wave run
                // both): the axis-aligned shortcut only if every active lane holds two merged boxes; the general composition
                // is exact for identity rotations too.
                const bool wave_aabb = __all(!need_exact || both_aabb);
                if (need_exact) {
                    const NodeRec &a = ua.r, &b = ub.r;
                    double R12[9], t12[3];
#pragma unroll
                    for (int k = 0; k < 9; ++k) R12[k] = s_pose[k];
#pragma unroll
                    for (int k = 0; k < 3; ++k) t12[k] = s_pose[9 + k];
                    if (wave_aabb) {
                        double aR12[9];
#pragma unroll
                        for (int k = 0; k < 9; ++k) aR12[k] = s_aR12[k];
                        hit = bb_bb_intersect_aabb(a.c, a.e, b.c, b.e, R12, aR12, t12);
                    } else {
                        hit = bb_bb_intersect(a, b, R12, t12);
                    }
                }
            }
            STAMP(u2);
            // live: the pair is one the reference tests (all its ancestors in this iteration overlap); fin: its expansion
            // is not already covered by lanes of this iteration
            const unsigned long long hm = __ballot(hit);
            const int gb = lane & ~(gl - 1);
            const bool live = act && (lvl == 0 || ((hm >> gb) & 1ull)) && (lvl <= 1 || ((hm >> (gb + 1 + c1)) & 1ull));
            const bool fin = live && hit && (lvl == LA - 1 || (la && lb));
            const bool is_cand = fin && la && lb;
            const bool two = fin && (la != lb);
            const bool four = fin && !la && !lb;
            const unsigned long long mc = __ballot(is_cand), m2 = __ballot(two), m4 = __ballot(four), ml = __ballot(live);
            const unsigned long long mu = __ballot(und && live);
            if (lane == 0) {
                s_cnt[wave][0] = __builtin_popcountll(mc);
                s_cnt[wave][1] = 2 * __builtin_popcountll(m2) + 4 * __builtin_popcountll(m4);
                s_cnt[wave][2] = __builtin_popcountll(ml);
                s_cnt[wave][3] = __builtin_popcountll(mu);
            }
            __syncthreads();
            STAMP(u3);
            int c_off = 0, p_off = 0, c_tot = 0, p_tot = 0, live_it = 0, und_it = 0;
#pragma unroll
            for (int w = 0; w < kFuWaves; ++w) {
                const int c = s_cnt[w][0], q = s_cnt[w][1];
                if (w < wave) { c_off += c; p_off += q; }
                c_tot += c; p_tot += q;
                live_it += s_cnt[w][2]; und_it += s_cnt[w][3];
            }
            n_test += live_it;
            if (4 * und_it > live_it) f32_on = false;      // uniform: see above
            if (is_cand) {
                const int pos = n_cand + c_off + prefix_count(mc);
                if (pos < kFuCand) cand[pos] = make_int2(la_id, lb_id);     // element indices
            }
            if (two | four) {
                const int pos = sp + p_off + 2 * prefix_count(m2) + 4 * prefix_count(m4);
                if (two) {
                    if (la) {  // leaf_1: descend tree_2 (:97-98)
                        B.stk[pos] = make_int2(e.x, b0); B.stk[pos + 1] = make_int2(e.x, b1);
                    } else {   // leaf_2: descend tree_1 (:101-103)
                        B.stk[pos] = make_int2(a0, e.y); B.stk[pos + 1] = make_int2(a1, e.y);
                    }
                } else {       // (1.1,2.1) (1.2,2.1) (1.1,2.2) (1.2,2.2) (:104-107)
                    B.stk[pos] = make_int2(a0, b0); B.stk[pos + 1] = make_int2(a1, b0);
                    B.stk[pos + 2] = make_int2(a0, b1); B.stk[pos + 3] = make_int2(a1, b1);
                }
            }
            n_cand += c_tot;
            sp += p_tot;
            if (n_cand > kFuCand) { ovf = true; sp = 0; hi = 0; lo = 0; bfs = false; }    // uniform: the item leaves for the batched path
            __syncthreads();
            if (MW && bfs && hi == lo && sp - hi0 < t_share && sp <= kFuStack / 2) {
                // the level is used up and the next one, stk[hi0, sp), is small: it stays where it is (no copy, no barrier)
                lo = hi0; hi = sp; hi0 = sp;
            } else if (MW && bfs && hi == lo) {
                // the level is used up: its children stk[hi0, sp) move down and become the next level -- or, if there are
                // enough of them, this rank's share of the depth-first descent
                const int n_next = sp - hi0;
                const bool share = n_next >= t_share;
                int2 keep[kFuStack / kFuBlock];
#pragma unroll
                for (int k = 0; k < kFuStack / kFuBlock; ++k) {
                    const int j = tid + k * kFuBlock;
                    keep[k] = j < n_next ? B.stk[hi0 + j] : make_int2(0, 0);
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < kFuStack / kFuBlock; ++k) {
                    const int j = tid + k * kFuBlock;
                    if (j < n_next && (!share || j % nw == wr)) B.stk[share ? j / nw : j] = keep[k];
                }
                if (share) {
                    sp = (n_next - wr + nw - 1) / nw;
                    if (wr != 0) { n_cand = 0; n_test = 0; }      // what has been counted and found so far stays with rank 0
                    bfs = false;
                } else {
                    sp = n_next; lo = 0; hi = n_next; hi0 = n_next;
                }
                __syncthreads();
            }
#ifdef PFC_STAMPS
            STAMP(u4);
            if (u1 == 0) u1 = u0;
            cy[0] += u1 - u0; cy[1] += u2 - u1; cy[2] += u3 - u2; cy[3] += u4 - u3; ++it_n;
#endif
        }
#ifdef PFC_STAMPS
        if (blockIdx.x == 0 && tid == 0 && g.stamps) {
            g.stamps[12] = cy[0]; g.stamps[13] = cy[1]; g.stamps[14] = cy[2]; g.stamps[15] = (cy[3] << 16) | it_n;
        }
#endif
#undef FU_FETCH
#undef FU_FETCHF
        if (MW && bfs) {      // the descent ended before it was shared out: every level is used up, rank 0 has it all
            sp = 0;
            if (wr != 0) { n_cand = 0; n_test = 0; }
        }
        if (ovf) status |= kStFusedOvf;
        else if (sp > 0) status |= kStAbort;
    }
    if (status & (kStFusedOvf | kStAbort)) n_cand = 0;   // uniform
    FSTAMP(3);
#ifdef PFC_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0 && g.stamps) g.stamps[8] = (unsigned long long)n_test;
#endif

    // ==== 2.-4. narrowphase ===========================================================================================
    // pass 0: normal wrench (+ regularized friction fused) + cop sums; bristle: pass 1 patch stiffness about the cop,
    // pass 2 friction.  With more than one round of candidates the polygons of earlier rounds are gone from the ring and
    // are clipped again in the later passes.
    // Items with more than one round of candidates first run ONLY the cheap front of the op on all of them (gather,
    // x_ζ²_r¹, tet coordinates, the bit-exact trivial reject: the expressions of the clip round below, hence the same
    // decisions) and compact the survivors in place at the front of the LDS list, so that the expensive clip rounds see
    // no trivially rejected candidate (C2: 463 candidates -> 254 survivors: one clip round instead of two).
    int n_list = n_cand;
    if (n_cand > kFuBlock && (!TT || I.ins.tet1 == nullptr)) {      // (tet-tet: the reject comes after the plane / tet polygon)
        int n_keep = 0;
        for (int base = 0; base < n_cand; base += kFuBlock) {
            const int ci = base + tid;
            bool surv = false;
            int2 cw = make_int2(0, 0);
            if (ci < n_cand) {
                cw = cand[ci];
                const GTetRec *tp = (const GTetRec *)(I.ins.tet + cw.y);
                const GTriRec *tr = (const GTriRec *)(I.ins.tri + cw.x);
                double Z[16], tv[9], R21[9], t21[3], X[16], z[3][4];
#pragma unroll
                for (int k = 0; k < 16; ++k) Z[k] = tp->xzr[k];
#pragma unroll
                for (int k = 0; k < 9; ++k) tv[k] = tr->v[k];
#pragma unroll
                for (int k = 0; k < 9; ++k) R21[k] = I.pose[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) t21[k] = I.pose[9 + k];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        X[i + 4 * j] = (Z[i] * R21[3 * j] + Z[i + 4] * R21[3 * j + 1]) + Z[i + 8] * R21[3 * j + 2];
                    X[i + 12] = ((Z[i] * t21[0] + Z[i + 4] * t21[1]) + Z[i + 8] * t21[2]) + Z[i + 12];
                }
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        z[k][i] = ((X[i] * tv[3 * k] + X[i + 4] * tv[3 * k + 1]) + X[i + 8] * tv[3 * k + 2]) + X[i + 12];
                bool finite = true;
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i) finite &= (__builtin_fabs(z[k][i]) <= 1.79769313486231570815e308);
                if (!finite) status |= kStNonFinite;
                bool reject = !finite;
#pragma unroll
                for (int i = 0; i < 4; ++i) reject |= (z[0][i] <= 0.0) && (z[1][i] <= 0.0) && (z[2][i] <= 0.0);
                surv = !reject;
            }
            const unsigned long long ms = __ballot(surv);
            if (lane == 0) s_cnt[wave][0] = __builtin_popcountll(ms);
            __syncthreads();                 // every entry of this round has been read: the front of the list may be overwritten
            int off = 0, tot = 0;
#pragma unroll
            for (int wv = 0; wv < kFuWaves; ++wv) { const int c = s_cnt[wv][0]; if (wv < wave) off += c; tot += c; }
            if (surv) cand[n_keep + off + prefix_count(ms)] = cw;     // n_keep + off + ... <= ci: never an entry still to be read
            n_keep += tot;
            __syncthreads();
        }
        n_list = n_keep;
    }
    const int n_round = (n_list + kFuBlock - 1) / kFuBlock;
    int n_nonempty = 0, n_trac = 0;
    double tot10[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) tot10[k] = 0.0;
    V3 cop = mk3(0.0, 0.0, 0.0), ts_c0 = cop, ts_e = cop;
    const V3 w = mk3(I.twist[0], I.twist[1], I.twist[2]), vl = mk3(I.twist[3], I.twist[4], I.twist[5]);
    const double chi = I.ins.chi, Ebar = I.ins.Ebar, v_c = I.ins.v_c, mu_s = I.ins.mu_s, mu_d = I.ins.mu_d;
    // Bristle items take TWO integration passes (round 3; three before): pass 0 sums, next to the normal wrench and the cop
    // sums, the 27 patch-stiffness moments (calc_patch_spatial_stiffness!, friction.jl:147-169) about a reference point c0
    // inside the patch -- the centroid of the first polygon this workgroup integrates -- and the parallel-axis terms
    // (shift_moments, pfc_br.h: what k_shift does for the batched path) move them to the cop once it is known; pass 2 is the
    // friction pass.  (The reference's own second pass over the TractionCache, x = r - cop, cost an integration pass, a
    // block reduction and, for a team, a team sum.)
    bool contact = false, have_c0 = false;
    V3 c0 = mk3(0.0, 0.0, 0.0);
    for (int pass = 0; pass < 3; pass += 2) {
        if (pass > 0 && (reg || !contact)) break;          // uniform
        double acc[37];
#pragma unroll
        for (int k = 0; k < 37; ++k) acc[k] = 0.0;
        int my_ne = 0, my_nt = 0;
        for (int rd = 0; rd < n_round; ++rd) {
            // ---- clip (only once if a single round holds every candidate) -------------------------------------------
            if (pass == 0 || n_round > 1) {
                __syncthreads();               // ring / fan of the previous round are free
                const int ci = rd * kFuBlock + tid;
                int n_poly = 0, rbase = 0;
                if (ci < n_list) {
                    const int2 cw = cand[ci];
                    const GTetRec *tp = (const GTetRec *)(I.ins.tet + cw.y);
                    const GTriRec *tr = (const GTriRec *)(I.ins.tri + cw.x);
                    double Z[16], tv[9], tn[3], V[12], er[4];
#pragma unroll
                    for (int k = 0; k < 16; ++k) Z[k] = tp->xzr[k];
                    if (!TT || I.ins.tet1 == nullptr) {
#pragma unroll
                        for (int k = 0; k < 9; ++k) tv[k] = tr->v[k];
#pragma unroll
                        for (int k = 0; k < 3; ++k) tn[k] = tr->n[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < 9; ++k) tv[k] = 0.0;
                        tn[0] = tn[1] = tn[2] = 0.0;
                    }
#pragma unroll
                    for (int k = 0; k < 12; ++k) V[k] = tp->xrz[k];
#pragma unroll
                    for (int k = 0; k < 4; ++k) er[k] = tp->epsr[k];
                    double R21[9], t21[3];
#pragma unroll
                    for (int k = 0; k < 9; ++k) R21[k] = I.pose[k];
#pragma unroll
                    for (int k = 0; k < 3; ++k) t21[k] = I.pose[9 + k];
                    double z[4][4];
                    int n_in = 3;
                    V3 nh_in = mk3(0.0, 0.0, 0.0);
                    if (!TT || I.ins.tet1 == nullptr) {
                        // ---- tri-tet op (non_friction.jl:196-215): x_ζ2_r1 = x_ζ2_r2 * x_r2_r1.mat (:204) ------------
                        double X[16];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
#pragma unroll
                            for (int j = 0; j < 3; ++j)
                                X[i + 4 * j] = (Z[i] * R21[3 * j] + Z[i + 4] * R21[3 * j + 1]) + Z[i + 8] * R21[3 * j + 2];
                            X[i + 12] = ((Z[i] * t21[0] + Z[i + 4] * t21[1]) + Z[i + 8] * t21[2]) + Z[i + 12];
                        }
#pragma unroll
                        for (int k = 0; k < 3; ++k)
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                z[k][i] = ((X[i] * tv[3 * k] + X[i + 4] * tv[3 * k + 1]) + X[i + 8] * tv[3 * k + 2]) + X[i + 12];
#pragma unroll
                        for (int i = 0; i < 4; ++i) z[3][i] = 0.0;
                        nh_in = mk3((R21[0] * tn[0] + R21[3] * tn[1]) + R21[6] * tn[2],
                                    (R21[1] * tn[0] + R21[4] * tn[1]) + R21[7] * tn[2],
                                    (R21[2] * tn[0] + R21[5] * tn[1]) + R21[8] * tn[2]);
                    } else {
                    // ---- tet-tet op (non_friction.jl:166-194) -----------------------------------------------------------
                    const GTetRec *t1 = (const GTetRec *)(I.ins.tet1 + cw.x);
                    double plane[4];
                    {
                        // ϵ_plane_r2 = (Ē2 ϵ2) x_ζ2_r2 - (Ē1 ϵ1) (x_ζ1_r1 x_r1_r2)   (find_plane_tet :164, :174-177)
                        double R12[9], t12[3], Z1[16], X1[16];
    #pragma unroll
                        for (int k = 0; k < 9; ++k) R12[k] = (g.bp_pose ? s_bp : I.pose + 12)[k];
    #pragma unroll
                        for (int k = 0; k < 3; ++k) t12[k] = (g.bp_pose ? s_bp : I.pose + 12)[9 + k];
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
                            Ee1[j] = I.ins.Ebar1 * ((const gdouble *)I.ins.eps1)[4 * (size_t)cw.x + j];
                            Ee2[j] = I.ins.Ebar * ((const gdouble *)I.ins.eps2)[4 * (size_t)cw.y + j];
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
                    if (!finite) status |= kStNonFinite;
                    // trivial reject (bit-exact shortcut of static_clip.jl:44, see k_narrow)
                    bool reject = !finite || n_in < 3;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        reject |= (z[0][i] <= 0.0) && (z[1][i] <= 0.0) && (z[2][i] <= 0.0) && (n_in < 4 || z[3][i] <= 0.0);
                    if (!reject) {
                        // ---- clip_in_tet_coordinates (static_clip.jl:7-23,34-201), in place in the LDS ring ----------
                        int n = n_in;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (k < n_in) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) FR(tid, rbase, k, i) = z[k][i];
                            }
                        bool err = false;
                        RingCol<kFuBlock> rg{ring, tid, rbase};
                        n = clip_ring_in_tet_coordinates(rg, n_in, err);     // pfc_clip.h
                        rbase = rg.rbase;
                        if (err) status |= kStNonFinite;
                        n_poly = n;
                    }
                    if (n_poly >= 3) {
                        // poly_r2 = mul_then_un_pad(x_r2_ζ2, poly_ζ2) (poly_eight.jl:83-98) fused with centroid(poly_r2, n̂2)
                        // (poly_eight.jl:35-52), as in k_narrow
                        const int n = n_poly;
                        auto conv = [&](int k) {
                            const double z0 = FR(tid, rbase, k, 0), z1 = FR(tid, rbase, k, 1), z2 = FR(tid, rbase, k, 2), z3 = FR(tid, rbase, k, 3);
                            const V3 x = mk3(((V[0] * z0 + V[3] * z1) + V[6] * z2) + V[9] * z3,
                                             ((V[1] * z0 + V[4] * z1) + V[7] * z2) + V[10] * z3,
                                             ((V[2] * z0 + V[5] * z1) + V[8] * z2) + V[11] * z3);
                            FR(tid, rbase, k, 0) = x.x; FR(tid, rbase, k, 1) = x.y; FR(tid, rbase, k, 2) = x.z;
                            return x;
                        };
                        const V3 a = conv(0);
                        V3 cc = conv(1);
                        double cum_sum = 0.0;
                        V3 cum_prod = mk3(0.0, 0.0, 0.0);
                        for (int k = 2; k < n; ++k) {
                            const V3 b = cc;
                            cc = conv(k);
                            const double ar = triangle_area(a, b, cc, nh_in);
                            cum_prod = cum_prod + ((a + b) + cc) * (1.0 / 3.0) * ar;
                            cum_sum += ar;
                        }
                        const V3 cen = (cum_sum == 0.0) ? a : cum_prod / cum_sum;
                        FX(tid, 0) = nh_in.x; FX(tid, 1) = nh_in.y; FX(tid, 2) = nh_in.z;
                        FX(tid, 3) = cen.x; FX(tid, 4) = cen.y; FX(tid, 5) = cen.z;
#pragma unroll
                        for (int k = 0; k < 4; ++k) s_epsr[k][tid] = er[k];
                        if (pass == 0) {
                            ++my_ne;
                            if (g.n_dir > 0 || g.emit_items) s_plist[atomicAdd(&s_npoly, 1)] = ci;
                        }
                    }
                }
                s_np[tid] = n_poly >= 3 ? n_poly : 0;
                s_rb[tid] = rbase;
                // ---- deal the fan triangles out: exclusive scan of the vertex counts over the workgroup -------------
                const int np = n_poly >= 3 ? n_poly : 0;
                const int incl = seg_incl_scan(np);
                if (lane == 63) s_scan[wave] = incl;
                __syncthreads();
                int off = incl - np;
                for (int wv = 0; wv < kFuWaves; ++wv) if (wv < wave) off += s_scan[wv];
                for (int k = 0; k < np; ++k) fan[off + k] = (unsigned)tid | ((unsigned)k << 16);
                __syncthreads();
            }
            if (pass == 0 && rd == 0) FSTAMP(4);
            const int n_fan = ((s_scan[0] + s_scan[1]) + s_scan[2]) + s_scan[3];
            if (!have_c0 && n_fan > 0) {       // uniform: the reference point of this workgroup's moments
                const int pt0 = (int)(fan[0] & 0xFFFFu);
                c0 = mk3(FX(pt0, 3), FX(pt0, 4), FX(pt0, 5));
                have_c0 = true;
            }
            // ---- integrate_over_polygon_patch! (non_friction.jl:217-265): one fan triangle per thread ------------------
            for (int wi = tid; wi < n_fan; wi += kFuBlock) {
                const unsigned fe = fan[wi];
                const int pt = (int)(fe & 0xFFFFu), k = (int)(fe >> 16);
                const int n = s_np[pt], rb = s_rb[pt];
                const int km = k == 0 ? n - 1 : k - 1;
                const V3 nh = mk3(FX(pt, 0), FX(pt, 1), FX(pt, 2)), cen = mk3(FX(pt, 3), FX(pt, 4), FX(pt, 5));
                const V3 v1 = mk3(FR(pt, rb, km, 0), FR(pt, rb, km, 1), FR(pt, rb, km, 2));
                const V3 v2 = mk3(FR(pt, rb, k, 0), FR(pt, rb, k, 1), FR(pt, rb, k, 2));
                const double er0 = s_epsr[0][pt], er1 = s_epsr[1][pt], er2 = s_epsr[2][pt], er3 = s_epsr[3][pt];
                double tW = 0.0, tm[3] = {0.0, 0.0, 0.0}, tq[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                PointParams pp;
                pp.w = w; pp.vl = vl; pp.chi = chi; pp.Ebar = Ebar; pp.er0 = er0; pp.er1 = er1; pp.er2 = er2; pp.er3 = er3; pp.nq = nq;
                // the traction points of this fan triangle: the shared statement of r, p, dA (fan_triangle_points, pfc_kernels.h)
                const int n_pt = fan_triangle_points(pp, v1, v2, cen, nh, [&](const V3 &r, const V3 &rdot, double p, double dA) {
                    const double p_dA = p * dA;
                    if (pass == 0) {
                        ++my_nt;
                        if (reg) {
                            // yes_contact!(::Regularized) (friction.jl:50-72), as fused in k_narrow
                            const V3 vt = vec_sub_vec_proj(rdot, nh);
                            const double m2 = dot_fma(vt, vt);
                            V3 T;
                            if (m2 < v_c * v_c) {
                                T = vt * (-(mu_s / v_c));
                            } else {
                                double ri = __builtin_amdgcn_rsq(m2);
                                const double hm = 0.5 * m2;
                                ri = ri * __builtin_fma(-hm * ri, ri, 1.5);
                                ri = ri * __builtin_fma(-hm * ri, ri, 1.5);
                                const double mg = m2 * ri;
                                const double mu = clamped_piecewise(mg, 2 * v_c, 3 * v_c, mu_s, mu_d);
                                T = vt * (-(mu * ri));
                            }
                            const V3 tk = (nh + T) * p_dA;
                            const V3 ta = cross_fma(r, tk);
                            acc[0] += ta.x; acc[1] += ta.y; acc[2] += ta.z;
                            acc[3] += tk.x; acc[4] += tk.y; acc[5] += tk.z;
                        } else {
                            // normal_wrench_cop (normal.jl:17-34) and the moments of calc_patch_spatial_stiffness!
                            // (friction.jl:147-169) about the polygon's own centroid: W, sum w x, sum w x x' with x = r - cen;
                            // n̂ is constant over the polygon
                            const V3 rc = r - cen;
                            const double wx = p_dA * rc.x, wy = p_dA * rc.y, wz = p_dA * rc.z;
                            tW += p_dA;
                            tm[0] += wx; tm[1] += wy; tm[2] += wz;
                            tq[0] = __builtin_fma(wx, rc.x, tq[0]); tq[1] = __builtin_fma(wx, rc.y, tq[1]);
                            tq[2] = __builtin_fma(wx, rc.z, tq[2]); tq[3] = __builtin_fma(wy, rc.y, tq[3]);
                            tq[4] = __builtin_fma(wy, rc.z, tq[4]); tq[5] = __builtin_fma(wz, rc.z, tq[5]);
                        }
                    } else {
                        // calc_spatial_bristle_force (friction.jl:171-201) + traction(::Bristle) (:32-48), as in k_fric
                        const V3 x = r - cop;
                        V3 Ts = ts_c0 + cross_fma(ts_e, r);
                        Ts = axpy_fma(-dot_fma(Ts, nh), nh, Ts);      // vec_sub_vec_proj
                        const double m2 = dot_fma(Ts, Ts);
                        V3 T;
                        if (m2 < mu_s * mu_s) {
                            T = Ts;
                        } else {
                            double ri = __builtin_amdgcn_rsq(m2);
                            const double hm = 0.5 * m2;
                            ri = ri * __builtin_fma(-hm * ri, ri, 1.5);
                            ri = ri * __builtin_fma(-hm * ri, ri, 1.5);
                            const double mg = m2 * ri;
                            const double mu_slope = (mu_d - mu_s) / (3 * mu_s - 2 * mu_s);
                            const double y = mu_s + (mg - 2 * mu_s) * mu_slope;
                            const double mu = (y > mu_s) ? mu_s : ((y < mu_d) ? mu_d : y);
                            T = Ts * (mu * ri);
                        }
                        const V3 Tc = T * p_dA;
                        const V3 ta = cross_fma(x, Tc);
                        acc[0] += ta.x; acc[1] += ta.y; acc[2] += ta.z;
                        acc[3] += Tc.x; acc[4] += Tc.y; acc[5] += Tc.z;
                    }
                });
                if (n_pt == 0) continue;
                if (pass == 0 && !reg) {
                    const V3 Sr = mk3(tm[0] + tW * cen.x, tm[1] + tW * cen.y, tm[2] + tW * cen.z);   // sum w r
                    const V3 ta = cross(Sr, nh);
                    acc[0] += ta.x; acc[1] += ta.y; acc[2] += ta.z;
                    acc[3] += nh.x * tW; acc[4] += nh.y * tW; acc[5] += nh.z * tW;
                    acc[6] += tW; acc[7] += Sr.x; acc[8] += Sr.y; acc[9] += Sr.z;
                    // the triangle's moments about the workgroup's reference point c0: x' = x + d, d = cen - c0 (within the patch)
                    const V3 d = cen - c0;
                    const V3 m1 = mk3(tm[0] + tW * d.x, tm[1] + tW * d.y, tm[2] + tW * d.z);
                    double q[6];
                    q[0] = tq[0] + 2.0 * tm[0] * d.x + tW * d.x * d.x;
                    q[1] = tq[1] + tm[0] * d.y + tm[1] * d.x + tW * d.x * d.y;
                    q[2] = tq[2] + tm[0] * d.z + tm[2] * d.x + tW * d.x * d.z;
                    q[3] = tq[3] + 2.0 * tm[1] * d.y + tW * d.y * d.y;
                    q[4] = tq[4] + tm[1] * d.z + tm[2] * d.y + tW * d.y * d.z;
                    q[5] = tq[5] + 2.0 * tm[2] * d.z + tW * d.z * d.z;
                    // n̂ is constant over the polygon: sum w n n' = W n n', sum w (x x n) n' = (m1 x n) n',
                    // sum w (x x n)(x x n)' = [n]x Q [n]x'   (acc[10 ..]: Snn 6, San 9, Saa 6, Srr 6 -- the kAccSnn.. layout)
                    double *am = acc + 10;
                    am[0] += tW * nh.x * nh.x; am[1] += tW * nh.x * nh.y; am[2] += tW * nh.x * nh.z;
                    am[3] += tW * nh.y * nh.y; am[4] += tW * nh.y * nh.z; am[5] += tW * nh.z * nh.z;
                    const V3 an = cross(m1, nh);
                    am[6] += an.x * nh.x; am[7] += an.y * nh.x; am[8] += an.z * nh.x;
                    am[9] += an.x * nh.y; am[10] += an.y * nh.y; am[11] += an.z * nh.y;
                    am[12] += an.x * nh.z; am[13] += an.y * nh.z; am[14] += an.z * nh.z;
                    const V3 q0 = mk3(q[0], q[1], q[2]), q1 = mk3(q[1], q[3], q[4]), q2 = mk3(q[2], q[4], q[5]);
                    const V3 m0 = cross(nh, q0), m1c = cross(nh, q1), m2c = cross(nh, q2);       // M = [n]x Q
                    const V3 r0 = cross(nh, mk3(m0.x, m1c.x, m2c.x)), r1 = cross(nh, mk3(m0.y, m1c.y, m2c.y));
                    const V3 r2 = cross(nh, mk3(m0.z, m1c.z, m2c.z));
                    am[15] += r0.x; am[16] += r0.y; am[17] += r0.z; am[18] += r1.y; am[19] += r1.z; am[20] += r2.z;
#pragma unroll
                    for (int kk = 0; kk < 6; ++kk) am[21 + kk] += q[kk];
                }
            }
        }
        // ---- block reductions and the per-pass epilogue --------------------------------------------------------------
        if (pass == 0) FSTAMP(5);
        if (pass == 0) {
            block_totals<10>(acc, tot10, red, tid);
            // counts: non-empty polygons and traction points
            const int ne_w = wave_total_i(my_ne, lane), nt_w = wave_total_i(my_nt, lane);
            // (a team-mate's per-thread status bits -- a non-finite vertex met by one of ITS clip lanes -- reach rank 0 only through
            // the exchange below: they ride on this reduction)
            const bool nf_w = MW && __ballot((status & kStNonFinite) != 0) != 0ull;
            __syncthreads();
            if (lane == 0) { s_cnt[wave][0] = ne_w; s_cnt[wave][1] = nt_w; s_cnt[wave][2] = nf_w ? 1 : 0; }
            __syncthreads();
            n_nonempty = ((s_cnt[0][0] + s_cnt[1][0]) + s_cnt[2][0]) + s_cnt[3][0];
            n_trac = ((s_cnt[0][1] + s_cnt[1][1]) + s_cnt[2][1]) + s_cnt[3][1];
            if (MW && (((s_cnt[0][2] | s_cnt[1][2]) | s_cnt[2][2]) | s_cnt[3][2])) status |= kStNonFinite;
            // what this workgroup found, one value per thread: the ten sums [0, 10), the four counters, the status word, and for
            // a bristle item the 27 moments about c0 [15, 42) and c0 itself [42, 45)
            double mine = 0.0;
            if (tid < 10) mine = ((red[tid] + red[32 + tid]) + red[64 + tid]) + red[96 + tid];
            else if (tid == 10) mine = (double)n_test;
            else if (tid == 11) mine = (double)n_cand;
            else if (tid == 12) mine = (double)n_nonempty;
            else if (tid == 13) mine = (double)n_trac;
            else if (tid == 14) mine = (double)status;
            else if (tid == 42) mine = c0.x;
            else if (tid == 43) mine = c0.y;
            else if (tid == 44) mine = c0.z;
            if (!reg) {
                block_partials<27>(acc + 10, red, tid);
                if (tid >= 15 && tid < 42) mine = ((red[tid - 15] + red[32 + tid - 15]) + red[64 + tid - 15]) + red[96 + tid - 15];
            }
            const int n_pub = reg ? 15 : 45;
            if (MW) {
                team_gather(g, item, nw, 0, n_pub, mine, tid, s_team, &s_tflag, status);      // every rank's values -> s_team
            } else {
                __syncthreads();
                if (tid < n_pub) s_team[tid] = mine;
                __syncthreads();
            }
            {
                const double tot = team_total(s_team, nw, 15, 14, tid);
                if (tid < 15) s_tres[tid] = tot;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 10; ++k) tot10[k] = s_tres[k];
            n_test = (int)s_tres[10]; n_cand = (int)s_tres[11]; n_nonempty = (int)s_tres[12]; n_trac = (int)s_tres[13];
            status |= (unsigned)s_tres[14];
            if (MW) FSTAMP(12);
            contact = n_trac > 0;
            if (!reg && contact) {
                const double iS = tot10[6];
                cop = mk3(tot10[7] / iS, tot10[8] / iS, tot10[9] / iS);      // normal.jl:33
                // every rank's moments move from its reference point to the cop (one thread per rank, in place), then add up
                // in rank order: the item's accumulator block in the kAcc* layout (read by eig_item)
                if (tid < nw) {
                    double *row = s_team + tid * kTeamSlots;
                    const double Wr = row[6];
                    double o27[27];
                    if (Wr > 0.0) {
                        const double cr[3] = {row[42], row[43], row[44]};
                        const double m1[3] = {row[7] - Wr * cr[0], row[8] - Wr * cr[1], row[9] - Wr * cr[2]};     // sum w (r - c0)
                        const double dd[3] = {cr[0] - cop.x, cr[1] - cop.y, cr[2] - cop.z};
                        shift_moments(row + 15, Wr, m1, dd, o27);
                    } else {
#pragma unroll
                        for (int k = 0; k < 27; ++k) o27[k] = 0.0;
                    }
#pragma unroll
                    for (int k = 0; k < 27; ++k) row[15 + k] = o27[k];
                }
                __syncthreads();
                if (tid < 27) {
                    double a27 = 0.0;
                    for (int w2 = 0; w2 < nw; ++w2) a27 += s_team[w2 * kTeamSlots + 15 + tid];
                    s_acc[kAccSnn + tid] = a27;
                }
                if (tid == 32) {
                    s_acc[kAccIp] = tot10[6];
                    s_acc[kAccIpc] = tot10[7]; s_acc[kAccIpc + 1] = tot10[8]; s_acc[kAccIpc + 2] = tot10[9];
                }
                __syncthreads();
                if (MW) FSTAMP(13);
                if (wave == 0) eig_item(s_acc, I.ins.k_bar, I.ins.magic, I.s, s_res, E, lane);
                __syncthreads();
                if (MW) FSTAMP(14);
                // per-item constants of the friction pass (k_fric): T̄s = c0 + e x r
                const V3 copr = ld3(s_res + kResCop), Da = ld3(s_res + kResDelta), Dl = ld3(s_res + kResDelta + 3);
                cop = copr;
                ts_c0 = ((Dl - cross(Da, copr)) + vl * I.ins.tau) * (-I.ins.k_bar);
                ts_e = (Da + w * I.ins.tau) * (-I.ins.k_bar);
            }
            __syncthreads();        // s_team is free again
        } else {
            block_partials<6>(acc, red, tid);
            if (MW) {
                // A rank whose wait after pass 0 timed out went on with stale totals and a status word only IT holds; rank 0, the
                // writer of the outputs, must learn of it here (before: it summed that rank's friction partials and reported
                // success).  Such a rank -- and one that met a non-finite vertex or an aborted descent after the first exchange --
                // publishes NaN instead of its sums: every rank's total is then NaN and the item reports kStFusedOvf (the host
                // re-issues on the batched path, which reports what there is to report).  No extra slot, no extra pass: a seventh
                // value OR-ed over the ranks cost the single pose 2.7 us.
                const bool bad = (status & (kStFusedOvf | kStNonFinite | kStAbort)) != 0;
                double mine = tid < 6 ? ((red[tid] + red[32 + tid]) + red[64 + tid]) + red[96 + tid] : 0.0;
                if (bad) mine = __builtin_nan("");
                const double tot = team_sum(g, item, nw, 2, 6, mine, -1, tid, s_team, &s_tflag, status);
                if (tid < 6) s_acc[kAccFric + tid] = tot;
                if (tid == 0) s_tres[6] = (tot != tot) ? 1.0 : 0.0;
                FSTAMP(15);
            } else if (tid < 6) s_acc[kAccFric + tid] = ((red[tid] + red[32 + tid]) + red[64 + tid]) + red[96 + tid];
            __syncthreads();
            if (MW && s_tres[6] != 0.0) status |= kStFusedOvf;
        }
    }

    // ==== 4a. hand-over to the batched Dual passes ======================================================================
    if (!MW && g.emit_items) {
        __shared__ int s_cbase;
        __syncthreads();
        const int n_pl = s_npoly;
        if (tid == 0) {
            ItemRec &r = s_it;
#pragma unroll
            for (int k = 0; k < 9; ++k) { r.R21[k] = I.pose[k]; r.R12[k] = (g.bp_pose ? s_bp : I.pose + 12)[k]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) { r.t21[k] = I.pose[9 + k]; r.t12[k] = (g.bp_pose ? s_bp : I.pose + 12)[9 + k]; r.w[k] = I.twist[k]; r.v[k] = I.twist[3 + k]; }
#pragma unroll
            for (int k = 0; k < 6; ++k) r.s[k] = (I.ins.model == PFC_BRISTLE) ? I.s[k] : 0.0;
            r.chi = I.ins.chi; r.Ebar = I.ins.Ebar; r.mu_s = I.ins.mu_s; r.mu_d = I.ins.mu_d; r.v_c = I.ins.v_c; r.tau = I.ins.tau;
            r.k_bar = I.ins.k_bar; r.magic = I.ins.magic; r.nodes1 = I.ins.nodes1; r.nodes2 = I.ins.nodes2; r.nf1 = I.ins.nf1;
            r.nf2 = I.ins.nf2; r.tri = I.ins.tri; r.tet = I.ins.tet; r.tet1 = I.ins.tet1; r.eps1 = I.ins.eps1; r.eps2 = I.ins.eps2;
            r.Ebar1 = I.ins.Ebar1; r.model = I.ins.model; r.nq = nq; r.ins = id; r.pad = 0;
            r.q12[0] = 1.0f; r.q12[1] = r.q12[2] = r.q12[3] = 0.0f; r.pose_exact = 1; r.bp_eabs = 0.0f;      // (the broadphase does not read handed-over records)
            r.pad2[0] = r.pad2[1] = 0;
            s_cbase = n_pl ? atomicAdd(g.emit_ctr, n_pl) : 0;
        }
        __syncthreads();
        constexpr int nw = (int)(sizeof(ItemRec) / sizeof(int));
        for (int k = tid; k < nw; k += kFuBlock)
            reinterpret_cast<int *>(g.emit_items + item)[k] = reinterpret_cast<const int *>(&s_it)[k];
        const int cbase = s_cbase;
        if (cbase + n_pl > g.emit_cap) {
            status |= kStCandOvf;          // uniform: the host re-issues on the batched path
        } else {
            for (int j = tid; j < n_pl; j += kFuBlock) {
                const int2 cw = cand[s_plist[j]];
                WorkRec c;
                c.item = item; c.a = cw.x; c.b = cw.y; c.pad = 0;
                g.emit_cand[cbase + j] = c;
                g.emit_surv[cbase + j] = cbase + j;
            }
        }
        if (tid == 0) {
            int *ic = g.emit_icnt + 4 * (size_t)item;
            ic[0] = n_test; ic[1] = n_cand; ic[2] = n_nonempty; ic[3] = n_trac;
        }
    }
    // ==== 4b. the evaluation on Dual numbers (regularized items; values above, one partial per direction here) =========
    // What calcXd! does on MechanismScenario.dual (src/mechanism_scenario.jl:187) for Radau's Jacobian, for the polygons
    // found above: lane = (polygon, direction) clips in (value, partial) arithmetic -- every branch on values, the value
    // half of every operation the instruction sequence of the value pass, as in k_narrow_dual -- then the fan triangles are
    // dealt out one per thread and the partial sums of each direction meet in LDS.
    if (!MW && g.n_dir > 0) {
        const int n_dir = g.n_dir;
        __syncthreads();
        const int n_pd = s_npoly * n_dir;
        if (tid < 16 * 6) (&s_dacc[0][0])[tid] = 0.0;
        if (!reg || n_pd > kFuDualMax || (TT && I.ins.tet1 != nullptr)) {
            if (contact) status |= kStFusedDualSkip;      // uniform: the host takes the batched Dual path
        } else if (contact) {
            if (tid == 0) {
                s_it.w[0] = I.twist[0]; s_it.w[1] = I.twist[1]; s_it.w[2] = I.twist[2];
                s_it.v[0] = I.twist[3]; s_it.v[1] = I.twist[4]; s_it.v[2] = I.twist[5];
                s_it.chi = I.ins.chi; s_it.Ebar = I.ins.Ebar; s_it.mu_s = I.ins.mu_s; s_it.mu_d = I.ins.mu_d; s_it.v_c = I.ins.v_c;
                s_it.tau = I.ins.tau; s_it.k_bar = I.ins.k_bar; s_it.magic = I.ins.magic; s_it.nq = nq; s_it.model = I.ins.model;
            }
            double *dv = S.ring, *dd = S.ring + 8 * 4 * kFuDualLanes;     // value ring, partial ring: [slot][coord][lane]
#define DV(l, rb, k, c) dv[(((((rb) + (k)) & 7) * 4 + (c)) * kFuDualLanes) + (l)]
#define DD(l, rb, k, c) dd[(((((rb) + (k)) & 7) * 4 + (c)) * kFuDualLanes) + (l)]
#define DXV(l, s) dv[((((s) * 4) + 3) * kFuDualLanes) + (l)]
#define DXD(l, s) dd[((((s) * 4) + 3) * kFuDualLanes) + (l)]
            for (int base = 0; base < n_pd; base += kFuDualLanes) {
                __syncthreads();           // rings and fan list of the previous round are free
                int n_poly = 0, rbase = 0;
                const int wq = base + tid;
                const bool lane_on = tid < kFuDualLanes && wq < n_pd;
                int dir = 0;
                if (lane_on) {
                    const int q = wq / n_dir;
                    dir = wq - q * n_dir;
                    const int2 cw = cand[s_plist[q]];
                    const GTetRec *tp = (const GTetRec *)(I.ins.tet + cw.y);
                    const GTriRec *tr = (const GTriRec *)(I.ins.tri + cw.x);
                    const double *dp = g.d_pose + ((size_t)item * n_dir + dir) * 24;
                    Du R21[9], t21[3];
#pragma unroll
                    for (int k = 0; k < 9; ++k) R21[k] = du(I.pose[k], dp[k]);
#pragma unroll
                    for (int k = 0; k < 3; ++k) t21[k] = du(I.pose[9 + k], dp[9 + k]);
                    double Z[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) Z[k] = tp->xzr[k];
                    // tri-tet op (non_friction.jl:196-215) on Duals, as k_narrow_dual
                    Du X[16], z[3][4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            X[i + 4 * j] = (Z[i] * R21[3 * j] + Z[i + 4] * R21[3 * j + 1]) + Z[i + 8] * R21[3 * j + 2];
                        X[i + 12] = ((Z[i] * t21[0] + Z[i + 4] * t21[1]) + Z[i + 8] * t21[2]) + du(Z[i + 12]);
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            z[k][i] = ((X[i] * tr->v[3 * k] + X[i + 4] * tr->v[3 * k + 1]) + X[i + 8] * tr->v[3 * k + 2]) + X[i + 12];
                    const Du3 nh_in = dmk((R21[0] * tr->n[0] + R21[3] * tr->n[1]) + R21[6] * tr->n[2],
                                          (R21[1] * tr->n[0] + R21[4] * tr->n[1]) + R21[7] * tr->n[2],
                                          (R21[2] * tr->n[0] + R21[5] * tr->n[1]) + R21[8] * tr->n[2]);
                    // clip_in_tet_coordinates (static_clip.jl:7-23,34-201): the value pass kept this candidate, so it is finite
                    // and not trivially rejected
                    int n = 3;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int i = 0; i < 4; ++i) { DV(tid, rbase, k, i) = z[k][i].v; DD(tid, rbase, k, i) = z[k][i].d; }
                    {
                        bool err = false;      // "Non-finite vertex likely": the value pass has reported it
                        RingDu<kFuDualLanes> rg{dv, dd, kFuDualLanes, tid, kFuDualLanes, tid, rbase};
                        n = clip_ring_in_tet_coordinates(rg, 3, err);     // pfc_clip.h
                        rbase = rg.rbase;
                    }
                    n_poly = n >= 3 ? n : 0;
                    if (n_poly) {
                        // poly_r2 = mul_then_un_pad(x_r2_ζ2, poly_ζ2), centroid(poly_r2, n̂2) on Duals
                        double V[12];
#pragma unroll
                        for (int k = 0; k < 12; ++k) V[k] = tp->xrz[k];
                        for (int k = 0; k < n; ++k) {
                            const Du z0 = du(DV(tid, rbase, k, 0), DD(tid, rbase, k, 0)), z1 = du(DV(tid, rbase, k, 1), DD(tid, rbase, k, 1)),
                                     z2 = du(DV(tid, rbase, k, 2), DD(tid, rbase, k, 2)), z3 = du(DV(tid, rbase, k, 3), DD(tid, rbase, k, 3));
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                const Du rr = ((V[c] * z0 + V[c + 3] * z1) + V[c + 6] * z2) + V[c + 9] * z3;
                                DV(tid, rbase, k, c) = rr.v; DD(tid, rbase, k, c) = rr.d;
                            }
                        }
#define DVT(k) dmk(du(DV(tid, rbase, k, 0), DD(tid, rbase, k, 0)), du(DV(tid, rbase, k, 1), DD(tid, rbase, k, 1)), du(DV(tid, rbase, k, 2), DD(tid, rbase, k, 2)))
                        const Du3 a = DVT(0);
                        Du3 cc = DVT(1);
                        Du cum_sum = du(0.0);
                        Du3 cum_prod = dmk(du(0.0), du(0.0), du(0.0));
                        for (int k = 2; k < n; ++k) {
                            const Du3 b = cc;
                            cc = DVT(k);
                            const Du ar = dtriangle_area(a, b, cc, nh_in);
                            cum_prod = cum_prod + (((a + b) + cc) * (1.0 / 3.0)) * ar;
                            cum_sum += ar;
                        }
#undef DVT
                        const Du3 cen = (cum_sum.v == 0.0) ? a : cum_prod / cum_sum;
                        DXV(tid, 0) = nh_in.x.v; DXV(tid, 1) = nh_in.y.v; DXV(tid, 2) = nh_in.z.v;
                        DXD(tid, 0) = nh_in.x.d; DXD(tid, 1) = nh_in.y.d; DXD(tid, 2) = nh_in.z.d;
                        DXV(tid, 3) = cen.x.v; DXV(tid, 4) = cen.y.v; DXV(tid, 5) = cen.z.v;
                        DXD(tid, 3) = cen.x.d; DXD(tid, 4) = cen.y.d; DXD(tid, 5) = cen.z.d;
#pragma unroll
                        for (int k = 0; k < 4; ++k) s_epsr[k][tid] = tp->epsr[k];
                    }
                }
                s_np[tid] = n_poly | (dir << 8);
                s_rb[tid] = rbase;
                // deal the fan triangles of the round's Dual polygons out, one per thread
                const int incl = seg_incl_scan(n_poly);
                if (lane == 63) s_scan[wave] = incl;
                __syncthreads();
                int off = incl - n_poly;
                for (int wv = 0; wv < kFuWaves; ++wv) if (wv < wave) off += s_scan[wv];
                for (int k = 0; k < n_poly; ++k) fan[off + k] = (unsigned)tid | ((unsigned)k << 16);
                __syncthreads();
                const int n_fan = ((s_scan[0] + s_scan[1]) + s_scan[2]) + s_scan[3];
                for (int wi = tid; wi < n_fan; wi += kFuBlock) {
                    const unsigned fe = fan[wi];
                    const int pt = (int)(fe & 0xFFFFu), k = (int)(fe >> 16);
                    const int n = s_np[pt] & 0xFF, d = s_np[pt] >> 8, rb = s_rb[pt];
                    const Du3 nh = dmk(du(DXV(pt, 0), DXD(pt, 0)), du(DXV(pt, 1), DXD(pt, 1)), du(DXV(pt, 2), DXD(pt, 2)));
                    const Du3 cen = dmk(du(DXV(pt, 3), DXD(pt, 3)), du(DXV(pt, 4), DXD(pt, 4)), du(DXV(pt, 5), DXD(pt, 5)));
                    const double er[4] = {s_epsr[0][pt], s_epsr[1][pt], s_epsr[2][pt], s_epsr[3][pt]};
                    Du sum[10];
#pragma unroll
                    for (int kk = 0; kk < 10; ++kk) sum[kk] = du(0.0);
                    int nt = 0;
                    const Du3 zero3 = dmk(du(0.0), du(0.0), du(0.0));
                    dual_integrate<0>(
                        [&](int kk) {
                            return dmk(du(DV(pt, rb, kk, 0), DD(pt, rb, kk, 0)), du(DV(pt, rb, kk, 1), DD(pt, rb, kk, 1)),
                                       du(DV(pt, rb, kk, 2), DD(pt, rb, kk, 2)));
                        },
                        n, nh, cen, er, &s_it, g.d_twist + ((size_t)item * n_dir + d) * 6, true, zero3, zero3, zero3, sum, nt, k, k + 1);
                    if (nt > 0) {
#pragma unroll
                        for (int kk = 0; kk < 6; ++kk) unsafeAtomicAdd(&s_dacc[d][kk], sum[kk].d);
                    }
                }
            }
#undef DV
#undef DD
#undef DXV
#undef DXD
        }
        __syncthreads();
        // partials of the outputs: regularized items have no bristle state (sdot = 0)
        if (tid < n_dir * 6) {
            const int d = tid / 6, k = tid - d * 6;
            g.d_wrench[((size_t)item * n_dir + d) * 6 + k] = s_dacc[d][k];
            g.d_sdot[((size_t)item * n_dir + d) * 6 + k] = 0.0;
        }
    }
    FSTAMP(6);
    // ==== 5. yes_contact! / no_contact! epilogue (friction.jl:76-81,119-143; non_friction.jl:77-83) ====================
    // status: every thread may have set bits
    {
        const unsigned long long any_nf = __ballot((status & kStNonFinite) != 0);
        if (any_nf) status |= kStNonFinite;
        __syncthreads();
        if (lane == 0) s_cnt[wave][0] = (int)status;
        __syncthreads();
        status = (unsigned)(((s_cnt[0][0] | s_cnt[1][0]) | s_cnt[2][0]) | s_cnt[3][0]);
    }
    if (tid == 0 && wr == 0) {     // one lane, static indices only (a register array indexed by the thread id would live in scratch)
        double wv[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, sd[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (reg) {
#pragma unroll
            for (int k = 0; k < 6; ++k) wv[k] = contact ? tot10[k] : 0.0;
        } else {
            const double tau_inv = 1.0 / I.ins.tau;
            if (!contact) {
#pragma unroll
                for (int k = 0; k < 6; ++k) sd[k] = -tau_inv * I.s[k];
            } else {
                const V3 fang = ld3(s_acc + kAccFric), flin = ld3(s_acc + kAccFric + 3);
                const V3 fang2 = fang + cross(cop, flin);
                wv[0] = tot10[0] + fang2.x; wv[1] = tot10[1] + fang2.y; wv[2] = tot10[2] + fang2.z;
                wv[3] = tot10[3] + flin.x; wv[4] = tot10[4] + flin.y; wv[5] = tot10[5] + flin.z;
                double sw[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) sw[k] = s_res[kResSinv + k] * s_acc[kAccFric + k];
#pragma unroll
                for (int ii = 0; ii < 6; ++ii) {
                    double a6 = 0.0;
#pragma unroll
                    for (int k = 0; k < 6; ++k) a6 += s_res[kResKis + ii + 6 * k] * sw[k];
                    sd[ii] = -tau_inv * (a6 + I.s[ii]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) { g.wrench[6 * (size_t)item + k] = wv[k]; g.sdot[6 * (size_t)item + k] = sd[k]; }
    }
    if (tid == 0 && wr == 0) {      // the same lane stored wrench and sdot above
        int *fo = g.fout + 8 * (size_t)item;
        fo[0] = (int)status; fo[1] = n_test; fo[2] = n_cand; fo[3] = n_nonempty; fo[4] = n_trac;
        fo[6] = fo[7] = 0;
        if (g.counts) {
            int *co = g.counts + 4 * (size_t)item;
            co[0] = n_test; co[1] = n_cand; co[2] = n_nonempty; co[3] = n_trac;
        }
        // completion word: everything this item writes is visible to the host before it (system-scope release)
        __threadfence_system();
        __hip_atomic_store(&fo[5], g.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    FSTAMP(9);
#ifdef PFC_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0 && g.stamps) g.stamps[11] = __builtin_amdgcn_s_memtime();
#endif
}
#undef FR
#undef FX
