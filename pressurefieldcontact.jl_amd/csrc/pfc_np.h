// pfc_np.h -- narrowphase: gather, clip, quadrature and per-item reductions (k_narrow), bristle friction over the kept polygons (k_fric).  Included by pfc_hip.hip inside namespace pfc (device code only).
#pragma once

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
    int *rgn;        // region counters of the record list (kRgn x kRgnStride ints, word 1)
    int rr_cap;      // record slots per region
    int *pcnt;       // kept polygons per candidate chunk (written for every chunk of the evaluation)
    int chunk_switch;  // number of chunks a batch must at least fall into (np_chunk picks the chunk size from it)
    int *poly_cand;  // clip-only mode with the Dual list on: candidate index of every kept polygon, or null
    int *icnt;
    int n_items;
    int *clip_n;     // per candidate, or null
    // clipped polygons of bristle items, kept for the friction pass (k_fric): SoA [field][slot], slot < pcap
    int *poly_item;  // item | n_poly << 28
    double *poly;    // 34 fields: n̂ 3, centroid 3, ϵ_r² 4, vertices 8 x 3 (frame r²)
    int pcap;        // the SoA stride of the polygon fields (>= ccap rounded up to a chunk)
    int *surv;       // candidate indices of the pairs that contributed traction points (work list of the Dual passes)
    int *scount;
    TracSoA trac;
    int *tcount;
    int tcap;
    unsigned *status;
    int debug;       // materialise traction points for every item
    unsigned long long *stamps;  // diagnostic builds: [0..5] cycles in gather / clip / reserve / integrate / reduce, rounds
};

// Fused-multiply-add forms of the vector helpers for quantities that are compared with the oracle by
// tolerance (friction force, patch moments) (the library is built with -ffp-contract=off because the traction points and every predicate upstream are
// bit-exact restatements; here an a*b + c*d costs two instructions instead of three).
__device__ __forceinline__ V3 cross_fma(V3 a, V3 b) {
    return V3{__builtin_fma(a.y, b.z, -(a.z * b.y)), __builtin_fma(a.z, b.x, -(a.x * b.z)), __builtin_fma(a.x, b.y, -(a.y * b.x))};
}
__device__ __forceinline__ double dot_fma(V3 a, V3 b) { return __builtin_fma(a.z, b.z, __builtin_fma(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ V3 axpy_fma(double a, V3 x, V3 y) {
    return V3{__builtin_fma(a, x.x, y.x), __builtin_fma(a, x.y, y.y), __builtin_fma(a, x.z, y.z)};
}

// Elimination builds (diagnostic, scripts/elimination.sh): -DPFC_EXP=n compiles one phase of k_narrow out behind a
// condition the compiler cannot fold, so that timing the variants against each other gives the phase costs including
// their overlap (results are wrong in these builds).  3: every candidate rejected after the gather; 4: polygons dropped
// after the clip; 9: slots reserved, no polygon set-up; 7: no quadrature points; 5: nothing after the integration.
#ifndef PFC_EXP
#define PFC_EXP 0
#endif
constexpr int kElim = PFC_EXP;  // 0 in the product: every `kElim == n` below folds away
constexpr int kNpBlock = 64;  // one wave per block: 16 KiB of LDS polygon staging per wave
// Candidates are dealt out in CHUNKS of consecutive list entries: one workgroup walks a chunk round by round and keeps
// the chunk's polygons in the chunk's own slot range [ch C, ch C + C) behind a counter it holds in a register -- no
// reservation atomic, and since the candidate list is grouped by item a chunk's polygons belong to one item (two at a run
// boundary), which is what lets the passes over the kept polygons (k_integ, k_fric) sum per item in registers.  Big
// batches use 512-candidate chunks; a small scene's chunk is one wave round (np_chunk: latency before density).
constexpr int kNpChunkBig = 512;
// the largest of 512 / 256 / 128 / 64 candidates that still leaves chunk_switch chunks (workgroups with work)
__device__ __forceinline__ int np_chunk(int n_c, int chunk_switch) {
    int C = kNpChunkBig;
    while (C > kNpBlock && (long long)C * chunk_switch > n_c) C >>= 1;
    return C;
}

// weightPoly (src/math_kernel/utility.jl:21-26) on 4-vectors held in LDS slots
// polygon ring in LDS: 8 physical slots x 4 coords per lane, [slot][coord][lane] layout (conflict-free per-lane
// dynamic indexing); logical vertex k of a lane lives in physical slot (rbase + k) & 7.  RC columns, a lane uses column
// rcol: its own lane number (RC = 64, 16 KiB: MODE 0, 1, 2), or its rank among the round's survivors of the trivial
// reject (MODE 3, the clip-only kernel of a half of a two-half evaluation: RC = kRingColsClip = 48, 12 KiB -- ~35 of 64
// candidates survive in the C3 batch, a round with more than 48 survivors takes a second pass for the rest).  The smaller
// footprint lets the kernel share CUs with the other half's broadphase (8 192-pose step 4.22 -> 4.14 ms); a launch that has
// the chip to itself is faster with a column per lane (2.01 vs 2.13 ms for the unsplit batch), hence both forms.
constexpr int kRingColsClip = 48;
#define PR(k, c) poly[((((rbase) + (k)) & 7) * 4 + (c)) * RC + rcol]

__device__ __forceinline__ double readlane_f64(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
                            __builtin_amdgcn_readlane(__double2loint(v), src));
}

// wave total of a double on DPP (row_shr 1/2/4/8, row_bcast 15/31), returned as a uniform value
__device__ __forceinline__ double wave_total(double v, int lane) {
    double t;
    t = dpp_move<0x111, 0xF>(v); v += t;
    t = dpp_move<0x112, 0xF>(v); v += t;
    t = dpp_move<0x114, 0xF>(v); v += t;
    t = dpp_move<0x118, 0xF>(v); v += t;
    t = dpp_move<0x142, 0xA>(v); v += (lane & 16) ? t : 0.0;
    t = dpp_move<0x143, 0xC>(v); v += (lane >= 32) ? t : 0.0;
    return readlane_f64(v, 63);
}
__device__ __forceinline__ int wave_total_i(int v, int lane) {
    return __builtin_amdgcn_readlane(seg_incl_scan(v), 63);
}

// Per-item accumulation of N per-lane partial sums.  Segmented scan per value, then the N totals of each run are
// transposed onto lanes 0..N-1 (readlane from the run's tail) and leave as ONE wave-wide FP64 atomic instruction
// on N consecutive accumulator slots: single-lane atomics are issue-bound (one wave instruction per ~50 ns per CU,
// MI355X guide 'Global float atomics'), a 37-lane one costs the same as a 1-lane one.
// FX (option "fixed_order", the Dual passes): a run's totals do not join the accumulator by an atomic; they leave as a RECORD
// -- key, the wave's position in the pass, the key's previous record, N values -- and k_fixed_reduce adds a key's records in the
// order of their positions (pfc_dual.h).
constexpr int kSinkHdr = 4, kSinkStride = 48;      // [0] key [1] position [2] previous record of the key [3] -; then up to 44 values
// Where a record goes: the first per_pos runs of the wave at position p of the pass (its 64-polygon piece / its group of pairs) have
// slots of their own, direct_base + p per_pos + k -- no counter; further runs (a wave that straddles several items) take slots
// behind ovf_base from ONE returning atomic per wave.  (A counter for every record took an atomic from each of the 125 000 pieces
// of a bench step on one address: 1 ms of k_fric_fixed.)
struct FixedSink {
    double *rec;
    int *count;          // overflow records handed out (beyond the capacity: dropped)
    int *head;           // per key: its last record, -1: none
    int direct_base, per_pos, n_pos, ovf_base, cap;
    unsigned *status;
};
template <int N, bool FX = false>
__device__ __forceinline__ void accumulate_items(double *acc, int item, bool listed, bool any, const double *v, int n0,
                                                 int stride = kAccStride, const FixedSink *fx = nullptr, int order = 0) {
    // listed: the lane holds a work-list entry (its item keys the run even if it contributes nothing, so empty
    // polygons do not chop an item's run into pieces); any: the lane has a contribution
    static_assert(N <= 64, "one value per lane");
    if (__ballot(any) == 0) return;
    if constexpr (FX) {
        // A lane without a list entry (a slot marker, a polygon that is not for this pass) takes the key of the nearest listed lane
        // below it, so that it does not chop that key's run in two: one record per (wave, key), and the bound of the record list --
        // distinct keys per wave summed over the waves -- holds.  (Such a lane contributes nothing: any is false for it.)
        const unsigned long long lm = __ballot(listed);
        const int ln = lane_id();
        const unsigned long long below = lm & (ln == 63 ? ~0ull : ((2ull << ln) - 1ull));
        const int src = below ? 63 - __builtin_clzll(below) : ln;
        const int filled = __shfl(item, src, 64);
        item = below ? filled : -1;
        listed = below != 0;
    }
    const Seg sg = seg_setup(listed ? item : -1);
    double tot[N];
#pragma unroll
    for (int k = 0; k < N; ++k) tot[k] = seg_sum(any ? v[k] : 0.0, sg);
    unsigned long long tails = __ballot(sg.tail && sg.valid);
    const int lane = lane_id();
    int fx_k = 0, fx_ovf = 0;
    if constexpr (FX) {
        const int n_ovf = __popcll(tails) - fx->per_pos;
        if (n_ovf > 0) {
            if (lane == 0) fx_ovf = atomicAdd(fx->count, n_ovf);
            fx_ovf = __builtin_amdgcn_readfirstlane(fx_ovf);
        }
    }
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
        if constexpr (FX) {
            static_assert(N <= kSinkStride - kSinkHdr, "a record holds the run's values");
            // (a run whose totals are all zero -- lanes that hold a list entry and contribute nothing -- adds nothing, as with the
            // atomics: its slot stays unlinked, k_fixed_reduce never sees it)
            const int k = fx_k++;
            int slot = k < fx->per_pos ? (order < fx->n_pos ? fx->direct_base + order * fx->per_pos + k : fx->cap)
                                       : fx->ovf_base + fx_ovf + (k - fx->per_pos);
            if (__ballot(lane < N && mine != 0.0) == 0) continue;
            if (slot < fx->cap) {
                double *r = fx->rec + (size_t)slot * kSinkStride;
                if (lane < N) r[kSinkHdr + lane] = mine;
                if (lane == 0) { r[0] = (double)item_t; r[1] = (double)order; r[2] = (double)atomicExch(&fx->head[item_t], slot); }
            }
            // (no room: the pass runs on more pairs than the capacity it was given was sized for -- the caller finds that out from
            // the pair count, as for the kept Dual polygons, and re-issues the passes)
        } else {
        if (lane < N && mine != 0.0) unsafeAtomicAdd(&acc[(size_t)item_t * stride + n0 + lane], mine);
        }
    }
}

// Sums of N per-lane values over a whole wave through LDS: lane `lane` writes column `lane` of N rows, then L
// lanes per row add up 64/L entries each (interleaved columns) and the L partials meet in a butterfly; the row totals are
// returned on lanes lane0 .. lane0+N-1.  N + ~130 instructions per wave
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
template <int N, int L>
__device__ __forceinline__ double lds_row_sums(double *buf, const double *v, bool any, int lane, int lane0) {
    constexpr int RS = 64 + L;   // row stride: the 32 lanes of a half-wave read 32 different banks
    static_assert(N * L <= 64 && N * RS <= 2048 && (L == 1 || L == 2 || L == 4), "rows x lanes per row must fit the wave");
#pragma unroll
    for (int k = 0; k < N; ++k) buf[k * RS + lane] = any ? v[k] : 0.0;
    wave_lds_sync();
    double t = 0.0;
    const int row = lane / L, part = lane % L;
    if (row < N) {
        const double *r = buf + row * RS + part;
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
        for (int j = 0; j < 64 / L; j += 8) {
            const double a0 = r[L * j], a1 = r[L * (j + 1)], a2 = r[L * (j + 2)], a3 = r[L * (j + 3)];
            const double a4 = r[L * (j + 4)], a5 = r[L * (j + 5)], a6 = r[L * (j + 6)], a7 = r[L * (j + 7)];
            t0 += a0; t1 += a1; t2 += a2; t3 += a3;
            t0 += a4; t1 += a5; t2 += a6; t3 += a7;
        }
        t = (t0 + t1) + (t2 + t3);
    }
    wave_lds_sync();
#pragma unroll
    for (int m = 1; m < L; m <<= 1) t += __shfl_xor(t, m, 64);
    // row r's total sits on lanes r L .. r L + L - 1; hand it to lane lane0 + r
    const int dst_row = lane - lane0;
    const double out = __shfl(t, (dst_row >= 0 && dst_row < N) ? dst_row * L : 0, 64);
    return (dst_row >= 0 && dst_row < N) ? out : 0.0;
}

// The front of one op (tri-tet: non_friction.jl:196-215 up to the clip; tet-tet: :166-190): gather the two mesh records,
// express the input polygon (3 or 4 vertices) in the coordinates of tet 2 -> z, its normal -> nh_in, and apply the bit-exact
// trivial reject.  Returns true if the candidate has to be clipped.  One statement for k_narrow (every mode) and
// k_clip_queue; `report`: raise kStNonFinite for a non-finite vertex ("Non-finite vertex likely", static_clip.jl:52).
// the arithmetic of the tri-tet front on loaded values (R21, t21: x_r2_r1; Z: x_ζ2_r2; TV, TN: the triangle's vertices and normal
// in frame r1): one statement for np_front and for k_clip_queue, which loads a round's records one round ahead
__device__ __forceinline__ void np_front_tri_values(const double (&R21)[9], const double (&t21)[3], const double (&Z)[16],
                                                    const double (&TV)[9], const double (&TN)[3], double (&z)[4][4], V3 &nh_in) {
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
            z[k][i] = ((X[i] * TV[3 * k] + X[i + 4] * TV[3 * k + 1]) + X[i + 8] * TV[3 * k + 2]) + X[i + 12];
#pragma unroll
    for (int i = 0; i < 4; ++i) z[3][i] = 0.0;
    // n̂2 = R(x_r2_r1) * n̂_r1 (:211-212)
    nh_in = mk3((R21[0] * TN[0] + R21[3] * TN[1]) + R21[6] * TN[2],
                (R21[1] * TN[0] + R21[4] * TN[1]) + R21[7] * TN[2],
                (R21[2] * TN[0] + R21[5] * TN[1]) + R21[8] * TN[2]);
}
// finite check + the bit-exact trivial reject of a front's tet coordinates (see np_front)
__device__ __forceinline__ bool np_front_accept(const double (&z)[4][4], int n_in, unsigned *status, bool report) {
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) finite &= (k >= n_in) || (__builtin_fabs(z[k][i]) <= 1.79769313486231570815e308);
    if (!finite) if (report) atomicOr(status, kStNonFinite);
    bool reject = !finite || n_in < 3;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        reject |= (z[0][i] <= 0.0) && (z[1][i] <= 0.0) && (z[2][i] <= 0.0) && (n_in < 4 || z[3][i] <= 0.0);
    if (kElim == 3) reject |= z[0][0] > -1e300;
    return !reject;
}

template <bool TT>
__device__ __forceinline__ bool np_front(const ItemRec *it, const WorkRec &cw, const GTetRec *tp, double (&z)[4][4], int &n_in,
                                         V3 &nh_in, unsigned *status, bool report) {
    double R21[9], t21[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R21[k] = it->R21[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) t21[k] = it->t21[k];
    double Z[16];        // x_ζ2_r2
#pragma unroll
    for (int k = 0; k < 16; ++k) Z[k] = tp->xzr[k];
    if (!TT || it->tet1 == nullptr) {
        // ---- tri-tet op (non_friction.jl:196-215) -----------------------------------------------------------
        const GTriRec *tr = (const GTriRec *)(it->tri + cw.a);
        double TV[9], TN[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) TV[k] = tr->v[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) TN[k] = tr->n[k];
        np_front_tri_values(R21, t21, Z, TV, TN, z, nh_in);
        n_in = 3;
    } else {
        // ---- tet-tet op (non_friction.jl:166-194) -----------------------------------------------------------
        const GTetRec *t1 = (const GTetRec *)(it->tet1 + cw.a);
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
                Ee1[j] = it->Ebar1 * ((const gdouble *)it->eps1)[4 * (size_t)cw.a + j];
                Ee2[j] = it->Ebar * ((const gdouble *)it->eps2)[4 * (size_t)cw.b + j];
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
    // Trivial reject: if every vertex is non-positive on some plane the clip is empty.  Bit-exact shortcut:
    // every clipped vertex is c1*p2 - c2*p1 with c1 >= 0 >= c2 (static_clip.jl:197-201), whose sign on that
    // plane is exact, so Sutherland-Hodgman returns the empty polygon at that plane (:44).
    return np_front_accept(z, n_in, status, report);
}

// Clip-only narrowphase: poly_r2 = mul_then_un_pad(x_r2_ζ2, poly_ζ2) (poly_eight.jl:83-98) fused with centroid(poly_r2, n̂2)
// (poly_eight.jl:35-52), as in the full kernel; every vertex leaves for the kept polygon (SoA slot `slot`, streaming stores)
// as it is converted.  k_integ and k_fric read what is written here.
// (V = x_r2_ζ2 and E = ϵ_r2 of the tet, the second line of its record, as the caller loaded them: k_clip_queue issues that gather
// before it clips)
template <class Ring>
__device__ __forceinline__ void np_keep_polygon(const Ring &R, int n, const double (&V)[12], const double (&E)[4], V3 nh, const NpArgs &g,
                                                int slot, int item, int idx) {
    const size_t P = (size_t)g.pcap;
    double *o = g.poly + slot;
#define NT_(p, v) __builtin_nontemporal_store((v), (p))
    auto conv = [&](int k) {
        const double z0 = R.get(k, 0), z1 = R.get(k, 1), z2 = R.get(k, 2), z3 = R.get(k, 3);
        const V3 x = mk3(((V[0] * z0 + V[3] * z1) + V[6] * z2) + V[9] * z3,
                         ((V[1] * z0 + V[4] * z1) + V[7] * z2) + V[10] * z3,
                         ((V[2] * z0 + V[5] * z1) + V[8] * z2) + V[11] * z3);
        NT_(o + (10 + 3 * k) * P, x.x); NT_(o + (11 + 3 * k) * P, x.y); NT_(o + (12 + 3 * k) * P, x.z);
        return x;
    };
    const V3 a = conv(0);
    V3 cc = conv(1);
    double cum_sum = 0.0;
    V3 cum_prod = mk3(0.0, 0.0, 0.0);
    for (int k = 2; k < n; ++k) {
        const V3 b = cc;
        cc = conv(k);
        const double ar = triangle_area(a, b, cc, nh);
        cum_prod = cum_prod + ((a + b) + cc) * (1.0 / 3.0) * ar;
        cum_sum += ar;
    }
    const V3 cen = (cum_sum == 0.0) ? a : cum_prod / cum_sum;
    NT_(&g.poly_item[slot], (int)((unsigned)item | ((unsigned)n << 28)));
    NT_(o, nh.x); NT_(o + P, nh.y); NT_(o + 2 * P, nh.z);
    NT_(o + 3 * P, cen.x); NT_(o + 4 * P, cen.y); NT_(o + 5 * P, cen.z);
    NT_(o + 6 * P, E[0]); NT_(o + 7 * P, E[1]); NT_(o + 8 * P, E[2]);
    NT_(o + 9 * P, E[3]);
    if (g.poly_cand) NT_(&g.poly_cand[slot], idx);
#undef NT_
}
template <class Ring>
__device__ __forceinline__ void np_keep_polygon(const Ring &R, int n, const GTetRec *tp, V3 nh, const NpArgs &g, int slot, int item,
                                                int idx) {
    double V[12], E[4];
#pragma unroll
    for (int k = 0; k < 12; ++k) V[k] = tp->xrz[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) E[k] = tp->epsr[k];
    np_keep_polygon(R, n, V, E, nh, g, slot, item, idx);
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
// MODE 0: everything up to the per-item sums in this kernel.  MODE 1: the same with option debug (per-candidate clip
// counts and the materialised TractionCache, pfc_debug_*): a build of its own, so that the default one does not keep the
// nine TractionCache pointers, their counters and the slot bookkeeping live (235 -> 223 VGPRs, 175 -> 145 scalar
// registers spilled into vector lanes; narrowphase 2.48 -> 2.41 ms).  MODE 2 (big batches): gather, clip, polygon set-up
// and the kept polygon only; the quadrature and the per-item sums are k_integ's, which walks the compacted polygons with
// every lane busy (here 46 % of the lanes are rejected candidates and a lane waits for the longest fan of its wave).
template <bool TT, int MODE>
__global__ void __launch_bounds__(kNpBlock) k_narrow(NpArgs g) {
    constexpr bool DBG = MODE == 1, CLIP = MODE == 2 || MODE == 3, COMPACT = MODE == 3;
    constexpr int RC = COMPACT ? kRingColsClip : kNpBlock;
    __shared__ double poly[8 * 4 * RC];
    const int lane = threadIdx.x;
    int n_c = *g.ccount;
    if (n_c > g.ccap) n_c = g.ccap;
    const int C = np_chunk(n_c, g.chunk_switch);
    const int n_chunk = (n_c + C - 1) / C;
#ifdef PFC_STAMPS
    unsigned long long st_c[6] = {0, 0, 0, 0, 0, 0}, st_r[3] = {0, 0, 0};
#endif
    for (int ch = blockIdx.x; ch < n_chunk; ch += gridDim.x) {
    int pc = 0;   // polygons kept so far in this chunk's slot range
    const int n_round = ((n_c - ch * C < C ? n_c - ch * C : C) + kNpBlock - 1) / kNpBlock;
    for (int rd = 0; rd < n_round; ++rd) {
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
        (void)t0; (void)t1; (void)t2; (void)t3; (void)t4; (void)t5;
        STAMP(t0);
        const int idx = ch * C + rd * kNpBlock + lane;
        bool active = idx < n_c;
        WorkRec cw;
        cw.item = 0; cw.a = 0; cw.b = 0; cw.pad = 0;
        if (active) cw = g.cand[idx];
        if ((unsigned)cw.item >= (unsigned)g.n_items) {     // an unwritten slot is reported, never followed
            atomicOr(g.status, kStHole);
            cw.item = 0; cw.a = 0; cw.b = 0;
            active = false;
        }
        const ItemRec *it = g.items + cw.item;
        const GTetRec *tp = (const GTetRec *)(it->tet + cw.b);
        const int nq = it->nq;
        const bool reg = it->model == PFC_REGULARIZED;
        const bool materialise = DBG && active;
        const bool work = active;
        int n_poly = 0, rbase = 0, rcol = lane;
        V3 nh = mk3(0.0, 0.0, 0.0);
        // input polygon (3 or 4 vertices) in the coordinates of tet 2, its normal, and whether it passed the trivial reject
        double z[4][4];
        int n_in = 0;
        V3 nh_in = mk3(0.0, 0.0, 0.0);
        bool survivor = false;
        // ---- clip_in_tet_coordinates (static_clip.jl:7-23,34-201), polygon ring in LDS (column rcol), clipped in place ----
        auto clip_ring = [&]() {
            int n = n_in;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < n_in) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) PR(k, i) = z[k][i];
                }
            bool err = false;
            RingCol<RC> ring{poly, rcol, rbase};
            n = clip_ring_in_tet_coordinates(ring, n_in, err);     // pfc_clip.h
            rbase = ring.rbase;
            if (err) atomicOr(g.status, kStNonFinite);
            n_poly = n;
            if (kElim == 4 && nh_in.x > -1e300) n_poly = 0;
            if (n >= 3) nh = nh_in;
        };
        // ==== phase 1 (divergent): gather, transform to tet coordinates, clip ========================================
        if (work) {
            if (np_front<TT>(it, cw, tp, z, n_in, nh_in, g.status, true)) {
                STAMP(t1);
                survivor = true;
                if constexpr (!COMPACT) clip_ring();      // (MODE 3 clips below, on the compacted ring)
            }
        }
        if (DBG && g.clip_n && active) g.clip_n[idx] = n_poly;
        STAMP(t2);
        if constexpr (CLIP) {
            // ==== clip-only mode: polygon set-up, kept polygon, non-empty count; k_integ does the rest ==================
            const unsigned long long sm = COMPACT ? __ballot(survivor) : 1ull, am = __ballot(active);
            const int rank = COMPACT ? __popcll(sm & ((1ull << lane) - 1ull)) : 0, n_surv = __popcll(sm);
            unsigned long long km_all = 0ull;
            bool has_poly_any = false;
            // MODE 3: wave-uniform passes over the survivors, RC at a time (a second pass is rare: > 48 survivors of 64);
            // MODE 2: one pass, every lane clipped above on its own column
            for (int pass0 = 0; pass0 < n_surv; pass0 += RC) {
            bool mine = survivor;
            if constexpr (COMPACT) {
                mine = survivor && rank >= pass0 && rank < pass0 + RC;
                rcol = rank - pass0;
                rbase = 0; n_poly = 0;
                if (mine) clip_ring();
            }
            const bool has_poly = mine && n_poly >= 3;
            has_poly_any |= has_poly;
            const unsigned long long km = __ballot(has_poly);
            km_all |= km;
            const int slot = ch * C + pc + __popcll(km & ((1ull << lane) - 1ull));
            pc += __popcll(km);
            if (has_poly) {
                RingCol<RC> ring{poly, rcol, rbase};
                np_keep_polygon(ring, n_poly, tp, nh, g, slot, cw.item, idx);
            }
            // (the LDS unit serves a wave's instructions in order: the next pass may overwrite the ring)
            }
            if (km_all) {
                const int item_first = __builtin_amdgcn_readlane(cw.item, __builtin_ctzll(am));
                if (__all(!active || cw.item == item_first)) {
                    if (lane == 0) atomicAdd(&g.icnt[4 * (size_t)item_first + 2], __popcll(km_all));
                } else {
                    count_per_item(g.icnt, cw.item, 2, active, has_poly_any);
                }
            }
            continue;
        }
        // ==== phase 2 (wave-uniform): reserve a contiguous run of traction slots for the whole wave ================
        // A lane with an n-gon owns n * nq consecutive slots, so the traction points of a wave (and, because the
        // candidate list is grouped by item, of an item) are contiguous: the later per-point passes then reduce
        // wave-uniformly with one atomic per wave instead of one per lane.
        const int slots = (materialise && n_poly >= 3) ? n_poly * nq : 0;
        int tbase = 0;
        if (DBG) {
            int incl = seg_incl_scan(slots);
            const int tot = __shfl(incl, 63, 64);
            int base = 0;
            if (tot > 0) {
                if (lane == 0) base = atomicAdd(g.tcount, tot);
                base = __shfl(base, 0, 64);
            }
            tbase = base + incl - slots;
        }
        // Slots of the lists this wave will append to at the END of the round are reserved HERE, before the integration,
        // with one returning atomic on the counters of this workgroup's list region (pfc_kernels.h, "Region-partitioned
        // append lists": on a single counter the 60 k atomics of a C3 batch were serialised by the L2 and cost a quarter
        // of the kernel).  The compiler waits for the result right away, so what is left of its latency (~3 % of the
        // kernel) is not hidden by the placement; reordering the integration's loads in front of it needs more than
        // 256 VGPRs.  What a lane will contribute is not known yet, so the reservation is for every lane that has a
        // polygon (bristle lanes: a kept-polygon slot; with the Dual list on: a contributing-pair slot) and for one
        // moment record per run of an item; slots that turn out unused get an empty marker.
        const bool has_poly = work && n_poly >= 3;
        const bool polyb = has_poly && !reg;
        const bool polys = has_poly && g.surv != nullptr;
        const unsigned long long km = __ballot(polyb), sm = __ballot(polys);
        // heads of the runs of equal items among the active lanes (the candidate list is grouped by item)
        const int item_prev = __shfl_up(cw.item, 1, 64);
        const unsigned long long am = __ballot(active);
        const unsigned long long heads = __ballot(active && (lane == 0 || item_prev != cw.item || !((am >> (lane - 1)) & 1ull)));
        int rbase_raw = 0;   // record offset in this workgroup's region
        int sbase_raw = 0;
        const int rgn_c = blockIdx.x & (kRgn - 1);
        if (lane == 0) {
            // one record per run of an item; the polygon slots need no reservation (the chunk's own range, counter pc)
            if (km) rbase_raw = atomicAdd(g.rgn + rgn_c * kRgnStride + 1, __popcll(heads));
            if (sm) sbase_raw = atomicAdd(g.scount, __popcll(sm));   // Dual evaluations only
        }
        const int pbase = pc;
        pc += __popcll(km);
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
        if (n_poly >= 3 && !(kElim == 9 && !(nh.x > 1e300))) {
            const int n = n_poly;
            // poly_r2 = mul_then_un_pad(x_r2_ζ2, poly_ζ2) (poly_eight.jl:83-98), converted in place (x, y, z), fused with
            // centroid(poly_r2, n̂2) (poly_eight.jl:35-52): vertex k is converted when the centroid fan first needs it
            {
                double V[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) V[k] = tp->xrz[k];
                auto conv = [&](int k) {
                    const double z0 = PR(k, 0), z1 = PR(k, 1), z2 = PR(k, 2), z3 = PR(k, 3);
                    const V3 x = mk3(((V[0] * z0 + V[3] * z1) + V[6] * z2) + V[9] * z3,
                                     ((V[1] * z0 + V[4] * z1) + V[7] * z2) + V[10] * z3,
                                     ((V[2] * z0 + V[5] * z1) + V[8] * z2) + V[11] * z3);
                    PR(k, 0) = x.x; PR(k, 1) = x.y; PR(k, 2) = x.z;
                    return x;
                };
                const V3 a = conv(0);
                V3 cc = conv(1);
                double cum_sum = 0.0;
                V3 cum_prod = mk3(0.0, 0.0, 0.0);
                for (int k = 2; k < n; ++k) {
                    const V3 b = cc;
                    cc = conv(k);
                    const double ar = triangle_area(a, b, cc, nh);
                    cum_prod = cum_prod + ((a + b) + cc) * (1.0 / 3.0) * ar;
                    cum_sum += ar;
                }
                cen = (cum_sum == 0.0) ? a : cum_prod / cum_sum;
            }
            const double er0 = tp->epsr[0], er1 = tp->epsr[1], er2 = tp->epsr[2], er3 = tp->epsr[3];
            const V3 w = ld3(it->w), vl = ld3(it->v);
            const double chi = it->chi, Ebar = it->Ebar;
            const double v_c = it->v_c, mu_s = it->mu_s, mu_d = it->mu_d;
            const bool store = DBG && materialise && (tbase + slots <= g.tcap);
            if (DBG && materialise && !store) atomicOr(g.status, kStTracOvf);
            int tpos = tbase;
            PointParams pp;
            pp.w = w; pp.vl = vl; pp.chi = chi; pp.Ebar = Ebar; pp.er0 = er0; pp.er1 = er1; pp.er2 = er2; pp.er3 = er3; pp.nq = nq;
            V3 v2 = mk3(PR(n - 1, 0), PR(n - 1, 1), PR(n - 1, 2));
            for (int k = 0; k < n; ++k) {
                V3 v1 = v2;
                v2 = mk3(PR(k, 0), PR(k, 1), PR(k, 2));
                if (kElim == 7 && nh.x > -1e300) continue;
                // fillTractionCacheForTriangle! / InnerLoop! (:236-265): the shared statement of r, p, dA (pfc_kernels.h)
                n_trac_lane += fan_triangle_points(pp, v1, v2, cen, nh, [&](const V3 &r, const V3 &rdot, double p, double dA) {
                    const double p_dA = p * dA;
                    if (DBG && store) {
                        g.trac.item[tpos] = cw.item;
                        g.trac.nx[tpos] = nh.x; g.trac.ny[tpos] = nh.y; g.trac.nz[tpos] = nh.z;
                        g.trac.rx[tpos] = r.x; g.trac.ry[tpos] = r.y; g.trac.rz[tpos] = r.z;
                        g.trac.dA[tpos] = dA; g.trac.p[tpos] = p;
                        ++tpos;
                    }
                    if (reg) {
                        // yes_contact!(::Regularized) (friction.jl:50-72) fused
                        V3 vt = vec_sub_vec_proj(rdot, nh);
                        double m2 = dot_fma(vt, vt);
                        V3 T;
                        // one division per point where the reference divides the three components (friction.jl:64-68):
                        // last-bit differences only
                        if (m2 < v_c * v_c) {
                            T = vt * (-(mu_s / v_c));
                        } else {
                            // as in k_fric: 1/|v_t| from the hardware rsqrt and two Newton steps
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
                        wrr[0] = __builtin_fma(wx, rc.x, wrr[0]); wrr[1] = __builtin_fma(wx, rc.y, wrr[1]);
                        wrr[2] = __builtin_fma(wx, rc.z, wrr[2]); wrr[3] = __builtin_fma(wy, rc.y, wrr[3]);
                        wrr[4] = __builtin_fma(wy, rc.z, wrr[4]); wrr[5] = __builtin_fma(wz, rc.z, wrr[5]);
                    }
                });
            }
            if (!reg) {
                const double W = sum[6];
                const V3 Sr = mk3(wr1[0] + W * cen.x, wr1[1] + W * cen.y, wr1[2] + W * cen.z);   // sum w r
                const V3 ta = cross(Sr, nh);
                sum[0] = ta.x; sum[1] = ta.y; sum[2] = ta.z;
                sum[3] = nh.x * W; sum[4] = nh.y * W; sum[5] = nh.z * W;
                sum[7] = Sr.x; sum[8] = Sr.y; sum[9] = Sr.z;
            }
            if (DBG && store)  // unused slots of this lane's run (area <= 0 or p <= 0 points)
                for (; tpos < tbase + slots; ++tpos) g.trac.item[tpos] = -1;
        }
        STAMP(t4);
        // ==== phase 4 (wave-uniform): per-item reductions =============================================================
        const bool contributed = work && n_trac_lane > 0 && !(kElim == 5 && !(sum[6] > 1e300));
        {
            // ---- (a) the polygons of contributing bristle pairs, kept for k_fric, and (b) when pfc_eval_dual asked for it,
            // the candidate indices of the contributing pairs, in the slots reserved before the integration (lanes that
            // had a polygon but no pressure point leave an empty marker).
            const bool keep = contributed && !reg;
            if (km | sm) {
                const int sbase = __builtin_amdgcn_readfirstlane(sbase_raw);
                const unsigned long long below = (1ull << lane) - 1ull;
                if (polys) g.surv[sbase + __popcll(sm & below)] = contributed ? idx : -1;   // <= ccap entries
                // the chunk's own slot range holds every candidate of the chunk; every slot below the chunk's count is written
                const int off = pbase + __popcll(km & below);
                const int slot = ch * C + off;
                if (polyb && off < C) {
                    const size_t P = (size_t)g.pcap;
                    double *o = g.poly + slot;
                    // streaming stores: 0.5 GB per C3 batch must not evict the mesh records from the XCD's 4 MiB L2
#define NT_(p, v) __builtin_nontemporal_store((v), (p))
                    NT_(&g.poly_item[slot], keep ? (int)((unsigned)cw.item | ((unsigned)n_poly << 28)) : 0);
                    if (keep) {
                        NT_(o, nh.x); NT_(o + P, nh.y); NT_(o + 2 * P, nh.z);
                        NT_(o + 3 * P, cen.x); NT_(o + 4 * P, cen.y); NT_(o + 5 * P, cen.z);
                        NT_(o + 6 * P, tp->epsr[0]); NT_(o + 7 * P, tp->epsr[1]); NT_(o + 8 * P, tp->epsr[2]);
                        NT_(o + 9 * P, tp->epsr[3]);
                        for (int k = 0; k < n_poly; ++k) {
                            NT_(o + (10 + 3 * k) * P, PR(k, 0)); NT_(o + (11 + 3 * k) * P, PR(k, 1));
                            NT_(o + (12 + 3 * k) * P, PR(k, 2));
                        }
                    }
#undef NT_
                }
            }
            unsigned long long r1 = 0, r2 = 0, r3 = 0; (void)r1; (void)r2; (void)r3;
            STAMP(r1);
            // ---- the ten per-item sums.  Single-item wave (the rule: an item has ~30 waves of candidates): LDS transpose;
            // otherwise segmented scans keyed by item.  The polygon ring is free from here on (its last reader was the
            // polygon store above).
            const int item_first = __builtin_amdgcn_readlane(cw.item, am ? __builtin_ctzll(am) : 0);
            const bool single = __all(!active || cw.item == item_first);
            double t10 = 0.0;   // single: lane k < 10 holds total k
            if (single) {
                if (__any(contributed)) {
                    t10 = lds_row_sums<10, 4>(poly, sum, contributed, lane, 0);
                    if (lane < 10 && t10 != 0.0) unsafeAtomicAdd(&g.acc[(size_t)item_first * kAccStride + lane], t10);
                }
            } else {
                accumulate_items<10>(g.acc, cw.item, active, contributed, sum, 0);
            }
            STAMP(r2);
            // ---- patch-stiffness moments of the bristle model, one record per run of an item in this wave, in the slots
            // reserved before the integration (run r of the wave -> slot rbase + r; a run without a record gets W = 0)
            const int rbase = __builtin_amdgcn_readfirstlane(rbase_raw);   // offset in the region
            const int rslot0 = rgn_c * g.rr_cap;
            unsigned long long rec_done = 0;   // bit r: run r has its record
            if (__any(contributed && !reg)) {
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
                    double mine = lds_row_sums<27, 2>(poly, v, cb, lane, 5);
                    if (Wt > 0.0) {
                        if (lane == 0) mine = (double)item_first;
                        if (lane == 1) mine = Wt;
                        if (lane == 2) mine = cwv.x;
                        if (lane == 3) mine = cwv.y;
                        if (lane == 4) mine = cwv.z;
                        if (rbase < g.rr_cap) {   // the wave's single run
                            if (lane < kRecStride) g.rec[(size_t)(rslot0 + rbase) * kRecStride + lane] = mine;
                        } else if (lane == 0) {
                            atomicOr(g.status, kStRecOvf);
                        }
                        rec_done |= 1ull;
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
                    const int run = __popcll(heads & ((2ull << t) - 1ull)) - 1;   // heads at or before the tail lane
                    if (rbase + run < g.rr_cap) {
                        if (lane < kRecStride) g.rec[(size_t)(rslot0 + rbase + run) * kRecStride + lane] = mine;
                    } else if (lane == 0) {
                        atomicOr(g.status, kStRecOvf);
                    }
                    rec_done |= 1ull << run;
                }
            }
            if (km) {   // reserved but unused record slots: W = 0
                const int n_run = __popcll(heads);
                if (lane < n_run && !((rec_done >> lane) & 1ull) && rbase + lane < g.rr_cap)
                    g.rec[(size_t)(rslot0 + rbase + lane) * kRecStride + 1] = 0.0;
            }
            STAMP(r3);
#ifdef PFC_STAMPS
            st_r[0] += r1 - t4; st_r[1] += r2 - r1; st_r[2] += r3 - r2;
#endif
            if (single) {   // one item: a popcount and one integer wave sum instead of two segmented scans
                const int n_ne = __popcll(__ballot(has_poly));
                // traction points of the wave: a lane has at most 8 x 3 = 24, i.e. five bits -> five ballots
                const int ntl = contributed ? n_trac_lane : 0;
                int nt = 0;
#pragma unroll
                for (int b = 0; b < 5; ++b) nt += __popcll(__ballot((ntl >> b) & 1)) << b;
                if (lane == 0 && am) {
                    if (n_ne) atomicAdd(&g.icnt[4 * (size_t)item_first + 2], n_ne);
                    if (nt) atomicAdd(&g.icnt[4 * (size_t)item_first + 3], nt);
                }
            } else {
                count_per_item(g.icnt, cw.item, 2, active, has_poly);
                count_per_item(g.icnt, cw.item, 3, active, contributed, n_trac_lane);
            }
        }
#ifdef PFC_STAMPS
        STAMP(t5);
        {   // summed in registers (an atomic per round on one address would itself dominate the timing)
            if (t1 == 0) t1 = t2;
            st_c[0] += t1 - t0; st_c[1] += t2 - t1; st_c[2] += t3 - t2; st_c[3] += t4 - t3; st_c[4] += t5 - t4; st_c[5] += 1;
        }
#endif
    }
    if (lane == 0) g.pcnt[ch] = pc;
    }
#ifdef PFC_STAMPS
    if (lane == 0 && g.stamps && st_c[5])
    {
        for (int k = 0; k < 6; ++k) atomicAdd(&g.stamps[k], st_c[k]);
        atomicAdd(&g.stamps[6], st_r[0]); atomicAdd(&g.stamps[7], st_r[1]); atomicAdd(&g.stamps[14], st_r[2]);
    }
#endif
}
#undef PR

// =================================================================================================================
// k_clip_queue (round 3) -- clip-only narrowphase of tri-tet batches: survivors of the trivial reject are QUEUED in the
// polygon ring itself and clipped 64 at a time, every lane busy.  k_narrow<.., 2 / 3> clips a round of 64 candidates with
// the ~35 lanes that pass the trivial reject (C3: 46 % of the candidates are rejected right after the gather); the other
// lanes idle through the divergent Sutherland-Hodgman, the polygon conversion and the kept-polygon stores (31 of 64 lanes
// active per VALU instruction, profiles/pmc_valu.json of round 2).  A first form of this kernel ran the front over the
// whole chunk, noted the survivors and evaluated the front a SECOND time for 64 dense survivors at a time: elimination
// builds put that second front at 0.25 ms of its 1.23 ms (profiles/r03_clip_elimination.txt), and it lost to
// k_narrow<.., 3> under the two-half overlap (4.22 vs 4.15 ms per step); this form keeps what the front computed.
//
// A queued survivor needs its 3 input vertices in tet coordinates (12 doubles), n̂ (3) and (item, tet) (one 8-byte word):
// 16 doubles -- half of a ring column (8 slots x 4 coordinates).  Entry e < 64 lives in column e, slots 0..2 (+ slot 6:
// n̂, word); entry 64 + e in column e, slots 3..5 (+ slot 7).  A round of the front adds at most 64 entries to fewer than
// 64 waiting ones, so 128 entries always suffice.  With 64 or more queued, lane c clips entry c in place (its vertices
// already sit at logical 0..2 of column c), having first taken the upper entry of its column into registers; afterwards it
// writes that entry back as entry c of the remaining queue.  16 KiB of LDS as before, no second gather, no second front.
// Tet-tet candidates enter with 4 vertices and do not fit the half column: scenarios with tet-tet instructions keep
// k_narrow<true, 2 / 3>.
// =================================================================================================================
__global__ void __launch_bounds__(kNpBlock) k_clip_queue(NpArgs g) {
    constexpr int RC = kNpBlock;
    __shared__ double poly[8 * 4 * RC];
    __shared__ unsigned short qpos[2 * kNpBlock];     // position in the chunk of every queued entry
    const int lane = threadIdx.x;
    int n_c = *g.ccount;
    if (n_c > g.ccap) n_c = g.ccap;
    const int C = np_chunk(n_c, g.chunk_switch);
    const int n_chunk = (n_c + C - 1) / C;
    const unsigned long long below = (1ull << lane) - 1ull;
#define QZ(e, k, i) poly[(((((e) >> 6) * 3 + (k)) * 4 + (i)) * RC) + ((e) & 63)]
#define QX(e, i) poly[(((6 + ((e) >> 6)) * 4 + (i)) * RC) + ((e) & 63)]
    for (int ch = blockIdx.x; ch < n_chunk; ch += gridDim.x) {
        const int n_here = n_c - ch * C < C ? n_c - ch * C : C;
        const int n_round = (n_here + kNpBlock - 1) / kNpBlock;
        int q = 0;    // queued entries
        int pc = 0;   // polygons kept so far in this chunk's slot range
        // clips entries 0 .. min(q, 64) - 1, one per lane; entries 64 .. q - 1 become entries 0 .. q - 65
        auto dense_round = [&]() {
            const int n_take = q < kNpBlock ? q : kNpBlock;
            const bool mine = lane < n_take, left = kNpBlock + lane < q;
            V3 nh = mk3(0.0, 0.0, 0.0);
            unsigned long long meta = 0ull;
            int pos = 0;
            if (mine) {
                nh = mk3(QX(lane, 0), QX(lane, 1), QX(lane, 2));
                meta = (unsigned long long)__double_as_longlong(QX(lane, 3));
                pos = (int)qpos[lane];
            }
            double zl[3][4], xl[4];
            int posl = 0;
            if (left) {
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i) zl[k][i] = QZ(kNpBlock + lane, k, i);
#pragma unroll
                for (int i = 0; i < 4; ++i) xl[i] = QX(kNpBlock + lane, i);
                posl = (int)qpos[kNpBlock + lane];
            }
            const int item = (int)(unsigned)(meta & 0xFFFFFFFFull), b = (int)(unsigned)(meta >> 32);
            const ItemRec *it = g.items + item;
            const GTetRec *tp = (const GTetRec *)(it->tet + b);
            RingCol<RC> ring{poly, lane, 0};
            int n_poly = 0;
            // the second line of the tet's record (vertices, ϵ_r2: what a kept polygon needs) is asked for NOW, a dependent gather
            // behind the queue entry that would otherwise start after the clip and be waited for in full
            double V[12] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, E[4] = {0.0, 0.0, 0.0, 0.0};
            if (mine) {
#pragma unroll
                for (int k = 0; k < 12; ++k) V[k] = tp->xrz[k];
#pragma unroll
                for (int k = 0; k < 4; ++k) E[k] = tp->epsr[k];
            }
            if (mine) {
                bool err = false;
                n_poly = clip_ring_in_tet_coordinates(ring, 3, err);     // pfc_clip.h
                if (err) atomicOr(g.status, kStNonFinite);
            }
            const bool has_poly = n_poly >= 3;
            const unsigned long long km = __ballot(has_poly);
            const int slot = ch * C + pc + __popcll(km & below);
            pc += __popcll(km);
            if (has_poly) np_keep_polygon(ring, n_poly, V, E, nh, g, slot, item, ch * C + pos);
            if (km) {
                const int item_first = __builtin_amdgcn_readfirstlane(item);      // lane 0 always holds an entry here
                if (__all(!mine || item == item_first)) {
                    if (lane == 0) atomicAdd(&g.icnt[4 * (size_t)item_first + 2], __popcll(km));
                } else {
                    count_per_item(g.icnt, item, 2, mine, has_poly);
                }
            }
            // the upper entry of this column moves down (a lane only touches its own column and its own two qpos entries)
            if (left) {
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i) QZ(lane, k, i) = zl[k][i];
#pragma unroll
                for (int i = 0; i < 4; ++i) QX(lane, i) = xl[i];
                qpos[lane] = (unsigned short)posl;
            }
            q -= n_take;
            wave_lds_sync();
        };
        for (int rd = 0; rd < n_round; ++rd) {
            const int pos = rd * kNpBlock + lane, idx = ch * C + pos;
            bool active = pos < n_here;
            WorkRec cw;
            cw.item = 0; cw.a = 0; cw.b = 0; cw.pad = 0;
            if (active) cw = g.cand[idx];
            if ((unsigned)cw.item >= (unsigned)g.n_items) {     // an unwritten slot is reported, never followed
                atomicOr(g.status, kStHole);
                cw.item = 0; cw.a = 0; cw.b = 0;
                active = false;
            }
            const ItemRec *it = g.items + cw.item;
            const GTetRec *tp = (const GTetRec *)(it->tet + cw.b);
            double z[4][4];
            int n_in = 0;
            V3 nh_in = mk3(0.0, 0.0, 0.0);
            bool survivor = false;
            if (active) survivor = np_front<false>(it, cw, tp, z, n_in, nh_in, g.status, true);
            const unsigned long long sm = __ballot(survivor);
            if (survivor) {
                const int e = q + __popcll(sm & below);
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i) QZ(e, k, i) = z[k][i];
                QX(e, 0) = nh_in.x; QX(e, 1) = nh_in.y; QX(e, 2) = nh_in.z;
                QX(e, 3) = __longlong_as_double((long long)(((unsigned long long)(unsigned)cw.b << 32) | (unsigned long long)(unsigned)cw.item));
                qpos[e] = (unsigned short)pos;
            }
            q += __popcll(sm);
            wave_lds_sync();
            if (q >= kNpBlock) dense_round();
        }
        if (q > 0) dense_round();
        if (lane == 0) g.pcnt[ch] = pc;
    }
#undef QZ
#undef QX
}

// =================================================================================================================
// k_integ -- integrate_over_polygon_patch! (non_friction.jl:217-265) over the polygons k_narrow<.., 2> kept: one lane per
// kept polygon, every lane busy (the clip kernel's waves are half rejected candidates), coalesced loads only.  The fan /
// quadrature arithmetic is the full kernel's expression for expression, so the traction points are bit-identical.
//
// Per-item sums stay in REGISTERS across the 64-polygon pieces of a chunk (a chunk's polygons belong to one item, two at a
// run boundary of the candidate list): the ten wrench / cop sums, and for bristle items the 27 patch-stiffness moments
// (normal_wrench_cop, normal.jl:17-34, and calc_patch_spatial_stiffness!, friction.jl:147-169) taken about c0, the centroid
// of the first polygon of the run -- a point inside the patch, so the shift to the cop (k_shift, parallel-axis terms with
// the first moment m1 = sum w (r - c0) carried in the record) is a shift by less than the patch size.  One LDS transpose
// reduction, one 10-lane and one record write per (chunk, item) instead of per wave round.
// =================================================================================================================
struct IntegArgs {
    const ItemRec *items;
    int n_items;
    const int *ccount;
    int ccap, chunk_switch;
    const int *pcnt;
    int *poly_item;          // key of a polygon that must not reach k_fric (no pressure point, regularized item) is cleared
    const double *poly;
    int pcap;
    const int *poly_cand;    // candidate index per kept polygon (Dual list on) or null
    int *surv, *scount;      // contributing candidates (work list of the Dual passes) or null
    double *acc, *rec;
    int *rgn;                // record counters (word 1 of a region)
    int rr_cap;
    int *icnt;
    unsigned *status;
    int *det;                // k_integ_fixed: per item (last record, first chunk, last chunk) -- k_fixed_init, k_shift_fixed
};

// the row-summing half of lds_row_sums on rows that already sit in LDS (row k at buf[k (64 + L) + lane])
template <int N, int L>
__device__ __forceinline__ double lds_rows_total(const double *buf, int lane, int lane0) {
    constexpr int RS = 64 + L;
    static_assert(N * L <= 64 && (L == 1 || L == 2 || L == 4), "rows x lanes per row must fit the wave");
    double t = 0.0;
    const int row = lane / L, part = lane % L;
    if (row < N) {
        const double *r = buf + row * RS + part;
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
        for (int j = 0; j < 64 / L; j += 8) {
            const double a0 = r[L * j], a1 = r[L * (j + 1)], a2 = r[L * (j + 2)], a3 = r[L * (j + 3)];
            const double a4 = r[L * (j + 4)], a5 = r[L * (j + 5)], a6 = r[L * (j + 6)], a7 = r[L * (j + 7)];
            t0 += a0; t1 += a1; t2 += a2; t3 += a3;
            t0 += a4; t1 += a5; t2 += a6; t3 += a7;
        }
        t = (t0 + t1) + (t2 + t3);
    }
#pragma unroll
    for (int m = 1; m < L; m <<= 1) t += __shfl_xor(t, m, 64);
    const int dst_row = lane - lane0;
    const double out = __shfl(t, (dst_row >= 0 && dst_row < N) ? dst_row * L : 0, 64);
    return (dst_row >= 0 && dst_row < N) ? out : 0.0;
}

// DET (option "fixed_order", the candidate list sorted: pfc_sort.hip): a run leaves NO atomic sum behind -- its ten sums travel in
// its record next to the moments, the record carries its chunk index and hangs in a list per item, and k_shift_fixed adds an
// item's records in chunk order.
template <bool DET>
__device__ __forceinline__ void integ_body(const IntegArgs &g, double *m27) {
    constexpr int RS = 66;                  // row stride of lds_row_sums / lds_rows_total with two lanes per row
    const int lane = threadIdx.x;
    int n_c = *g.ccount;
    if (n_c > g.ccap) n_c = g.ccap;
    const int C = np_chunk(n_c, g.chunk_switch);
    const int n_chunk = (n_c + C - 1) / C;
    const size_t P = (size_t)g.pcap;
    for (int ch = blockIdx.x; ch < n_chunk; ch += gridDim.x) {
        int cnt = __builtin_amdgcn_readfirstlane(g.pcnt[ch]);
        if (cnt > C) cnt = C;
        // the running sums of the current (chunk, item) run: ten in registers, the 27 moments in this lane's LDS column
        int cur = -1, n_tr = 0;
        bool cur_reg = false;
        V3 c0 = mk3(0.0, 0.0, 0.0);
        double s10[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) s10[k] = 0.0;
        auto flush = [&]() {
            if (cur < 0 || !__any(n_tr > 0)) return;
            // the ten sums: wave totals on DPP (once per run of a chunk), total k on lane k
            double t10 = 0.0, Wt = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                const double t = wave_total(s10[k], lane);
                if (lane == (DET ? kRecTen + k : k)) t10 = t;
                if (k == 6) Wt = t;
                if (k == 7) sx = t;
                if (k == 8) sy = t;
                if (k == 9) sz = t;
            }
            if constexpr (DET) {
                // one record per run whatever the model: [0] item [1] W (0: no moments) [2..4] c0 [5..31] moments [32..34] m1
                // [kRecTen..+10) the ten sums [kRecChunk] chunk [kRecNext] the item's previous record
                const bool mom = !cur_reg && Wt > 0.0;
                double mine = 0.0;
                if (mom) {
                    wave_lds_sync();
                    mine = lds_rows_total<27, 2>(m27, lane, 5);
                    if (lane == 1) mine = Wt;
                    if (lane == 2) mine = c0.x;
                    if (lane == 3) mine = c0.y;
                    if (lane == 4) mine = c0.z;
                    if (lane == 32) mine = sx - Wt * c0.x;
                    if (lane == 33) mine = sy - Wt * c0.y;
                    if (lane == 34) mine = sz - Wt * c0.z;
                }
                if (lane == 0) mine = (double)cur;
                if (lane >= kRecTen && lane < kRecTen + 10) mine = t10;
                if (lane == kRecChunk) mine = (double)ch;
                const int rgn_c = ch & (kRgn - 1);
                int rs = 0;
                if (lane == 0) rs = atomicAdd(g.rgn + rgn_c * kRgnStride + 1, 1);
                rs = __builtin_amdgcn_readfirstlane(rs);
                if (rs < g.rr_cap) {
                    const int slot = rgn_c * g.rr_cap + rs;
                    if (lane == kRecNext) mine = (double)atomicExch(&g.det[3 * (size_t)cur], slot);
                    if (lane == kRecNext + 1) { atomicMin(&g.det[3 * (size_t)cur + 1], ch); atomicMax(&g.det[3 * (size_t)cur + 2], ch); }
                    if (lane < kRecStrideFixed) g.rec[(size_t)slot * kRecStrideFixed + lane] = mine;
                } else if (lane == 0) {
                    atomicOr(g.status, kStRecOvf);
                }
                if (mom) wave_lds_sync();
            } else {
            if (lane < 10 && t10 != 0.0) unsafeAtomicAdd(&g.acc[(size_t)cur * kAccStride + lane], t10);
            if (!cur_reg && Wt > 0.0) {
                wave_lds_sync();
                double mine = lds_rows_total<27, 2>(m27, lane, 5);             // lanes 5..31: moments about c0
                if (lane == 0) mine = (double)cur;
                if (lane == 1) mine = Wt;
                if (lane == 2) mine = c0.x;
                if (lane == 3) mine = c0.y;
                if (lane == 4) mine = c0.z;
                if (lane == 32) mine = sx - Wt * c0.x;      // m1 = sum w (r - c0)
                if (lane == 33) mine = sy - Wt * c0.y;
                if (lane == 34) mine = sz - Wt * c0.z;
                const int rgn_c = ch & (kRgn - 1);
                int rs = 0;
                if (lane == 0) rs = atomicAdd(g.rgn + rgn_c * kRgnStride + 1, 1);
                rs = __builtin_amdgcn_readfirstlane(rs);
                if (rs < g.rr_cap) {
                    if (lane < kRecStride) g.rec[((size_t)rgn_c * g.rr_cap + rs) * kRecStride + lane] = mine;
                } else if (lane == 0) {
                    atomicOr(g.status, kStRecOvf);
                }
                wave_lds_sync();
            }
            }
            const int nt = wave_total_i(n_tr, lane);
            if (lane == 0 && nt) atomicAdd(&g.icnt[4 * (size_t)cur + 3], nt);
        };
        for (int p0 = 0; p0 < cnt; p0 += 64) {
            const int idx = ch * C + p0 + lane;
            const bool active = p0 + lane < cnt;
            unsigned pk = active ? (unsigned)g.poly_item[idx] : 0u;
            if ((pk >> 28) > 8u || ((pk >> 28) >= 3u && (pk & 0x0FFFFFFFu) >= (unsigned)g.n_items)) {
                atomicOr(g.status, kStHole);     // not a key k_narrow writes: an unwritten slot is reported, never followed
                pk = 0u;
            }
            const int n = (int)(pk >> 28);
            const bool has = n >= 3;
            const int item = has ? (int)(pk & 0x0FFFFFFFu) : -1;
            const double *o = g.poly + idx;
            V3 nh = mk3(0.0, 0.0, 0.0), cen = nh;
            double er0 = 0.0, er1 = 0.0, er2 = 0.0, er3 = 0.0;
            if (has) {
                nh = mk3(o[0], o[P], o[2 * P]);
                cen = mk3(o[3 * P], o[4 * P], o[5 * P]);
                er0 = o[6 * P]; er1 = o[7 * P]; er2 = o[8 * P]; er3 = o[9 * P];
            }
            // the item's constants per lane (vector loads issued together with the polygon's: no dependent scalar round trip)
            const ItemRec *it = g.items + (has ? item : 0);
            const bool lane_reg = it->model == PFC_REGULARIZED;
            const int nq = it->nq;
            const V3 w = ld3(it->w), vl = ld3(it->v);
            const double chi = it->chi, Ebar = it->Ebar;
            // ---- (A) the traction points of every polygon of the piece, ONCE: regularized lanes sum their wrench, bristle
            // lanes W and the moments of w about their OWN polygon centroid (normal_wrench_cop, normal.jl:17-34) ----------
            int ntl = 0;
            double W = 0.0, wr1[3] = {0.0, 0.0, 0.0}, wrr[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            double fs[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (has) {
                V3 v2 = mk3(o[(10 + 3 * (n - 1)) * P], o[(11 + 3 * (n - 1)) * P], o[(12 + 3 * (n - 1)) * P]);
                V3 vn = mk3(o[10 * P], o[11 * P], o[12 * P]);
                PointParams pp;
                pp.w = w; pp.vl = vl; pp.chi = chi; pp.Ebar = Ebar; pp.er0 = er0; pp.er1 = er1; pp.er2 = er2; pp.er3 = er3; pp.nq = nq;
                auto points = [&](auto &&body) {
                    for (int k = 0; k < n; ++k) {
                        const V3 v1 = v2;
                        v2 = vn;
                        // the next vertex is fetched while this triangle's points are evaluated (slot k + 1 <= 7 always exists)
                        if (k + 1 < n) vn = mk3(o[(13 + 3 * k) * P], o[(14 + 3 * k) * P], o[(15 + 3 * k) * P]);
                        ntl += fan_triangle_points(pp, v1, v2, cen, nh, body);
                    }
                };
                if (lane_reg) {
                    // yes_contact!(::Regularized) (friction.jl:50-72) fused, as in the full kernel
                    const double v_c = it->v_c, mu_s = it->mu_s, mu_d = it->mu_d;
                    points([&](const V3 &r, const V3 &rdot, double p, double dA) {
                        const double p_dA = p * dA;
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
                        fs[0] += ta.x; fs[1] += ta.y; fs[2] += ta.z;
                        fs[3] += tk.x; fs[4] += tk.y; fs[5] += tk.z;
                    });
                } else {
                    points([&](const V3 &r, const V3 &, double p, double dA) {
                        const double p_dA = p * dA;
                        W += p_dA;
                        const V3 rc = r - cen;
                        const double wx = p_dA * rc.x, wy = p_dA * rc.y, wz = p_dA * rc.z;
                        wr1[0] += wx; wr1[1] += wy; wr1[2] += wz;
                        wrr[0] = __builtin_fma(wx, rc.x, wrr[0]); wrr[1] = __builtin_fma(wx, rc.y, wrr[1]);
                        wrr[2] = __builtin_fma(wx, rc.z, wrr[2]); wrr[3] = __builtin_fma(wy, rc.y, wrr[3]);
                        wrr[4] = __builtin_fma(wy, rc.z, wrr[4]); wrr[5] = __builtin_fma(wz, rc.z, wrr[5]);
                    });
                }
            }
            const bool contributed = ntl > 0;
            // ---- (B) the runs of equal items of this piece (one; two at a run boundary of the candidate list; several when
            // the items are small, as in a pile of boxes): each lane's sums join its run's, the moments shifted from the
            // polygon centroid to the run's reference point c0 (parallel axis, |d| below the patch size) -------------------
            unsigned long long todo = __ballot(has);
            while (todo) {
                const int f = __builtin_ctzll(todo);
                const int run_item = __builtin_amdgcn_readlane(item, f);
                const bool in_run = has && item == run_item;
                todo &= ~__ballot(in_run);
                const bool reg = __builtin_amdgcn_readlane((int)lane_reg, f) != 0;
                if (run_item != cur) {
                    flush();
                    cur = run_item;
                    cur_reg = reg;
                    c0 = mk3(readlane_f64(cen.x, f), readlane_f64(cen.y, f), readlane_f64(cen.z, f));
                    n_tr = 0;
#pragma unroll
                    for (int k = 0; k < 10; ++k) s10[k] = 0.0;
                    if (!reg) {
#pragma unroll
                        for (int k = 0; k < 27; ++k) m27[k * RS + lane] = 0.0;
                    }
                }
                if (!in_run || !contributed) continue;
                n_tr += ntl;
                if (reg) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) s10[k] += fs[k];
                } else {
                    // the polygon's share of the ten sums: torque (sum w r) x n̂, force n̂ W, W, sum w r
                    const V3 Sr = mk3(wr1[0] + W * cen.x, wr1[1] + W * cen.y, wr1[2] + W * cen.z);
                    const V3 ta = cross(Sr, nh);
                    s10[0] += ta.x; s10[1] += ta.y; s10[2] += ta.z;
                    s10[3] += nh.x * W; s10[4] += nh.y * W; s10[5] += nh.z * W;
                    s10[6] += W; s10[7] += Sr.x; s10[8] += Sr.y; s10[9] += Sr.z;
                    // moments about c0: x' = x + d, d = cen - c0
                    const V3 d = cen - c0;
                    const V3 m1 = mk3(wr1[0] + W * d.x, wr1[1] + W * d.y, wr1[2] + W * d.z);
                    double q[6];
                    q[0] = wrr[0] + 2.0 * wr1[0] * d.x + W * d.x * d.x;
                    q[1] = wrr[1] + wr1[0] * d.y + wr1[1] * d.x + W * d.x * d.y;
                    q[2] = wrr[2] + wr1[0] * d.z + wr1[2] * d.x + W * d.x * d.z;
                    q[3] = wrr[3] + 2.0 * wr1[1] * d.y + W * d.y * d.y;
                    q[4] = wrr[4] + wr1[1] * d.z + wr1[2] * d.y + W * d.y * d.z;
                    q[5] = wrr[5] + 2.0 * wr1[2] * d.z + W * d.z * d.z;
                    // and the 27 moments about c0 (this lane's LDS column); n̂ is constant over the polygon:
                    // sum w n n' = W n n', sum w (x x n) n' = (m1 x n) n', sum w (x x n)(x x n)' = [n]x Q [n]x'
                    double *mc = m27 + lane;
#define M27_(k, v) mc[(k) * RS] += (v)
                    M27_(0, W * nh.x * nh.x); M27_(1, W * nh.x * nh.y); M27_(2, W * nh.x * nh.z);
                    M27_(3, W * nh.y * nh.y); M27_(4, W * nh.y * nh.z); M27_(5, W * nh.z * nh.z);
                    const V3 an = cross(m1, nh);
                    M27_(6, an.x * nh.x); M27_(7, an.y * nh.x); M27_(8, an.z * nh.x);
                    M27_(9, an.x * nh.y); M27_(10, an.y * nh.y); M27_(11, an.z * nh.y);
                    M27_(12, an.x * nh.z); M27_(13, an.y * nh.z); M27_(14, an.z * nh.z);
                    const V3 q0c = mk3(q[0], q[1], q[2]), q1c = mk3(q[1], q[3], q[4]), q2c = mk3(q[2], q[4], q[5]);
                    const V3 m0 = cross(nh, q0c), m1c = cross(nh, q1c), m2c = cross(nh, q2c);       // M = [n]x Q
                    const V3 r0 = cross(nh, mk3(m0.x, m1c.x, m2c.x)), r1 = cross(nh, mk3(m0.y, m1c.y, m2c.y));
                    const V3 r2 = cross(nh, mk3(m0.z, m1c.z, m2c.z));
                    M27_(15, r0.x); M27_(16, r0.y); M27_(17, r0.z); M27_(18, r1.y); M27_(19, r1.z); M27_(20, r2.z);
#pragma unroll
                    for (int k = 0; k < 6; ++k) M27_(21 + k, q[k]);
#undef M27_
                }
            }
            // a polygon without a pressure point, or of a regularized item, is not for the friction pass
            if (has && (!contributed || lane_reg)) g.poly_item[idx] = 0;
            if (g.surv != nullptr) {
                const unsigned long long sm = __ballot(has);
                if (sm) {
                    int sbase = 0;
                    if (lane == 0) sbase = atomicAdd(g.scount, __popcll(sm));
                    sbase = __builtin_amdgcn_readfirstlane(sbase);
                    if (has) g.surv[sbase + __popcll(sm & ((1ull << lane) - 1ull))] = contributed ? g.poly_cand[idx] : -1;
                }
            }
        }
        flush();
    }
}
__global__ void __launch_bounds__(64, 2) k_integ(IntegArgs g) {
    __shared__ double m27[27 * 66];         // the 27 running moments of the current run, one column per lane
    integ_body<false>(g, m27);
}
__global__ void __launch_bounds__(64, 2) k_integ_fixed(IntegArgs g) {
    __shared__ double m27[27 * 66];
    integ_body<true>(g, m27);
}

// Bristle friction pass (after k_eig): calc_spatial_bristle_force (friction.jl:171-201) + traction(::Bristle) (:32-48)
// over the polygons k_narrow kept.  One lane per kept polygon, every load is a coalesced read of consecutive slots;
// the fan / quadrature arithmetic is the one of k_narrow, so the traction points are bit-identical.
struct FricArgs {
    const ItemRec *items;
    const int *poly_item;
    const double *poly;
    const int *ccount;   // candidate count and the chunk rule (np_chunk): the kept polygons sit in per-chunk slot ranges
    int ccap, chunk_switch;
    const int *pcnt;     // kept polygons per chunk
    int pcap;
    int n_items;
    unsigned *status;
    const double *res;
    double *acc;
    FixedSink sink;      // k_fric_fixed (option "fixed_order"): the six sums of a piece's runs leave as records (k_fixed_reduce)
};
template <bool FX>
__device__ __forceinline__ void fric_body(const FricArgs &g) {
    const int lane = threadIdx.x;
    const size_t P = (size_t)g.pcap;
    int n_c = *g.ccount;
    if (n_c > g.ccap) n_c = g.ccap;
    const int C = np_chunk(n_c, g.chunk_switch);
    const int n_chunk = (n_c + C - 1) / C;
    for (int ch = blockIdx.x; ch < n_chunk; ch += gridDim.x) {
      int cnt = __builtin_amdgcn_readfirstlane(g.pcnt[ch]);
      if (cnt > C) cnt = C;
      for (int p0 = 0; p0 < cnt; p0 += 64) {
        const int idx = ch * C + p0 + lane;
        const bool active = p0 + lane < cnt;
        double sum[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) sum[k] = 0.0;
        int item = 0;
        bool contributed = false;
        unsigned pk = active ? (unsigned)g.poly_item[idx] : 0u;
        if ((pk >> 28) > 8u || ((pk >> 28) >= 3u && (pk & 0x0FFFFFFFu) >= (unsigned)g.n_items)) {
            atomicOr(g.status, kStHole);     // not a key k_narrow writes: an unwritten slot is reported, never followed
            pk = 0u;
        }
        if ((pk >> 28) >= 3u) {     // 0: a reserved slot whose pair had a polygon but no pressure point
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
            const double mu_slope = (mu_d - mu_s) / (3 * mu_s - 2 * mu_s);   // clamped_piecewise(x, 2 mu_s, 3 mu_s, mu_s, mu_d)
            const double *res = g.res + (size_t)item * kResStride;
            const V3 cop = ld3(res + kResCop), Da = ld3(res + kResDelta), Dl = ld3(res + kResDelta + 3);
            const V3 ts_c0 = ((Dl - cross(Da, cop)) + vl * tau) * (-k_bar), ts_e = (Da + w * tau) * (-k_bar);
            V3 v2 = mk3(o[(10 + 3 * (n - 1)) * P], o[(11 + 3 * (n - 1)) * P], o[(12 + 3 * (n - 1)) * P]);
            V3 vn = mk3(o[10 * P], o[11 * P], o[12 * P]);
            PointParams pp;
            pp.w = w; pp.vl = vl; pp.chi = chi; pp.Ebar = Ebar; pp.er0 = er0; pp.er1 = er1; pp.er2 = er2; pp.er3 = er3; pp.nq = nq;
            for (int k = 0; k < n; ++k) {
                const V3 v1 = v2;
                v2 = vn;
                // the next vertex is fetched while this triangle's points are evaluated (slot k + 1 <= 7 always exists)
                if (k + 1 < n) vn = mk3(o[(13 + 3 * k) * P], o[(14 + 3 * k) * P], o[(15 + 3 * k) * P]);
                // the traction points of k_narrow / k_integ, bit for bit (fan_triangle_points, pfc_kernels.h)
                if (fan_triangle_points(pp, v1, v2, cen, nh, [&](const V3 &r, const V3 &, double p, double dA) {
                    const double p_dA = p * dA;
                    const V3 x = r - cop;
                    // T̄s = -k̄ (Δ_lin + Δ_ang x (r - cop) + τ (v + ω x r)) (friction.jl:186-190) = c0 + e x r with the
                    // per-item constants c0 = -k̄ (Δ_lin - Δ_ang x cop + τ v), e = -k̄ (Δ_ang + τ ω): one cross product
                    // per point instead of two (rounding differs in the last bits; T̄s only feeds the friction force)
                    V3 Ts = ts_c0 + cross_fma(ts_e, r);
                    Ts = axpy_fma(-dot_fma(Ts, nh), nh, Ts);      // vec_sub_vec_proj
                    const double m2 = dot_fma(Ts, Ts);
                    V3 T;
                    if (m2 < mu_s * mu_s) {
                        T = Ts;
                    } else {
                        // mu / |T| once (the reference divides the three components: friction.jl:44-46), the slope of the
                        // clamp once per item
                        // 1/|T̄s| by the hardware reciprocal square root and two Newton steps (error ~1e-16; m2 >= mu_s^2
                        // here, far from the denormal range), |T̄s| = m2 / |T̄s|: a third of the instructions of an
                        // IEEE sqrt followed by an IEEE division
                        double ri = __builtin_amdgcn_rsq(m2);
                        const double hm = 0.5 * m2;
                        ri = ri * __builtin_fma(-hm * ri, ri, 1.5);
                        ri = ri * __builtin_fma(-hm * ri, ri, 1.5);
                        const double mg = m2 * ri;
                        const double y = mu_s + (mg - 2 * mu_s) * mu_slope;
                        const double mu = (y > mu_s) ? mu_s : ((y < mu_d) ? mu_d : y);
                        T = Ts * (mu * ri);
                    }
                    const V3 Tc = T * p_dA;
                    const V3 ta = cross_fma(x, Tc);
                    sum[0] += ta.x; sum[1] += ta.y; sum[2] += ta.z;
                    sum[3] += Tc.x; sum[4] += Tc.y; sum[5] += Tc.z;
                }))
                    contributed = true;
            }
        }
        accumulate_items<6, FX>(g.acc, item, FX ? (active && (pk >> 28) >= 3u) : active, contributed, sum, kAccFric, kAccStride, &g.sink, (ch * C + p0) >> 6);
      }
    }
}
__global__ void __launch_bounds__(64, 3) k_fric(FricArgs g) { fric_body<false>(g); }
__global__ void __launch_bounds__(64, 3) k_fric_fixed(FricArgs g) { fric_body<true>(g); }

