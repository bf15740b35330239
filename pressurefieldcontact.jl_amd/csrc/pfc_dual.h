// pfc_dual.h — the hot path on ForwardDiff.Dual numbers (value + partials): what calcXd! runs when Radau builds its
// Jacobian (src/mechanism_scenario.jl:187, src/radau/radau_functions.jl:2-40).  Included by pfc_hip.hip.
//
// Layout of the work.  The broadphase does not depend on partials (src/contact_algorithms_non_friction.jl:95), so
// the candidate list of the value evaluation is reused.  A Dual with N partials is carried as N independent
// (value, one partial) pairs: the work item is (candidate, seed direction) and one lane evaluates one direction.
// A wave takes 64 / n_dir consecutive candidates; its lanes are direction-major (lane = dir * cpw + c), so the lanes
// of one (item, direction) are contiguous for the segmented reductions while the n_dir lanes of one candidate gather
// the same mesh records (one fetch per wave instruction).  Per-lane state is 2x the value kernel's, so the polygon
// ring stays in LDS (32 KiB per wave) and the occupancy does not collapse as it would with 7-wide Duals in one lane.
//
// All branches compare values (ForwardDiff's comparison operators do), and the value part of every operation is the
// same instruction sequence as the value kernel's, so both take the same branches.
//
// Bristle model: three passes over the candidates, as in the reference (normal_wrench_cop; then
// calc_patch_spatial_stiffness! about the cop; then calc_spatial_bristle_force), src/contact_algorithms_friction.jl
// :119-201.  The partials of K̄^{-1/2} are the Frechet derivative of the matrix function on the eigen-basis
// (Daleckii-Krein); the reference differentiates through GenericLinearAlgebra's eigen-solver instead (not vendored),
// which agrees wherever the unclamped eigenvalues are distinct ("Dual path").
#pragma once

struct Du { double v, d; };
__device__ __forceinline__ Du du(double v) { return Du{v, 0.0}; }
__device__ __forceinline__ Du du(double v, double d) { return Du{v, d}; }
__device__ __forceinline__ Du operator+(Du a, Du b) { return Du{a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Du operator-(Du a, Du b) { return Du{a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Du operator-(Du a) { return Du{-a.v, -a.d}; }
// values: the reference's operations, unfused (they take the same branches as the value pass); partials: fused
// multiply-adds (compared with the Dual oracle by tolerance; these kernels are bound by Dual arithmetic)
__device__ __forceinline__ Du operator*(Du a, Du b) { return Du{a.v * b.v, __builtin_fma(a.v, b.d, a.d * b.v)}; }
__device__ __forceinline__ Du operator*(double a, Du b) { return Du{a * b.v, a * b.d}; }
__device__ __forceinline__ Du operator*(Du a, double b) { return Du{a.v * b, a.d * b}; }
__device__ __forceinline__ Du operator/(Du a, Du b) {
    const double q = a.v / b.v;
    return Du{q, __builtin_fma(-q, b.d, a.d) / b.v};
}

// Ring accessor of the shared clip (pfc_clip.h) for a polygon of (value, partial) pairs: values in one LDS ring (column
// vcol of VS columns; shared by the directions of a candidate in k_narrow_dual), partials in another (column dcol of DS).
template <int VS_STATIC = 0>
struct RingDu {
    typedef Du scalar;
    double *pv, *pd;
    int vs, vcol, ds, dcol, rbase;
    __device__ __forceinline__ int iv(int k, int c) const { return ((((rbase + k) & 7) * 4 + c) * (VS_STATIC ? VS_STATIC : vs)) + vcol; }
    __device__ __forceinline__ int id(int k, int c) const { return ((((rbase + k) & 7) * 4 + c) * ds) + dcol; }
    __device__ __forceinline__ double val(int k, int c) const { return pv[iv(k, c)]; }
    __device__ __forceinline__ Du get(int k, int c) const { return Du{pv[iv(k, c)], pd[id(k, c)]}; }
    __device__ __forceinline__ void set(int k, int c, Du x) const { pv[iv(k, c)] = x.v; pd[id(k, c)] = x.d; }
    __device__ __forceinline__ void move(int src, int dst, int c) const {
        const double tv = pv[iv(src, c)], td = pd[id(src, c)];
        pv[iv(dst, c)] = tv; pd[id(dst, c)] = td;
    }
    __device__ __forceinline__ void rotate(int st) { rbase = (rbase + st) & 7; }
};
__device__ __forceinline__ Du operator/(Du a, double b) { return Du{a.v / b, a.d / b}; }
__device__ __forceinline__ Du dsqrt(Du a) {
    const double s = __builtin_sqrt(a.v);
    return Du{s, a.d / (2.0 * s)};
}
// muladd(a, x, c) with a constant a
__device__ __forceinline__ Du dfma(double a, Du x, Du c) { return Du{__builtin_fma(a, x.v, c.v), __builtin_fma(a, x.d, c.d)}; }
__device__ __forceinline__ Du dfma(Du a, Du x, Du c) {
    return Du{__builtin_fma(a.v, x.v, c.v), __builtin_fma(a.v, x.d, __builtin_fma(a.d, x.v, c.d))};
}
__device__ __forceinline__ void operator+=(Du &a, Du b) { a.v += b.v; a.d += b.d; }

struct Du3 { Du x, y, z; };
__device__ __forceinline__ Du3 dmk(Du x, Du y, Du z) { return Du3{x, y, z}; }
__device__ __forceinline__ Du3 operator+(Du3 a, Du3 b) { return Du3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ Du3 operator-(Du3 a, Du3 b) { return Du3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ Du3 operator*(Du3 a, Du s) { return Du3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ Du3 operator*(Du3 a, double s) { return Du3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ Du3 operator/(Du3 a, Du s) { return Du3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ Du3 operator/(Du3 a, double s) { return Du3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ Du ddot(Du3 a, Du3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ Du3 dcross(Du3 a, Du3 b) {
    return Du3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ Du3 dnormalize(Du3 a) {
    const Du s = du(1.0) / dsqrt(ddot(a, a));
    return Du3{s * a.x, s * a.y, s * a.z};
}
__device__ __forceinline__ Du dtriangle_area(Du3 a, Du3 b, Du3 c, Du3 n) { return ddot(n, dcross(b - a, c - b) * 0.5); }
__device__ __forceinline__ Du3 dvec_sub_vec_proj(Du3 v, Du3 n) {
    const Du t = -ddot(v, n);
    return Du3{dfma(t, n.x, v.x), dfma(t, n.y, v.y), dfma(t, n.z, v.z)};
}
// mu(|T|) / |T| for a sliding point, |T|^2 = m2 given: the friction coefficient ramp of calc_clamped_piecewise
// (friction.jl:2-10) over the norm, with 1/|T| from the hardware reciprocal square root and two Newton steps (as in
// k_fric) -- no IEEE sqrt and no division; the partials follow from d|T| = dm2 / (2 |T|).
__device__ __forceinline__ Du dmu_over_norm(Du m2, double x1, double x2, double y1, double y2) {
    double ri = __builtin_amdgcn_rsq(m2.v);
    const double hm = 0.5 * m2.v;
    ri = ri * __builtin_fma(-hm * ri, ri, 1.5);
    ri = ri * __builtin_fma(-hm * ri, ri, 1.5);
    const Du mg = Du{m2.v * ri, (0.5 * m2.d) * ri};
    const double k = (y2 - y1) / (x2 - x1);
    const Du y = du(y1) + (mg - du(x1)) * k;
    const Du mu = (y.v > y1) ? du(y1) : ((y.v < y2) ? du(y2) : y);
    const double q = mu.v * ri;
    return Du{q, (mu.d - q * mg.d) * ri};
}
__device__ __forceinline__ Du dclamped_piecewise(Du x, double x1, double x2, double y1, double y2) {
    const double k = (y2 - y1) / (x2 - x1);
    const Du y = du(y1) + (x - du(x1)) * k;
    return (y.v > y1) ? du(y1) : ((y.v < y2) ? du(y2) : y);
}

// accumulator slots per (item, direction): values then partials of each group
constexpr int kDaA = 0;     // pass A: 10 Duals = wrench 6 (regularized total / bristle normal), sum w, sum w r (3)
constexpr int kDaB = 20;    // pass B: 21 Duals = K11 (xx xy xz yy yz zz), K12 (9, column-major), K22 (6) about the cop
constexpr int kDaC = 62;    // pass C: 6 Duals = friction wrench about the cop
constexpr int kDaStride = 76;
// derived per (item, direction): cop 3, Delta 6, Sinv 6, Kis 36 (values then partials each)
constexpr int kDrCop = 0, kDrDelta = 6, kDrSinv = 18, kDrKis = 30, kDrStride = 102;

struct DualArgs {
    const ItemRec *items;
    const WorkRec *cand;
    const int *ccount;
    int ccap;
    const int *surv;         // candidate indices of contributing pairs (from the value pass)
    const int *scount;
    int n_items, n_dir;
    const double *d_pose;    // (item, dir) x 24: dR21 9, dt21 3, dR12 9, dt12 3
    const double *d_twist;   // (item, dir) x 6
    const double *d_s;       // (item, dir) x 6
    const int *icnt;
    double *dacc;
    double *dres;
    double *d_wrench;        // OUT (item, dir) x 6
    double *d_sdot;          // OUT (item, dir) x 6
    unsigned *status;
    // Dual polygons of contributing bristle (pair, direction) lanes, kept by pass A for passes B and C:
    // SoA [field][slot], 64 fields: n̂ (3 values, 3 partials), centroid (3, 3), ϵ_r² (4), 8 vertices x (3, 3)
    double *dpoly;
    int2 *dpoly_key;         // (item * n_dir + dir, n_poly)
    long long dpcap;
    int tri_split;           // k_dual_poly: 1 = one lane per (kept polygon, fan triangle) instead of one per polygon (small scenes)
    // Fold of pass B into pass A (batched value pass only): the value pass's result rows (kRes* layout; the cop VALUE c0 of a
    // bristle item in contact).  Non-null: k_narrow_dual<.., FOLD> also forms the 21 stiffness sums, about the fixed point
    // c0 instead of the Dual cop (not known before pass A ends), and k_dual_eig moves them to the Dual cop by the
    // parallel-axis rule -- the shift d = cop - c0 has a rounding-sized value and the cop's partials, nothing cancels.
    // k_dual_poly<1> and its read of every kept polygon are not launched.  Null (hand-over from the fused small-scene
    // kernel, which keeps its cop in LDS): three passes as before.
    const double *vres;
    // Option "fixed_order" (or PFC_DUAL_VALUE_K=1): the value pass's result rows again (cop, K, the eigen-decomposition of K̄).
    // Non-null: k_dual_eig takes the VALUE of the cop and of
    // K from them -- the same numbers for every direction and every chunk of a Jacobian (src/radau/radau_functions.jl:2-14), summed
    // in a fixed order by the value pass -- and only the partials from the Dual sums, whose own value parts are the same quantities
    // summed in another, run-dependent order.  Where decompose_K! clamps an eigenvalue that is zero up to rounding (a flat patch,
    // friction.jl:92) the last bits of K decide the branch: this way they decide it once per item and evaluation.
    const double *vres_k;
    int stored_v;            // (with vres_k) 1: also the value pass's eigenvectors / eigenvalues / scaling; 0: k_dual_eig iterates itself (A/B)
    // ... and the sums that feed the eigen-decomposition -- pass A's (cop) and pass B's (patch stiffness) -- leave the passes as
    // records (FixedSink, pfc_np.h) that k_fixed_reduce adds per key in the order of the waves' positions in the contributing-pair
    // list, which pfc_sort.hip has sorted: the partials of K are then the same numbers in every evaluation of the same inputs
    // too.  Pass C's six friction sums go the same way (sink_c): nothing amplifies their last bits, but with them every output of
    // the evaluation is the same bit pattern in every run.
    FixedSink sink_a, sink_b, sink_c;
};
constexpr int kDpFields = 64;

// Zero seeds.  Every output partial of a key (item, direction) is LINEAR in that key's 36 seed components (d_pose 24,
// d_twist 6, d_s 6), so a key whose seeds are all zero has zero partials, whatever the values.  A chunk of a Radau
// Jacobian seeds N_chunk state variables (src/radau/radau_functions.jl:2-14) and a contact instruction only depends on
// the states of its two bodies (and its own bristle state), so for a scene of many bodies nearly all keys of nearly all
// chunks are such zeros: the passes skip them (no gather, no clip, no kept polygon, no eigen-derivative) and
// k_dual_final writes the zeros.  NaN != 0, so a non-finite seed is still evaluated.  The three kernels that decide
// (k_narrow_dual per lane, k_dual_eig per wave, k_dual_final per thread) read the same 36 numbers: one verdict.
__device__ __forceinline__ bool seed_nonzero(const DualArgs &g, int key) {
    const double *p = g.d_pose + (size_t)key * 24, *t = g.d_twist + (size_t)key * 6, *s = g.d_s + (size_t)key * 6;
    bool nz = false;
#pragma unroll
    for (int k = 0; k < 24; ++k) nz |= p[k] != 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) nz |= (t[k] != 0.0) | (s[k] != 0.0);
    return nz;
}
// the same verdict formed by a wave: lane k < 36 looks at one component
__device__ __forceinline__ bool seed_nonzero_wave(const DualArgs &g, int key, int lane) {
    double x = 0.0;
    if (lane < 24) x = g.d_pose[(size_t)key * 24 + lane];
    else if (lane < 30) x = g.d_twist[(size_t)key * 6 + (lane - 24)];
    else if (lane < 36) x = g.d_s[(size_t)key * 6 + (lane - 30)];
    return __ballot(x != 0.0) != 0ull;
}

// fan quadrature of one Dual polygon (integrate_over_polygon_patch!, non_friction.jl:217-265) with the pass-specific
// integrand: MODE 0 normal wrench + regularized friction + cop sums, 1 patch stiffness about the cop, 2 bristle force
// [k0, k1): the fan triangles (v_{k-1}, v_k, centroid) this lane integrates: all of them (0, n) with one lane per
// polygon; one of them when the fused small-scene kernel deals the triangles out one per thread.
// the 21 patch-stiffness entries of one polygon (n̂ constant) from W = sum w, m1 = sum w x, Q = sum w x x' (x = r - reference point):
//   K22 = W (I - n n'),  K12 = [m1]x - (m1 x n) n',  K11 = -(Q - tr(Q) I) - [n]x Q [n]x'   (Q symmetric)
__device__ __forceinline__ void dual_stiffness_entries(Du kW, const Du *km, const Du *kQ, Du3 nh, Du *sum) {
    const Du3 m1 = dmk(km[0], km[1], km[2]);
    const Du3 mn = dcross(m1, nh);
    const Du tr = (kQ[0] + kQ[3]) + kQ[5];
    const Du3 c0 = dmk(kQ[0], kQ[1], kQ[2]), c1 = dmk(kQ[1], kQ[3], kQ[4]), c2 = dmk(kQ[2], kQ[4], kQ[5]);
    const Du3 a0 = dcross(nh, c0), a1 = dcross(nh, c1), a2 = dcross(nh, c2);                 // M = [n]x Q (columns)
    // S = M [n]x': row i of S = n x (row i of M)
    const Du3 r0 = dcross(nh, dmk(a0.x, a1.x, a2.x)), r1 = dcross(nh, dmk(a0.y, a1.y, a2.y));
    const Du3 r2 = dcross(nh, dmk(a0.z, a1.z, a2.z));
    sum[0] += -((kQ[0] - tr) + r0.x); sum[1] += -(kQ[1] + r0.y); sum[2] += -(kQ[2] + r0.z);
    sum[3] += -((kQ[3] - tr) + r1.y); sum[4] += -(kQ[4] + r1.z); sum[5] += -((kQ[5] - tr) + r2.z);
    sum[6] += du(0.0) - mn.x * nh.x; sum[7] += m1.z - mn.y * nh.x; sum[8] += -m1.y - mn.z * nh.x;
    sum[9] += -m1.z - mn.x * nh.y; sum[10] += du(0.0) - mn.y * nh.y; sum[11] += m1.x - mn.z * nh.y;
    sum[12] += m1.y - mn.x * nh.z; sum[13] += -m1.x - mn.y * nh.z; sum[14] += du(0.0) - mn.z * nh.z;
    sum[15] += kW * (du(1.0) - nh.x * nh.x); sum[16] += kW * (du(0.0) - nh.x * nh.y);
    sum[17] += kW * (du(0.0) - nh.x * nh.z); sum[18] += kW * (du(1.0) - nh.y * nh.y);
    sum[19] += kW * (du(0.0) - nh.y * nh.z); sum[20] += kW * (du(1.0) - nh.z * nh.z);
}

// FOLD (MODE 0, bristle lanes): also m1 = sum w x and Q = sum w x x' about the FIXED point c0 (fold[0..2], fold[3..8]): what
// pass B would accumulate, had it the cop c0 (DualArgs::vres).
template <int MODE, bool FOLD = false, class VF>
__device__ __forceinline__ void dual_integrate(VF vert, int n, Du3 nh, Du3 cen, const double *er, const ItemRec *it,
                                       const double *dt, bool reg, Du3 cop, Du3 Da, Du3 Dl, Du *sum, int &n_trac,
                                       int k0, int k1, const double *c0 = nullptr, Du *fold = nullptr) {
    const double er0 = er[0], er1 = er[1], er2 = er[2], er3 = er[3];
    const Du3 w = dmk(du(it->w[0], dt[0]), du(it->w[1], dt[1]), du(it->w[2], dt[2]));
    const Du3 vl = dmk(du(it->v[0], dt[3]), du(it->v[1], dt[4]), du(it->v[2], dt[5]));
    const double chi = it->chi, Ebar = it->Ebar;
    const double v_c = it->v_c, mu_s = it->mu_s, mu_d = it->mu_d;
    const double tau = it->tau, k_bar = it->k_bar;
    const int nq = it->nq;
    Du kW = du(0.0), km[3] = {du(0.0), du(0.0), du(0.0)};   // MODE 1 accumulators
    Du kQ[6] = {du(0.0), du(0.0), du(0.0), du(0.0), du(0.0), du(0.0)};
    Du3 v2 = vert(k0 == 0 ? n - 1 : k0 - 1);
    for (int k = k0; k < k1; ++k) {
        const Du3 v1 = v2;
        v2 = vert(k);
        const Du area = dtriangle_area(v1, v2, cen, nh);
        if (!(0.0 < area.v)) continue;
        for (int q = 0; q < nq; ++q) {
            double q0, q1, q2, qw;
            if (nq == 1) {
                q0 = q1 = q2 = 0.33333333333333331483; qw = 1.0;
            } else {
                const double qa = 0.16666666666666674068, qb = 0.66666666666666651864;
                q0 = (q == 1) ? qb : qa; q1 = (q == 0) ? qb : qa; q2 = (q == 2) ? qb : qa;
                qw = 0.33333333333333331483;
            }
            const Du3 r = dmk((v1.x * q0 + v2.x * q1) + cen.x * q2, (v1.y * q0 + v2.y * q1) + cen.y * q2,
                              (v1.z * q0 + v2.z * q1) + cen.z * q2);
            Du eq = dfma(er0, r.x, du(er3));
            eq = dfma(er1, r.y, eq);
            eq = dfma(er2, r.z, eq);
            const Du3 rdot = vl + dcross(w, r);
            const Du ee = -((er0 * rdot.x + er1 * rdot.y) + er2 * rdot.z);
            const Du darg = du(1.0) + chi * ee;
            const Du damp = (darg.v > 0.0) ? darg : du(0.0);   // max(0.0, .)
            const Du p = (eq * Ebar) * damp;
            const Du dA = qw * area;
            if (!(0.0 < p.v)) continue;
            ++n_trac;
            const Du p_dA = p * dA;
            if constexpr (MODE == 2) {
                // calc_spatial_bristle_force (friction.jl:171-201) + traction(::Bristle) (:32-48)
                const Du3 x = r - cop;
                const Du3 del = Dl + dcross(Da, x);
                Du3 Ts = (del + rdot * tau) * (-k_bar);
                Ts = dvec_sub_vec_proj(Ts, nh);
                const Du m2 = ddot(Ts, Ts);
                Du3 T;
                if (m2.v < mu_s * mu_s) {
                    T = Ts;
                } else {
                    T = Ts * dmu_over_norm(m2, 2 * mu_s, 3 * mu_s, mu_s, mu_d);
                }
                const Du3 Tc = T * p_dA;
                const Du3 ta = dcross(x, Tc);
                sum[0] += ta.x; sum[1] += ta.y; sum[2] += ta.z;
                sum[3] += Tc.x; sum[4] += Tc.y; sum[5] += Tc.z;
            } else if constexpr (MODE == 1) {
                // calc_patch_spatial_stiffness! (friction.jl:147-169) with x = r - cop.  n̂ is constant over the polygon, so
                // per point only W = sum w, m1 = sum w x and Q = sum w x x' are accumulated; the 21 entries follow after
                // the loop (below).
                const Du3 x = r - cop;
                const Du wx = p_dA * x.x, wy = p_dA * x.y, wz = p_dA * x.z;
                kW += p_dA;
                km[0] += wx; km[1] += wy; km[2] += wz;
                kQ[0] += wx * x.x; kQ[1] += wx * x.y; kQ[2] += wx * x.z;
                kQ[3] += wy * x.y; kQ[4] += wy * x.z; kQ[5] += wz * x.z;
            } else {
            if (reg) {
                // yes_contact!(::Regularized) (friction.jl:50-72)
                const Du3 vt = dvec_sub_vec_proj(rdot, nh);
                const Du m2 = ddot(vt, vt);
                Du3 T;
                if (m2.v < v_c * v_c) {
                    T = (vt * (-mu_s)) / v_c;
                } else {
                    // (pass A sits at the 256-register limit: the division-free form of pass C costs it a wave per SIMD)
                    const Du mg = dsqrt(m2);
                    const Du mu = dclamped_piecewise(mg, 2 * v_c, 3 * v_c, mu_s, mu_d);
                    T = (vt * (-mu)) / mg;
                }
                const Du3 tk = nh * p_dA + T * p_dA;
                const Du3 ta = dcross(r, tk);
                sum[0] += ta.x; sum[1] += ta.y; sum[2] += ta.z;
                sum[3] += tk.x; sum[4] += tk.y; sum[5] += tk.z;
            } else {
                // normal_wrench_cop (normal.jl:17-34): the traction is n̂ w with n̂ constant over the polygon; force and
                // torque follow from W and sum w r after the loop
                sum[6] += p_dA;
                sum[7] += p_dA * r.x; sum[8] += p_dA * r.y; sum[9] += p_dA * r.z;
                if constexpr (FOLD) {
                    const Du3 x = dmk(r.x - du(c0[0]), r.y - du(c0[1]), r.z - du(c0[2]));
                    const Du wx = p_dA * x.x, wy = p_dA * x.y, wz = p_dA * x.z;
                    fold[0] += wx; fold[1] += wy; fold[2] += wz;
                    fold[3] += wx * x.x; fold[4] += wx * x.y; fold[5] += wx * x.z;
                    fold[6] += wy * x.y; fold[7] += wy * x.z; fold[8] += wz * x.z;
                }
            }
            }
        }
    }
    if constexpr (MODE == 0) {
        if (!reg) {
            const Du3 ta = dcross(dmk(sum[7], sum[8], sum[9]), nh);
            sum[0] = ta.x; sum[1] = ta.y; sum[2] = ta.z;
            sum[3] = nh.x * sum[6]; sum[4] = nh.y * sum[6]; sum[5] = nh.z * sum[6];
        }
    }
    if constexpr (MODE == 1) dual_stiffness_entries(kW, km, kQ, nh, sum);
}

// Polygon rings of pass A in LDS.  The VALUE ring is shared by the n_dir lanes of a candidate: they hold the same
// values and take the same branches (Dual comparisons look at values only), so every ring write is the same number to
// the same address from each of them, and a read is an LDS broadcast.  With at most 16 candidates per wave (n_dir >= 4)
// the value ring is 4 KiB instead of 16 (column stride 16: the 4 s + c part of an address is a multiple of 16 doubles,
// so distinct candidates sit in distinct banks); 20 KiB per wave let eight waves share a CU instead of five.
#define PV(k, c) pv[((((rbase) + (k)) & 7) * 4 + (c)) * pvs + pvi]
#define PD(k, c) pd[((((rbase) + (k)) & 7) * 4 + (c)) * 64 + lane]
__host__ __device__ inline int dual_pv_stride(int n_dir) { return 64 / n_dir <= 16 ? 16 : 64; }
__host__ inline size_t dual_lds_bytes(int n_dir) { return sizeof(double) * (size_t)(8 * 4 * 64 + 8 * 4 * dual_pv_stride(n_dir)); }

// Per-key sums of a wave of the Dual passes.  Lane l of a wave of k_narrow_dual / k_dual_poly holds candidate l % cpw in
// direction l / cpw (cpw = 64 / n_dir), so a key's lanes are the cpw consecutive lanes of its direction -- if the wave's
// contributing candidates all belong to ONE item, which is nearly always so (an item's pairs sit in runs).  Then the N sums go
// through LDS: every lane stores its values as a column, lane (d, j) adds the cpw entries of direction d in row j and issues
// the atomic: ~6 N instructions per wave instead of the ~30 N of N segmented scans (accumulate_items, the fallback for a
// wave that straddles items).  buf: 21 x 65 doubles (row stride 65: the row walk of consecutive lanes is conflict-free).
constexpr int kDualRedRows = 21, kDualRedStride = 65;
template <int N, bool FX = false>
__device__ __forceinline__ void dual_accumulate(double *buf, double *acc, int key, bool listed, bool any, const double *v, int n0,
                                                int n_dir, int lane, const FixedSink *fx = nullptr, int order = 0) {
    const unsigned long long am = __ballot(any);
    if (am == 0) return;
    const int item = key >= 0 ? key / n_dir : -1;
    const int item0 = __builtin_amdgcn_readlane(item, __builtin_ctzll(am));
    if (__ballot(any && item != item0) != 0) {      // uniform
        accumulate_items<N, FX>(acc, key, listed, any, v, n0, kDaStride, fx, order);
        return;
    }
    const int cpw = 64 / n_dir;
    constexpr int R = (N + kDualRedRows - 1) / kDualRedRows, NB = (N + R - 1) / R;
    int slot0 = 0;
    bool room = true;
    if constexpr (FX) {      // one record per direction of the wave's item (see accumulate_items)
        // the wave's item in its n_dir directions: the direct slots of the wave's position (per_pos = n_dir) -- no counter
        slot0 = fx->direct_base + order * n_dir;
        room = order < fx->n_pos && fx->per_pos == n_dir;
    }
    wave_lds_sync();      // (the buffer may be the polygon ring the lanes have just read)
#pragma unroll
    for (int rd = 0; rd < R; ++rd) {
        const int base = rd * NB;
        const int nb = (N - base < NB) ? N - base : NB;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (j < nb) buf[j * kDualRedStride + lane] = any ? v[base + j] : 0.0;
        wave_lds_sync();
        for (int t = lane; t < n_dir * nb; t += 64) {
            const int d = t / nb, j = t - d * nb;
            const double *row = buf + j * kDualRedStride + d * cpw;
            double x = 0.0;
            for (int c = 0; c < cpw; ++c) x += row[c];
            if constexpr (FX) {
                if (room) fx->rec[(size_t)(slot0 + d) * kSinkStride + kSinkHdr + base + j] = x;
            } else {
            if (x != 0.0) unsafeAtomicAdd(&acc[(size_t)(item0 * n_dir + d) * kDaStride + n0 + base + j], x);
            }
        }
        wave_lds_sync();
    }
    if constexpr (FX) {
        if (room && lane < n_dir) {
            double *r = fx->rec + (size_t)(slot0 + lane) * kSinkStride;
            const int k = item0 * n_dir + lane;
            r[0] = (double)k; r[1] = (double)order; r[2] = (double)atomicExch(&fx->head[k], slot0 + lane);
        }
    }
}

// Selection of the contributing pairs a chunk has work for (scenes of many items): k_dual_flags marks the items with a
// non-zero seed in any direction (one wave per item), k_dual_select copies the contributing pairs of marked items into a
// list of their own (compacted 2 048 entries at a time, order kept, so an item's pairs stay in runs).  The
// passes then walk that list instead of all contributing pairs: the cost of a chunk follows the instructions its state
// variables touch (C5, seeds of one body: 63 of 2 016 instructions).  Directions of a marked item whose own seeds are
// zero are still skipped per key (seed_nonzero).
__global__ void __launch_bounds__(64) k_dual_flags(DualArgs g, int *flag, int *selcount) {
    const int item = blockIdx.x, lane = threadIdx.x;
    if (item == 0 && lane == 0) *selcount = 0;      // k_dual_select runs behind this kernel on the stream
    if (item >= g.n_items) return;
    const size_t k0 = (size_t)item * g.n_dir;
    const double *p = g.d_pose + k0 * 24, *t = g.d_twist + k0 * 6, *s = g.d_s + k0 * 6;
    bool nz = false;
    for (int k = lane; k < g.n_dir * 24; k += 64) nz |= p[k] != 0.0;
    for (int k = lane; k < g.n_dir * 6; k += 64) nz |= (t[k] != 0.0) | (s[k] != 0.0);
    const bool any = __ballot(nz) != 0ull;
    if (lane == 0) flag[item] = any ? 1 : 0;
}
constexpr int kSelRounds = 8;      // a workgroup of k_dual_select compacts 256 x 8 consecutive list entries at a time
__global__ void __launch_bounds__(256) k_dual_select(DualArgs g, const int *flag, int *sel, int *selcount) {
    __shared__ int s_cnt[kSelRounds][4], s_base;
    int n_c = *g.scount;
    if (n_c > g.ccap) n_c = g.ccap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kChunk = 256 * kSelRounds;
    // One returning atomic per 2 048 entries, and the entries of a chunk keep their order: the list stays in runs of an
    // item's pairs in traversal order (a wave-granular append scattered 64-entry pieces over the list and cost the passes
    // of a densely seeded 2 048-pose batch 7 %).
    for (int c0 = (int)blockIdx.x * kChunk; c0 < n_c; c0 += (int)gridDim.x * kChunk) {      // uniform over the workgroup
        int ci[kSelRounds];
        unsigned keepm = 0;
#pragma unroll
        for (int r = 0; r < kSelRounds; ++r) {
            const int idx = c0 + r * 256 + tid;
            ci[r] = idx < n_c ? g.surv[idx] : -1;
        }
#pragma unroll
        for (int r = 0; r < kSelRounds; ++r) {
            bool keep = false;
            if (ci[r] >= 0 && ci[r] < g.ccap) {
                const int item = g.cand[ci[r]].item;
                keep = (unsigned)item < (unsigned)g.n_items && flag[item] != 0;
            }
            const unsigned long long km = __ballot(keep);
            if (keep) keepm |= 1u << r;
            // position among the kept entries of this wave and round, kept in the upper bits of ci's companion
            ci[r] = keep ? ci[r] : -1;
            if (lane == 0) s_cnt[r][wave] = __popcll(km);
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int r = 0; r < kSelRounds; ++r)
                for (int w = 0; w < 4; ++w) { const int c = s_cnt[r][w]; s_cnt[r][w] = tot; tot += c; }
            s_base = tot ? atomicAdd(selcount, tot) : 0;
        }
        __syncthreads();
        const int base = s_base;
#pragma unroll
        for (int r = 0; r < kSelRounds; ++r) {
            const bool keep = (keepm >> r) & 1u;
            const unsigned long long km = __ballot(keep);
            if (keep) sel[base + s_cnt[r][wave] + __popcll(km & ((1ull << lane) - 1ull))] = ci[r];
        }
        __syncthreads();      // s_cnt / s_base are rewritten by the next chunk
    }
}

// Pass A: gather, clip and integrate in Dual arithmetic (normal wrench, regularized friction fused, cop sums); the
// Dual polygon of every contributing bristle lane is kept for passes B and C (k_dual_poly).  TT as in k_narrow.
// PVS: the column stride of the shared value ring, dual_pv_stride(n_dir), as a compile-time constant: with a run-time stride every
// ring slot the code names -- (slot, coordinate) pairs with rbase = 0 -- became a loop-invariant address in a register of its
// own, and those registers were what the tet-tet instantiation spilled (round 3: 15 VGPRs + 64 bytes of scratch per lane).
template <bool TT, int PVS, bool FOLD = false, bool FX = false>
__global__ void __launch_bounds__(64, 2) k_narrow_dual(DualArgs g) {
    extern __shared__ double dual_lds[];     // dual_lds_bytes(n_dir): partial ring (16 KiB), then the value ring
    double *pd = dual_lds, *pv = dual_lds + 8 * 4 * 64;
    const int lane = threadIdx.x;
    int n_c = *g.scount;     // contributing pairs only: no lane gathers for a pair that will be rejected
    if (n_c > g.ccap) n_c = g.ccap;
    const int n_dir = g.n_dir;
    const int cpw = 64 / n_dir;
    const int n_group = (n_c + cpw - 1) / cpw;
    const int dir = lane / cpw, cl = lane - dir * cpw;
    constexpr int pvs = PVS;
    const int pvi = PVS == 16 ? cl : lane;
    for (int grp = blockIdx.x; grp < n_group; grp += gridDim.x) {
        const int idx = grp * cpw + cl;
        const bool active = dir < n_dir && idx < n_c;
        WorkRec cw;
        cw.item = 0; cw.a = 0; cw.b = 0; cw.pad = 0;
        int ci = -1;
        if (active) ci = g.surv[idx];   // -1: the pair had a polygon but did not contribute in the value pass
        if (ci >= g.ccap) { atomicOr(g.status, kStHole); ci = -1; }      // not an index the value pass writes: reported, never followed
        if (ci >= 0) cw = g.cand[ci];
        if ((unsigned)cw.item >= (unsigned)g.n_items) {                  // an unwritten (or overwritten) candidate slot
            atomicOr(g.status, kStHole);
            cw.item = 0; cw.a = 0; cw.b = 0;
            ci = -1;
        }
        const ItemRec *it = g.items + cw.item;
        const GTetRec *tp = (const GTetRec *)(it->tet + cw.b);
        const bool reg = it->model == PFC_REGULARIZED;
        const bool live = active && ci >= 0;
        const int key = live ? cw.item * n_dir + dir : -1;
        const bool work = live && g.icnt[4 * (size_t)cw.item + 3] > 0 && seed_nonzero(g, key);   // zero seeds: zero partials (above)
        int n_poly = 0, rbase = 0;
        Du3 nh = dmk(du(0.0), du(0.0), du(0.0));
        if (work) {
            const double *dp = g.d_pose + (size_t)key * 24;
            Du R21[9], t21[3];
#pragma unroll
            for (int k = 0; k < 9; ++k) R21[k] = du(it->R21[k], dp[k]);
#pragma unroll
            for (int k = 0; k < 3; ++k) t21[k] = du(it->t21[k], dp[9 + k]);
            // Each op ends with its polygon in the ring (slots 0 .. n_in - 1, rbase 0) and the verdict of the bit-exact trivial
            // reject; nothing but n_in, nh_in and that verdict lives across the two forms (with the tri-tet op's z[4][4] and the
            // tet-tet op's vertices in common variables the TT kernel needed 15 ... 75 spilled registers and scratch).
            int n_in = 0;
            Du3 nh_in = nh;
            bool reject = true;
            double Z[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) Z[k] = tp->xzr[k];
            if (!TT || it->tet1 == nullptr) {
                // tri-tet op (non_friction.jl:196-215)
                const GTriRec *tr = (const GTriRec *)(it->tri + cw.a);
                Du z[3][4];
                Du X[16];
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
                n_in = 3;
                nh_in = dmk((R21[0] * tr->n[0] + R21[3] * tr->n[1]) + R21[6] * tr->n[2],
                            (R21[1] * tr->n[0] + R21[4] * tr->n[1]) + R21[7] * tr->n[2],
                            (R21[2] * tr->n[0] + R21[5] * tr->n[1]) + R21[8] * tr->n[2]);
                bool finite = true;
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i) finite &= (__builtin_fabs(z[k][i].v) <= 1.79769313486231570815e308);
                reject = !finite;
#pragma unroll
                for (int i = 0; i < 4; ++i) reject |= (z[0][i].v <= 0.0) && (z[1][i].v <= 0.0) && (z[2][i].v <= 0.0);
                if (!reject) {
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int i = 0; i < 4; ++i) { PV(k, i) = z[k][i].v; PD(k, i) = z[k][i].d; }
                }
            } else {
                // tet-tet op (non_friction.jl:166-194)
                const GTetRec *t1 = (const GTetRec *)(it->tet1 + cw.a);
                Du plane[4];
                {
                    // x_zeta1_r2 = x_zeta1_r1 * x_r1_r2 one COLUMN at a time, each reduced to its plane coefficient at once (the same
                    // expressions, entry for entry, as the whole 4 x 4 product held in 64 registers)
                    double Z1[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) Z1[k] = t1->xzr[k];
                    double Ee1[4], Ee2[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        Ee1[j] = it->Ebar1 * ((const gdouble *)it->eps1)[4 * (size_t)cw.a + j];
                        Ee2[j] = it->Ebar * ((const gdouble *)it->eps2)[4 * (size_t)cw.b + j];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        Du Xc[4];
                        if (j < 3) {
                            const Du r0 = du(it->R12[3 * j], dp[12 + 3 * j]), r1 = du(it->R12[3 * j + 1], dp[12 + 3 * j + 1]),
                                     r2 = du(it->R12[3 * j + 2], dp[12 + 3 * j + 2]);
#pragma unroll
                            for (int i = 0; i < 4; ++i) Xc[i] = (Z1[i] * r0 + Z1[i + 4] * r1) + Z1[i + 8] * r2;
                        } else {
                            const Du r0 = du(it->t12[0], dp[21]), r1 = du(it->t12[1], dp[22]), r2 = du(it->t12[2], dp[23]);
#pragma unroll
                            for (int i = 0; i < 4; ++i) Xc[i] = ((Z1[i] * r0 + Z1[i + 4] * r1) + Z1[i + 8] * r2) + du(Z1[i + 12]);
                        }
                        const Du p1 = ((Ee1[0] * Xc[0] + Ee1[1] * Xc[1]) + Ee1[2] * Xc[2]) + Ee1[3] * Xc[3];
                        const double p2 = ((Ee2[0] * Z[4 * j] + Ee2[1] * Z[4 * j + 1]) + Ee2[2] * Z[4 * j + 2]) + Ee2[3] * Z[4 * j + 3];
                        plane[j] = du(p2) - p1;
                    }
                }
                // The four vertices of tet 1 in frame r2 and their projections wait in the upper half of the lane's polygon ring
                // (slots 4..7: coordinates 0..2 the vertex, 3 its projection) instead of in 64 registers: with them the kernel needed
                // 256 VGPRs + 15 spilled ones and 64 bytes of scratch per lane (round 3).  rbase is 0 here; the clip below starts
                // from slots 0..3.  (The value ring is shared by the directions of a candidate: they write the same numbers.)
                int n_neg = 0, n_pos = 0;
                unsigned posm = 0, negm = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double vx = t1->xrz[3 * j], vy = t1->xrz[3 * j + 1], vz = t1->xrz[3 * j + 2];
                    const Du3 Pj = dmk(((R21[0] * vx + R21[3] * vy) + R21[6] * vz) + t21[0],
                                       ((R21[1] * vx + R21[4] * vy) + R21[7] * vz) + t21[1],
                                       ((R21[2] * vx + R21[5] * vy) + R21[8] * vz) + t21[2]);
                    const Du pj = ((plane[0] * Pj.x + plane[1] * Pj.y) + plane[2] * Pj.z) + plane[3];
                    PV(4 + j, 0) = Pj.x.v; PV(4 + j, 1) = Pj.y.v; PV(4 + j, 2) = Pj.z.v; PV(4 + j, 3) = pj.v;
                    PD(4 + j, 0) = Pj.x.d; PD(4 + j, 1) = Pj.y.d; PD(4 + j, 2) = Pj.z.d; PD(4 + j, 3) = pj.d;
                    if (pj.v < 0.0) { ++n_neg; negm |= 1u << j; }
                    if (0.0 < pj.v) { ++n_pos; posm |= 1u << j; }
                }
                Du3 q[4];
                q[0] = q[1] = q[2] = q[3] = dmk(du(0.0), du(0.0), du(0.0));
                int n_q = 0;
#define PJ_(j) dmk(du(PV(4 + (j), 0), PD(4 + (j), 0)), du(PV(4 + (j), 1), PD(4 + (j), 1)), du(PV(4 + (j), 2), PD(4 + (j), 2)))
#define PR_(j) du(PV(4 + (j), 3), PD(4 + (j), 3))
#define PW_(i1, i2) (PJ_(i2) * (PR_(i1) / (PR_(i1) - PR_(i2))) - PJ_(i1) * (PR_(i2) / (PR_(i1) - PR_(i2))))
                if (n_pos != 0 && n_neg != 0) {
                    int lone = -1;
                    if (n_pos == 1) lone = __builtin_ctz(posm);
                    else if (n_neg == 1) lone = __builtin_ctz(negm);
                    if (lone >= 0) {
                        Du3 a, b, c;
                        if (lone == 0) { a = PW_(1, 0); b = PW_(3, 0); c = PW_(2, 0); }
                        else if (lone == 1) { a = PW_(0, 1); b = PW_(2, 1); c = PW_(3, 1); }
                        else if (lone == 2) { a = PW_(0, 2); b = PW_(3, 2); c = PW_(1, 2); }
                        else { a = PW_(0, 3); b = PW_(1, 3); c = PW_(2, 3); }
                        const double pl = (lone == 0) ? PV(4, 3) : (lone == 1) ? PV(5, 3) : (lone == 2) ? PV(6, 3) : PV(7, 3);
                        n_q = 3;
                        if (0.0 < pl) { q[0] = a; q[1] = b; q[2] = c; } else { q[0] = c; q[1] = b; q[2] = a; }
                    } else {
                        Du3 a, b, c, d;
                        const bool p0 = (posm & 1u) != 0, p1 = (posm & 2u) != 0, p2 = (posm & 4u) != 0;
                        if (p0 == p1) { a = PW_(1, 2); b = PW_(1, 3); c = PW_(0, 3); d = PW_(0, 2); }
                        else if (p0 == p2) { a = PW_(0, 1); b = PW_(0, 3); c = PW_(2, 3); d = PW_(2, 1); }
                        else { a = PW_(0, 2); b = PW_(0, 1); c = PW_(3, 1); d = PW_(3, 2); }
                        n_q = 4;
                        if (0.0 < PV(4, 3)) { q[0] = a; q[1] = b; q[2] = c; q[3] = d; }
                        else { q[0] = d; q[1] = c; q[2] = b; q[3] = a; }
                    }
                }
#undef PW_
#undef PJ_
#undef PR_
                // (written straight into the ring, where the clip wants them: as 32 more live doubles they were what spilled)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const Du v = ((Z[i] * q[k].x + Z[i + 4] * q[k].y) + Z[i + 8] * q[k].z) + du(Z[i + 12]);
                        const Du vz = v * ((1.0e-14 < __builtin_fabs(v.v)) ? 1.0 : 0.0);   // zero_small_coordinates
                        PV(k, i) = vz.v; PD(k, i) = vz.d;
                    }
                n_in = n_q;
                nh_in = dnormalize(dmk(plane[0], plane[1], plane[2]));
                bool finite = true;
                reject = n_in < 3;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double z0 = PV(0, i), z1 = PV(1, i), z2 = PV(2, i), z3 = PV(3, i);      // the values just written
                    finite &= (__builtin_fabs(z0) <= 1.79769313486231570815e308) && (__builtin_fabs(z1) <= 1.79769313486231570815e308) &&
                              (__builtin_fabs(z2) <= 1.79769313486231570815e308) && (n_in < 4 || __builtin_fabs(z3) <= 1.79769313486231570815e308);
                    reject |= (z0 <= 0.0) && (z1 <= 0.0) && (z2 <= 0.0) && (n_in < 4 || z3 <= 0.0);
                }
                reject |= !finite;
            }
            if (!reject) {
                // clip_in_tet_coordinates (static_clip.jl:7-23,34-201), polygon ring in LDS, clipped in place
                int n = n_in;
                {
                    bool err = false;      // "Non-finite vertex likely": the value pass has reported it
                    RingDu<PVS> ring{pv, pd, pvs, pvi, 64, lane, rbase};
                    n = clip_ring_in_tet_coordinates(ring, n_in, err);     // pfc_clip.h
                    rbase = ring.rbase;
                }
                n_poly = n;
                if (n >= 3) nh = nh_in;
            }
        }
        // ---- integrate_over_polygon_patch! (non_friction.jl:217-234) ------------------------------------------------
        constexpr int NS = 10;
        Du sum[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) sum[k] = du(0.0);
        int n_trac_lane = 0;
        Du3 cen_keep = dmk(du(0.0), du(0.0), du(0.0));
        Du fold[9];      // FOLD: m1 (3), Q (6) about the value cop
#pragma unroll
        for (int k = 0; k < 9; ++k) fold[k] = du(0.0);
        if (n_poly >= 3) {
            const int n = n_poly;
            {
                double V[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) V[k] = tp->xrz[k];
                for (int k = 0; k < n; ++k) {
                    const Du z0 = du(PV(k, 0), PD(k, 0)), z1 = du(PV(k, 1), PD(k, 1)), z2 = du(PV(k, 2), PD(k, 2)),
                             z3 = du(PV(k, 3), PD(k, 3));
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const Du r = ((V[c] * z0 + V[c + 3] * z1) + V[c + 6] * z2) + V[c + 9] * z3;
                        PV(k, c) = r.v; PD(k, c) = r.d;
                    }
                }
            }
#define PVT(k) dmk(du(PV(k, 0), PD(k, 0)), du(PV(k, 1), PD(k, 1)), du(PV(k, 2), PD(k, 2)))
            Du3 cen;
            {
                const Du3 a = PVT(0);
                Du3 cc = PVT(1);
                Du cum_sum = du(0.0);
                Du3 cum_prod = dmk(du(0.0), du(0.0), du(0.0));
                for (int k = 2; k < n; ++k) {
                    const Du3 b = cc;
                    cc = PVT(k);
                    const Du ar = dtriangle_area(a, b, cc, nh);
                    cum_prod = cum_prod + (((a + b) + cc) * (1.0 / 3.0)) * ar;
                    cum_sum += ar;
                }
                cen = (cum_sum.v == 0.0) ? a : cum_prod / cum_sum;
            }
            const double er[4] = {tp->epsr[0], tp->epsr[1], tp->epsr[2], tp->epsr[3]};
            Du3 cop = dmk(du(0.0), du(0.0), du(0.0)), Da = cop, Dl = cop;
            double c0[3] = {0.0, 0.0, 0.0};
            if (FOLD && !reg) {
                const double *vr = g.vres + (size_t)cw.item * kResStride + kResCop;
                c0[0] = vr[0]; c0[1] = vr[1]; c0[2] = vr[2];
            }
            dual_integrate<0, FOLD>([&](int k) { return PVT(k); }, n, nh, cen, er, it, g.d_twist + (size_t)key * 6, reg, cop, Da,
                                    Dl, sum, n_trac_lane, 0, n, c0, fold);
            cen_keep = cen;
#undef PVT
        }
        // ---- keep the Dual polygon of contributing bristle lanes.  The work list of this pass holds contributing pairs
        // only, so nearly every lane keeps its polygon: a wave owns the 64 slots grp * 64 + lane (coalesced SoA stores)
        // and lanes without a polygon leave a marker -- no slot counter, hence no returning atomic per wave.
        {
            const bool keep = work && n_trac_lane > 0 && !reg;
            const long long slot = (long long)grp * 64 + lane;
            if (slot < g.dpcap) {
                if (!keep) g.dpoly_key[slot] = make_int2(-1, 0);
                if (keep) {     // dpcap >= 64 x groups of the contributing pairs: cannot overflow
                    const size_t P = (size_t)g.dpcap;
                    double *o = g.dpoly + slot;
                    g.dpoly_key[slot] = make_int2(key, n_poly);
                    o[0] = nh.x.v; o[P] = nh.y.v; o[2 * P] = nh.z.v; o[3 * P] = nh.x.d; o[4 * P] = nh.y.d; o[5 * P] = nh.z.d;
                    o[6 * P] = cen_keep.x.v; o[7 * P] = cen_keep.y.v; o[8 * P] = cen_keep.z.v;
                    o[9 * P] = cen_keep.x.d; o[10 * P] = cen_keep.y.d; o[11 * P] = cen_keep.z.d;
                    o[12 * P] = tp->epsr[0]; o[13 * P] = tp->epsr[1]; o[14 * P] = tp->epsr[2]; o[15 * P] = tp->epsr[3];
                    for (int k = 0; k < n_poly; ++k) {
                        double *q = o + (size_t)(16 + 6 * k) * P;
                        q[0] = PV(k, 0); q[P] = PV(k, 1); q[2 * P] = PV(k, 2);
                        q[3 * P] = PD(k, 0); q[4 * P] = PD(k, 1); q[5 * P] = PD(k, 2);
                    }
                }
            }
        }
        // ---- per (item, direction) reductions ---------------------------------------------------------------------
        double flat[2 * NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) { flat[k] = sum[k].v; flat[NS + k] = sum[k].d; }
        if constexpr (FOLD) dual_accumulate<2 * NS, FX>(pd, g.dacc, key, live, work && n_trac_lane > 0, flat, kDaA, n_dir, lane, &g.sink_a, grp);
        else accumulate_items<2 * NS, FX>(g.dacc, key, live, work && n_trac_lane > 0, flat, kDaA, kDaStride, &g.sink_a, grp);
        if constexpr (FOLD) {
            // pass B's sums of this polygon, about c0 (a wave without a contributing bristle lane adds nothing)
            const bool cb = work && n_trac_lane > 0 && !reg;
            if (__ballot(cb) != 0) {
                Du k21[21];
#pragma unroll
                for (int k = 0; k < 21; ++k) k21[k] = du(0.0);
                dual_stiffness_entries(sum[6], fold, fold + 3, nh, k21);
                double flat_b[42];
#pragma unroll
                for (int k = 0; k < 21; ++k) { flat_b[k] = k21[k].v; flat_b[21 + k] = k21[k].d; }
                dual_accumulate<42, FX>(pd, g.dacc, key, live, cb, flat_b, kDaB, n_dir, lane, &g.sink_b, grp);
            }
        }
    }
}
#undef PV
#undef PD

// Passes B (MODE 1: calc_patch_spatial_stiffness! about the Dual cop) and C (MODE 2: calc_spatial_bristle_force) over
// the kept Dual polygons: one lane per slot, coalesced loads, same fan / quadrature arithmetic as pass A.
template <int MODE, bool FX = false>
__global__ void __launch_bounds__(64) k_dual_poly(DualArgs g) {
    __shared__ double red[MODE == 2 ? kDualRedRows * kDualRedStride : 1];
    const int lane = threadIdx.x;
    // the slots k_narrow_dual owns: 64 per group of 64 / n_dir contributing pairs (markers where no polygon was kept)
    int n_c = *g.scount;
    if (n_c > g.ccap) n_c = g.ccap;
    const int cpw = 64 / g.n_dir;
    long long n_p = (long long)((n_c + cpw - 1) / cpw) * 64;
    if (n_p > g.dpcap) n_p = g.dpcap;
    const size_t P = (size_t)g.dpcap;
    const long long stride = (long long)gridDim.x * 64;
    constexpr int NS = MODE == 1 ? 21 : 6;
    // Smallest scenes (tri_split): a lane takes ONE fan triangle of a kept polygon (8 lanes per slot) -- a chunk of a
    // pencil-scale bristle pair is a few waves whose time is the serial walk of a lane over its polygon's <= 8 triangles x 3
    // points of Dual arithmetic; the sums are linear, so the per-key reduction below adds the triangles up.  Everything
    // larger keeps one lane per polygon (header and vertices loaded once, an eighth of the waves and of their atomics).
    const int ts = g.tri_split ? 8 : 1;
    n_p *= ts;
    for (long long idx0 = (long long)blockIdx.x * 64; idx0 < n_p; idx0 += stride) {
        const long long lidx = idx0 + lane;
        const bool active = lidx < n_p;
        const long long idx = ts == 8 ? (lidx >> 3) : lidx;      // the slot
        const int tri = ts == 8 ? (int)(lidx & 7) : 0;
        Du sum[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) sum[k] = du(0.0);
        int key = -1, n_trac = 0;
        int2 kn = make_int2(-1, 0);
        if (active) kn = g.dpoly_key[idx];
        if (ts == 8 && tri >= kn.y) kn.x = -1;      // this polygon has fewer fan triangles
        if (kn.x >= 0) {
            key = kn.x;
            const int n = kn.y;
            const ItemRec *it = g.items + key / g.n_dir;
            const double *o = g.dpoly + idx;
            const Du3 nh = dmk(du(o[0], o[3 * P]), du(o[P], o[4 * P]), du(o[2 * P], o[5 * P]));
            const Du3 cen = dmk(du(o[6 * P], o[9 * P]), du(o[7 * P], o[10 * P]), du(o[8 * P], o[11 * P]));
            const double er[4] = {o[12 * P], o[13 * P], o[14 * P], o[15 * P]};
            Du3 cop, Da = dmk(du(0.0), du(0.0), du(0.0)), Dl = Da;
            if (MODE == 1) {
                const double *a = g.dacc + (size_t)key * kDaStride + kDaA;
                const Du ip = du(a[6], a[16]);
                cop = dmk(du(a[7], a[17]), du(a[8], a[18]), du(a[9], a[19])) / ip;
            } else {
                const double *r = g.dres + (size_t)key * kDrStride;
                cop = dmk(du(r[kDrCop], r[kDrCop + 3]), du(r[kDrCop + 1], r[kDrCop + 4]), du(r[kDrCop + 2], r[kDrCop + 5]));
                Da = dmk(du(r[kDrDelta], r[kDrDelta + 6]), du(r[kDrDelta + 1], r[kDrDelta + 7]), du(r[kDrDelta + 2], r[kDrDelta + 8]));
                Dl = dmk(du(r[kDrDelta + 3], r[kDrDelta + 9]), du(r[kDrDelta + 4], r[kDrDelta + 10]), du(r[kDrDelta + 5], r[kDrDelta + 11]));
            }
            dual_integrate<MODE>(
                [&](int k) {
                    const double *q = o + (size_t)(16 + 6 * k) * P;
                    return dmk(du(q[0], q[3 * P]), du(q[P], q[4 * P]), du(q[2 * P], q[5 * P]));
                },
                n, nh, cen, er, it, g.d_twist + (size_t)key * 6, false, cop, Da, Dl, sum, n_trac, ts == 8 ? tri : 0, ts == 8 ? tri + 1 : n);
        }
        double flat[2 * NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) { flat[k] = sum[k].v; flat[NS + k] = sum[k].d; }
        // (pass B keeps the scans: with both forms in the kernel it would need more than 256 registers, and it only runs behind the
        // fused small-scene kernel's hand-over and for tet-tet scenes now)
        if (MODE == 2 && ts == 1) dual_accumulate<2 * NS, FX>(red, g.dacc, key, key >= 0, n_trac > 0, flat, MODE == 1 ? kDaB : kDaC, g.n_dir, lane, &g.sink_c, (int)(idx0 >> 6));
        else accumulate_items<2 * NS, FX>(g.dacc, key, key >= 0, n_trac > 0, flat, MODE == 1 ? kDaB : kDaC, kDaStride, MODE == 1 ? &g.sink_b : &g.sink_c,
                                          (int)(idx0 >> 6));
    }
}

// Option "fixed_order": the records of one accumulator block (pass A's 20 numbers at kDaA, pass B's 42 at kDaB) added per key in
// the order of the positions they carry.  One wave per key; a key's records are as many as waves of the pass held pairs of its item.
// (Also the value pass's friction sums: key = item, accumulator rows of kAccStride.)
constexpr int kSinkSpan = 2048;
__global__ void __launch_bounds__(64) k_fixed_reduce(FixedSink fx, int n_keys, double *dacc, int stride, int n0, int n_val) {
    __shared__ int ord[kSinkSpan], slt[kSinkSpan], perm[kSinkSpan];
    const int key = blockIdx.x, lane = threadIdx.x;
    if (key >= n_keys) return;
    const int head = fx.head[key];
    if (head < 0) return;
    const int n_rec = fx.cap;
    int n = 0;
    if (lane == 0) {
        for (int s = head; s >= 0 && s < n_rec && n < kSinkSpan; ++n) {
            const double *r = fx.rec + (size_t)s * kSinkStride;
            ord[n] = (int)r[1]; slt[n] = s;
            s = (int)r[2];
            if (n + 1 == kSinkSpan && s >= 0) atomicOr(fx.status, kStFixedList);
        }
    }
    n = __builtin_amdgcn_readfirstlane(n);
    wave_lds_sync();
    for (int i = lane; i < n; i += 64) {      // positions are distinct (one record per wave and key): the rank is a permutation
        int rank = 0;
        const int oi = ord[i];
        for (int j = 0; j < n; ++j) rank += (ord[j] < oi || (ord[j] == oi && j < i)) ? 1 : 0;
        perm[rank] = slt[i];
    }
    wave_lds_sync();
    if (lane < n_val) {
        double t = 0.0;
        for (int q = 0; q < n; ++q) t += fx.rec[(size_t)perm[q] * kSinkStride + kSinkHdr + lane];
        dacc[(size_t)key * stride + n0 + lane] = t;
    }
}

// forward declarations (pfc_br.h, included after this header)
__device__ __forceinline__ void jacobi6_wave(double *A, double *V, int lane);

// per (item, direction), one wave: cop, decompose_K! (friction.jl:96-117) with the Frechet derivative of K̄^{-1/2},
// Delta (:130-131).  Lane e = i + 6 j (e < 36) owns entry (i, j) of every 6 x 6 matrix; the matrices live in LDS.
// (The first version ran one thread per (item, direction) with its matrices in scratch: a 0.4 ms dependency chain,
// half of the latency of a Dual evaluation of a single bristle scene.)
__global__ void __launch_bounds__(64) k_dual_eig(DualArgs g) {
    __shared__ double Kv[36], Kd[36], A[36], dK[36], V[36], T[36], M[36], KisV[36], KisD[36];
    __shared__ double SinvV[6], SinvD[6], f[6], fp[6], fx[6], lam[6];
    __shared__ int clamped[6];
    __shared__ double copV[3], copD[3];
    const int key = blockIdx.x, lane = threadIdx.x;
    if (key >= g.n_items * g.n_dir) return;
    const int item = key / g.n_dir;
    const ItemRec *it = g.items + item;
    if (it->model == PFC_REGULARIZED || g.icnt[4 * (size_t)item + 3] <= 0) return;   // uniform over the wave
    if (!seed_nonzero_wave(g, key, lane)) return;      // zero seeds: the passes kept nothing for this key, k_dual_final writes zeros
    const double *a = g.dacc + (size_t)key * kDaStride;
    double *res = g.dres + (size_t)key * kDrStride;
    const double *vk = g.vres_k ? g.vres_k + (size_t)item * kResStride : nullptr;      // uniform
    if (lane < 3) {
        const Du ip = du(a[kDaA + 6], a[kDaA + 16]);
        Du c = du(a[kDaA + 7 + lane], a[kDaA + 17 + lane]) / ip;
        if (vk) c.v = vk[kResCop + lane];
        res[kDrCop + lane] = c.v; res[kDrCop + 3 + lane] = c.d;
        copV[lane] = c.v; copD[lane] = c.d;
    }
    const bool fold = g.vres != nullptr;      // uniform
    if (fold) wave_lds_sync();
    const bool ent = lane < 36;
    const int i = ent ? lane % 6 : 0, j = ent ? lane / 6 : 0;
    if (ent) {
        // K from the 21 sums of pass B: K11 (xx xy xz yy yz zz), K12 (9, column-major), K22 (6); K21 = K12'
        // packed index of entry (r, c) of a symmetric 3 x 3 block (xx xy xz yy yz zz) -- arithmetic, not a table: a local array
        // indexed by the lane lives in scratch memory
        auto u11 = [](int r, int c) { const int lo = r < c ? r : c, hi = r < c ? c : r; return lo * 3 - (lo * (lo - 1)) / 2 + (hi - lo); };
        const int bi = i % 3, bj = j % 3;
        int k;
        if (i < 3 && j < 3) k = u11(bi, bj);
        else if (i >= 3 && j >= 3) k = 15 + u11(bi, bj);
        else if (i < 3) k = 6 + bi + 3 * bj;
        else k = 6 + bj + 3 * bi;
        const double *b = a + kDaB;
        Du kx = du(b[k], b[21 + k]);
        if (fold && (i < 3 || j < 3)) {
            // The sums were formed about c0 (the value pass's cop); about the Dual cop = c0 + d, with D = [d]x:
            //   K22' = K22      K12' = K12 - D K22      K11' = K11 + K12 D + D' K12' + D' K22 D
            // (x' = x - d in calc_patch_spatial_stiffness!, friction.jl:147-169; checked entry by entry against the three-pass form)
            const double *c0 = g.vres + (size_t)item * kResStride + kResCop;
            auto D = [&](int r, int c) -> Du {      // [d]x, d = cop - c0 (read from LDS / the result row by index: no private array)
                if (r == c) return du(0.0);
                const int m = 3 - r - c;
                const Du x = du(copV[m] - c0[m], copD[m]);
                return ((c - r + 3) % 3 == 1) ? -x : x;
            };
            auto K12 = [&](int r, int c) -> Du { return du(b[6 + r + 3 * c], b[21 + 6 + r + 3 * c]); };
            auto K22 = [&](int r, int c) -> Du { return du(b[15 + u11(r, c)], b[21 + 15 + u11(r, c)]); };
            if (i < 3 && j < 3) {
                for (int m = 0; m < 3; ++m) {
                    kx += K12(bi, m) * D(m, bj) + D(m, bi) * K12(bj, m);
                    Du t = du(0.0);
                    for (int l = 0; l < 3; ++l) t += K22(m, l) * D(l, bj);
                    kx += D(m, bi) * t;
                }
            } else {
                const int r = i < 3 ? bi : bj, c = i < 3 ? bj : bi;      // entry (r, c) of K12
                for (int m = 0; m < 3; ++m) kx = kx - D(r, m) * K22(m, c);
            }
        }
        const Du kv = kx * it->k_bar;
        Kv[lane] = vk ? vk[kResK + lane] : kv.v; Kd[lane] = kv.d;
    }
    wave_lds_sync();
    if (lane < 6) {
        const Du t1 = (du(Kv[0], Kd[0]) + du(Kv[7], Kd[7])) + du(Kv[14], Kd[14]);
        const Du t2 = (du(Kv[21], Kd[21]) + du(Kv[28], Kd[28])) + du(Kv[35], Kd[35]);
        const Du sv = lane < 3 ? (du(1.0) / dsqrt(t1)) * it->magic : du(1.0) / dsqrt(t2);
        SinvV[lane] = (vk && g.stored_v) ? vk[kResSinv + lane] : sv.v; SinvD[lane] = sv.d;      // (vk: the scaling the stored decomposition belongs to)
    }
    wave_lds_sync();
    if (ent) {
        const int e = (i <= j) ? i + 6 * j : j + 6 * i;   // Hermitian: upper triangle authoritative
        const Du kb = (du(SinvV[i], SinvD[i]) * du(Kv[e], Kd[e])) * du(SinvV[j], SinvD[j]);
        A[lane] = kb.v; dK[lane] = kb.d;
    }
    wave_lds_sync();
    if (vk && g.stored_v) {
        // the decomposition the value pass made of this very matrix (k_eig: same K, same scaling, same iteration): the six
        // directions of an item -- and every chunk of a Jacobian -- need not repeat its ~8 us dependency chain
        if (ent) { V[lane] = vk[kResV + lane]; if (i == j) A[lane] = vk[kResLam + i]; }
        wave_lds_sync();
    } else {
        jacobi6_wave(A, V, lane);
    }
    // M = V' dK V
    if (ent) {
        double x = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) x += dK[i + 6 * k] * V[k + 6 * j];
        T[lane] = x;
    }
    wave_lds_sync();
    if (ent) {
        double x = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) x += V[k + 6 * i] * T[k + 6 * j];
        M[lane] = x;
    }
    wave_lds_sync();
    if (lane < 6) {
        double mx = A[0];
        int imx = 0;
#pragma unroll
        for (int k = 1; k < 6; ++k) if (A[7 * k] > mx) { mx = A[7 * k]; imx = k; }
        const double floor_v = mx * 1.0e-16, dfloor = M[imx + 6 * imx] * 1.0e-16;   // d(sigma_max) = v_max' dK v_max
        const double l = A[7 * lane];
        const bool cl = !(l > floor_v);          // max(sigma, floor): ties take the floor
        const double x = cl ? floor_v : l;
        const double fv = 1.0 / __builtin_sqrt(x), dfdx = -0.5 * fv / x;
        lam[lane] = l; f[lane] = fv; clamped[lane] = cl ? 1 : 0;
        fp[lane] = cl ? 0.0 : dfdx;              // df/dsigma of the own eigenvalue
        fx[lane] = cl ? dfdx * dfloor : 0.0;     // df through the floor
    }
    wave_lds_sync();
    double gm = 0.0;
    if (ent) {
        double gij;
        if (i == j) gij = fp[i];
        else if (clamped[i] && clamped[j]) gij = 0.0;
        else if (lam[i] != lam[j]) gij = (f[i] - f[j]) / (lam[i] - lam[j]);
        else gij = fp[i];
        gm = gij * M[lane] + (i == j ? fx[i] : 0.0);
        double x = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) x += (V[i + 6 * k] * f[k]) * V[j + 6 * k];
        KisV[lane] = x;
    }
    wave_lds_sync();
    if (ent) M[lane] = gm;          // G o M (+ floor terms on the diagonal)
    wave_lds_sync();
    if (ent) {
        double x = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) x += M[i + 6 * k] * V[j + 6 * k];
        T[lane] = x;
    }
    wave_lds_sync();
    if (ent) {
        double x = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) x += V[i + 6 * k] * T[k + 6 * j];
        KisD[lane] = x;
        res[kDrKis + lane] = KisV[lane]; res[kDrKis + 36 + lane] = x;
    }
    wave_lds_sync();
    if (lane < 6) {
        const double *ds = g.d_s + (size_t)key * 6;
        Du acc = du(0.0);
#pragma unroll
        for (int k = 0; k < 6; ++k) acc += du(KisV[lane + 6 * k], KisD[lane + 6 * k]) * du(it->s[k], ds[k]);
        const Du sv = du(SinvV[lane], SinvD[lane]);
        const Du dl = sv * acc;
        res[kDrDelta + lane] = dl.v; res[kDrDelta + 6 + lane] = dl.d;
        res[kDrSinv + lane] = sv.v; res[kDrSinv + 6 + lane] = sv.d;
    }
}

// per (item, direction): total wrench and sdot partials (friction.jl:134-143, :77-81)
__global__ void __launch_bounds__(64) k_dual_final(DualArgs g) {
    const int key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= g.n_items * g.n_dir) return;
    const int item = key / g.n_dir;
    const ItemRec *it = g.items + item;
    const double *a = g.dacc + (size_t)key * kDaStride;
    double *ow = g.d_wrench + (size_t)key * 6, *os = g.d_sdot + (size_t)key * 6;
    const bool contact = g.icnt[4 * (size_t)item + 3] > 0;
    if (!seed_nonzero(g, key)) {      // zero seeds: zero partials (the sums of this key were never formed)
        for (int k = 0; k < 6; ++k) { ow[k] = 0.0; os[k] = 0.0; }
        return;
    }
    if (it->model == PFC_REGULARIZED) {
        for (int k = 0; k < 6; ++k) { ow[k] = contact ? a[kDaA + 10 + k] : 0.0; os[k] = 0.0; }
        return;
    }
    const double *ds = g.d_s + (size_t)key * 6;
    const double tau_inv = 1.0 / it->tau;
    if (!contact) {
        for (int k = 0; k < 6; ++k) { ow[k] = 0.0; os[k] = -tau_inv * ds[k]; }
        return;
    }
    const double *res = g.dres + (size_t)key * kDrStride;
    Du wc[6];
    for (int k = 0; k < 6; ++k) wc[k] = du(a[kDaC + k], a[kDaC + 6 + k]);
    const Du3 cop = dmk(du(res[kDrCop], res[kDrCop + 3]), du(res[kDrCop + 1], res[kDrCop + 4]), du(res[kDrCop + 2], res[kDrCop + 5]));
    const Du3 fang = dmk(wc[0], wc[1], wc[2]), flin = dmk(wc[3], wc[4], wc[5]);
    const Du3 fang2 = fang + dcross(cop, flin);
    Du sw[6];
    for (int k = 0; k < 6; ++k) sw[k] = du(res[kDrSinv + k], res[kDrSinv + 6 + k]) * wc[k];
    for (int i = 0; i < 6; ++i) {
        Du acc = du(0.0);
        for (int k = 0; k < 6; ++k) acc += du(res[kDrKis + i + 6 * k], res[kDrKis + 36 + i + 6 * k]) * sw[k];
        const Du sd = (acc + du(it->s[i], ds[i])) * (-tau_inv);
        os[i] = sd.d;
    }
    ow[0] = a[kDaA + 10] + fang2.x.d; ow[1] = a[kDaA + 11] + fang2.y.d; ow[2] = a[kDaA + 12] + fang2.z.d;
    ow[3] = a[kDaA + 13] + flin.x.d; ow[4] = a[kDaA + 14] + flin.y.d; ow[5] = a[kDaA + 15] + flin.z.d;
}
