// pfc_br.h -- per-item kernels of the bristle model (record shift, 6x6 eigen, finalisation), result packing, generalized-force scatter, arithmetic self-test.  Included by pfc_hip.hip inside namespace pfc (device code only).
#pragma once

// =================================================================================================================
// bristle model: cop, patch stiffness, 6x6 eigen, friction pass, finalisation
// =================================================================================================================
struct BrArgs {
    int *ctr;            // counter block, packed into tail and zeroed by k_final
    int *rgn;            // region counters of the polygon / record lists: summed into tail[12 + i_pcount], tail[12 + 3], zeroed
    int i_pcount;
    int n_ctr;
    unsigned *status;
    int *tail;
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

// Parallel-axis shift of the 27 patch-stiffness moments (Snn 6, San 9, Saa 6, Srr 6: the kAccSnn.. layout) of a set of
// traction points from a reference point c to c - d (d = c - target), given W = sum w and m1 = sum w (r - c):
//   Snn' = Snn            San' = San + [d]x Snn            Srr' = Srr + m1 d' + d m1' + W d d'
//   Saa' = Saa + San [d]x' + [d]x San' + [d]x Snn [d]x'
// Every shift distance in this library is of the order of the patch size (reference points lie inside the patch), so
// nothing cancels.  Used by k_shift (records of the batched narrowphase) and by the one-launch kernel (pfc_fused.h).
__device__ __forceinline__ void shift_moments(const double *m, double W, const double *m1, const double *d, double *o) {
    const int s6[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
    double Snn[9], San[9], Saa[9], Srr[9];
    const double dx[9] = {0.0, d[2], -d[1], -d[2], 0.0, d[0], d[1], -d[0], 0.0};   // [d]x column-major
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        Snn[k] = m[s6[k]]; San[k] = m[6 + k]; Saa[k] = m[15 + s6[k]]; Srr[k] = m[21 + s6[k]];
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
    const int u6[6] = {0, 3, 6, 4, 7, 8};   // xx xy xz yy yz zz in a column-major 3x3
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = Snn[u6[k]];
#pragma unroll
    for (int k = 0; k < 9; ++k) o[6 + k] = San[k] + dS[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int ii = u6[k] % 3, j = u6[k] / 3;
        o[15 + k] = Saa[u6[k]] + Sd[ii + 3 * j] + Sd[j + 3 * ii] + dSd[u6[k]];
        // sum w (x + d)(x + d)' = Srr + m1 d' + d m1' + W d d'  (m1 = sum w x about the reference point)
        o[21 + k] = Srr[u6[k]] + (m1[ii] * d[j] + d[ii] * m1[j]) + W * d[ii] * d[j];
    }
}

// Moves every moment record from its run centroid c_w to the item's cop and adds it to the item accumulators.
// With d = c_w - cop and sum w (r - c_w) = 0 by construction of c_w:
//   Snn' = Snn            San' = San + [d]x Snn            Srr' = Srr + W d d'
//   Saa' = Saa + San [d]x' + [d]x San' + [d]x Snn [d]x'
// One lane computes one record, the block transposes through LDS so that each record leaves as ONE 27-lane atomic.
struct ShiftArgs {
    const double *rec;
    const int *rgn;      // region counters (word 1: moment records)
    int rr_cap;          // record slots per region
    double *acc;
};
__global__ void __launch_bounds__(64) k_shift(ShiftArgs g) {
    __shared__ double out[64 * 28];
    __shared__ int items[64];
    const int lane = threadIdx.x;
    const RgnScan rs = rgn_scan(g.rgn, 1, g.rr_cap, lane);
    for (int w = blockIdx.x; w < rs.total; w += gridDim.x) {
        int slot0, n_here;
        rgn_locate(rs, w, g.rr_cap, slot0, n_here);
        const int i = slot0 + lane;
        items[lane] = -1;
        if (lane < n_here && g.rec[(size_t)i * kRecStride + 1] > 0.0) {   // W = 0: a reserved slot that was not needed
            const double *r = g.rec + (size_t)i * kRecStride;
            const int item = (int)r[0];
            const double W = r[1];
            const double *a = g.acc + (size_t)item * kAccStride;
            const double S = a[kAccIp];
            const double d[3] = {r[2] - a[kAccIpc] / S, r[3] - a[kAccIpc + 1] / S, r[4] - a[kAccIpc + 2] / S};
            const double m1[3] = {r[32], r[33], r[34]};
            shift_moments(r + 5, W, m1, d, out + lane * 28);
            items[lane] = item;
        }
        __syncthreads();
        for (int q = 0; q < n_here; ++q) {
            if (lane < 27 && items[q] >= 0) {
                const double x = out[q * 28 + lane];
                if (x != 0.0) unsafeAtomicAdd(&g.acc[(size_t)items[q] * kAccStride + kAccSnn + lane], x);
            }
        }
        __syncthreads();
    }
}

// ---- option "fixed_order": an item's run records added in CHUNK ORDER ----------------------------------------------------
// k_integ_fixed leaves one record per (chunk, item) run -- the ten sums, and for a bristle run the 27 moments about the run's
// reference point -- in a list per item; with the candidate list sorted (pfc_sort.hip) the runs and their values are the same in
// every evaluation of the same inputs, and so is the result of adding them in the order of their chunks: one wave per item, no
// atomic.  The chunks of an item are consecutive (its candidates are), so chunk - first chunk indexes a table.
constexpr int kFixedSpan = 4096;      // chunks one item may span (x 512 candidates)
struct FixedArgs {
    const double *rec;
    int *det;            // per item: last record, first chunk, last chunk
    int n_items;
    int n_slots;         // record slots in all (kRgn x rr_cap)
    const ItemRec *items;
    double *acc;
    unsigned *status;
};
__global__ void __launch_bounds__(256) k_fixed_init(int n_items, int *det) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_items) { det[3 * (size_t)i] = -1; det[3 * (size_t)i + 1] = 0x7FFFFFFF; det[3 * (size_t)i + 2] = -1; }
}
__global__ void __launch_bounds__(64) k_shift_fixed(FixedArgs g) {
    __shared__ int tab[kFixedSpan];
    __shared__ double stage[64 * 28];
    const int item = blockIdx.x, lane = threadIdx.x;
    if (item >= g.n_items) return;
    const int head = g.det[3 * (size_t)item], ch0 = g.det[3 * (size_t)item + 1], ch1 = g.det[3 * (size_t)item + 2];
    if (head < 0 || ch1 < ch0) return;      // no run with a traction point
    const int span = ch1 - ch0 + 1;
    if (span > kFixedSpan) {
        if (lane == 0) atomicOr(g.status, kStFixedSpan);
        return;
    }
    for (int k = lane; k < span; k += 64) tab[k] = -1;
    wave_lds_sync();
    if (lane == 0) {
        // (the list is as long as the item has runs; the guard bounds the walk should a slot ever hold something else)
        int guard = 0;
        for (int s = head; s >= 0 && s < g.n_slots && guard <= span; ++guard) {
            const double *r = g.rec + (size_t)s * kRecStrideFixed;
            const int k = (int)r[kRecChunk] - ch0;
            if (k >= 0 && k < span && (int)r[0] == item) tab[k] = s;
            s = (int)r[kRecNext];
        }
    }
    wave_lds_sync();
    // (1) the ten sums, chunk by chunk: 64 records fetched at a time (lane l: record l of the window), added in order by lane k
    double t10 = 0.0;
    for (int w0 = 0; w0 < span; w0 += 64) {
        const int s = (w0 + lane < span) ? tab[w0 + lane] : -1;
#pragma unroll
        for (int k = 0; k < 10; ++k) stage[lane * 11 + k] = s >= 0 ? g.rec[(size_t)s * kRecStrideFixed + kRecTen + k] : 0.0;
        wave_lds_sync();
        const int n_here = span - w0 < 64 ? span - w0 : 64;
        if (lane < 10)
            for (int q = 0; q < n_here; ++q) t10 += stage[q * 11 + lane];
        wave_lds_sync();
    }
    double *a = g.acc + (size_t)item * kAccStride;
    if (lane < 10) a[lane] = t10;
    if (g.items[item].model != PFC_BRISTLE) return;
    // (2) the moments, each record moved from its reference point to the cop (k_shift's arithmetic), added in chunk order
    const double S = __shfl(t10, kAccIp, 64);
    if (!(S > 0.0)) return;
    const double cx = __shfl(t10, kAccIpc, 64) / S, cy = __shfl(t10, kAccIpc + 1, 64) / S, cz = __shfl(t10, kAccIpc + 2, 64) / S;
    double m = 0.0;
    for (int w0 = 0; w0 < span; w0 += 64) {
        const int s = (w0 + lane < span) ? tab[w0 + lane] : -1;
        bool live = false;
        if (s >= 0) {
            const double *r = g.rec + (size_t)s * kRecStrideFixed;
            const double W = r[1];
            if (W > 0.0) {
                const double d[3] = {r[2] - cx, r[3] - cy, r[4] - cz};
                const double m1[3] = {r[32], r[33], r[34]};
                shift_moments(r + 5, W, m1, d, stage + lane * 28);
                live = true;
            }
        }
        if (!live)
            for (int k = 0; k < 27; ++k) stage[lane * 28 + k] = 0.0;
        wave_lds_sync();
        const int n_here = span - w0 < 64 ? span - w0 : 64;
        if (lane < 27)
            for (int q = 0; q < n_here; ++q) m += stage[q * 28 + lane];
        wave_lds_sync();
    }
    if (lane < 27) a[kAccSnn + lane] = m;
}

// Jacobi eigen-solver for a symmetric 6x6 (stands in for LAPACK eigen!(Hermitian), friction.jl:88): rotation angle of
// one pivot, then the wave-cooperative round-robin iteration.
__device__ __forceinline__ void jacobi_angle(double app, double aqq, double apq, double &cs, double &sn) {
    // apq == 0: identity rotation.  Reciprocals and square roots by the hardware estimates + two Newton steps each (~1e-16:
    // a Jacobi rotation only has to be orthogonal to rounding, c^2 + s^2 = 1; the IEEE division / sqrt sequences were 2/3 of
    // the ~250 instructions of a round).  A huge |theta| (pivot at the rounding floor) takes the limit t = 1 / (2 theta).
    double r = __builtin_amdgcn_rcp(2.0 * apq);
    r = r * __builtin_fma(-(2.0 * apq), r, 2.0);
    r = r * __builtin_fma(-(2.0 * apq), r, 2.0);
    const double theta = (aqq - app) * r;
    const double at = __builtin_fabs(theta);
    const double h = __builtin_fma(theta, theta, 1.0);
    double y = __builtin_amdgcn_rsq(h);
    y = y * __builtin_fma(-0.5 * h * y, y, 1.5);
    y = y * __builtin_fma(-0.5 * h * y, y, 1.5);
    const double den = at + h * y;                   // |theta| + sqrt(theta^2 + 1)
    double q = __builtin_amdgcn_rcp(den);
    q = q * __builtin_fma(-den, q, 2.0);
    q = q * __builtin_fma(-den, q, 2.0);
    double t = theta >= 0 ? q : -q;
    if (!(at < 1.0e100)) t = 0.5 * __builtin_amdgcn_rcp(theta);      // (also NaN / inf from a vanishing pivot)
    if (apq == 0.0 || !(__builtin_fabs(t) <= 1.0)) t = 0.0;
    const double g = __builtin_fma(t, t, 1.0);
    double c = __builtin_amdgcn_rsq(g);
    c = c * __builtin_fma(-0.5 * g * c, c, 1.5);
    c = c * __builtin_fma(-0.5 * g * c, c, 1.5);
    cs = c;
    sn = t * cs;
}
// Round-robin Jacobi (5 rounds of 3 index-disjoint pivots per sweep) carried by ONE WAVE: lane e = i + 6 j (e < 36) owns
// A[i][j] and V[i][j] in LDS.  The three rotations of a round touch disjoint index pairs, so together they map entry
// (i, j) through the rotation of i's pair (rows) and of j's pair (columns):
//     new = a_i (a_j A_ij + b_j A_i,pj) + b_i (a_j A_pi,j + b_j A_pi,pj)        (columns first, then rows)
// with (a, b) = (c, -s) for the smaller index of a pair and (c, +s) for the larger.  Every lane recomputes the two
// rotation angles it needs from the shared diagonal / pivot entries (broadcast LDS reads) instead of waiting for
// another lane to publish them.  A round is ~200 instructions and two LDS round trips; the first version (one thread
// per item, matrices in registers, three overlapped rotations per round) had a 45 us dependency chain, a fifth of the
// latency of a single bristle evaluation.  Converged when every off-diagonal entry is below the rounding floor of the
// matrix (eps * largest diagonal) or negligible against its own two diagonal entries; also stops once (after 4
// sweeps) a sweep no longer halves the off-diagonal mass (nothing but rounding noise is left to annihilate).
__device__ __forceinline__ void jacobi6_wave(double *A, double *V, int lane) {
    const bool ent = lane < 36;
    const int i = ent ? lane % 6 : 0, j = ent ? lane / 6 : 0;
    if (ent) V[lane] = (i == j) ? 1.0 : 0.0;
    wave_lds_sync();
    // partner of every index in the five rounds (0,5)(1,4)(2,3) | (0,4)(3,5)(1,2) | (0,3)(2,4)(1,5) | (0,2)(1,3)(4,5) | (0,1)(2,5)(3,4)
    const int PT[5][6] = {{5, 4, 3, 2, 1, 0}, {4, 2, 1, 5, 0, 3}, {3, 5, 4, 0, 2, 1}, {2, 3, 0, 1, 5, 4}, {1, 0, 5, 4, 3, 2}};
    double off_prev = 1.79769313486231570815e308;
    for (int sweep = 0; sweep < 40; ++sweep) {
        double dmax = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) dmax = fmax(dmax, __builtin_fabs(A[7 * k]));
        const bool upper = ent && i < j;
        const double a = upper ? __builtin_fabs(A[lane]) : 0.0;
        const bool ok = !upper || a <= 2.3e-16 * dmax || a * a <= 1e-30 * __builtin_fabs(A[7 * i] * A[7 * j]);
        const bool done = __all(ok);
        const double off = wave_sum(a * a);
        if (done || (sweep >= 4 && !(off < 0.5 * off_prev))) break;
        off_prev = off;
        for (int r = 0; r < 5; ++r) {
            const int pi = PT[r][i], pj = PT[r][j];
            double ai, bi, aj, bj;
            {
                const int p = i < pi ? i : pi, q = i < pi ? pi : i;
                double cs, sn;
                jacobi_angle(A[7 * p], A[7 * q], A[p + 6 * q], cs, sn);
                ai = cs; bi = (i == p) ? -sn : sn;
            }
            // the column rotation (j, pj) is the row rotation of lane j (= entry (j, 0)): the same numbers, two lane reads
            // instead of a second angle (35 dependent Float64 instructions of a round's ~200)
            aj = __shfl(ai, j, 64); bj = __shfl(bi, j, 64);
            const double t0 = aj * A[i + 6 * j] + bj * A[i + 6 * pj];      // columns first ...
            const double t1 = aj * A[pi + 6 * j] + bj * A[pi + 6 * pj];
            const double an = ai * t0 + bi * t1;                            // ... then rows
            const double vn = aj * V[i + 6 * j] + bj * V[i + 6 * pj];
            wave_lds_sync();
            if (ent) { A[lane] = an; V[lane] = vn; }
            wave_lds_sync();
        }
    }
}

// decompose_K! / calc_K̄_sqrt_inv / Δ² (friction.jl:85-132) of one bristle item in contact, carried by ONE wave.
// a: the item's accumulator block (kAcc* layout, moments about the cop), r: its result block (kRes* layout, global or
// LDS), s: the bristle state; E: LDS scratch of the wave.
struct EigScratch { double K[36], A[36], V[36], Kis[36], sig[6], Sinv[6]; };
__device__ __forceinline__ void eig_item(const double *a, double k_bar, double magic, const double *s, double *r,
                                         EigScratch &E, int lane) {
    double *K = E.K, *A = E.A, *V = E.V, *Kis = E.Kis, *sig = E.sig, *Sinv = E.Sinv;
    // cop = sum w r / sum w (normal.jl:33)
    const double S = a[kAccIp];
    if (lane < 3) r[kResCop + lane] = a[kAccIpc + lane] / S;
    // calc_patch_spatial_stiffness! (friction.jl:147-169) from the moments about the cop (x = r - cop):
    //   K22 = S I - sum w n n'      K12 = -sum w (x x n) n'   (sum w [x]x = 0 about the cop)
    //   K11 = -(sum w x x' - tr(.) I + sum w (x x n)(x x n)')
    if (lane < 36) {
        const int s6[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};  // symmetric 3x3 from 6 unique
        const int ii = lane % 6, jj = lane / 6;
        const int bi = ii % 3, bj = jj % 3;
        const double I = (bi == bj) ? 1.0 : 0.0;
        double kv;
        if (ii < 3 && jj < 3) {
            const double trC = (a[kAccSrr] + a[kAccSrr + 3]) + a[kAccSrr + 5];
            kv = -(a[kAccSrr + s6[bi + 3 * bj]] - trC * I + a[kAccSaa + s6[bi + 3 * bj]]);
        } else if (ii >= 3 && jj >= 3) {
            kv = S * I - a[kAccSnn + s6[bi + 3 * bj]];
        } else if (ii < 3) {
            kv = -a[kAccSan + bi + 3 * bj];        // K12[bi][bj]
        } else {
            kv = -a[kAccSan + bj + 3 * bi];        // K21 = K12'
        }
        kv *= k_bar;
        K[lane] = kv;
        r[kResK + lane] = kv;
    }
    wave_lds_sync();
    if (lane < 6) {
        const double t1 = (K[0] + K[7]) + K[14], t2 = (K[21] + K[28]) + K[35];
        Sinv[lane] = lane < 3 ? (1.0 / __builtin_sqrt(t1)) * magic : 1.0 / __builtin_sqrt(t2);
    }
    wave_lds_sync();
    if (lane < 36) {
        const int ii = lane % 6, jj = lane / 6;
        const double kij = (ii <= jj) ? K[ii + 6 * jj] : K[jj + 6 * ii];   // Hermitian: upper triangle authoritative
        A[lane] = (Sinv[ii] * kij) * Sinv[jj];
    }
    wave_lds_sync();
    jacobi6_wave(A, V, lane);
    if (lane < 36) r[kResV + lane] = V[lane];
    if (lane < 6) r[kResLam + lane] = A[7 * lane];
    if (lane < 6) {
        double mx = A[0];
#pragma unroll
        for (int k = 1; k < 6; ++k) mx = fmax(mx, A[7 * k]);
        sig[lane] = 1.0 / __builtin_sqrt(fmax(A[7 * lane], mx * 1.0e-16));
    }
    wave_lds_sync();
    if (lane < 36) {
        const int ii = lane % 6, jj = lane / 6;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc += (V[ii + 6 * k] * sig[k]) * V[jj + 6 * k];
        Kis[lane] = acc;
        r[kResKis + lane] = acc;
    }
    wave_lds_sync();
    if (lane < 6) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc += Kis[lane + 6 * k] * s[k];
        r[kResDelta + lane] = Sinv[lane] * acc;
        r[kResSinv + lane] = Sinv[lane];
    }
}

// one wave per bristle item in contact
__global__ void __launch_bounds__(64) k_eig(BrArgs g) {
    __shared__ EigScratch E;
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= g.n_items) return;
    const ItemRec *it = g.items + i;
    if (it->model != PFC_BRISTLE || g.icnt[4 * (size_t)i + 3] == 0) return;   // uniform over the wave
    eig_item(g.acc + (size_t)i * kAccStride, it->k_bar, it->magic, it->s, g.res + (size_t)i * kResStride, E, lane);
}

// yes_contact! / no_contact! epilogue (friction.jl:76-81,119-143; non_friction.jl:77-83)
__device__ __forceinline__ void final_item(const BrArgs &g, int i) {
    const ItemRec *it = g.items + i;
    const double *a = g.acc + (size_t)i * kAccStride;
    const double *r = g.res + (size_t)i * kResStride;
    double *w = g.wrench + 6 * (size_t)i, *sd = g.sdot + 6 * (size_t)i;
    // everything the common paths need is loaded before the first branch (one round trip to memory instead of three)
    const int4 cnt = *reinterpret_cast<const int4 *>(g.icnt + 4 * (size_t)i);
    const int model = it->model;
    double aw[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) aw[k] = a[kAccWrench + k];
    const bool contact = cnt.w > 0;
    if (g.counts) {   // the caller's buffer: only int alignment may be assumed
        int *co = g.counts + 4 * (size_t)i;
        co[0] = cnt.x; co[1] = cnt.y; co[2] = cnt.z; co[3] = cnt.w;
    }
    if (model == PFC_REGULARIZED) {
        for (int k = 0; k < 6; ++k) { w[k] = contact ? aw[k] : 0.0; sd[k] = 0.0; }
        return;
    }
    const double tau_inv = 1.0 / it->tau;
    if (!contact) {
        double s0[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) s0[k] = it->s[k];
        for (int k = 0; k < 6; ++k) { w[k] = 0.0; sd[k] = -tau_inv * s0[k]; }
        return;
    }
    // every operand is in registers before the first store (the outputs are plain double pointers: behind a store the compiler
    // must re-issue what it could otherwise have fetched in one go -- the 6 x 6 product below was six dependent round trips)
    double fr[6], sinv[6], kis[36], s0[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { fr[k] = a[kAccFric + k]; sinv[k] = r[kResSinv + k]; s0[k] = it->s[k]; }
#pragma unroll
    for (int k = 0; k < 36; ++k) kis[k] = r[kResKis + k];
    const V3 cop = ld3(r + kResCop);
    const V3 fang = mk3(fr[0], fr[1], fr[2]), flin = mk3(fr[3], fr[4], fr[5]);
    const V3 fang2 = fang + cross(cop, flin);
    double sw[6], sdv[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) sw[k] = sinv[k] * fr[k];
#pragma unroll
    for (int ii = 0; ii < 6; ++ii) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc += kis[ii + 6 * k] * sw[k];
        sdv[ii] = -tau_inv * (acc + s0[ii]);
    }
    w[0] = aw[0] + fang2.x; w[1] = aw[1] + fang2.y; w[2] = aw[2] + fang2.z;
    w[3] = aw[3] + flin.x; w[4] = aw[4] + flin.y; w[5] = aw[5] + flin.z;
#pragma unroll
    for (int ii = 0; ii < 6; ++ii) sd[ii] = sdv[ii];
}

// k_final: the per-item epilogue, and -- in block 0, after its own items -- the packing of everything the host needs
// to judge an evaluation into one small block (one D2H copy instead of five): tail[0..3] status words, tail[4..11] totals
// {node tests, non-empty pairs, traction points, 0} as 64-bit, tail[12..] the counter block (candidates, traction slots,
// seed ticket, records, frontier sizes per level).  The packing only reads what earlier kernels wrote, so it does not
// wait for the other blocks; it also leaves the counters and the status word zeroed for the next evaluation.  (What a
// small scene pays is launches, not kernels: this used to be a kernel of its own plus two memset nodes.)
__global__ void __launch_bounds__(128) k_final(BrArgs g) {
    __shared__ unsigned long long tot[4];
    __shared__ int rtot[2];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * blockDim.x + tid;
    const bool b0 = blockIdx.x == 0;
    // Block 0 issues every load of the packing FIRST: they depend on nothing this kernel computes, and a small scene
    // pays each dependent round trip to memory in full (this kernel was 14.6 us of a 68 us C1 evaluation when the
    // packing loaded after the epilogue and between barriers).
    unsigned stw = 0;
    int c0 = 0, c1 = 0, r0 = 0, r1 = 0;
    unsigned long long a = 0, b = 0, c = 0, d = 0;
    if (b0) {
        if (tid < 4) stw = g.status[tid];
        if (tid < g.n_ctr) c0 = g.ctr[tid];
        if (tid + 128 < g.n_ctr) c1 = g.ctr[tid + 128];
        if (tid < kRgn) { r0 = g.rgn[tid * kRgnStride]; r1 = g.rgn[tid * kRgnStride + 1]; }
        // (one 16-byte load per item, eight items in flight per thread)
        const int4 *ic = reinterpret_cast<const int4 *>(g.icnt);
        auto tally = [&](const int4 &v) {
            a += (unsigned)v.x; b += (unsigned)v.z; c += (unsigned)v.w;
            d += (unsigned)v.x > 1u ? 1ull : 0ull;      // items whose root pair overlapped (pile_mode's hint)
        };
        int k = tid;
        for (; k + 7 * 128 < g.n_items; k += 8 * 128) {
            int4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ic[k + u * 128];
#pragma unroll
            for (int u = 0; u < 8; ++u) tally(v[u]);
        }
        for (; k < g.n_items; k += 128) tally(ic[k]);
        if (tid < 4) tot[tid] = 0ull;
        if (tid < 2) rtot[tid] = 0;
    }
    if (i < g.n_items) final_item(g, i);
    if (!b0) return;
    __syncthreads();
    // Wave sums first, then one LDS atomic per wave and total: 128 threads adding to the same four 64-bit words one by one were
    // ~13 us of this kernel's ~20 (elimination build: 21.4 -> 6.7 us at 2 016 items without the packing).  The sums are integers
    // far below 2^53: exact as doubles.
    {
        const double sa = wave_sum((double)a), sb = wave_sum((double)b), sc = wave_sum((double)c), sd_ = wave_sum((double)d);
        const double s0 = wave_sum((double)r0), s1 = wave_sum((double)r1);
        if ((tid & 63) == 0) {
            if (sa != 0.0) atomicAdd(&tot[0], (unsigned long long)sa);
            if (sb != 0.0) atomicAdd(&tot[1], (unsigned long long)sb);
            if (sc != 0.0) atomicAdd(&tot[2], (unsigned long long)sc);
            if (sd_ != 0.0) atomicAdd(&tot[3], (unsigned long long)sd_);
            if (s0 != 0.0) atomicAdd(&rtot[0], (int)s0);
            if (s1 != 0.0) atomicAdd(&rtot[1], (int)s1);
        }
    }
    int *tail = g.tail;
    if (tid < 4) { tail[tid] = (int)stw; g.status[tid] = 0u; }
    if (tid < g.n_ctr && tid != g.i_pcount && tid != 3) tail[12 + tid] = c0;
    if (tid + 128 < g.n_ctr) tail[12 + tid + 128] = c1;     // i_pcount, 3 < 128
    if (tid < g.n_ctr) g.ctr[tid] = 0;
    if (tid + 128 < g.n_ctr) g.ctr[tid + 128] = 0;
    for (int k = tid + 256; k < g.n_ctr; k += blockDim.x) { tail[12 + k] = g.ctr[k]; g.ctr[k] = 0; }   // very deep trees
    if (tid < kRgn) { g.rgn[tid * kRgnStride] = 0; g.rgn[tid * kRgnStride + 1] = 0; }
    __syncthreads();
    if (tid < 4) reinterpret_cast<unsigned long long *>(tail + 4)[tid] = tot[tid];
    // region counters -> totals (the host sizes the record list by them)
    if (tid == 0) { tail[12 + g.i_pcount] = rtot[0]; tail[12 + 3] = rtot[1]; }
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

