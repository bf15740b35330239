// pfc_br.h -- per-item kernels of the bristle model (record shift, 6x6 eigen, finalisation), result packing, generalized-force scatter, arithmetic self-test.  Included by pfc_hip.hip inside namespace pfc (device code only).
#pragma once

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

