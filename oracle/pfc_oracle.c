/*
 * pfc_oracle.c — plain-C, scalar, single-thread restatement of the reference hot path.
 * TEST INFRASTRUCTURE ONLY (see pfc_oracle.h).  Every function cites the reference lines it follows
 * (paths relative to /root/reference).  Loop structure deliberately mirrors the Julia code: recursive dual-tree
 * descent with a full 4x4 pose composition per node pair, a 4x4 inverse per candidate pair, a materialised
 * traction-point list, one pass over it for regularized friction and three for bristle friction.
 */
#include "pfc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------------------ */
/* small vectors                                                                                                  */
/* ------------------------------------------------------------------------------------------------------------ */
typedef struct { double x, y, z; } v3;
typedef struct { double c[4]; } v4;

static inline v3 V3(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 scl3(v3 a, double s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 div3(v3 a, double s) { return V3(a.x / s, a.y / s, a.z / s); }
/* StaticArrays dot/cross: left-to-right sums, no contraction */
static inline double dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
/* StaticArrays normalize(a) = inv(norm(a)) * a */
static inline v3 normalize3(v3 a) { double s = 1.0 / sqrt(dot3(a, a)); return V3(s * a.x, s * a.y, s * a.z); }
static inline v3 ld3(const double *p) { return V3(p[0], p[1], p[2]); }
static inline void st3(double *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

/* column-major 4x4: m[i + 4 j] */
typedef struct { double m[16]; } m4;

/* StaticArrays 4x4 * 4x4: C[i,j] = ((A[i,1]B[1,j] + A[i,2]B[2,j]) + A[i,3]B[3,j]) + A[i,4]B[4,j] */
static m4 mul44(const m4 *A, const m4 *B)
{
    m4 C;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            C.m[i + 4 * j] = ((A->m[i] * B->m[4 * j] + A->m[i + 4] * B->m[1 + 4 * j]) + A->m[i + 8] * B->m[2 + 4 * j]) +
                             A->m[i + 12] * B->m[3 + 4 * j];
    return C;
}

static v4 mul4v(const m4 *A, v4 v)
{
    v4 r;
    for (int i = 0; i < 4; ++i)
        r.c[i] = ((A->m[i] * v.c[0] + A->m[i + 4] * v.c[1]) + A->m[i + 8] * v.c[2]) + A->m[i + 12] * v.c[3];
    return r;
}

/* basic_dh(R, t): src/math_kernel/basic_dh.jl:38-45 */
static m4 dh_from_Rt(const double R[9], const double t[3])
{
    m4 M;
    M.m[0] = R[0]; M.m[1] = R[1]; M.m[2] = R[2]; M.m[3] = 0.0;
    M.m[4] = R[3]; M.m[5] = R[4]; M.m[6] = R[5]; M.m[7] = 0.0;
    M.m[8] = R[6]; M.m[9] = R[7]; M.m[10] = R[8]; M.m[11] = 0.0;
    M.m[12] = t[0]; M.m[13] = t[1]; M.m[14] = t[2]; M.m[15] = 1.0;
    return M;
}

/* 4x4 inverse, adjugate x (1/det) from 2x2 minors.  Stands in for StaticArrays inv(::SMatrix{4,4}) at
 * src/contact_algorithms_non_friction.jl:160 (bit-level rounding unpinned, see header).  Variant 0 of pfo_inv4 (the default of rounds 1-2;
 * the Laplace-expansion form later StaticArrays versions adopted). */
static int inv4_minors(const double a[16], double b[16])
{
#define A(i, j) a[(i) + 4 * (j)]
    double s0 = A(0, 0) * A(1, 1) - A(1, 0) * A(0, 1);
    double s1 = A(0, 0) * A(1, 2) - A(1, 0) * A(0, 2);
    double s2 = A(0, 0) * A(1, 3) - A(1, 0) * A(0, 3);
    double s3 = A(0, 1) * A(1, 2) - A(1, 1) * A(0, 2);
    double s4 = A(0, 1) * A(1, 3) - A(1, 1) * A(0, 3);
    double s5 = A(0, 2) * A(1, 3) - A(1, 2) * A(0, 3);
    double c5 = A(2, 2) * A(3, 3) - A(3, 2) * A(2, 3);
    double c4 = A(2, 1) * A(3, 3) - A(3, 1) * A(2, 3);
    double c3 = A(2, 1) * A(3, 2) - A(3, 1) * A(2, 2);
    double c2 = A(2, 0) * A(3, 3) - A(3, 0) * A(2, 3);
    double c1 = A(2, 0) * A(3, 2) - A(3, 0) * A(2, 2);
    double c0 = A(2, 0) * A(3, 1) - A(3, 0) * A(2, 1);
    double det = ((((s0 * c5 - s1 * c4) + s2 * c3) + s3 * c2) - s4 * c1) + s5 * c0;
    double id = 1.0 / det;
#define B(i, j) b[(i) + 4 * (j)]
    B(0, 0) = ((A(1, 1) * c5 - A(1, 2) * c4) + A(1, 3) * c3) * id;
    B(0, 1) = ((-A(0, 1) * c5 + A(0, 2) * c4) - A(0, 3) * c3) * id;
    B(0, 2) = ((A(3, 1) * s5 - A(3, 2) * s4) + A(3, 3) * s3) * id;
    B(0, 3) = ((-A(2, 1) * s5 + A(2, 2) * s4) - A(2, 3) * s3) * id;
    B(1, 0) = ((-A(1, 0) * c5 + A(1, 2) * c2) - A(1, 3) * c1) * id;
    B(1, 1) = ((A(0, 0) * c5 - A(0, 2) * c2) + A(0, 3) * c1) * id;
    B(1, 2) = ((-A(3, 0) * s5 + A(3, 2) * s2) - A(3, 3) * s1) * id;
    B(1, 3) = ((A(2, 0) * s5 - A(2, 2) * s2) + A(2, 3) * s1) * id;
    B(2, 0) = ((A(1, 0) * c4 - A(1, 1) * c2) + A(1, 3) * c0) * id;
    B(2, 1) = ((-A(0, 0) * c4 + A(0, 1) * c2) - A(0, 3) * c0) * id;
    B(2, 2) = ((A(3, 0) * s4 - A(3, 1) * s2) + A(3, 3) * s0) * id;
    B(2, 3) = ((-A(2, 0) * s4 + A(2, 1) * s2) - A(2, 3) * s0) * id;
    B(3, 0) = ((-A(1, 0) * c3 + A(1, 1) * c1) - A(1, 2) * c0) * id;
    B(3, 1) = ((A(0, 0) * c3 - A(0, 1) * c1) + A(0, 2) * c0) * id;
    B(3, 2) = ((-A(3, 0) * s3 + A(3, 1) * s1) - A(3, 2) * s0) * id;
    B(3, 3) = ((A(2, 0) * s3 - A(2, 1) * s1) + A(2, 2) * s0) * id;
#undef A
#undef B
    return isfinite(id) ? 0 : 1;
}

/* Variant 1: the explicit cofactor expansion x (1/det) -- the form StaticArrays published for inv(::SMatrix{4,4}) at the
 * version the reference pins (0.10.3, Manifest.toml; src/inv.jl `_inv(::Size{(4,4)}, A)` with `idet = 1/det(A)` and
 * src/det.jl `_det(::Size{(4,4)}, A)`): every entry is six left-to-right triple products times idet, the determinant 24
 * left-to-right quadruple products.  StaticArrays is NOT vendored under /root/reference and there is no network, so the
 * term ORDER below is written from the published source as recalled and cannot be diffed here; what the variant measures
 * (tests/test_inv4_exposure.py) is how many predicate outcomes of the path depend on WHICH correctly-rounded-to-a-few-ulp
 * inverse is used -- the only unpinned rounding of the oracle. */
static int inv4_cofactor(const double a[16], double b[16])
{
#define A(i, j) a[((i) - 1) + 4 * ((j) - 1)]
    /* det: linear (column-major) indices A[1..16] as in the published _det */
#define L(k) a[(k) - 1]
    double det =
        L(13) * L(10) * L(7) * L(4) - L(9) * L(14) * L(7) * L(4) -
        L(13) * L(6) * L(11) * L(4) + L(5) * L(14) * L(11) * L(4) +
        L(9) * L(6) * L(15) * L(4) - L(5) * L(10) * L(15) * L(4) -
        L(13) * L(10) * L(3) * L(8) + L(9) * L(14) * L(3) * L(8) +
        L(13) * L(2) * L(11) * L(8) - L(1) * L(14) * L(11) * L(8) -
        L(9) * L(2) * L(15) * L(8) + L(1) * L(10) * L(15) * L(8) +
        L(13) * L(6) * L(3) * L(12) - L(5) * L(14) * L(3) * L(12) -
        L(13) * L(2) * L(7) * L(12) + L(1) * L(14) * L(7) * L(12) +
        L(5) * L(2) * L(15) * L(12) - L(1) * L(6) * L(15) * L(12) -
        L(9) * L(6) * L(3) * L(16) + L(5) * L(10) * L(3) * L(16) +
        L(9) * L(2) * L(7) * L(16) - L(1) * L(10) * L(7) * L(16) -
        L(5) * L(2) * L(11) * L(16) + L(1) * L(6) * L(11) * L(16);
#undef L
    double idet = 1.0 / det;
#define B(i, j) b[((i) - 1) + 4 * ((j) - 1)]
    B(1, 1) = (A(2,3)*A(3,4)*A(4,2) - A(2,4)*A(3,3)*A(4,2) + A(2,4)*A(3,2)*A(4,3) - A(2,2)*A(3,4)*A(4,3) - A(2,3)*A(3,2)*A(4,4) + A(2,2)*A(3,3)*A(4,4)) * idet;
    B(2, 1) = (A(2,4)*A(3,3)*A(4,1) - A(2,3)*A(3,4)*A(4,1) - A(2,4)*A(3,1)*A(4,3) + A(2,1)*A(3,4)*A(4,3) + A(2,3)*A(3,1)*A(4,4) - A(2,1)*A(3,3)*A(4,4)) * idet;
    B(3, 1) = (A(2,2)*A(3,4)*A(4,1) - A(2,4)*A(3,2)*A(4,1) + A(2,4)*A(3,1)*A(4,2) - A(2,1)*A(3,4)*A(4,2) - A(2,2)*A(3,1)*A(4,4) + A(2,1)*A(3,2)*A(4,4)) * idet;
    B(4, 1) = (A(2,3)*A(3,2)*A(4,1) - A(2,2)*A(3,3)*A(4,1) - A(2,3)*A(3,1)*A(4,2) + A(2,1)*A(3,3)*A(4,2) + A(2,2)*A(3,1)*A(4,3) - A(2,1)*A(3,2)*A(4,3)) * idet;
    B(1, 2) = (A(1,4)*A(3,3)*A(4,2) - A(1,3)*A(3,4)*A(4,2) - A(1,4)*A(3,2)*A(4,3) + A(1,2)*A(3,4)*A(4,3) + A(1,3)*A(3,2)*A(4,4) - A(1,2)*A(3,3)*A(4,4)) * idet;
    B(2, 2) = (A(1,3)*A(3,4)*A(4,1) - A(1,4)*A(3,3)*A(4,1) + A(1,4)*A(3,1)*A(4,3) - A(1,1)*A(3,4)*A(4,3) - A(1,3)*A(3,1)*A(4,4) + A(1,1)*A(3,3)*A(4,4)) * idet;
    B(3, 2) = (A(1,4)*A(3,2)*A(4,1) - A(1,2)*A(3,4)*A(4,1) - A(1,4)*A(3,1)*A(4,2) + A(1,1)*A(3,4)*A(4,2) + A(1,2)*A(3,1)*A(4,4) - A(1,1)*A(3,2)*A(4,4)) * idet;
    B(4, 2) = (A(1,2)*A(3,3)*A(4,1) - A(1,3)*A(3,2)*A(4,1) + A(1,3)*A(3,1)*A(4,2) - A(1,1)*A(3,3)*A(4,2) - A(1,2)*A(3,1)*A(4,3) + A(1,1)*A(3,2)*A(4,3)) * idet;
    B(1, 3) = (A(1,3)*A(2,4)*A(4,2) - A(1,4)*A(2,3)*A(4,2) + A(1,4)*A(2,2)*A(4,3) - A(1,2)*A(2,4)*A(4,3) - A(1,3)*A(2,2)*A(4,4) + A(1,2)*A(2,3)*A(4,4)) * idet;
    B(2, 3) = (A(1,4)*A(2,3)*A(4,1) - A(1,3)*A(2,4)*A(4,1) - A(1,4)*A(2,1)*A(4,3) + A(1,1)*A(2,4)*A(4,3) + A(1,3)*A(2,1)*A(4,4) - A(1,1)*A(2,3)*A(4,4)) * idet;
    B(3, 3) = (A(1,2)*A(2,4)*A(4,1) - A(1,4)*A(2,2)*A(4,1) + A(1,4)*A(2,1)*A(4,2) - A(1,1)*A(2,4)*A(4,2) - A(1,2)*A(2,1)*A(4,4) + A(1,1)*A(2,2)*A(4,4)) * idet;
    B(4, 3) = (A(1,3)*A(2,2)*A(4,1) - A(1,2)*A(2,3)*A(4,1) - A(1,3)*A(2,1)*A(4,2) + A(1,1)*A(2,3)*A(4,2) + A(1,2)*A(2,1)*A(4,3) - A(1,1)*A(2,2)*A(4,3)) * idet;
    B(1, 4) = (A(1,4)*A(2,3)*A(3,2) - A(1,3)*A(2,4)*A(3,2) - A(1,4)*A(2,2)*A(3,3) + A(1,2)*A(2,4)*A(3,3) + A(1,3)*A(2,2)*A(3,4) - A(1,2)*A(2,3)*A(3,4)) * idet;
    B(2, 4) = (A(1,3)*A(2,4)*A(3,1) - A(1,4)*A(2,3)*A(3,1) + A(1,4)*A(2,1)*A(3,3) - A(1,1)*A(2,4)*A(3,3) - A(1,3)*A(2,1)*A(3,4) + A(1,1)*A(2,3)*A(3,4)) * idet;
    B(3, 4) = (A(1,4)*A(2,2)*A(3,1) - A(1,2)*A(2,4)*A(3,1) - A(1,4)*A(2,1)*A(3,2) + A(1,1)*A(2,4)*A(3,2) + A(1,2)*A(2,1)*A(3,4) - A(1,1)*A(2,2)*A(3,4)) * idet;
    B(4, 4) = (A(1,2)*A(2,3)*A(3,1) - A(1,3)*A(2,2)*A(3,1) + A(1,3)*A(2,1)*A(3,2) - A(1,1)*A(2,3)*A(3,2) - A(1,2)*A(2,1)*A(3,3) + A(1,1)*A(2,2)*A(3,3)) * idet;
#undef A
#undef B
    return isfinite(idet) ? 0 : 1;
}

/* Variant 2: Gauss-Jordan elimination with partial pivoting on [A | I] (what a LAPACK-style getrf/getri pair amounts to
 * for a 4x4: a third, structurally different rounding). */
static int inv4_lu(const double a[16], double b[16])
{
    double M[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { M[i][j] = a[i + 4 * j]; M[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int k = 0; k < 4; ++k) {
        int p = k;
        for (int i = k + 1; i < 4; ++i)
            if (fabs(M[i][k]) > fabs(M[p][k])) p = i;
        if (M[p][k] == 0.0) return 1;
        if (p != k)
            for (int j = 0; j < 8; ++j) { double t = M[k][j]; M[k][j] = M[p][j]; M[p][j] = t; }
        double piv = M[k][k];
        for (int j = 0; j < 8; ++j) M[k][j] = M[k][j] / piv;
        for (int i = 0; i < 4; ++i) {
            if (i == k) continue;
            double f = M[i][k];
            for (int j = 0; j < 8; ++j) M[i][j] = M[i][j] - f * M[k][j];
        }
    }
    int bad = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { b[i + 4 * j] = M[i][4 + j]; bad |= !isfinite(M[i][4 + j]); }
    return bad;
}

/* The inverse every caller of the oracle goes through.  Variant 1 (default since round 3) is the form StaticArrays 0.10.3
 * published and the HIP path's form (k_prep_tet / inv4, pfc_kernels.h); 0 (rounds 1-2: 2x2 minors, the form later
 * StaticArrays versions use) and 2 exist to MEASURE the path's exposure to the one rounding the reference takes from an
 * un-vendored dependency. */
static int g_inv4_variant = 1;
int pfo_set_inv4_variant(int v)
{
    if (v < 0 || v > 2) return PFO_ERR_BAD_ARG;
    g_inv4_variant = v;
    return PFO_OK;
}
int pfo_inv4(const double a[16], double b[16])
{
    if (g_inv4_variant == 1) return inv4_cofactor(a, b);
    if (g_inv4_variant == 2) return inv4_lu(a, b);
    return inv4_minors(a, b);
}

/* The reference's tests run with set_zero_subnormals(true) (test/runtests.jl:13): flush-to-zero + denormals-are-zero in
 * MXCSR for the duration of an evaluation (per calling thread), restored afterwards.  Off by default. */
#if defined(__x86_64__) || defined(__i386__)
#include <xmmintrin.h>
static int g_ftz = 0;
int pfo_set_ftz(int on) { g_ftz = on != 0; return PFO_OK; }
static unsigned ftz_enter(void)
{
    unsigned old = _mm_getcsr();
    if (g_ftz) _mm_setcsr(old | 0x8040u);   /* FTZ (bit 15) | DAZ (bit 6) */
    return old;
}
static void ftz_leave(unsigned old) { _mm_setcsr(old); }
#else
int pfo_set_ftz(int on) { return on ? PFO_ERR_BAD_ARG : PFO_OK; }
static unsigned ftz_enter(void) { return 0; }
static void ftz_leave(unsigned old) { (void)old; }
#endif

/* ------------------------------------------------------------------------------------------------------------ */
/* math kernel                                                                                                    */
/* ------------------------------------------------------------------------------------------------------------ */
/* src/math_kernel/utility.jl:21-26 */
void pfo_weight_poly(int n, const double *p1, const double *p2, double w1, double w2, double *out)
{
    double sw = w1 - w2, c1 = w1 / sw, c2 = w2 / sw;
    for (int k = 0; k < n; ++k) out[k] = c1 * p2[k] - c2 * p1[k];
}
static inline v4 weight_poly4(v4 p1, v4 p2, double w1, double w2)
{
    v4 r;
    pfo_weight_poly(4, p1.c, p2.c, w1, w2, r.c);
    return r;
}
static inline v3 weight_poly3(v3 p1, v3 p2, double w1, double w2)
{
    double sw = w1 - w2, c1 = w1 / sw, c2 = w2 / sw;
    return V3(c1 * p2.x - c2 * p1.x, c1 * p2.y - c2 * p1.y, c1 * p2.z - c2 * p1.z);
}

/* src/math_kernel/vector_projections.jl:9-13 */
double pfo_a_dot_one_pad_b(const double a[4], const double b[3])
{
    double d = fma(a[0], b[0], a[3]);
    d = fma(a[1], b[1], d);
    return fma(a[2], b[2], d);
}

/* src/math_kernel/vector_projections.jl:2-7 */
static inline v3 vec_sub_vec_proj(v3 v, v3 n)
{
    double t = -dot3(v, n);
    return V3(fma(t, n.x, v.x), fma(t, n.y, v.y), fma(t, n.z, v.z));
}
void pfo_vec_sub_vec_proj(const double v[3], const double n[3], double out[3]) { st3(out, vec_sub_vec_proj(ld3(v), ld3(n))); }

/* src/math_kernel/geometry_kernel.jl:5-10 */
static inline v3 vector_area(v3 a, v3 b, v3 c) { return scl3(cross3(sub3(b, a), sub3(c, b)), 0.5); }
static inline double triangle_area(v3 a, v3 b, v3 c, v3 n) { return dot3(n, vector_area(a, b, c)); }
static inline v3 triangle_normal(v3 a, v3 b, v3 c) { return normalize3(vector_area(a, b, c)); }
/* :4 — (v1 + v2 + v3) * Float64(1/3) */
static inline v3 centroid3(v3 a, v3 b, v3 c) { return scl3(add3(add3(a, b), c), 1.0 / 3.0); }

double pfo_triangle_area(const double v[9], const double n[3]) { return triangle_area(ld3(v), ld3(v + 3), ld3(v + 6), ld3(n)); }
void pfo_triangle_normal(const double v[9], double n[3]) { st3(n, triangle_normal(ld3(v), ld3(v + 3), ld3(v + 6))); }

/* src/math_kernel/geometry_kernel.jl:22-38 */
double pfo_volume(const double v[12])
{
    double a1 = v[0], a2 = v[1], a3 = v[2], b1 = v[3], b2 = v[4], b3 = v[5];
    double c1 = v[6], c2 = v[7], c3 = v[8], d1 = v[9], d2 = v[10], d3 = v[11];
    double V = (b1 - a1) * (c2 * d3 - c3 * d2);
    V = fma(b2 - a2, c3 * d1 - c1 * d3, V);
    V = fma(b3 - a3, c1 * d2 - c2 * d1, V);
    V = fma(c1 - d1, a3 * b2 - a2 * b3, V);
    V = fma(c2 - d2, a1 * b3 - a3 * b1, V);
    V = fma(c3 - d3, a2 * b1 - a1 * b2, V);
    return V * (1.0 / 6.0);
}

/* src/clip/quadrature.jl:21-41 — literal decimals of the reference */
int pfo_tri_quad_rule(int n_rule, double *zeta, double *w)
{
    if (n_rule == 1) {
        zeta[0] = zeta[1] = zeta[2] = 0.33333333333333331483;
        w[0] = 1.00000000000000000000;
        return 1;
    }
    if (n_rule == 2) {
        const double a = 0.16666666666666674068, b = 0.66666666666666651864;
        const double z[9] = {a, b, a, b, a, a, a, a, b};
        memcpy(zeta, z, sizeof z);
        w[0] = w[1] = w[2] = 0.33333333333333331483;
        return 3;
    }
    return -1; /* ContactInstructions restricts to rules 1 and 2: src/mechanism_scenario.jl:45 */
}

/* ------------------------------------------------------------------------------------------------------------ */
/* clip: Sutherland-Hodgman in tet coordinates, src/clip/static_clip.jl                                          */
/* ------------------------------------------------------------------------------------------------------------ */
typedef struct { int n; v4 v[8]; } poly4;
typedef struct { int n; v3 v[8]; } poly3;

/* clip_node :197-201 */
static inline v4 clip_node(v4 z_non, v4 z_pos, int i) { return weight_poly4(z_non, z_pos, z_non.c[i], z_pos.c[i]); }

/* cut_clip for arity n (:135-195).  z[0] is non-positive and z[1] positive on plane i.  Writes the cut polygon to
 * out and returns its arity; *final is set when the reference returns the polygon without visiting the
 * remaining planes (7-vertex method, :185-195). */
static int cut_clip(const v4 *z, int n, int i, v4 *out, int *final)
{
    /* (z_{n-1}[i] <= 0) && return cut_clip(z_1..z_{n-1}) : the trailing vertex is dropped (:148,160,172,186) */
    while (n > 3 && z[n - 2].c[i] <= 0.0) --n;
    v4 z_start = clip_node(z[0], z[1], i);
    double last = z[n - 1].c[i];
    /* inside test of the last vertex: strict for arity 3..5 (:140,150,162), non-strict for 6..7 (:176,188) */
    int inside = (n <= 5) ? (0.0 < last) : (0.0 <= last);
    int m;
    out[0] = z_start;
    if (inside) {
        for (int k = 1; k < n; ++k) out[k] = z[k];
        out[n] = clip_node(z[0], z[n - 1], i);
        m = n + 1;
    } else {
        for (int k = 1; k < n - 1; ++k) out[k] = z[k];
        out[n - 1] = clip_node(z[n - 1], z[n - 2], i);
        m = n;
    }
    *final = (n == 7);
    return m;
}

/* clip (:34-128) iterated over planes i = 0..3 (the reference recurses with i+1) */
static int clip_poly(poly4 *p)
{
    v4 z[8], rot[8], out[8];
    int n = p->n;
    for (int k = 0; k < n; ++k) z[k] = p->v[k];
    for (int i = 0; i < 4; ++i) {
        int all_non_pos = 1, all_non_neg = 1;
        for (int k = 0; k < n; ++k) {
            all_non_pos &= (z[k].c[i] <= 0.0);
            all_non_neg &= (0.0 <= z[k].c[i]);
        }
        if (all_non_pos) { p->n = 0; return 0; }
        if (all_non_neg) continue;
        int start = -1;
        for (int k = 0; k < n; ++k)
            if (z[k].c[i] <= 0.0 && !(z[(k + 1) % n].c[i] <= 0.0)) { start = k; break; }
        if (start < 0) return PFO_ERR_NONFINITE; /* "Non-finite vertex likely" :52 */
        for (int k = 0; k < n; ++k) rot[k] = z[(start + k) % n];
        int final = 0;
        n = cut_clip(rot, n, i, out, &final);
        for (int k = 0; k < n; ++k) z[k] = out[k];
        if (final) break;
    }
    p->n = n;
    for (int k = 0; k < n; ++k) p->v[k] = z[k];
    for (int k = n; k < 8; ++k) p->v[k] = z[0]; /* unused slots = vertex 1: src/clip/poly_eight.jl:17-22 */
    return 0;
}

int pfo_clip_in_tet_coordinates(int n_in, const double *z_in, double *z_out)
{
    poly4 p;
    if (n_in != 3 && n_in != 4) return -PFO_ERR_BAD_ARG; /* "something is wrong" :13 */
    p.n = n_in;
    for (int k = 0; k < n_in; ++k) memcpy(p.v[k].c, z_in + 4 * k, 4 * sizeof(double));
    if (clip_poly(&p)) return -PFO_ERR_NONFINITE;
    for (int k = 0; k < 8; ++k) memcpy(z_out + 4 * k, p.v[k < p.n ? k : 0].c, 4 * sizeof(double));
    return p.n;
}

/* src/clip/poly_eight.jl:106-126 */
void pfo_zero_small_coordinates(int n, double *z)
{
    for (int k = 0; k < 4 * n; ++k) z[k] = z[k] * ((1.0e-14 < fabs(z[k])) ? 1.0 : 0.0);
}

/* centroid(poly, n̂): src/clip/poly_eight.jl:35-52.  Returns the signed area, writes the centroid. */
static double poly_centroid(const poly3 *p, v3 nh, v3 *c)
{
    v3 a = p->v[0], cc = p->v[1];
    double cum_sum = 0.0;
    v3 cum_prod = V3(0, 0, 0);
    for (int k = 2; k < p->n; ++k) {
        v3 b = cc;
        cc = p->v[k];
        double ar = triangle_area(a, b, cc, nh);
        cum_prod = add3(cum_prod, scl3(centroid3(a, b, cc), ar));
        cum_sum += ar;
    }
    *c = (cum_sum == 0.0) ? a : div3(cum_prod, cum_sum);
    return cum_sum;
}
double pfo_poly_centroid(int n, const double *v, const double nhat[3], double c[3])
{
    poly3 p;
    v3 cc;
    p.n = n;
    for (int k = 0; k < 8; ++k) p.v[k] = ld3(v + 3 * k);
    double a = poly_centroid(&p, ld3(nhat), &cc);
    st3(c, cc);
    return a;
}

/* clip_plane_tet: src/clip/plane_tet_intersection.jl:9-106.  tet is the 4x4 [v1 v2 v3 v4; 1 1 1 1] column-major. */
static int clip_plane_tet(const double plane[4], const m4 *tet, poly3 *out)
{
    double proj[4];
    v3 v[4];
    int neg[4], pos[4], n_neg = 0, n_pos = 0;
    for (int j = 0; j < 4; ++j) {
        proj[j] = ((plane[0] * tet->m[4 * j] + plane[1] * tet->m[1 + 4 * j]) + plane[2] * tet->m[2 + 4 * j]) +
                  plane[3] * tet->m[3 + 4 * j];
        v[j] = V3(tet->m[4 * j], tet->m[1 + 4 * j], tet->m[2 + 4 * j]);
        neg[j] = proj[j] < 0.0;
        pos[j] = 0.0 < proj[j];
        n_neg += neg[j];
        n_pos += pos[j];
    }
    out->n = 0;
    if (n_pos == 0 || n_neg == 0) return 0;
#define PW(i1, i2) weight_poly3(v[i1], v[i2], proj[i1], proj[i2])
    int lone = -1;
    if (n_pos == 1) { for (int j = 0; j < 4; ++j) if (pos[j]) { lone = j; break; } }
    else if (n_neg == 1) { for (int j = 0; j < 4; ++j) if (neg[j]) { lone = j; break; } }
    if (lone >= 0) {
        /* :52-79 (0-based pairs) */
        static const int tab[4][3] = {{1, 3, 2}, {0, 2, 3}, {0, 3, 1}, {0, 1, 2}};
        v3 a = PW(tab[lone][0], lone), b = PW(tab[lone][1], lone), c = PW(tab[lone][2], lone);
        out->n = 3;
        if (0.0 < proj[lone]) { out->v[0] = a; out->v[1] = b; out->v[2] = c; }
        else { out->v[0] = c; out->v[1] = b; out->v[2] = a; }
    } else {
        v3 a, b, c, d;
        if (pos[0] == pos[1]) { a = PW(1, 2); b = PW(1, 3); c = PW(0, 3); d = PW(0, 2); }      /* _12 :81-88 */
        else if (pos[0] == pos[2]) { a = PW(0, 1); b = PW(0, 3); c = PW(2, 3); d = PW(2, 1); } /* _13 :90-97 */
        else { a = PW(0, 2); b = PW(0, 1); c = PW(3, 1); d = PW(3, 2); }                       /* _14 :99-106 */
        out->n = 4;
        if (0.0 < proj[0]) { out->v[0] = a; out->v[1] = b; out->v[2] = c; out->v[3] = d; }
        else { out->v[0] = d; out->v[1] = c; out->v[2] = b; out->v[3] = a; }
    }
#undef PW
    for (int k = out->n; k < 8; ++k) out->v[k] = out->v[0];
    return out->n;
}
int pfo_clip_plane_tet(const double plane[4], const double tet_cm[16], double *o)
{
    m4 t;
    poly3 p;
    memcpy(t.m, tet_cm, sizeof t.m);
    clip_plane_tet(plane, &t, &p);
    for (int k = 0; k < p.n; ++k) st3(o + 3 * k, p.v[k]);
    return p.n;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* OBB-OBB SAT, src/obb/bb_intersection.jl                                                                        */
/* ------------------------------------------------------------------------------------------------------------ */
static int sat(const double ea[3], const double eb[3], const double t[3], const double R[9], const double aR[9])
{
    /* rows of R (R0_012 = R[1],R[4],R[7] in 1-based linear indexing): :21-26 */
#define Rr(i, j) R[(i) + 3 * (j)]
#define Ar(i, j) aR[(i) + 3 * (j)]
    /* face test 1/2 :29-32 */
    for (int i = 0; i < 3; ++i) {
        double rb = (Ar(i, 0) * eb[0] + Ar(i, 1) * eb[1]) + Ar(i, 2) * eb[2];
        if ((ea[i] + rb) < fabs(t[i])) return 0;
    }
    /* face test 2/2 :35-38 */
    for (int j = 0; j < 3; ++j) {
        double tl = fabs((Rr(0, j) * t[0] + Rr(1, j) * t[1]) + Rr(2, j) * t[2]);
        double ra = (Ar(0, j) * ea[0] + Ar(1, j) * ea[1]) + Ar(2, j) * ea[2];
        if ((ra + eb[j]) < tl) return 0;
    }
    /* s100(r) = (r2, r1, r1), s221(r) = (r3, r3, r2) in 1-based terms: :14-15 */
    static const int i100[3] = {1, 0, 0}, i221[3] = {2, 2, 1};
    /* cross test 1/3 :56-60 : L = A0 x Bj */
    for (int j = 0; j < 3; ++j) {
        double tl = fabs(t[2] * Rr(1, j) - t[1] * Rr(2, j));
        double ra = ea[1] * Ar(2, j) + ea[2] * Ar(1, j);
        double rb = eb[i100[j]] * Ar(0, i221[j]) + eb[i221[j]] * Ar(0, i100[j]);
        if ((ra + rb) < tl) return 0;
    }
    /* cross test 2/3 :62-66 */
    for (int j = 0; j < 3; ++j) {
        double tl = fabs(t[0] * Rr(2, j) - t[2] * Rr(0, j));
        double ra = ea[0] * Ar(2, j) + ea[2] * Ar(0, j);
        double rb = eb[i100[j]] * Ar(1, i221[j]) + eb[i221[j]] * Ar(1, i100[j]);
        if ((ra + rb) < tl) return 0;
    }
    /* cross test 3/3 :68-72 */
    for (int j = 0; j < 3; ++j) {
        double tl = fabs(t[1] * Rr(0, j) - t[0] * Rr(1, j));
        double ra = ea[0] * Ar(1, j) + ea[1] * Ar(0, j);
        double rb = eb[i100[j]] * Ar(2, i221[j]) + eb[i221[j]] * Ar(2, i100[j]);
        if ((ra + rb) < tl) return 0;
    }
#undef Rr
#undef Ar
    return 1;
}

/* BB_BB_intersect(tt, a, b): :2-12 — three homogeneous transforms composed per node pair */
int pfo_bb_bb_intersect(const double ca[3], const double ea[3], const double Ra[9], const double cb[3],
                        const double eb[3], const double Rb[9], const double R_a_b[9], const double t_a_b[3])
{
    double RaT[9], nt[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) RaT[i + 3 * j] = Ra[j + 3 * i];
    /* -a.R' * a.c  ==  (-(R')) * c */
    for (int i = 0; i < 3; ++i)
        nt[i] = ((-RaT[i]) * ca[0] + (-RaT[i + 3]) * ca[1]) + (-RaT[i + 6]) * ca[2];
    m4 i_dh_a = dh_from_Rt(RaT, nt);
    m4 dh_a_b = dh_from_Rt(R_a_b, t_a_b);
    m4 dh_b = dh_from_Rt(Rb, cb);
    m4 tmp = mul44(&i_dh_a, &dh_a_b);
    m4 fin = mul44(&tmp, &dh_b);
    double R[9], aR[9], t[3];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) {
            R[i + 3 * j] = fin.m[i + 4 * j];
            aR[i + 3 * j] = fabs(R[i + 3 * j]) + 1.0e-14;
        }
    t[0] = fin.m[12]; t[1] = fin.m[13]; t[2] = fin.m[14];
    return sat(ea, eb, t, R, aR);
}

/* ------------------------------------------------------------------------------------------------------------ */
/* evaluation context                                                                                             */
/* ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    const pfo_mesh *m1, *m2;
    const pfo_ins *ins;
    m4 x21, x12;             /* x_r2_r1.mat, x_r1_r2.mat */
    double R21[9], R12[9], t12[3];
    v3 w, vlin;              /* twist_r2_r1_r2 */
    double chi, Ebar;
    int nq;
    double qz[9], qw[3];
    pfo_debug *d;
    int err;
} ctx;

pfo_debug *pfo_debug_new(void)
{
    pfo_debug *d = (pfo_debug *)calloc(1, sizeof *d);
    if (!d) return NULL;
    d->cap_pair = 64; /* VectorCache starts at 64 and doubles: src/obb/vector_cache.jl:7,13-17 */
    d->cap_trac = 64;
    d->pair = (int *)malloc(sizeof(int) * 2 * d->cap_pair);
    d->clip_n = (int *)malloc(sizeof(int) * d->cap_pair);
    d->trac = (pfo_trac *)malloc(sizeof(pfo_trac) * d->cap_trac);
    if (!d->pair || !d->clip_n || !d->trac) { pfo_debug_free(d); return NULL; }
    return d;
}
void pfo_debug_free(pfo_debug *d)
{
    if (!d) return;
    free(d->pair); free(d->clip_n); free(d->trac); free(d);
}
static void add_pair(ctx *c, int i1, int i2)
{
    pfo_debug *d = c->d;
    if (d->n_pair == d->cap_pair) {
        d->cap_pair += d->cap_pair;
        d->pair = (int *)realloc(d->pair, sizeof(int) * 2 * d->cap_pair);
        d->clip_n = (int *)realloc(d->clip_n, sizeof(int) * d->cap_pair);
        if (!d->pair || !d->clip_n) { c->err = PFO_ERR_NOMEM; return; }
    }
    d->pair[2 * d->n_pair] = i1;
    d->pair[2 * d->n_pair + 1] = i2;
    d->clip_n[d->n_pair] = 0;
    d->n_pair++;
}
static void add_trac(ctx *c, v3 n, v3 r, double dA, double p)
{
    pfo_debug *d = c->d;
    if (d->n_trac == d->cap_trac) {
        d->cap_trac += d->cap_trac;
        d->trac = (pfo_trac *)realloc(d->trac, sizeof(pfo_trac) * d->cap_trac);
        if (!d->trac) { c->err = PFO_ERR_NOMEM; return; }
    }
    pfo_trac *t = &d->trac[d->n_trac++];
    st3(t->n, n); st3(t->r, r); t->dA = dA; t->p = p;
}

/* tree_tree_intersect: src/obb/tree_types.jl:88-111 */
static void tree_tree(ctx *c, int a, int b)
{
    const pfo_mesh *m1 = c->m1, *m2 = c->m2;
    if (c->err) return;
    c->d->n_node_tests++;
    if (!pfo_bb_bb_intersect(m1->node_c + 3 * a, m1->node_e + 3 * a, m1->node_R + 9 * a, m2->node_c + 3 * b,
                             m2->node_e + 3 * b, m2->node_R + 9 * b, c->R12, c->t12))
        return;
    int la = m1->node_leaf[a] != PFO_INTERNAL, lb = m2->node_leaf[b] != PFO_INTERNAL;
    const int *ca = m1->node_child + 2 * a, *cb = m2->node_child + 2 * b;
    if (la) {
        if (lb) add_pair(c, m1->node_leaf[a], m2->node_leaf[b]);
        else { tree_tree(c, a, cb[0]); tree_tree(c, a, cb[1]); }
    } else {
        if (lb) { tree_tree(c, ca[0], b); tree_tree(c, ca[1], b); }
        else {
            tree_tree(c, ca[0], cb[0]); tree_tree(c, ca[1], cb[0]);
            tree_tree(c, ca[0], cb[1]); tree_tree(c, ca[1], cb[1]);
        }
    }
}

/* tetrahedron_vertices_ϵ + calc_ζ_transforms: src/contact_algorithms_non_friction.jl:150-162 */
static int tet_transforms(const pfo_mesh *m, int i_tet, m4 *x_r_z, m4 *x_z_r, double eps[4])
{
    const int *it = m->tet + 4 * i_tet;
    for (int j = 0; j < 4; ++j) {
        const double *p = m->pt + 3 * it[j];
        x_r_z->m[4 * j] = p[0]; x_r_z->m[1 + 4 * j] = p[1]; x_r_z->m[2 + 4 * j] = p[2]; x_r_z->m[3 + 4 * j] = 1.0;
        eps[j] = m->eps[it[j]];
    }
    return pfo_inv4(x_r_z->m, x_z_r->m);
}

/* 1x4 row times 4x4 */
static void row_mul44(const double r[4], const m4 *M, double out[4])
{
    for (int j = 0; j < 4; ++j)
        out[j] = ((r[0] * M->m[4 * j] + r[1] * M->m[1 + 4 * j]) + r[2] * M->m[2 + 4 * j]) + r[3] * M->m[3 + 4 * j];
}

/* fillTractionCacheForTriangle! / fillTractionCacheInnerLoop!: src/contact_algorithms_non_friction.jl:236-265 */
static void fill_triangle(ctx *c, double area, v3 nh, v3 a0, v3 a1, v3 a2, const double eps_r[4])
{
    for (int k = 0; k < c->nq; ++k) {
        const double *z = c->qz + 3 * k;
        v3 r = V3((a0.x * z[0] + a1.x * z[1]) + a2.x * z[2], (a0.y * z[0] + a1.y * z[1]) + a2.y * z[2],
                  (a0.z * z[0] + a1.z * z[1]) + a2.z * z[2]);
        double rr[3] = {r.x, r.y, r.z};
        double eps_quad = pfo_a_dot_one_pad_b(eps_r, rr);
        v3 rdot = add3(c->vlin, cross3(c->w, r));
        double ee = -dot3(V3(eps_r[0], eps_r[1], eps_r[2]), rdot);
        double damp = fmax(0.0, 1.0 + c->chi * ee);
        double p = eps_quad * c->Ebar * damp;
        double dA = c->qw[k] * area;
        if (0.0 < p) add_trac(c, nh, r, dA, p);
    }
}

/* integrate_over_polygon_patch!: :217-234 */
static void integrate_patch(ctx *c, v3 nh, const poly4 *pz, const m4 *x_r_z, const double eps_r[4])
{
    poly3 pr;
    pr.n = pz->n;
    for (int k = 0; k < 8; ++k) {
        v4 r = mul4v(x_r_z, pz->v[k < pz->n ? k : 0]);
        pr.v[k] = V3(r.c[0], r.c[1], r.c[2]);
    }
    v3 cen;
    poly_centroid(&pr, nh, &cen);
    int N = pz->n;
    v3 v2 = pr.v[N - 1];
    for (int k = 0; k < N; ++k) {
        v3 v1 = v2;
        v2 = pr.v[k];
        double area = triangle_area(v1, v2, cen, nh);
        if (0.0 < area) fill_triangle(c, area, nh, v1, v2, cen, eps_r);
    }
}

/* tri-tet op: :196-215 */
static void op_tri_tet(ctx *c, int k_pair, int i1, int i2)
{
    const pfo_mesh *m1 = c->m1, *m2 = c->m2;
    const int *it = m1->tri + 3 * i1;
    v3 t0 = ld3(m1->pt + 3 * it[0]), t1 = ld3(m1->pt + 3 * it[1]), t2 = ld3(m1->pt + 3 * it[2]);
    m4 x_r_z, x_z_r;
    double eps2[4], eps_r[4];
    if (tet_transforms(m2, i2, &x_r_z, &x_z_r, eps2)) { c->err = PFO_ERR_NONFINITE; return; }
    row_mul44(eps2, &x_z_r, eps_r);
    m4 x_z_r1 = mul44(&x_z_r, &c->x21);
    poly4 p;
    p.n = 3;
    v4 a = {{t0.x, t0.y, t0.z, 1.0}}, b = {{t1.x, t1.y, t1.z, 1.0}}, d = {{t2.x, t2.y, t2.z, 1.0}};
    p.v[0] = mul4v(&x_z_r1, a); p.v[1] = mul4v(&x_z_r1, b); p.v[2] = mul4v(&x_z_r1, d);
    if (clip_poly(&p)) { c->err = PFO_ERR_NONFINITE; return; }
    c->d->clip_n[k_pair] = p.n;
    if (3 <= p.n) {
        v3 n1 = triangle_normal(t0, t1, t2);
        const double *R = c->R21; /* transform(FreeVector3D, x_r2_r1) = R * v */
        v3 nh = V3((R[0] * n1.x + R[3] * n1.y) + R[6] * n1.z, (R[1] * n1.x + R[4] * n1.y) + R[7] * n1.z,
                   (R[2] * n1.x + R[5] * n1.y) + R[8] * n1.z);
        integrate_patch(c, nh, &p, &x_r_z, eps_r);
    }
}

/* tet-tet op: :166-194 */
static void op_tet_tet(ctx *c, int k_pair, int i1, int i2)
{
    m4 x_r1_z1, x_z1_r1, x_r2_z2, x_z2_r2;
    double e1[4], e2[4], eps_r[4];
    if (tet_transforms(c->m1, i1, &x_r1_z1, &x_z1_r1, e1) || tet_transforms(c->m2, i2, &x_r2_z2, &x_z2_r2, e2)) {
        c->err = PFO_ERR_NONFINITE;
        return;
    }
    /* find_plane_tet(E, ϵ, X) = (E * ϵ) * X : :164,174-177 */
    m4 X1 = mul44(&x_z1_r1, &c->x12);
    double Ee1[4], Ee2[4], pl1[4], pl2[4], plane[4];
    for (int j = 0; j < 4; ++j) { Ee1[j] = c->m1->Ebar * e1[j]; Ee2[j] = c->m2->Ebar * e2[j]; }
    row_mul44(Ee1, &X1, pl1);
    row_mul44(e2, &x_z2_r2, eps_r);
    row_mul44(Ee2, &x_z2_r2, pl2);
    for (int j = 0; j < 4; ++j) plane[j] = pl2[j] - pl1[j];
    m4 x_r2_z1 = mul44(&c->x21, &x_r1_z1);
    poly3 pr;
    clip_plane_tet(plane, &x_r2_z1, &pr);
    if (3 <= pr.n) {
        poly4 pz;
        pz.n = pr.n;
        for (int k = 0; k < 8; ++k) {
            v3 q = pr.v[k < pr.n ? k : 0];
            v4 o = {{q.x, q.y, q.z, 1.0}};
            pz.v[k] = mul4v(&x_z2_r2, o);
        }
        for (int k = 0; k < 8; ++k) pfo_zero_small_coordinates(1, pz.v[k].c);
        if (clip_poly(&pz)) { c->err = PFO_ERR_NONFINITE; return; }
        c->d->clip_n[k_pair] = pz.n;
        if (3 <= pz.n) {
            v3 nh = normalize3(V3(plane[0], plane[1], plane[2]));
            integrate_patch(c, nh, &pz, &x_r2_z2, eps_r);
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------ */
/* friction, src/contact_algorithms_friction.jl                                                                   */
/* ------------------------------------------------------------------------------------------------------------ */
/* :2-10 ; clamp(y, lo, hi) */
double pfo_calc_clamped_piecewise(double x, double x1, double x2, double y1, double y2)
{
    double k = (y2 - y1) / (x2 - x1);
    double y = y1 + (x - x1) * k;
    return (y > y1) ? y1 : ((y < y2) ? y2 : y);
}

/* :13-30 */
static v3 traction_reg(double mu_s, double mu_d, double v_c, v3 vt, double p_dA)
{
    double m2 = dot3(vt, vt);
    v3 T;
    if (m2 < v_c * v_c) {
        T = div3(scl3(vt, -mu_s), v_c);
    } else {
        double m = sqrt(m2);
        double mu = pfo_calc_clamped_piecewise(m, 2 * v_c, 3 * v_c, mu_s, mu_d); /* v_μs, v_μd: mechanism_scenario.jl:30-31 */
        T = div3(scl3(vt, -mu), m);
    }
    return scl3(T, p_dA);
}
void pfo_traction_regularized(double mu_s, double mu_d, double v_c, const double vel_t[3], double p_dA, double out[3])
{
    st3(out, traction_reg(mu_s, mu_d, v_c, ld3(vel_t), p_dA));
}

/* :32-48 */
static v3 traction_bri(double mu_s, double mu_d, v3 Ts, double p_dA)
{
    double m2 = dot3(Ts, Ts);
    v3 T;
    if (m2 < mu_s * mu_s) {
        T = Ts;
    } else {
        double m = sqrt(m2);
        double mu = pfo_calc_clamped_piecewise(m, 2 * mu_s, 3 * mu_s, mu_s, mu_d); /* T̄s_μs, T̄s_μd: :16-17 */
        T = div3(scl3(Ts, mu), m);
    }
    return scl3(T, p_dA);
}
void pfo_traction_bristle(double mu_s, double mu_d, const double Ts[3], double p_dA, double out[3])
{
    st3(out, traction_bri(mu_s, mu_d, ld3(Ts), p_dA));
}

/* yes_contact!(::Regularized): :50-72 */
static void yes_contact_regularized(ctx *c, double wrench[6])
{
    const pfo_ins *in = c->ins;
    v3 lin = V3(0, 0, 0), ang = V3(0, 0, 0);
    for (int k = 0; k < c->d->n_trac; ++k) {
        const pfo_trac *t = &c->d->trac[k];
        v3 r = ld3(t->r), n = ld3(t->n);
        v3 vel = add3(c->vlin, cross3(c->w, r));
        v3 vt = vec_sub_vec_proj(vel, n);
        double p_dA = t->p * t->dA;
        v3 Tc = traction_reg(in->mu_s, in->mu_d, in->v_c, vt, p_dA);
        v3 trk = add3(scl3(n, p_dA), Tc);
        lin = add3(lin, trk);
        ang = add3(ang, cross3(r, trk));
    }
    st3(wrench, ang);
    st3(wrench + 3, lin);
}

/* cyclic Jacobi for a symmetric 6x6 (stands in for LAPACK eigen!(Hermitian), :88; K̄^{-1/2} does not depend
 * on eigenvector order/sign) */
static void jacobi6(double A[36], double V[36], double w[6])
{
    for (int i = 0; i < 36; ++i) V[i] = 0.0;
    for (int i = 0; i < 6; ++i) V[7 * i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dia = 0.0;
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j)
                if (i != j) off += A[i + 6 * j] * A[i + 6 * j]; else dia += A[7 * i] * A[7 * i];
        if (off <= 1e-300 || off <= 1e-34 * dia) break;
        for (int p = 0; p < 5; ++p)
            for (int q = p + 1; q < 6; ++q) {
                double apq = A[p + 6 * q];
                if (apq == 0.0) continue;
                double theta = (A[7 * q] - A[7 * p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 6; ++k) {
                    double akp = A[k + 6 * p], akq = A[k + 6 * q];
                    A[k + 6 * p] = cs * akp - sn * akq;
                    A[k + 6 * q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < 6; ++k) {
                    double apk = A[p + 6 * k], aqk = A[q + 6 * k];
                    A[p + 6 * k] = cs * apk - sn * aqk;
                    A[q + 6 * k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < 6; ++k) {
                    double vkp = V[k + 6 * p], vkq = V[k + 6 * q];
                    V[k + 6 * p] = cs * vkp - sn * vkq;
                    V[k + 6 * q] = sn * vkp + cs * vkq;
                }
            }
    }
    for (int i = 0; i < 6; ++i) w[i] = A[7 * i];
}

/* decompose_K! + calc_K̄_sqrt_inv: :85-117.  K column-major 6x6, upper triangle authoritative (Hermitian). */
void pfo_decompose_K(const double K[36], double magic, double Kis[36], double Sinv[6])
{
    double t1 = (K[0] + K[7]) + K[14], t2 = (K[21] + K[28]) + K[35];
    double s1 = 1.0 / sqrt(t1), s2 = 1.0 / sqrt(t2);
    for (int k = 0; k < 3; ++k) { Sinv[k] = s1 * magic; Sinv[k + 3] = s2; }
    double Kb[36], V[36], sig[6];
    for (int j = 0; j < 6; ++j)
        for (int i = 0; i < 6; ++i) {
            double kij = (i <= j) ? K[i + 6 * j] : K[j + 6 * i];
            Kb[i + 6 * j] = (Sinv[i] * kij) * Sinv[j];
        }
    jacobi6(Kb, V, sig);
    double mx = sig[0];
    for (int k = 1; k < 6; ++k) mx = fmax(mx, sig[k]);
    for (int k = 0; k < 6; ++k) sig[k] = 1.0 / sqrt(fmax(sig[k], mx * 1.0e-16));
    for (int j = 0; j < 6; ++j)
        for (int i = 0; i < 6; ++i) {
            double a = 0.0;
            for (int k = 0; k < 6; ++k) a += (V[i + 6 * k] * sig[k]) * V[j + 6 * k];
            Kis[i + 6 * j] = a;
        }
}

static void mat6v(const double M[36], const double v[6], double o[6])
{
    for (int i = 0; i < 6; ++i) {
        double a = 0.0;
        for (int k = 0; k < 6; ++k) a += M[i + 6 * k] * v[k];
        o[i] = a;
    }
}

/* yes_contact!(::Bristle): :119-201 + src/contact_algorithms_normal.jl:17-34 */
static void yes_contact_bristle(ctx *c, const double s[6], double wrench[6], double sdot[6])
{
    const pfo_ins *in = c->ins;
    pfo_debug *d = c->d;
    /* pass 1: normal_wrench_cop */
    v3 lin = V3(0, 0, 0), ang = V3(0, 0, 0), ipc = V3(0, 0, 0);
    double ip = 0.0;
    for (int k = 0; k < d->n_trac; ++k) {
        const pfo_trac *t = &d->trac[k];
        v3 r = ld3(t->r), n = ld3(t->n);
        double p_dA = t->p * t->dA;
        v3 ls = scl3(n, p_dA);
        lin = add3(lin, ls);
        ang = add3(ang, cross3(r, ls));
        ip += p_dA;
        ipc = add3(ipc, scl3(r, p_dA));
    }
    v3 cop = div3(ipc, ip);
    /* pass 2: calc_patch_spatial_stiffness! :147-169 */
    double K11[9] = {0}, K12[9] = {0}, K22[9] = {0};
    for (int k = 0; k < d->n_trac; ++k) {
        const pfo_trac *t = &d->trac[k];
        v3 n = ld3(t->n), r = sub3(ld3(t->r), cop);
        double p_dA = t->p * t->dA;
        double nn[3] = {n.x, n.y, n.z};
        v3 rxn = cross3(r, n);
        double rn[3] = {rxn.x, rxn.y, rxn.z};
        double sk[9] = {0.0, r.z, -r.y, -r.z, 0.0, r.x, r.y, -r.x, 0.0}; /* column-major [r]x */
        double q1 = r.x * r.x, q2 = r.y * r.y, q3 = r.z * r.z;
        double sk2[9] = {-q2 - q3, r.x * r.y, r.x * r.z, r.x * r.y, -q1 - q3, r.y * r.z, r.x * r.z, r.y * r.z, -q1 - q2};
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) {
                double I = (i == j) ? 1.0 : 0.0;
                K22[i + 3 * j] += p_dA * (I - nn[i] * nn[j]);
                K12[i + 3 * j] += p_dA * (sk[i + 3 * j] - rn[i] * nn[j]);
                K11[i + 3 * j] -= p_dA * (sk2[i + 3 * j] + rn[i] * rn[j]);
            }
    }
    double K[36];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) {
            K[i + 6 * j] = K11[i + 3 * j];
            K[(i + 3) + 6 * j] = K12[j + 3 * i];
            K[i + 6 * (j + 3)] = K12[i + 3 * j];
            K[(i + 3) + 6 * (j + 3)] = K22[i + 3 * j];
        }
    for (int k = 0; k < 36; ++k) K[k] *= in->k_bar;
    /* decompose_K!, Δ² :130-132 */
    double Kis[36], Sinv[6], tmp[6], Delta[6];
    pfo_decompose_K(K, in->magic, Kis, Sinv);
    mat6v(Kis, s, tmp);
    for (int k = 0; k < 6; ++k) Delta[k] = Sinv[k] * tmp[k];
    /* pass 3: calc_spatial_bristle_force :171-201 */
    v3 Da = V3(Delta[0], Delta[1], Delta[2]), Dl = V3(Delta[3], Delta[4], Delta[5]);
    v3 flin = V3(0, 0, 0), fang = V3(0, 0, 0);
    for (int k = 0; k < d->n_trac; ++k) {
        const pfo_trac *t = &d->trac[k];
        v3 n = ld3(t->n), r = ld3(t->r);
        v3 x = sub3(r, cop);
        v3 del = add3(Dl, cross3(Da, x));
        v3 rp = add3(c->vlin, cross3(c->w, r));
        double p_dA = t->p * t->dA;
        v3 Ts = scl3(add3(del, scl3(rp, in->tau)), -in->k_bar);
        Ts = vec_sub_vec_proj(Ts, n);
        v3 Tc = traction_bri(in->mu_s, in->mu_d, Ts, p_dA);
        flin = add3(flin, Tc);
        fang = add3(fang, cross3(x, Tc));
    }
    double wcop[6] = {fang.x, fang.y, fang.z, flin.x, flin.y, flin.z};
    v3 fang2 = add3(fang, cross3(cop, flin));
    /* ṡ = -τ⁻¹ (K̄^{-1/2} (S⁻¹ w_cop) + s)  :134 */
    double sw[6], ks[6];
    double tau_inv = 1.0 / in->tau;
    for (int k = 0; k < 6; ++k) sw[k] = Sinv[k] * wcop[k];
    mat6v(Kis, sw, ks);
    for (int k = 0; k < 6; ++k) sdot[k] = -tau_inv * (ks[k] + s[k]);
    /* wrench = normal + friction :140-143 */
    wrench[0] = ang.x + fang2.x; wrench[1] = ang.y + fang2.y; wrench[2] = ang.z + fang2.z;
    wrench[3] = lin.x + flin.x; wrench[4] = lin.y + flin.y; wrench[5] = lin.z + flin.z;
    d->has_K = 1;
    memcpy(d->K, K, sizeof K); memcpy(d->Kbar_inv_sqrt, Kis, sizeof Kis); memcpy(d->Sinv, Sinv, sizeof Sinv);
    st3(d->cop, cop); st3(d->wrench_normal, ang); st3(d->wrench_normal + 3, lin);
    memcpy(d->wrench_fric_cop, wcop, sizeof wcop); memcpy(d->Delta, Delta, sizeof Delta);
}

/* ------------------------------------------------------------------------------------------------------------ */
/* force_single_elastic_intersection!: src/contact_algorithms_non_friction.jl:70-84                              */
/* ------------------------------------------------------------------------------------------------------------ */
static int eval_impl(const pfo_mesh *m1, const pfo_mesh *m2, const pfo_ins *ins, const double *pose, const double *bp_pose,
                     const double *twist, const double *s, double *wrench, double *sdot, int *counts, pfo_debug *dbg);
int pfo_eval(const pfo_mesh *m1, const pfo_mesh *m2, const pfo_ins *ins, const double *pose, const double *twist,
             const double *s, double *wrench, double *sdot, int *counts, pfo_debug *dbg)
{
    const unsigned csr = ftz_enter();
    const int rc = eval_impl(m1, m2, ins, pose, NULL, twist, s, wrench, sdot, counts, dbg);
    ftz_leave(csr);
    return rc;
}
/* The same with the broadphase run on bp_pose's x_r1_r2 (24 doubles, the packing of pose; NULL: pose itself): what
 * calcTriTetIntersections! does when the scenario being evaluated is not m.float -- it always culls with m.float's transforms
 * (src/contact_algorithms_non_friction.jl:94-101: `refreshBodyBodyTransform!(m, m.float, c_ins)`), the narrowphase then runs on
 * the evaluated scenario's own pose (:120, refreshBodyBodyCache!(m, tm, c_ins)). */
int pfo_eval_bp(const pfo_mesh *m1, const pfo_mesh *m2, const pfo_ins *ins, const double *pose, const double *bp_pose,
                const double *twist, const double *s, double *wrench, double *sdot, int *counts, pfo_debug *dbg)
{
    const unsigned csr = ftz_enter();
    const int rc = eval_impl(m1, m2, ins, pose, bp_pose, twist, s, wrench, sdot, counts, dbg);
    ftz_leave(csr);
    return rc;
}
static int eval_impl(const pfo_mesh *m1, const pfo_mesh *m2, const pfo_ins *ins, const double *pose, const double *bp_pose,
                     const double *twist, const double *s, double *wrench, double *sdot, int *counts, pfo_debug *dbg)
{
    ctx c;
    pfo_debug *own = NULL;
    if (!m1 || !m2 || !ins || !pose || !twist || !wrench || !sdot) return PFO_ERR_BAD_ARG;
    if (!m2->tet || !m2->eps || (!m1->tri && !m1->tet)) return PFO_ERR_BAD_ARG; /* id_2 is always a tet mesh: mechanism_scenario.jl:402-416 */
    if (ins->n_quad < 1 || ins->n_quad > 2) return PFO_ERR_BAD_ARG;
    if (ins->model == PFO_BRISTLE && !s) return PFO_ERR_BAD_ARG;
    if (!dbg) { own = dbg = pfo_debug_new(); if (!dbg) return PFO_ERR_NOMEM; }
    memset(&c, 0, sizeof c);
    c.m1 = m1; c.m2 = m2; c.ins = ins; c.d = dbg;
    memcpy(c.R21, pose, 9 * sizeof(double));
    /* TT_Cache.R_a_b, t_a_b (update_TT_Cache!, tree_types.jl:43-50): read by the tree descent only */
    memcpy(c.R12, (bp_pose ? bp_pose : pose) + 12, 9 * sizeof(double));
    memcpy(c.t12, (bp_pose ? bp_pose : pose) + 21, 3 * sizeof(double));
    c.x21 = dh_from_Rt(pose, pose + 9);
    c.x12 = dh_from_Rt(pose + 12, pose + 21);
    c.w = ld3(twist); c.vlin = ld3(twist + 3);
    c.chi = ins->chi; c.Ebar = m2->Ebar; /* Ē of mesh_2 only: :131 */
    c.nq = pfo_tri_quad_rule(ins->n_quad, c.qz, c.qw);
    dbg->n_pair = 0; dbg->n_trac = 0; dbg->n_node_tests = 0; dbg->has_K = 0; /* empty! keeps capacity */
    for (int k = 0; k < 6; ++k) { wrench[k] = 0.0; sdot[k] = 0.0; }

    /* calcTriTetIntersections! :94-101 */
    tree_tree(&c, 0, 0);
    int n_nonempty = 0;
    if (!c.err && dbg->n_pair != 0) {
        /* integrate_over! :136-143 */
        for (int k = 0; k < dbg->n_pair && !c.err; ++k) {
            if (m1->tri) op_tri_tet(&c, k, dbg->pair[2 * k], dbg->pair[2 * k + 1]);
            else op_tet_tet(&c, k, dbg->pair[2 * k], dbg->pair[2 * k + 1]);
            n_nonempty += (dbg->clip_n[k] >= 3);
        }
    }
    if (!c.err) {
        if (dbg->n_trac != 0) {
            if (ins->model == PFO_REGULARIZED) yes_contact_regularized(&c, wrench);
            else yes_contact_bristle(&c, s, wrench, sdot);
        } else if (ins->model == PFO_BRISTLE) {
            /* no_contact!(::Bristle): friction.jl:77-81 */
            double ti = 1.0 / ins->tau;
            for (int k = 0; k < 6; ++k) sdot[k] = -ti * s[k];
        }
    }
    if (counts) {
        counts[0] = dbg->n_node_tests > 2147483647LL ? 2147483647 : (int)dbg->n_node_tests;
        counts[1] = dbg->n_pair; counts[2] = n_nonempty; counts[3] = dbg->n_trac;
    }
    int err = c.err;
    if (own) pfo_debug_free(own);
    return err;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* Batch driver for the CPU baseline: items are independent, so they are spread over host threads   */
/* with OpenMP when the library is built with -fopenmp (n_threads <= 1: plain serial loop, the reference's own  */
/* execution model).  meshes: array of pfo_mesh; ins: array of pfo_ins; item k uses ins[ins_ids[k]] whose      */
/* meshes are (id_1[.], id_2[.]).  Returns the first non-zero status.                                            */
/* ------------------------------------------------------------------------------------------------------------ */
#ifdef _OPENMP
#include <omp.h>
#endif
int pfo_eval_batch(int n_items, const pfo_mesh *meshes, const pfo_ins *ins, const int *ins_m1, const int *ins_m2,
                   const int *ins_ids, const double *pose, const double *twist, const double *s, double *wrench,
                   double *sdot, int *counts, int n_threads)
{
    int status = 0;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
    {
        pfo_debug *d = pfo_debug_new();
#pragma omp for schedule(dynamic, 4)
        for (int k = 0; k < n_items; ++k) {
            int id = ins_ids ? ins_ids[k] : k;
            int rc = d ? pfo_eval(&meshes[ins_m1[id]], &meshes[ins_m2[id]], &ins[id], pose + 24 * (size_t)k,
                                  twist + 6 * (size_t)k, s ? s + 6 * (size_t)k : NULL, wrench + 6 * (size_t)k,
                                  sdot + 6 * (size_t)k, counts ? counts + 4 * (size_t)k : NULL, d)
                        : PFO_ERR_NOMEM;
            if (rc) {
#pragma omp critical
                if (!status) status = rc;
            }
        }
        pfo_debug_free(d);
    }
#else
    (void)n_threads;
    pfo_debug *d = pfo_debug_new();
    if (!d) return PFO_ERR_NOMEM;
    for (int k = 0; k < n_items; ++k) {
        int id = ins_ids ? ins_ids[k] : k;
        int rc = pfo_eval(&meshes[ins_m1[id]], &meshes[ins_m2[id]], &ins[id], pose + 24 * (size_t)k, twist + 6 * (size_t)k,
                          s ? s + 6 * (size_t)k : NULL, wrench + 6 * (size_t)k, sdot + 6 * (size_t)k,
                          counts ? counts + 4 * (size_t)k : NULL, d);
        if (rc && !status) status = rc;
    }
    pfo_debug_free(d);
#endif
    return status;
}

int pfo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* addGeneralizedForcesThirdLaw!: src/contact_algorithms_non_friction.jl:267-286 (RigidBodyDynamics transform(wrench,.)
 * and torque!(tau, jac, wrench) restated: lin = R lin, ang = R ang + t x lin, tau_j = J_ang[:,j].ang + J_lin[:,j].lin) */
void pfo_scatter_generalized(int n_items, const double *wrench, const double *x_w_r2, const int *body_1, const int *body_2,
                             const int *scene, int nv, const double *jac, double *f)
{
    for (int i = 0; i < n_items; ++i) {
        const double *w = wrench + 6 * (size_t)i, *x = x_w_r2 + 12 * (size_t)i;
        v3 ang = ld3(w), lin = ld3(w + 3);
        v3 lw = V3((x[0] * lin.x + x[3] * lin.y) + x[6] * lin.z, (x[1] * lin.x + x[4] * lin.y) + x[7] * lin.z,
                   (x[2] * lin.x + x[5] * lin.y) + x[8] * lin.z);
        v3 aw = add3(V3((x[0] * ang.x + x[3] * ang.y) + x[6] * ang.z, (x[1] * ang.x + x[4] * ang.y) + x[7] * ang.z,
                        (x[2] * ang.x + x[5] * ang.y) + x[8] * ang.z), cross3(ld3(x + 9), lw));
        double *fs = f + (size_t)(scene ? scene[i] : 0) * nv;
        for (int pass = 0; pass < 2; ++pass) {       /* +wrench on body 2, then -wrench on body 1 (:271-272) */
            int b = pass == 0 ? body_2[i] : body_1[i];
            double sgn = pass == 0 ? 1.0 : -1.0;
            if (b < 0) continue;                     /* jac::Nothing (:275-279) */
            for (int j = 0; j < nv; ++j) {
                const double *J = jac + ((size_t)b * nv + j) * 6;
                fs[j] += sgn * (dot3(ld3(J), aw) + dot3(ld3(J + 3), lw));
            }
        }
    }
}
