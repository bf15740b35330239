/*
 * pfc_oracle_dual.cpp — CPU restatement of the reference hot path evaluated on ForwardDiff.Dual numbers
 * (value + partials), the scalar type Radau's Jacobian evaluation uses (src/mechanism_scenario.jl:187,
 * src/radau/radau_functions.jl:2-40).  TEST INFRASTRUCTURE ONLY (see pfc_oracle.h).
 *
 * One seed direction at a time: every scalar is (value, one partial); an N-partial Dual evaluation is N of these with
 * the same values.  The broadphase runs on values only (src/contact_algorithms_non_friction.jl:95), so the pair list
 * comes from pfo_eval(); all branches (clip inside tests, 0 < area, 0 < p, friction regimes, clamps, max) compare
 * values, as ForwardDiff's comparison operators do.
 *
 * Parity status: "parity unpinned" for one step.  The reference differentiates THROUGH GenericLinearAlgebra's
 * symmetric eigen-solver (eigen!(Hermitian{Dual}), src/contact_algorithms_friction.jl:88; GenericLinearAlgebra >= 0.1.0
 * is a Project.toml dependency that is not vendored under /root/reference).  Here the partials of K̄^{-1/2} are the
 * analytic Frechet derivative of the same matrix function (Daleckii-Krein divided differences on the eigen-basis,
 * clamp max(sigma, 1e-16 sigma_max) differentiated as ForwardDiff's max does).  They agree with the reference wherever
 * the eigenvalues that are not clamped are distinct; for clustered eigenvalues the reference's partials are
 * themselves rounding noise.  Everything else follows the reference operation by operation; the partials are pinned in
 * tests/test_oracle_dual.py by central differences of the value oracle.
 */
#include "pfc_oracle.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct D {
    double v, d;
    D() : v(0.0), d(0.0) {}
    D(double v_) : v(v_), d(0.0) {}
    D(double v_, double d_) : v(v_), d(d_) {}
};
inline D operator+(D a, D b) { return D(a.v + b.v, a.d + b.d); }
inline D operator-(D a, D b) { return D(a.v - b.v, a.d - b.d); }
inline D operator-(D a) { return D(-a.v, -a.d); }
inline D operator*(D a, D b) { return D(a.v * b.v, a.v * b.d + a.d * b.v); }
inline D operator/(D a, D b) {
    const double q = a.v / b.v;
    return D(q, (a.d - q * b.d) / b.v);
}
inline D dsqrt(D a) {
    const double s = std::sqrt(a.v);
    return D(s, a.d / (2.0 * s));
}
inline D dfma(D a, D b, D c) { return D(std::fma(a.v, b.v, c.v), a.v * b.d + a.d * b.v + c.d); }  /* muladd on Duals */
inline D& operator+=(D &a, D b) { a = a + b; return a; }

struct D3 { D x, y, z; };
inline D3 mk(D x, D y, D z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
inline D3 operator+(D3 a, D3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline D3 operator-(D3 a, D3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline D3 operator*(D3 a, D s) { return mk(a.x * s, a.y * s, a.z * s); }
inline D3 operator/(D3 a, D s) { return mk(a.x / s, a.y / s, a.z / s); }
inline D dot(D3 a, D3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline D3 cross(D3 a, D3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline D3 normalize(D3 a) { D s = D(1.0) / dsqrt(dot(a, a)); return mk(s * a.x, s * a.y, s * a.z); }
inline D3 cst3(const double *p) { return mk(D(p[0]), D(p[1]), D(p[2])); }

struct D4 { D c[4]; };
struct M4 { D m[16]; };  /* column-major */

M4 mul44(const M4 &A, const M4 &B) {
    M4 C;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            C.m[i + 4 * j] = ((A.m[i] * B.m[4 * j] + A.m[i + 4] * B.m[1 + 4 * j]) + A.m[i + 8] * B.m[2 + 4 * j]) +
                             A.m[i + 12] * B.m[3 + 4 * j];
    return C;
}
D4 mul4v(const M4 &A, const D4 &v) {
    D4 r;
    for (int i = 0; i < 4; ++i)
        r.c[i] = ((A.m[i] * v.c[0] + A.m[i + 4] * v.c[1]) + A.m[i + 8] * v.c[2]) + A.m[i + 12] * v.c[3];
    return r;
}
M4 cst44(const double *a) {
    M4 r;
    for (int k = 0; k < 16; ++k) r.m[k] = D(a[k]);
    return r;
}
/* Transform3D as a 4x4 with last row (0 0 0 1); R column-major */
M4 dh(const D *R, const D *t) {
    M4 r;
    for (int j = 0; j < 3; ++j) {
        for (int i = 0; i < 3; ++i) r.m[i + 4 * j] = R[i + 3 * j];
        r.m[3 + 4 * j] = D(0.0);
    }
    r.m[12] = t[0]; r.m[13] = t[1]; r.m[14] = t[2]; r.m[15] = D(1.0);
    return r;
}

/* weightPoly: src/math_kernel/utility.jl:21-26 */
inline D4 weight_poly4(const D4 &p1, const D4 &p2, D w1, D w2) {
    D sw = w1 - w2, c1 = w1 / sw, c2 = w2 / sw;
    D4 r;
    for (int k = 0; k < 4; ++k) r.c[k] = c1 * p2.c[k] - c2 * p1.c[k];
    return r;
}
inline D3 weight_poly3(D3 p1, D3 p2, D w1, D w2) {
    D sw = w1 - w2, c1 = w1 / sw, c2 = w2 / sw;
    return mk(c1 * p2.x - c2 * p1.x, c1 * p2.y - c2 * p1.y, c1 * p2.z - c2 * p1.z);
}
/* vec_sub_vec_proj: src/math_kernel/vector_projections.jl:2-7 */
inline D3 vec_sub_vec_proj(D3 v, D3 n) {
    D t = -dot(v, n);
    return mk(dfma(t, n.x, v.x), dfma(t, n.y, v.y), dfma(t, n.z, v.z));
}
inline D3 vector_area(D3 a, D3 b, D3 c) { return cross(b - a, c - b) * D(0.5); }
inline D triangle_area(D3 a, D3 b, D3 c, D3 n) { return dot(n, vector_area(a, b, c)); }

struct Poly4 { int n; D4 v[8]; };
struct Poly3 { int n; D3 v[8]; };

inline D4 clip_node(const D4 &z_non, const D4 &z_pos, int i) { return weight_poly4(z_non, z_pos, z_non.c[i], z_pos.c[i]); }

/* cut_clip: src/clip/static_clip.jl:135-195 */
int cut_clip(const D4 *z, int n, int i, D4 *out, int *final) {
    while (n > 3 && z[n - 2].c[i].v <= 0.0) --n;
    D4 z_start = clip_node(z[0], z[1], i);
    const double last = z[n - 1].c[i].v;
    const bool inside = (n <= 5) ? (0.0 < last) : (0.0 <= last);
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
/* clip: src/clip/static_clip.jl:34-128 */
int clip_poly(Poly4 &p) {
    D4 z[8], rot[8], out[8];
    int n = p.n;
    for (int k = 0; k < n; ++k) z[k] = p.v[k];
    for (int i = 0; i < 4; ++i) {
        bool all_non_pos = true, all_non_neg = true;
        for (int k = 0; k < n; ++k) {
            all_non_pos &= (z[k].c[i].v <= 0.0);
            all_non_neg &= (0.0 <= z[k].c[i].v);
        }
        if (all_non_pos) { p.n = 0; return 0; }
        if (all_non_neg) continue;
        int start = -1;
        for (int k = 0; k < n; ++k)
            if (z[k].c[i].v <= 0.0 && !(z[(k + 1) % n].c[i].v <= 0.0)) { start = k; break; }
        if (start < 0) return PFO_ERR_NONFINITE;
        for (int k = 0; k < n; ++k) rot[k] = z[(start + k) % n];
        int final = 0;
        n = cut_clip(rot, n, i, out, &final);
        for (int k = 0; k < n; ++k) z[k] = out[k];
        if (final) break;
    }
    p.n = n;
    for (int k = 0; k < n; ++k) p.v[k] = z[k];
    return 0;
}

/* clip_plane_tet: src/clip/plane_tet_intersection.jl:9-106 */
int clip_plane_tet(const D plane[4], const M4 &tet, Poly3 &out) {
    D proj[4];
    D3 v[4];
    bool neg[4], pos[4];
    int n_neg = 0, n_pos = 0;
    for (int j = 0; j < 4; ++j) {
        proj[j] = ((plane[0] * tet.m[4 * j] + plane[1] * tet.m[1 + 4 * j]) + plane[2] * tet.m[2 + 4 * j]) +
                  plane[3] * tet.m[3 + 4 * j];
        v[j] = mk(tet.m[4 * j], tet.m[1 + 4 * j], tet.m[2 + 4 * j]);
        neg[j] = proj[j].v < 0.0;
        pos[j] = 0.0 < proj[j].v;
        n_neg += neg[j];
        n_pos += pos[j];
    }
    out.n = 0;
    if (n_pos == 0 || n_neg == 0) return 0;
#define PW(i1, i2) weight_poly3(v[i1], v[i2], proj[i1], proj[i2])
    int lone = -1;
    if (n_pos == 1) { for (int j = 0; j < 4; ++j) if (pos[j]) { lone = j; break; } }
    else if (n_neg == 1) { for (int j = 0; j < 4; ++j) if (neg[j]) { lone = j; break; } }
    if (lone >= 0) {
        static const int tab[4][3] = {{1, 3, 2}, {0, 2, 3}, {0, 3, 1}, {0, 1, 2}};
        D3 a = PW(tab[lone][0], lone), b = PW(tab[lone][1], lone), c = PW(tab[lone][2], lone);
        out.n = 3;
        if (0.0 < proj[lone].v) { out.v[0] = a; out.v[1] = b; out.v[2] = c; }
        else { out.v[0] = c; out.v[1] = b; out.v[2] = a; }
    } else {
        D3 a, b, c, d;
        if (pos[0] == pos[1]) { a = PW(1, 2); b = PW(1, 3); c = PW(0, 3); d = PW(0, 2); }
        else if (pos[0] == pos[2]) { a = PW(0, 1); b = PW(0, 3); c = PW(2, 3); d = PW(2, 1); }
        else { a = PW(0, 2); b = PW(0, 1); c = PW(3, 1); d = PW(3, 2); }
        out.n = 4;
        if (0.0 < proj[0].v) { out.v[0] = a; out.v[1] = b; out.v[2] = c; out.v[3] = d; }
        else { out.v[0] = d; out.v[1] = c; out.v[2] = b; out.v[3] = a; }
    }
#undef PW
    return out.n;
}

struct Trac { D3 n, r; D dA, p; };

struct Ctx {
    const pfo_mesh *m1, *m2;
    const pfo_ins *ins;
    D R21[9], t21[3], R12[9], t12[3];
    M4 x21, x12;
    D3 w, vlin;
    double qz[9], qw[3];
    int nq;
    std::vector<Trac> trac;
    int err;
};

int tet_transforms(const pfo_mesh *m, int i_tet, double x_r_z[16], double x_z_r[16], double eps[4]) {
    const int *it = m->tet + 4 * i_tet;
    for (int j = 0; j < 4; ++j) {
        const double *p = m->pt + 3 * it[j];
        x_r_z[4 * j] = p[0]; x_r_z[1 + 4 * j] = p[1]; x_r_z[2 + 4 * j] = p[2]; x_r_z[3 + 4 * j] = 1.0;
        eps[j] = m->eps[it[j]];
    }
    return pfo_inv4(x_r_z, x_z_r);
}
void row_mul44(const D r[4], const M4 &M, D out[4]) {
    for (int j = 0; j < 4; ++j)
        out[j] = ((r[0] * M.m[4 * j] + r[1] * M.m[1 + 4 * j]) + r[2] * M.m[2 + 4 * j]) + r[3] * M.m[3 + 4 * j];
}

/* fillTractionCacheForTriangle! / InnerLoop!: src/contact_algorithms_non_friction.jl:236-265 */
void fill_triangle(Ctx &c, D area, D3 nh, D3 a0, D3 a1, D3 a2, const double eps_r[4]) {
    for (int k = 0; k < c.nq; ++k) {
        const double *z = c.qz + 3 * k;
        D3 r = mk((a0.x * D(z[0]) + a1.x * D(z[1])) + a2.x * D(z[2]), (a0.y * D(z[0]) + a1.y * D(z[1])) + a2.y * D(z[2]),
                  (a0.z * D(z[0]) + a1.z * D(z[1])) + a2.z * D(z[2]));
        D eq = dfma(D(eps_r[0]), r.x, D(eps_r[3]));
        eq = dfma(D(eps_r[1]), r.y, eq);
        eq = dfma(D(eps_r[2]), r.z, eq);
        D3 rdot = c.vlin + cross(c.w, r);
        D ee = -dot(mk(D(eps_r[0]), D(eps_r[1]), D(eps_r[2])), rdot);
        D damp_arg = D(1.0) + D(c.ins->chi) * ee;
        D damp = (damp_arg.v > 0.0) ? damp_arg : D(0.0);  /* max(0.0, .) */
        D p = eq * D(c.m2->Ebar) * damp;
        D dA = D(c.qw[k]) * area;
        if (0.0 < p.v) {
            Trac t;
            t.n = nh; t.r = r; t.dA = dA; t.p = p;
            c.trac.push_back(t);
        }
    }
}

/* integrate_over_polygon_patch!: :217-234 ; centroid: src/clip/poly_eight.jl:35-52 */
void integrate_patch(Ctx &c, D3 nh, const Poly4 &pz, const double x_r_z[16], const double eps_r[4]) {
    Poly3 pr;
    pr.n = pz.n;
    const M4 X = cst44(x_r_z);
    for (int k = 0; k < pz.n; ++k) {
        D4 r = mul4v(X, pz.v[k]);
        pr.v[k] = mk(r.c[0], r.c[1], r.c[2]);
    }
    D3 a = pr.v[0], cc = pr.v[1];
    D cum_sum(0.0);
    D3 cum_prod = mk(D(0.0), D(0.0), D(0.0));
    for (int k = 2; k < pr.n; ++k) {
        D3 b = cc;
        cc = pr.v[k];
        D ar = triangle_area(a, b, cc, nh);
        cum_prod = cum_prod + (((a + b) + cc) * D(1.0 / 3.0)) * ar;
        cum_sum += ar;
    }
    D3 cen = (cum_sum.v == 0.0) ? a : cum_prod / cum_sum;
    const int N = pz.n;
    D3 v2 = pr.v[N - 1];
    for (int k = 0; k < N; ++k) {
        D3 v1 = v2;
        v2 = pr.v[k];
        D area = triangle_area(v1, v2, cen, nh);
        if (0.0 < area.v) fill_triangle(c, area, nh, v1, v2, cen, eps_r);
    }
}

/* tri-tet op: :196-215 */
void op_tri_tet(Ctx &c, int i1, int i2) {
    const int *it = c.m1->tri + 3 * i1;
    const double *q0 = c.m1->pt + 3 * it[0], *q1 = c.m1->pt + 3 * it[1], *q2 = c.m1->pt + 3 * it[2];
    double x_r_z[16], x_z_r[16], eps2[4], eps_r[4];
    if (tet_transforms(c.m2, i2, x_r_z, x_z_r, eps2)) { c.err = PFO_ERR_NONFINITE; return; }
    for (int j = 0; j < 4; ++j)
        eps_r[j] = ((eps2[0] * x_z_r[4 * j] + eps2[1] * x_z_r[1 + 4 * j]) + eps2[2] * x_z_r[2 + 4 * j]) + eps2[3] * x_z_r[3 + 4 * j];
    const M4 x_z_r1 = mul44(cst44(x_z_r), c.x21);
    Poly4 p;
    p.n = 3;
    const double *q[3] = {q0, q1, q2};
    for (int k = 0; k < 3; ++k) {
        D4 a;
        a.c[0] = D(q[k][0]); a.c[1] = D(q[k][1]); a.c[2] = D(q[k][2]); a.c[3] = D(1.0);
        p.v[k] = mul4v(x_z_r1, a);
    }
    if (clip_poly(p)) { c.err = PFO_ERR_NONFINITE; return; }
    if (3 <= p.n) {
        D3 n1 = normalize(vector_area(cst3(q0), cst3(q1), cst3(q2)));
        const D *R = c.R21;
        D3 nh = mk((R[0] * n1.x + R[3] * n1.y) + R[6] * n1.z, (R[1] * n1.x + R[4] * n1.y) + R[7] * n1.z,
                   (R[2] * n1.x + R[5] * n1.y) + R[8] * n1.z);
        integrate_patch(c, nh, p, x_r_z, eps_r);
    }
}

/* tet-tet op: :166-194 */
void op_tet_tet(Ctx &c, int i1, int i2) {
    double x_r1_z1[16], x_z1_r1[16], x_r2_z2[16], x_z2_r2[16], e1[4], e2[4], eps_r[4];
    if (tet_transforms(c.m1, i1, x_r1_z1, x_z1_r1, e1) || tet_transforms(c.m2, i2, x_r2_z2, x_z2_r2, e2)) {
        c.err = PFO_ERR_NONFINITE;
        return;
    }
    const M4 X1 = mul44(cst44(x_z1_r1), c.x12);
    const M4 Z2 = cst44(x_z2_r2);
    D Ee1[4], Ee2[4], pl1[4], pl2[4], plane[4];
    for (int j = 0; j < 4; ++j) { Ee1[j] = D(c.m1->Ebar * e1[j]); Ee2[j] = D(c.m2->Ebar * e2[j]); }
    row_mul44(Ee1, X1, pl1);
    row_mul44(Ee2, Z2, pl2);
    for (int j = 0; j < 4; ++j) {
        eps_r[j] = ((e2[0] * x_z2_r2[4 * j] + e2[1] * x_z2_r2[1 + 4 * j]) + e2[2] * x_z2_r2[2 + 4 * j]) + e2[3] * x_z2_r2[3 + 4 * j];
        plane[j] = pl2[j] - pl1[j];
    }
    const M4 x_r2_z1 = mul44(c.x21, cst44(x_r1_z1));
    Poly3 pr;
    clip_plane_tet(plane, x_r2_z1, pr);
    if (3 <= pr.n) {
        Poly4 pz;
        pz.n = pr.n;
        for (int k = 0; k < pr.n; ++k) {
            D4 o;
            o.c[0] = pr.v[k].x; o.c[1] = pr.v[k].y; o.c[2] = pr.v[k].z; o.c[3] = D(1.0);
            pz.v[k] = mul4v(Z2, o);
            for (int i = 0; i < 4; ++i)   /* zero_small_coordinates: src/clip/poly_eight.jl:106-126 */
                pz.v[k].c[i] = pz.v[k].c[i] * D((1.0e-14 < std::fabs(pz.v[k].c[i].v)) ? 1.0 : 0.0);
        }
        if (clip_poly(pz)) { c.err = PFO_ERR_NONFINITE; return; }
        if (3 <= pz.n) {
            D3 nh = normalize(mk(plane[0], plane[1], plane[2]));
            integrate_patch(c, nh, pz, x_r2_z2, eps_r);
        }
    }
}

/* calc_clamped_piecewise: src/contact_algorithms_friction.jl:2-10 */
D clamped_piecewise(D x, double x1, double x2, double y1, double y2) {
    const double k = (y2 - y1) / (x2 - x1);
    D y = D(y1) + (x - D(x1)) * D(k);
    return (y.v > y1) ? D(y1) : ((y.v < y2) ? D(y2) : y);
}
/* traction(::Regularized): :13-30 */
D3 traction_reg(double mu_s, double mu_d, double v_c, D3 vt, D p_dA) {
    D m2 = dot(vt, vt);
    D3 T;
    if (m2.v < v_c * v_c) {
        T = (vt * D(-mu_s)) / D(v_c);
    } else {
        D m = dsqrt(m2);
        D mu = clamped_piecewise(m, 2 * v_c, 3 * v_c, mu_s, mu_d);
        T = (vt * (-mu)) / m;
    }
    return T * p_dA;
}
/* traction(::Bristle): :32-48 */
D3 traction_bri(double mu_s, double mu_d, D3 Ts, D p_dA) {
    D m2 = dot(Ts, Ts);
    D3 T;
    if (m2.v < mu_s * mu_s) {
        T = Ts;
    } else {
        D m = dsqrt(m2);
        D mu = clamped_piecewise(m, 2 * mu_s, 3 * mu_s, mu_s, mu_d);
        T = (Ts * mu) / m;
    }
    return T * p_dA;
}

void yes_contact_regularized(Ctx &c, D wrench[6]) {
    const pfo_ins *in = c.ins;
    D3 lin = mk(D(0.0), D(0.0), D(0.0)), ang = lin;
    for (const Trac &t : c.trac) {
        D3 vel = c.vlin + cross(c.w, t.r);
        D3 vt = vec_sub_vec_proj(vel, t.n);
        D p_dA = t.p * t.dA;
        D3 Tc = traction_reg(in->mu_s, in->mu_d, in->v_c, vt, p_dA);
        D3 trk = t.n * p_dA + Tc;
        lin = lin + trk;
        ang = ang + cross(t.r, trk);
    }
    wrench[0] = ang.x; wrench[1] = ang.y; wrench[2] = ang.z;
    wrench[3] = lin.x; wrench[4] = lin.y; wrench[5] = lin.z;
}

void jacobi6(double A[36], double V[36], double w[6]) {
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
                const double apq = A[p + 6 * q];
                if (apq == 0.0) continue;
                const double theta = (A[7 * q] - A[7 * p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 6; ++k) {
                    const double akp = A[k + 6 * p], akq = A[k + 6 * q];
                    A[k + 6 * p] = cs * akp - sn * akq; A[k + 6 * q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < 6; ++k) {
                    const double apk = A[p + 6 * k], aqk = A[q + 6 * k];
                    A[p + 6 * k] = cs * apk - sn * aqk; A[q + 6 * k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < 6; ++k) {
                    const double vkp = V[k + 6 * p], vkq = V[k + 6 * q];
                    V[k + 6 * p] = cs * vkp - sn * vkq; V[k + 6 * q] = sn * vkp + cs * vkq;
                }
            }
    }
    for (int i = 0; i < 6; ++i) w[i] = A[7 * i];
}

/* decompose_K! + calc_K̄_sqrt_inv (:85-117) on Duals; see the header for how the eigen step is differentiated. */
void decompose_K(const D K[36], double magic, D Kis[36], D Sinv[6]) {
    D t1 = (K[0] + K[7]) + K[14], t2 = (K[21] + K[28]) + K[35];
    D s1 = D(1.0) / dsqrt(t1), s2 = D(1.0) / dsqrt(t2);
    for (int k = 0; k < 3; ++k) { Sinv[k] = s1 * D(magic); Sinv[k + 3] = s2; }
    D Kb[36];
    for (int j = 0; j < 6; ++j)
        for (int i = 0; i < 6; ++i) {
            D kij = (i <= j) ? K[i + 6 * j] : K[j + 6 * i];
            Kb[i + 6 * j] = (Sinv[i] * kij) * Sinv[j];
        }
    double A[36], V[36], lam[6], dK[36];
    for (int k = 0; k < 36; ++k) { A[k] = Kb[k].v; dK[k] = Kb[k].d; }
    jacobi6(A, V, lam);
    double mx = lam[0];
    int imx = 0;
    for (int k = 1; k < 6; ++k) if (lam[k] > mx) { mx = lam[k]; imx = k; }
    /* M = V' dK V */
    double T[36], M[36];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double a = 0.0;
            for (int k = 0; k < 6; ++k) a += dK[i + 6 * k] * V[k + 6 * j];
            T[i + 6 * j] = a;
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double a = 0.0;
            for (int k = 0; k < 6; ++k) a += V[k + 6 * i] * T[k + 6 * j];
            M[i + 6 * j] = a;
        }
    const double floor_v = mx * 1.0e-16, dfloor = M[imx + 6 * imx] * 1.0e-16;   /* d(sigma_max) = v_max' dK v_max */
    double f[6], fp[6], fx[6];  /* f(sigma), df/dsigma (own eigenvalue), df via the floor */
    bool clamped[6];
    for (int k = 0; k < 6; ++k) {
        clamped[k] = !(lam[k] > floor_v);   /* max(sigma, floor): ties take the floor */
        const double x = clamped[k] ? floor_v : lam[k];
        f[k] = 1.0 / std::sqrt(x);
        const double dfdx = -0.5 * f[k] / x;
        fp[k] = clamped[k] ? 0.0 : dfdx;
        fx[k] = clamped[k] ? dfdx * dfloor : 0.0;
    }
    double G[36];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double g;
            if (i == j) g = fp[i];
            else if (clamped[i] && clamped[j]) g = 0.0;
            else if (lam[i] != lam[j]) g = (f[i] - f[j]) / (lam[i] - lam[j]);
            else g = fp[i];
            G[i + 6 * j] = g * M[i + 6 * j];
        }
    for (int k = 0; k < 6; ++k) G[7 * k] += fx[k];
    /* Kis = V f V',  dKis = V G V' */
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double a = 0.0;
            for (int k = 0; k < 6; ++k) a += (V[i + 6 * k] * f[k]) * V[j + 6 * k];
            Kis[i + 6 * j].v = a;
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double a = 0.0;
            for (int k = 0; k < 6; ++k) a += G[i + 6 * k] * V[j + 6 * k];
            T[i + 6 * j] = a;
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double a = 0.0;
            for (int k = 0; k < 6; ++k) a += V[i + 6 * k] * T[k + 6 * j];
            Kis[i + 6 * j].d = a;
        }
}

void mat6v(const D M[36], const D v[6], D o[6]) {
    for (int i = 0; i < 6; ++i) {
        D a(0.0);
        for (int k = 0; k < 6; ++k) a += M[i + 6 * k] * v[k];
        o[i] = a;
    }
}

/* yes_contact!(::Bristle): :119-201 + src/contact_algorithms_normal.jl:17-34 */
void yes_contact_bristle(Ctx &c, const D s[6], D wrench[6], D sdot[6]) {
    const pfo_ins *in = c.ins;
    const D zero(0.0);
    D3 lin = mk(zero, zero, zero), ang = lin, ipc = lin;
    D ip(0.0);
    for (const Trac &t : c.trac) {
        D p_dA = t.p * t.dA;
        D3 ls = t.n * p_dA;
        lin = lin + ls;
        ang = ang + cross(t.r, ls);
        ip += p_dA;
        ipc = ipc + t.r * p_dA;
    }
    D3 cop = ipc / ip;
    D K11[9], K12[9], K22[9];
    for (const Trac &t : c.trac) {
        D3 n = t.n, r = t.r - cop;
        D p_dA = t.p * t.dA;
        D nn[3] = {n.x, n.y, n.z};
        D3 rxn = cross(r, n);
        D rn[3] = {rxn.x, rxn.y, rxn.z};
        D sk[9] = {zero, r.z, -r.y, -r.z, zero, r.x, r.y, -r.x, zero};
        D q1 = r.x * r.x, q2 = r.y * r.y, q3 = r.z * r.z;
        D sk2[9] = {-q2 - q3, r.x * r.y, r.x * r.z, r.x * r.y, -q1 - q3, r.y * r.z, r.x * r.z, r.y * r.z, -q1 - q2};
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) {
                D I((i == j) ? 1.0 : 0.0);
                K22[i + 3 * j] += p_dA * (I - nn[i] * nn[j]);
                K12[i + 3 * j] += p_dA * (sk[i + 3 * j] - rn[i] * nn[j]);
                K11[i + 3 * j] = K11[i + 3 * j] - p_dA * (sk2[i + 3 * j] + rn[i] * rn[j]);
            }
    }
    D K[36];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) {
            K[i + 6 * j] = K11[i + 3 * j];
            K[(i + 3) + 6 * j] = K12[j + 3 * i];
            K[i + 6 * (j + 3)] = K12[i + 3 * j];
            K[(i + 3) + 6 * (j + 3)] = K22[i + 3 * j];
        }
    for (int k = 0; k < 36; ++k) K[k] = K[k] * D(in->k_bar);
    D Kis[36], Sinv[6], tmp[6], Delta[6];
    decompose_K(K, in->magic, Kis, Sinv);
    mat6v(Kis, s, tmp);
    for (int k = 0; k < 6; ++k) Delta[k] = Sinv[k] * tmp[k];
    D3 Da = mk(Delta[0], Delta[1], Delta[2]), Dl = mk(Delta[3], Delta[4], Delta[5]);
    D3 flin = mk(zero, zero, zero), fang = flin;
    for (const Trac &t : c.trac) {
        D3 x = t.r - cop;
        D3 del = Dl + cross(Da, x);
        D3 rp = c.vlin + cross(c.w, t.r);
        D p_dA = t.p * t.dA;
        D3 Ts = (del + rp * D(in->tau)) * D(-in->k_bar);
        Ts = vec_sub_vec_proj(Ts, t.n);
        D3 Tc = traction_bri(in->mu_s, in->mu_d, Ts, p_dA);
        flin = flin + Tc;
        fang = fang + cross(x, Tc);
    }
    D wcop[6] = {fang.x, fang.y, fang.z, flin.x, flin.y, flin.z};
    D3 fang2 = fang + cross(cop, flin);
    D sw[6], ks[6];
    const double tau_inv = 1.0 / in->tau;
    for (int k = 0; k < 6; ++k) sw[k] = Sinv[k] * wcop[k];
    mat6v(Kis, sw, ks);
    for (int k = 0; k < 6; ++k) sdot[k] = D(-tau_inv) * (ks[k] + s[k]);
    wrench[0] = ang.x + fang2.x; wrench[1] = ang.y + fang2.y; wrench[2] = ang.z + fang2.z;
    wrench[3] = lin.x + flin.x; wrench[4] = lin.y + flin.y; wrench[5] = lin.z + flin.z;
}

}  // namespace

/*
 * force_single_elastic_intersection! on Duals with n_dir partials.  pose/twist/s as in pfo_eval; d_pose (n_dir x 24),
 * d_twist (n_dir x 6), d_s (n_dir x 6) are the partials of each input; d_wrench, d_sdot (n_dir x 6) the partials of
 * the outputs.  The values written to wrench / sdot come from the Dual evaluation itself.
 */
extern "C" int pfo_eval_dual_bp(const pfo_mesh *m1, const pfo_mesh *m2, const pfo_ins *ins, const double *pose, const double *bp_pose,
                                const double *twist, const double *s, int n_dir, const double *d_pose,
                                const double *d_twist, const double *d_s, double *wrench, double *sdot, double *d_wrench,
                                double *d_sdot);
extern "C" int pfo_eval_dual(const pfo_mesh *m1, const pfo_mesh *m2, const pfo_ins *ins, const double *pose,
                             const double *twist, const double *s, int n_dir, const double *d_pose,
                             const double *d_twist, const double *d_s, double *wrench, double *sdot, double *d_wrench,
                             double *d_sdot) {
    return pfo_eval_dual_bp(m1, m2, ins, pose, nullptr, twist, s, n_dir, d_pose, d_twist, d_s, wrench, sdot, d_wrench, d_sdot);
}
/* ... with the pair list of the value pass taken at bp_pose (m.float's transforms, non_friction.jl:94-101; NULL: pose) */
extern "C" int pfo_eval_dual_bp(const pfo_mesh *m1, const pfo_mesh *m2, const pfo_ins *ins, const double *pose, const double *bp_pose,
                                const double *twist, const double *s, int n_dir, const double *d_pose,
                                const double *d_twist, const double *d_s, double *wrench, double *sdot, double *d_wrench,
                                double *d_sdot) {
    if (!m1 || !m2 || !ins || !pose || !twist || !wrench || !sdot || n_dir < 0) return PFO_ERR_BAD_ARG;
    if (n_dir > 0 && (!d_pose || !d_twist || !d_wrench || !d_sdot)) return PFO_ERR_BAD_ARG;
    static const double zero6[6] = {0, 0, 0, 0, 0, 0};
    if (!s) s = zero6;
    pfo_debug *dbg = pfo_debug_new();
    if (!dbg) return PFO_ERR_NOMEM;
    int counts[4];
    int st = pfo_eval_bp(m1, m2, ins, pose, bp_pose, twist, s, wrench, sdot, counts, dbg);   /* values + the pair list */
    if (st) { pfo_debug_free(dbg); return st; }
    for (int dir = 0; dir < n_dir; ++dir) {
        Ctx c;
        c.m1 = m1; c.m2 = m2; c.ins = ins; c.err = 0;
        const double *dp = d_pose + 24 * (size_t)dir, *dt = d_twist + 6 * (size_t)dir;
        for (int k = 0; k < 9; ++k) { c.R21[k] = D(pose[k], dp[k]); c.R12[k] = D(pose[12 + k], dp[12 + k]); }
        for (int k = 0; k < 3; ++k) { c.t21[k] = D(pose[9 + k], dp[9 + k]); c.t12[k] = D(pose[21 + k], dp[21 + k]); }
        c.x21 = dh(c.R21, c.t21);
        c.x12 = dh(c.R12, c.t12);
        c.w = mk(D(twist[0], dt[0]), D(twist[1], dt[1]), D(twist[2], dt[2]));
        c.vlin = mk(D(twist[3], dt[3]), D(twist[4], dt[4]), D(twist[5], dt[5]));
        c.nq = pfo_tri_quad_rule(ins->n_quad, c.qz, c.qw);
        for (int k = 0; k < dbg->n_pair && !c.err; ++k) {
            if (m1->tri) op_tri_tet(c, dbg->pair[2 * k], dbg->pair[2 * k + 1]);
            else op_tet_tet(c, dbg->pair[2 * k], dbg->pair[2 * k + 1]);
        }
        if (c.err) { pfo_debug_free(dbg); return c.err; }
        D w[6], sd[6], sD[6];
        for (int k = 0; k < 6; ++k) sD[k] = D(s[k], d_s ? d_s[6 * (size_t)dir + k] : 0.0);
        if (!c.trac.empty()) {
            if (ins->model == PFO_REGULARIZED) yes_contact_regularized(c, w);
            else yes_contact_bristle(c, sD, w, sd);
        } else if (ins->model == PFO_BRISTLE) {
            for (int k = 0; k < 6; ++k) sd[k] = D(-(1.0 / ins->tau)) * sD[k];
        }
        for (int k = 0; k < 6; ++k) {
            d_wrench[6 * (size_t)dir + k] = w[k].d;
            d_sdot[6 * (size_t)dir + k] = sd[k].d;
            if (dir == 0) { wrench[k] = w[k].v; sdot[k] = sd[k].v; }
        }
    }
    pfo_debug_free(dbg);
    return 0;
}
