/*
 * pfc_oracle.h — CPU restatement of the reference contact hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the parity oracle for the HIP path and (as "kind": "port") the CPU baseline of bench.py.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product library
 * (pressurefieldcontact.jl_amd/csrc) never includes, links or calls anything from oracle/.
 *
 * Parity status: the reference is pure Julia and no Julia toolchain exists in the build container or on the
 * GPU box, and the reference ships no golden vectors for this path.  The restatement is pinned
 * by the reference's own analytic known-answer and property tests, restated in tests/test_oracle_*.py.
 * "parity unpinned" applies to exactly one thing: the bit-level rounding of the 4x4 inverse, which the
 * reference takes from StaticArrays 0.10.3 `inv(::SMatrix{4,4})` (not vendored under /root/reference); here it
 * is the explicit cofactor expansion x (1/det) that StaticArrays 0.10.3 published (term order as recalled; variants 0 and
 * 2 of pfo_set_inv4_variant and tests/test_inv4_exposure.py count how many predicate outcomes depend on the choice).
 *
 * Floating-point discipline: compile with -ffp-contract=off and without -ffast-math.  Julia never contracts
 * a*b+c on its own; the reference's explicit `muladd` sites are written here as fma().
 */
#ifndef PFC_ORACLE_H
#define PFC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define PFO_INTERNAL (-9999) /* src/obb/tree_types.jl:11,56 */

enum { PFO_REGULARIZED = 0, PFO_BRISTLE = 1 };
enum { PFO_OK = 0, PFO_ERR_NONFINITE = 1, PFO_ERR_BAD_ARG = 3, PFO_ERR_NOMEM = 4 };

typedef struct {
    int n_pt, n_tri, n_tet, n_node;
    const double *pt;        /* n_pt  x 3 */
    const int *tri;          /* n_tri x 3, 0-based, or NULL */
    const int *tet;          /* n_tet x 4, 0-based, or NULL */
    const double *eps;       /* n_pt, or NULL */
    double Ebar;             /* ContactProperties.Ē of a tet mesh (src/structs.jl:9-15) */
    const double *node_c;    /* n_node x 3 */
    const double *node_e;    /* n_node x 3 */
    const double *node_R;    /* n_node x 9, column-major */
    const int *node_child;   /* n_node x 2 */
    const int *node_leaf;    /* n_node; element index or PFO_INTERNAL */
} pfo_mesh;

/* ContactInstructions + friction model (src/mechanism_scenario.jl:5-49) */
typedef struct {
    double chi;
    int n_quad;              /* 1 or 2 */
    int model;               /* PFO_REGULARIZED / PFO_BRISTLE */
    double mu_s, mu_d;
    double v_c;              /* Regularized: v_tol */
    double tau, k_bar, magic;/* Bristle */
} pfo_ins;

/* TractionCache (src/mechanism_scenario.jl:51-58) */
typedef struct { double n[3], r[3], dA, p; } pfo_trac;

typedef struct {
    int n_pair, cap_pair;
    int *pair;               /* 2 x n_pair: (i_1, i_2) in traversal (DFS) order */
    int *clip_n;             /* n_pair: vertex count of the clipped polygon (0, 3..8) */
    int n_trac, cap_trac;
    pfo_trac *trac;
    long long n_node_tests;
    int has_K;
    double K[36], Kbar_inv_sqrt[36], Sinv[6], cop[3], wrench_normal[6], wrench_fric_cop[6], Delta[6];
} pfo_debug;

pfo_debug *pfo_debug_new(void);
void pfo_debug_free(pfo_debug *);

/*
 * One force_single_elastic_intersection! (src/contact_algorithms_non_friction.jl:70-84).
 * pose[24] = x_r2_r1 (R 9 column-major, t 3) then x_r1_r2 (R 9, t 3); twist[6] = twist_r2_r1_r2 [ang; lin];
 * s[6] = bristle state.  wrench[6] = [ang; lin] on body 2 in frame r2; sdot[6]; counts[4] =
 * {node tests (saturated), candidate pairs, pairs with a non-empty clipped polygon, traction points}.
 */
int pfo_eval(const pfo_mesh *m1, const pfo_mesh *m2, const pfo_ins *ins, const double *pose, const double *twist,
             const double *s, double *wrench, double *sdot, int *counts, pfo_debug *dbg);

/* Batch driver (CPU baseline): independent items over n_threads OpenMP threads (1 = the reference's serial model). */
/* pfo_eval with the broadphase culling on bp_pose (NULL: pose): non_friction.jl:94-101 on a scenario that is not m.float */
int pfo_eval_bp(const pfo_mesh *m1, const pfo_mesh *m2, const pfo_ins *ins, const double *pose, const double *bp_pose,
                const double *twist, const double *s, double *wrench, double *sdot, int *counts, pfo_debug *dbg);
int pfo_eval_batch(int n_items, const pfo_mesh *meshes, const pfo_ins *ins, const int *ins_m1, const int *ins_m2,
                   const int *ins_ids, const double *pose, const double *twist, const double *s, double *wrench,
                   double *sdot, int *counts, int n_threads);
int pfo_max_threads(void);

void pfo_scatter_generalized(int n_items, const double *wrench, const double *x_w_r2, const int *body_1, const int *body_2,
                             const int *scene, int nv, const double *jac, double *f);

/* Unit entry points used by the restated reference tests */
double pfo_calc_clamped_piecewise(double x, double x1, double x2, double y1, double y2);
void pfo_traction_regularized(double mu_s, double mu_d, double v_c, const double vel_t[3], double p_dA, double out[3]);
void pfo_traction_bristle(double mu_s, double mu_d, const double Ts[3], double p_dA, double out[3]);
void pfo_weight_poly(int n, const double *p1, const double *p2, double w1, double w2, double *out);
double pfo_a_dot_one_pad_b(const double a[4], const double b[3]);
void pfo_vec_sub_vec_proj(const double v[3], const double n[3], double out[3]);
double pfo_volume(const double v[12]);
double pfo_triangle_area(const double v[9], const double n[3]);
void pfo_triangle_normal(const double v[9], double n[3]);
int pfo_clip_in_tet_coordinates(int n_in, const double *z_in, double *z_out);          /* n_in 3|4; z: n x 4 */
int pfo_clip_plane_tet(const double plane[4], const double tet_cm[16], double *out);   /* out: up to 4 x 3 */
void pfo_zero_small_coordinates(int n, double *z);
double pfo_poly_centroid(int n, const double *v, const double nhat[3], double c[3]);   /* v: 8 x 3 */
int pfo_inv4(const double a_cm[16], double b_cm[16]);
/* Which 4x4 inverse pfo_inv4 (and with it every evaluation) uses: 0 = adjugate from 2x2 minors x 1/det (rounds 1-2),
 * 1 = explicit cofactor expansion x 1/det (DEFAULT, and the HIP path's form: what StaticArrays 0.10.3 publishes for
 * inv(::SMatrix{4,4}), the reference's call at src/contact_algorithms_non_friction.jl:160), 2 = Gauss-Jordan with partial pivoting.
 * Process-wide; for tests/test_inv4_exposure.py, which counts the predicate outcomes that depend on it. */
int pfo_set_inv4_variant(int v);
/* 1: evaluate with flush-to-zero / denormals-are-zero, as the reference's tests do (set_zero_subnormals(true),
 * test/runtests.jl:13); 0 (default): IEEE subnormals.  Process-wide flag, applied per pfo_eval call. */
int pfo_set_ftz(int on);
int pfo_bb_bb_intersect(const double ca[3], const double ea[3], const double Ra[9],
                        const double cb[3], const double eb[3], const double Rb[9],
                        const double R_a_b[9], const double t_a_b[3]);
void pfo_decompose_K(const double K[36], double magic, double Kbar_inv_sqrt[36], double Sinv[6]);
int pfo_tri_quad_rule(int n_rule, double *zeta, double *w);                            /* returns n points */

#ifdef __cplusplus
}
#endif
#endif
