/*
 * box_on_plane.c -- the drop-in boundary exercised from plain C, no Python, no torch.
 *
 * The scene of the reference's test/test_normal.jl:2-49: a rigid 12-triangle box (half-width 0.05 m) pressed
 * 5 mm into the compliant half-plane tetrahedron (Ē = 1e9 Pa), shifted by (0.1, 0.2).  The exact normal wrench is
 * f_z = Ē * penetration * (2 r)^2 on the plane, torque = r x f.  Build (the library is built by
 * `python -c "import __graft_entry__ as g; g.build()"`):
 *
 *   gcc -O2 -I include examples/box_on_plane.c -L pressurefieldcontact.jl_amd/csrc -lpfc_hip \
 *       -Wl,-rpath,$PWD/pressurefieldcontact.jl_amd/csrc -lm -o /tmp/box_on_plane && /tmp/box_on_plane
 *
 * Exit status 0 iff the wrench matches the analytic value to 1e-8.  `box_on_plane N` additionally times N further
 * evaluations: what one call of this small scene costs a C (or ccall) host -- 61 us on an MI355X box.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "pfc.h"

/* status-returning calls: 0 = ok */
#define CHECK(call)                                                           \
    do {                                                                      \
        int rc_ = (call);                                                     \
        if (rc_ != PFC_OK) {                                                  \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, pfc_last_error(h)); \
            return 2;                                                         \
        }                                                                     \
    } while (0)

static int add_mesh(pfc_handle h, int n_pt, const double *xyz, int n_tri, const int *tri, int n_tet, const int *tet,
                    const double *eps, double Ebar) {
    const int n_elem = n_tri ? n_tri : n_tet, arity = n_tri ? 3 : 4, n_node = 2 * n_elem - 1;
    double *c = malloc(sizeof(double) * 3 * n_node), *e = malloc(sizeof(double) * 3 * n_node);
    double *R = malloc(sizeof(double) * 9 * n_node);
    int *child = malloc(sizeof(int) * 2 * n_node), *leaf = malloc(sizeof(int) * n_node);
    /* eMesh_to_tree: the open half-plane surface has a single element, the box surface is closed */
    int rc = pfc_build_tree(n_pt, xyz, n_elem, arity, n_tri ? tri : tet, eps, PFC_TREE_BLOB, c, e, R, child, leaf);
    if (rc < 0) { fprintf(stderr, "pfc_build_tree: %s\n", pfc_tree_last_error()); return rc; }
    rc = pfc_add_mesh(h, n_pt, xyz, n_tri, tri, n_tet, tet, eps, Ebar, n_node, c, e, R, child, leaf);
    free(c); free(e); free(R); free(child); free(leaf);
    return rc;
}

#include <time.h>

int main(int argc, char **argv) {
    const double r = 0.05, Ebar = 1.0e9, pene = 0.1 * 0.05, px = 0.1, py = 0.2;
    pfc_handle h = NULL;
    if (pfc_create(0, &h) != PFC_OK) { fprintf(stderr, "no HIP device: the hot path has no CPU fallback\n"); return 3; }

    /* half-plane: one tet, top face on z = 0, apex at z = -1, eps = depth (src/geometry/mesh.jl:430-442) */
    const double pi = 3.14159265358979323846;
    double hp[12];
    for (int k = 0; k < 3; ++k) { hp[3 * k] = cos(2 * pi * k / 3); hp[3 * k + 1] = sin(2 * pi * k / 3); hp[3 * k + 2] = 0.0; }
    hp[9] = 0.0; hp[10] = 0.0; hp[11] = -1.0;
    const int hp_tet[4] = {3, 0, 1, 2};
    const double hp_eps[4] = {0.0, 0.0, 0.0, 1.0};
    /* box surface, centre at (0, 0, r) in its own frame: 8 corners, 12 outward triangles (mesh.jl:527-575) */
    double bx[24];
    for (int k = 0; k < 8; ++k) {
        bx[3 * k] = (k & 1) ? r : -r; bx[3 * k + 1] = (k & 2) ? r : -r; bx[3 * k + 2] = r + ((k & 4) ? r : -r);
    }
    static const int face[6][4] = {{0, 2, 4, 6}, {1, 5, 3, 7}, {0, 4, 1, 5}, {2, 3, 6, 7}, {0, 1, 2, 3}, {4, 6, 5, 7}};
    int bt[36];
    for (int f = 0; f < 6; ++f) {
        bt[6 * f] = face[f][0]; bt[6 * f + 1] = face[f][2]; bt[6 * f + 2] = face[f][3];
        bt[6 * f + 3] = face[f][0]; bt[6 * f + 4] = face[f][3]; bt[6 * f + 5] = face[f][1];
    }
    const int id_box = add_mesh(h, 8, bx, 12, bt, 0, NULL, NULL, 0.0);
    const int id_plane = add_mesh(h, 4, hp, 0, NULL, 1, hp_tet, hp_eps, Ebar);
    if (id_box < 0 || id_plane < 0) { fprintf(stderr, "add_mesh failed: %s\n", pfc_last_error(h)); return 2; }

    /* add_friction_regularize!(box, plane, mu_d = 0.3, chi = 0.6, v_tol = 1e-2), quadrature rule 2 */
    const double params[8] = {0.3, 0.3, 1.0e-2, 0, 0, 0, 0, 0};
    if (pfc_add_instruction(h, id_box, id_plane, 0.6, 2, PFC_REGULARIZED, params) < 0) {   /* returns the id */
        fprintf(stderr, "add_instruction failed: %s\n", pfc_last_error(h));
        return 2;
    }
    CHECK(pfc_finalize(h));

    /* x_r2_r1: box frame -> plane frame = translation (px, py, -pene); x_r1_r2 = its inverse */
    double pose[24] = {1, 0, 0, 0, 1, 0, 0, 0, 1, px, py, -pene, 1, 0, 0, 0, 1, 0, 0, 0, 1, -px, -py, pene};
    double twist[6] = {0, 0, 0, 0, 0, 0}, wrench[6], sdot[6];
    int counts[4];
    CHECK(pfc_eval(h, 1, NULL, pose, twist, NULL, wrench, sdot, counts));

    const double fz = Ebar * pene * 4 * r * r;
    const double expect[6] = {-(py * fz), px * fz, 0.0, 0.0, 0.0, -fz};   /* -[r x f; f], f = (0, 0, fz) */
    double err = 0.0, nrm = 0.0;
    for (int k = 0; k < 6; ++k) { err += (wrench[k] - expect[k]) * (wrench[k] - expect[k]); nrm += expect[k] * expect[k]; }
    printf("node tests %d, candidate pairs %d, non-empty %d, traction points %d\n", counts[0], counts[1], counts[2], counts[3]);
    printf("wrench  % .9e % .9e % .9e | % .9e % .9e % .9e\n", wrench[0], wrench[1], wrench[2], wrench[3], wrench[4], wrench[5]);
    printf("expect  % .9e % .9e % .9e | % .9e % .9e % .9e\n", expect[0], expect[1], expect[2], expect[3], expect[4], expect[5]);
    printf("relative error %.3e\n", sqrt(err / nrm));
    if (argc > 1) {     /* box_on_plane N: time N more evaluations (what one ccall of a small scene costs the host) */
        const int n_rep = atoi(argv[1]);
        struct timespec t0, t1;
        for (int k = 0; k < 20; ++k) CHECK(pfc_eval(h, 1, NULL, pose, twist, NULL, wrench, sdot, counts));
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int k = 0; k < n_rep; ++k) CHECK(pfc_eval(h, 1, NULL, pose, twist, NULL, wrench, sdot, counts));
        clock_gettime(CLOCK_MONOTONIC, &t1);
        printf("%d evaluations: %.1f us per pfc_eval\n", n_rep,
               ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / 1e3 / (n_rep > 0 ? n_rep : 1));
        /* the same scene on Dual numbers with 6 seed directions (what one chunk of Radau's Jacobian costs, src/radau/
         * radau_functions.jl:2-14): unit seeds on the translation and on the twist */
        enum { ND = 6 };
        double d_pose[ND * 24] = {0}, d_twist[ND * 6] = {0}, d_wrench[ND * 6], d_sdot[ND * 6];
        for (int d = 0; d < 3; ++d) { d_pose[d * 24 + 9 + d] = 1.0; d_pose[d * 24 + 21 + d] = -1.0; d_twist[(3 + d) * 6 + d] = 1.0; }
        for (int k = 0; k < 20; ++k)
            CHECK(pfc_eval_dual(h, 1, ND, NULL, pose, twist, NULL, d_pose, d_twist, NULL, wrench, sdot, d_wrench, d_sdot, counts));
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int k = 0; k < n_rep; ++k)
            CHECK(pfc_eval_dual(h, 1, ND, NULL, pose, twist, NULL, d_pose, d_twist, NULL, wrench, sdot, d_wrench, d_sdot, counts));
        clock_gettime(CLOCK_MONOTONIC, &t1);
        printf("%d evaluations: %.1f us per pfc_eval_dual (6 directions); d f_z / d z = %.6e (analytic %.6e)\n", n_rep,
               ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) / 1e3 / (n_rep > 0 ? n_rep : 1), d_wrench[2 * 6 + 5],
               Ebar * 4 * r * r);
    }
    pfc_destroy(h);
    return sqrt(err / nrm) < 1.0e-8 ? 0 : 1;
}
