// pfc_kernels.h — device records and device-side math of libpfc_hip (gfx950 / CDNA4, wave64).
//
// Everything here is Float64 vector ALU work: branchy small-polygon geometry, no MFMA.  The arithmetic of every
// predicate-relevant expression follows the reference's operation order (cited per function, paths relative to
// the reference repository) so that candidate pairs and clipped-polygon vertex counts are bit-identical to the
// CPU oracle; the file is compiled with -ffp-contract=off and the reference's `muladd` sites are explicit fma.
#pragma once
#include <hip/hip_runtime.h>

namespace pfc {

constexpr int kInternal = -9999;  // src/obb/tree_types.jl:11,56
constexpr int kWave = 64;

// status word bits (device -> host)
enum : unsigned {
    kStNonFinite = 1u,
    kStFrontierOvf = 2u,
    kStCandOvf = 4u,
    kStTracOvf = 8u,
    kStBadIns = 16u,
    kStAbort = 32u,
    kStRecOvf = 64u,
    kStPolyOvf = 128u,   // a kept-polygon region ran full (sized so that it cannot: an internal error)
    kStHole = 256u,      // a work-list entry with an item index out of range was read (unwritten slot: an internal error)
    // (512, 1024: the one-launch kernel's own bits, pfc_fused.h)
    kStFixedSpan = 2048u,   // option "fixed_order": the candidates of one item span more chunks than k_shift_fixed's table holds
    kStFixedCover = 8192u,  // option "fixed_order": more candidates than the slots the sort was asked to cover
    kStFixedBig = 16384u,   // option "fixed_order": an item with more candidates than the per-item segment sort takes
    kStFixedList = 4096u,   // option "fixed_order": a key has more sum records than k_fixed_reduce orders at a time
};

// ---------------------------------------------------------------------------------------------------------------
// HBM records.  Element-expanded AoS: a lane that owns one (triangle, tet) pair gathers two contiguous records
// (96 B + 256 B) instead of 7 index-chased vertices; the per-tet zeta transforms are precomputed once at
// pfc_finalize because meshes are immutable afterwards (src/mechanism_scenario.jl:206).
// ---------------------------------------------------------------------------------------------------------------
struct alignas(16) NodeRec {  // OBB{c,e,R} + bin_BB_Tree links: src/obb/box_types.jl:4-9, tree_types.jl:1-16
    // first 64 bytes: everything an axis-aligned (merged / internal) box needs
    double c[3];
    double e[3];
    int child0, child1;   // links: index of an internal child, ~index of a leaf child
    int leaf;   // element index or kInternal
    int aabb;   // 1 if R is exactly the identity (all merged/internal boxes: src/obb/util.jl:47-51)
    double R[9];  // column-major; only read when aabb == 0 (tight-fitted leaves)
    double pad;
};
static_assert(sizeof(NodeRec) == 144, "NodeRec layout");
// child links on the device: ~index (negative) marks a LEAF child (encoded at pfc_add_mesh)
__host__ __device__ inline int node_index(int link) { return link < 0 ? ~link : link; }

// 48 bytes per node for the single-precision broadphase (three 16-byte loads): Float32 centre and extents, the box
// rotation as a unit quaternion (identity for every merged box) and the links.  For a leaf link0 is the element index.
// A node whose quaternion does not reproduce R to 4 u (improper / non-orthonormal R) carries e[0] = NaN: the sum S of
// the magnitudes of a test with such a node is NaN, which sends the pair to the exact Float64 test (test_pair_f32).
// (Round 3: the centre used to be Float64, 64 bytes per node, and the centre offset of two boxes was formed in Float64:
// 12 Float64 operations + 3 conversions of the test's ~230 instructions at twice the issue cost each, and a fourth load
// per node.  The Float32 offset has an ABSOLUTE error -- of the order of u x the size of the scene, not of the boxes --
// which the error radius carries as a per-item constant, ItemRec.bp_eabs.)
struct alignas(16) NodeF {
    float c[3];
    float e[3];
    float q[4];        // w, x, y, z
    int link0, link1;  // children (index, or ~index for a leaf child); leaf: link0 = element index
};
static_assert(sizeof(NodeF) == 48, "NodeF layout");

// Records are sized and aligned to the 128-byte L2 line: a lane's gather touches exactly one line per triangle and one
// per tet (x_ζ²_r², all the clip needs); the second line of a tet is read only by pairs that survive the clip.  With
// the previous packing (96-byte triangles, x_ζ²_r² straddling both lines of a tet) the per-XCD hot set of the C3
// batch was 4.1 MB against a 4 MiB L2 and 43 % of the narrowphase's L2 requests missed (rocprofv3 TCC_HIT/TCC_MISS).
struct alignas(128) TriRec {  // triangle_vertices (non_friction.jl:145) + triangleNormal (geometry_kernel.jl:10)
    double v[9];  // v1, v2, v3 in frame r1
    double n[3];  // unit normal in frame r1
    double pad[4];
};
static_assert(sizeof(TriRec) == 128, "TriRec layout");

struct alignas(128) TetRec {  // tetrahedron_vertices_ϵ + calc_ζ_transforms (non_friction.jl:150-162)
    double xzr[16];  // line 0: x_ζ2_r2 = inv([v1 v2 v3 v4; 1 1 1 1]), column-major 4x4
    double xrz[12];  // line 1: x_r2_ζ2 rows 1..3, column-major 3x4 (= the 4 vertices); row 4 is all ones
    double epsr[4];  //         ϵ_r2 = ϵ2 * x_ζ2_r2   (1x4 affine pressure functional), non_friction.jl:202
};
static_assert(sizeof(TetRec) == 256, "TetRec layout");

// A pointer read out of an ItemRec is "generic" to the compiler: it emits flat_load (counted on vmcnt AND lgkmcnt, aperture
// check in the address path).  Hot gathers go through pointers cast to the global address space instead (global_load).
typedef int vec4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) vec4i gvec4i;
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) TriRec GTriRec;
typedef __attribute__((address_space(1))) TetRec GTetRec;
typedef __attribute__((address_space(1))) NodeRec GNodeRec;

struct MeshDev {
    const NodeRec *nodes;
    const NodeF *nodesf;
    const TriRec *tri;
    const TetRec *tet;
    const double *tet_eps;   // 4 per tet: raw ϵ of the tet's vertices (needed by the tet-tet equal-pressure plane)
    double Ebar;
    double cmax;             // max over the nodes of |c|_1 (box centres, mesh frame): enters ItemRec.bp_eabs
    int n_tri, n_tet, n_node, depth;
};

struct InsDev {  // ContactInstructions + friction model: src/mechanism_scenario.jl:5-49
    int m1, m2, model, nq;
    double chi, mu_s, mu_d, v_c, tau, k_bar, magic, pad;
};

struct alignas(16) ItemRec {  // per (instruction, pose) item: outputs of refreshBodyBodyTransform!/Cache!
    double R21[9], t21[3];    // x_r2_r1
    double R12[9], t12[3];    // x_r1_r2  (= TT_Cache.R_a_b, t_a_b: src/obb/tree_types.jl:43-50)
    double w[3], v[3];        // twist_r2_r1_r2
    double s[6];              // bristle state
    double chi, Ebar, mu_s, mu_d, v_c, tau, k_bar, magic;
    const NodeRec *nodes1, *nodes2;
    const NodeF *nf1, *nf2;
    const TriRec *tri;       // mesh_1 triangles, or null for a tet-tet instruction
    const TetRec *tet;       // mesh_2 tets
    const TetRec *tet1;      // mesh_1 tets (tet-tet) or null
    const double *eps1, *eps2;
    double Ebar1;
    int model, nq, ins, pad;  // nq = number of quadrature points (1 or 3)
    // the broadphase's single-precision filter composes rotations as quaternions: R_a_b as a unit quaternion, formed and
    // checked in Float64 once per item (pose_quat); pose_exact: the pose is not a proper rotation, every node test of the
    // item is settled by the exact Float64 test
    float q12[4];
    int pose_exact;
    // absolute part of the error radius of the single-precision test: 24 u (cmax_1 + cmax_2 + max |t12_i|), rounded up
    // (pfc_bp.h, "Error radius E", (0))
    float bp_eabs;
    int pad2[2];
};

struct alignas(16) WorkRec {  // frontier entry (item, node_a, node_b) or candidate (item, i_1, i_2)
    int item, a, b, pad;
};

// per-item accumulator slots (doubles).  The bristle model's patch stiffness (calc_patch_spatial_stiffness!,
// src/contact_algorithms_friction.jl:147-169) is a sum over traction points of polynomials in r - cop.  It is
// accumulated in the same pass that finds the cop, without cancellation: every lane sums its polygon's moments about
// the polygon centroid, a wave run shifts them to the run's own pressure centroid c_w and writes one 32-double
// record, and k_shift moves each record from c_w to the item's cop (parallel-axis terms only: all shift distances
// are of the order of the patch size) before adding it here.
constexpr int kAccWrench = 0;   // 6: regularized total wrench, or bristle normal wrench [ang; lin]
constexpr int kAccIp = 6;       // 1: S   = sum w            (w = p dA)
constexpr int kAccIpc = 7;      // 3: Sr  = sum w r          (about the frame origin: gives the cop)
constexpr int kAccSnn = 10;     // 6: sum w n n'             (xx xy xz yy yz zz)
constexpr int kAccSan = 16;     // 9: sum w (x x n) n'       (column-major 3x3), x = r - cop
constexpr int kAccSaa = 25;     // 6: sum w (x x n)(x x n)'
constexpr int kAccSrr = 31;     // 6: sum w x x'
constexpr int kAccFric = 37;    // 6: friction wrench about the cop [ang; lin]
constexpr int kAccStride = 44;
// moment record of one run of an item: item, W, c(3), Snn 6, San 9, Saa 6, Srr 6 (about the reference point c), and the
// first moment m1 = sum w (r - c) (3; zero when c is the run's own pressure centroid: k_narrow's full modes), padding
constexpr int kRecStride = 36;
// records of option "fixed_order" (k_integ_fixed -> k_shift_fixed): the same 35 numbers, then the run's ten sums, its chunk index and
// the slot of the item's previous record
constexpr int kRecTen = 35, kRecChunk = 45, kRecNext = 46, kRecStrideFixed = 48;
// per-item derived results (doubles)
constexpr int kResCop = 0;      // 3
constexpr int kResSinv = 3;     // 6
constexpr int kResDelta = 9;    // 6
constexpr int kResKis = 15;     // 36  K̄^{-1/2}
constexpr int kResK = 51;       // 36
constexpr int kResV = 88;       // 36  eigenvectors of K̄ (columns) ...
constexpr int kResLam = 124;    // 6   ... and its eigenvalues, as the Jacobi iteration left them: k_dual_eig differentiates THIS decomposition
constexpr int kResStride = 130;

// ---------------------------------------------------------------------------------------------------------------
// small vector helpers (StaticArrays evaluation order: left-to-right sums, no contraction)
// ---------------------------------------------------------------------------------------------------------------
struct V3 { double x, y, z; };
__device__ __forceinline__ V3 mk3(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 ld3(const double *p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator/(V3 a, double s) { return V3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// StaticArrays normalize(a) = inv(norm(a)) * a
__device__ __forceinline__ V3 normalize(V3 a) {
    double s = 1.0 / __builtin_sqrt(dot(a, a));
    return V3{s * a.x, s * a.y, s * a.z};
}
// src/math_kernel/geometry_kernel.jl:5-9
__device__ __forceinline__ V3 vector_area(V3 a, V3 b, V3 c) { return cross(b - a, c - b) * 0.5; }
__device__ __forceinline__ double triangle_area(V3 a, V3 b, V3 c, V3 n) { return dot(n, vector_area(a, b, c)); }
// src/math_kernel/vector_projections.jl:2-7
__device__ __forceinline__ V3 vec_sub_vec_proj(V3 v, V3 n) {
    double t = -dot(v, n);
    return V3{__builtin_fma(t, n.x, v.x), __builtin_fma(t, n.y, v.y), __builtin_fma(t, n.z, v.z)};
}
// src/contact_algorithms_friction.jl:2-10
__device__ __forceinline__ double clamped_piecewise(double x, double x1, double x2, double y1, double y2) {
    double k = (y2 - y1) / (x2 - x1);
    double y = y1 + (x - x1) * k;
    return (y > y1) ? y1 : ((y < y2) ? y2 : y);
}

// 4x4 inverse: explicit cofactor expansion x (1/det) -- the form StaticArrays 0.10.3 (the version the reference pins,
// Manifest.toml) published for inv(::SMatrix{4,4}) (src/inv.jl, src/det.jl), the reference's call at
// src/contact_algorithms_non_friction.jl:160: every entry six left-to-right triple products times idet, the determinant 24
// left-to-right quadruple products.  StaticArrays is not vendored under the reference, so the term order is written from the
// published source as recalled (the one unpinned rounding of the path: DESIGN.md section 2 counts the predicate outcomes
// that depend on it; the test suite's CPU restatement uses the same expressions).  Rounds 1-2 used the adjugate from
// 2x2 minors.  a, b column-major.  Returns 1/det.
__device__ __forceinline__ double inv4(const double *a, double *b) {
#define A(i, j) a[((i) - 1) + 4 * ((j) - 1)]
#define B(i, j) b[((i) - 1) + 4 * ((j) - 1)]
#define L(k) a[(k) - 1]
    const double det =
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
    const double idet = 1.0 / det;
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
#undef L
    return idet;
}

// ---------------------------------------------------------------------------------------------------------------
// OBB-OBB separating-axis test, src/obb/bb_intersection.jl:2-74.
// The reference composes three 4x4 homogeneous transforms per node pair (:3-7); products with the structural
// zeros/ones of those matrices are exact, so they are skipped here without changing any bit of R_tot / t.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool sat15(const double *ea, const double *eb, const double *t, const double *R,
                                      const double *aR) {
#define R_(i, j) R[(i) + 3 * (j)]
#define AR_(i, j) aR[(i) + 3 * (j)]
    bool sep = false;
    // face test 1/2 (:29-32)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double rb = (AR_(i, 0) * eb[0] + AR_(i, 1) * eb[1]) + AR_(i, 2) * eb[2];
        sep |= (ea[i] + rb) < __builtin_fabs(t[i]);
    }
    if (sep) return false;
    // face test 2/2 (:35-38)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double tl = __builtin_fabs((R_(0, j) * t[0] + R_(1, j) * t[1]) + R_(2, j) * t[2]);
        double ra = (AR_(0, j) * ea[0] + AR_(1, j) * ea[1]) + AR_(2, j) * ea[2];
        sep |= (ra + eb[j]) < tl;
    }
    if (sep) return false;
    // s100(r) = (r2, r1, r1), s221(r) = (r3, r3, r2) (1-based; :14-15)
    constexpr int i100[3] = {1, 0, 0}, i221[3] = {2, 2, 1};
    // cross tests 1/3..3/3 (:56-72)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double tl = __builtin_fabs(t[2] * R_(1, j) - t[1] * R_(2, j));
        double ra = ea[1] * AR_(2, j) + ea[2] * AR_(1, j);
        double rb = eb[i100[j]] * AR_(0, i221[j]) + eb[i221[j]] * AR_(0, i100[j]);
        sep |= (ra + rb) < tl;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double tl = __builtin_fabs(t[0] * R_(2, j) - t[2] * R_(0, j));
        double ra = ea[0] * AR_(2, j) + ea[2] * AR_(0, j);
        double rb = eb[i100[j]] * AR_(1, i221[j]) + eb[i221[j]] * AR_(1, i100[j]);
        sep |= (ra + rb) < tl;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double tl = __builtin_fabs(t[1] * R_(0, j) - t[0] * R_(1, j));
        double ra = ea[0] * AR_(1, j) + ea[1] * AR_(0, j);
        double rb = eb[i100[j]] * AR_(2, i221[j]) + eb[i221[j]] * AR_(2, i100[j]);
        sep |= (ra + rb) < tl;
    }
#undef R_
#undef AR_
    return !sep;
}

// Single-precision filter in front of sat15().  The reference's predicate is "separated on axis L iff
// (r_a + r_b) < |T.L|" in Float64.  With d = |T.L| - (r_a + r_b) evaluated in FP32 from inputs rounded to FP32, the
// absolute error of d is below 16 u S, u = 2^-24, S = |t|_1 + sum e_a + sum e_b (every axis is <= 12 roundings of
// terms bounded by S, |R| <= 1; the +1e-14 of abs_R is far below u).  So d > E proves separation, d < -E on all 15
// axes proves overlap, anything else is undecided and goes to the Float64 test: the boolean returned for a node pair
// is bit for bit the reference's.  t and R come from the Float64 composition, i.e. t is the centre offset of two
// nearby boxes, so S is of the order of the box sizes and the undecided band is ~1e-6 of the typical margin.
// FP32 vector ops issue at twice the FP64 rate on CDNA4 and use explicit fma.
// returns 0 = separated, 1 = overlapping, 2 = undecided
__device__ __forceinline__ int sat15_f32(const double *ea64, const double *eb64, const double *t64, const double *R64) {
    float ea[3], eb[3], t[3], R[9], aR[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) { ea[k] = (float)ea64[k]; eb[k] = (float)eb64[k]; t[k] = (float)t64[k]; }
#pragma unroll
    for (int k = 0; k < 9; ++k) { R[k] = (float)R64[k]; aR[k] = __builtin_fabsf(R[k]); }
    const float S = ((__builtin_fabsf(t[0]) + __builtin_fabsf(t[1])) + __builtin_fabsf(t[2])) + ((ea[0] + ea[1]) + ea[2]) +
                    ((eb[0] + eb[1]) + eb[2]);
    const float E = 9.6e-7f * S;   // 16 * 2^-24 = 9.54e-7
    bool sep = false, hit = true;
#define R_(i, j) R[(i) + 3 * (j)]
#define AR_(i, j) aR[(i) + 3 * (j)]
#define AXIS_(tl, rsum)                      \
    do {                                     \
        const float d_ = (tl) - (rsum);      \
        sep |= d_ > E;                       \
        hit &= d_ < -E;                      \
    } while (0)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float rb = __builtin_fmaf(AR_(i, 2), eb[2], __builtin_fmaf(AR_(i, 1), eb[1], AR_(i, 0) * eb[0]));
        AXIS_(__builtin_fabsf(t[i]), ea[i] + rb);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float tl = __builtin_fabsf(__builtin_fmaf(R_(2, j), t[2], __builtin_fmaf(R_(1, j), t[1], R_(0, j) * t[0])));
        const float ra = __builtin_fmaf(AR_(2, j), ea[2], __builtin_fmaf(AR_(1, j), ea[1], AR_(0, j) * ea[0]));
        AXIS_(tl, ra + eb[j]);
    }
    constexpr int i100[3] = {1, 0, 0}, i221[3] = {2, 2, 1};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float tl = __builtin_fabsf(__builtin_fmaf(t[2], R_(1, j), -(t[1] * R_(2, j))));
        const float ra = __builtin_fmaf(ea[1], AR_(2, j), ea[2] * AR_(1, j));
        const float rb = __builtin_fmaf(eb[i100[j]], AR_(0, i221[j]), eb[i221[j]] * AR_(0, i100[j]));
        AXIS_(tl, ra + rb);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float tl = __builtin_fabsf(__builtin_fmaf(t[0], R_(2, j), -(t[2] * R_(0, j))));
        const float ra = __builtin_fmaf(ea[0], AR_(2, j), ea[2] * AR_(0, j));
        const float rb = __builtin_fmaf(eb[i100[j]], AR_(1, i221[j]), eb[i221[j]] * AR_(1, i100[j]));
        AXIS_(tl, ra + rb);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float tl = __builtin_fabsf(__builtin_fmaf(t[1], R_(0, j), -(t[0] * R_(1, j))));
        const float ra = __builtin_fmaf(ea[0], AR_(1, j), ea[1] * AR_(0, j));
        const float rb = __builtin_fmaf(eb[i100[j]], AR_(2, i221[j]), eb[i221[j]] * AR_(2, i100[j]));
        AXIS_(tl, ra + rb);
    }
#undef AXIS_
#undef R_
#undef AR_
    return sep ? 0 : (hit ? 1 : 2);
}

// Unit quaternion (Float32) of the item's R_a_b, formed in Float64 (largest of 4w^2, 4x^2, 4y^2, 4z^2 as pivot, no
// square root before the normalisation), and the check the error bound above rests on, IN FLOAT64: the rounded quaternion
// reproduces R_a_b to 4 u per entry and has |q|^2 within 2.25 u of 1.  A pose that fails it (not a proper rotation) returns
// false: every node test of the item then goes to the exact Float64 test.
__device__ __forceinline__ bool pose_quat(const double *R, float *qf) {
    const double d0 = ((1.0 + R[0]) + R[4]) + R[8], d1 = ((1.0 + R[0]) - R[4]) - R[8];
    const double d2 = ((1.0 - R[0]) + R[4]) - R[8], d3 = ((1.0 - R[0]) - R[4]) + R[8];
    double q0, q1, q2, q3;
    if (d0 >= d1 && d0 >= d2 && d0 >= d3) { q0 = d0; q1 = R[5] - R[7]; q2 = R[6] - R[2]; q3 = R[1] - R[3]; }
    else if (d1 >= d2 && d1 >= d3) { q0 = R[5] - R[7]; q1 = d1; q2 = R[3] + R[1]; q3 = R[6] + R[2]; }
    else if (d2 >= d3) { q0 = R[6] - R[2]; q1 = R[3] + R[1]; q2 = d2; q3 = R[7] + R[5]; }
    else { q0 = R[1] - R[3]; q1 = R[6] + R[2]; q2 = R[7] + R[5]; q3 = d3; }
    const double qn = __builtin_sqrt(((q0 * q0 + q1 * q1) + q2 * q2) + q3 * q3);
    qf[0] = (float)(q0 / qn); qf[1] = (float)(q1 / qn); qf[2] = (float)(q2 / qn); qf[3] = (float)(q3 / qn);
    const double w = qf[0], x = qf[1], y = qf[2], z = qf[3];
    const double Rq[9] = {1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w),
                          2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w),
                          2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)};
    double worst = 0.0;
#pragma unroll
    for (int j = 0; j < 9; ++j) worst = fmax(worst, __builtin_fabs(Rq[j] - R[j]));
    const double n2 = ((w * w + x * x) + y * y) + z * z;
    const double u = 5.9604644775390625e-8;
    // written so that a NaN anywhere fails the check
    return (worst <= 4.0 * u) && (__builtin_fabs(n2 - 1.0) <= 2.25 * u);
}

// The traction points of ONE fan triangle (v1, v2, polygon centroid): fillTractionCacheForTriangle! +
// fillTractionCacheInnerLoop! (src/contact_algorithms_non_friction.jl:236-265) with TriTetQuadRule rules 1 and 2 (literal
// decimals of src/clip/quadrature.jl:24-39).  The ONE statement of r, p and dA shared by every value kernel (k_narrow,
// k_integ, k_fric, k_fused): the per-item traction counts and the bit-identity of the traction points between the
// passes rest on these expressions being the same everywhere.  body(r, rdot, p, dA) is called for every point with
// 0 < p (:245); returns the number of such points.
struct PointParams {
    V3 w, vl;                          // twist of the item: angular, linear
    double chi, Ebar, er0, er1, er2, er3;   // damping, modulus, the tet's strain row eps_r2
    int nq;                            // quadrature points per triangle: 1 or 3
};
template <class F>
__device__ __forceinline__ int fan_triangle_points(const PointParams &c, const V3 &v1, const V3 &v2, const V3 &cen, const V3 &nh,
                                                   F &&body) {
    const double area = triangle_area(v1, v2, cen, nh);
    if (!(0.0 < area)) return 0;       // :232
    int n_pt = 0;
    for (int q = 0; q < c.nq; ++q) {
        double q0, q1, q2, qw;
        if (c.nq == 1) {
            q0 = q1 = q2 = 0.33333333333333331483; qw = 1.0;
        } else {
            const double qa = 0.16666666666666674068, qb = 0.66666666666666651864;
            q0 = (q == 1) ? qb : qa; q1 = (q == 0) ? qb : qa; q2 = (q == 2) ? qb : qa;
            qw = 0.33333333333333331483;
        }
        const V3 r = mk3((v1.x * q0 + v2.x * q1) + cen.x * q2, (v1.y * q0 + v2.y * q1) + cen.y * q2,
                         (v1.z * q0 + v2.z * q1) + cen.z * q2);
        double eq = __builtin_fma(c.er0, r.x, c.er3);
        eq = __builtin_fma(c.er1, r.y, eq);
        eq = __builtin_fma(c.er2, r.z, eq);
        const V3 rdot = c.vl + cross(c.w, r);
        const double ee = -dot(mk3(c.er0, c.er1, c.er2), rdot);
        const double damp = fmax(0.0, 1.0 + c.chi * ee);
        const double p = eq * c.Ebar * damp;
        const double dA = qw * area;
        if (!(0.0 < p)) continue;      // :245
        ++n_pt;
        body(r, rdot, p, dA);
    }
    return n_pt;
}

// The 15 axes on Float32 inputs with a caller-supplied error radius E (see k_bp_dfs32).
// returns 0 = separated, 1 = overlapping, 2 = undecided.  "Some axis has d > E" and "every axis has d < -E" are both
// statements about max d, so only the maximum is carried (v_max3_f32: 8 instructions instead of 30 compares and as many
// scalar mask operations).  v_max drops a NaN operand, so the caller sends non-finite or huge inputs (S >= 1e18: products
// could overflow to inf - inf) to the exact test itself; for finite d the verdict is the one of the compare chain.
__device__ __forceinline__ int sat15_f32_core(const float *ea, const float *eb, const float *t, const float *R, float E) {
    float aR[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) aR[k] = __builtin_fabsf(R[k]);
    float dmax;
#define R_(i, j) R[(i) + 3 * (j)]
#define AR_(i, j) aR[(i) + 3 * (j)]
#define AXIS_(tl, rsum) dmax = __builtin_fmaxf(dmax, (tl) - (rsum))
    {
        const float rb = __builtin_fmaf(AR_(0, 2), eb[2], __builtin_fmaf(AR_(0, 1), eb[1], AR_(0, 0) * eb[0]));
        dmax = __builtin_fabsf(t[0]) - (ea[0] + rb);
    }
#pragma unroll
    for (int i = 1; i < 3; ++i) {
        const float rb = __builtin_fmaf(AR_(i, 2), eb[2], __builtin_fmaf(AR_(i, 1), eb[1], AR_(i, 0) * eb[0]));
        AXIS_(__builtin_fabsf(t[i]), ea[i] + rb);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float tl = __builtin_fabsf(__builtin_fmaf(R_(2, j), t[2], __builtin_fmaf(R_(1, j), t[1], R_(0, j) * t[0])));
        const float ra = __builtin_fmaf(AR_(2, j), ea[2], __builtin_fmaf(AR_(1, j), ea[1], AR_(0, j) * ea[0]));
        AXIS_(tl, ra + eb[j]);
    }
    constexpr int i100[3] = {1, 0, 0}, i221[3] = {2, 2, 1};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float tl = __builtin_fabsf(__builtin_fmaf(t[2], R_(1, j), -(t[1] * R_(2, j))));
        const float ra = __builtin_fmaf(ea[1], AR_(2, j), ea[2] * AR_(1, j));
        const float rb = __builtin_fmaf(eb[i100[j]], AR_(0, i221[j]), eb[i221[j]] * AR_(0, i100[j]));
        AXIS_(tl, ra + rb);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float tl = __builtin_fabsf(__builtin_fmaf(t[0], R_(2, j), -(t[2] * R_(0, j))));
        const float ra = __builtin_fmaf(ea[0], AR_(2, j), ea[2] * AR_(0, j));
        const float rb = __builtin_fmaf(eb[i100[j]], AR_(1, i221[j]), eb[i221[j]] * AR_(1, i100[j]));
        AXIS_(tl, ra + rb);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float tl = __builtin_fabsf(__builtin_fmaf(t[1], R_(0, j), -(t[0] * R_(1, j))));
        const float ra = __builtin_fmaf(ea[0], AR_(1, j), ea[1] * AR_(0, j));
        const float rb = __builtin_fmaf(eb[i100[j]], AR_(2, i221[j]), eb[i221[j]] * AR_(2, i100[j]));
        AXIS_(tl, ra + rb);
    }
#undef AXIS_
#undef R_
#undef AR_
    return (dmax > E) ? 0 : ((dmax < -E) ? 1 : 2);
}

// Hamilton product r = a (x) b of quaternions (w, x, y, z), Float32: each component is a four-term dot product evaluated
// as one multiplication and three fused multiply-adds (four roundings, each of a partial sum bounded by |a| |b|).
// CONJ_A: a is taken as its conjugate (sign flips are source modifiers, free).
template <bool CONJ_A>
__device__ __forceinline__ void quat_mul(const float *a, const float *b, float *r) {
    const float aw = a[0], ax = CONJ_A ? -a[1] : a[1], ay = CONJ_A ? -a[2] : a[2], az = CONJ_A ? -a[3] : a[3];
    const float bw = b[0], bx = b[1], by = b[2], bz = b[3];
    r[0] = __builtin_fmaf(-az, bz, __builtin_fmaf(-ay, by, __builtin_fmaf(-ax, bx, aw * bw)));
    r[1] = __builtin_fmaf(-az, by, __builtin_fmaf(ay, bz, __builtin_fmaf(ax, bw, aw * bx)));
    r[2] = __builtin_fmaf(az, bx, __builtin_fmaf(ay, bw, __builtin_fmaf(-ax, bz, aw * by)));
    r[3] = __builtin_fmaf(az, bw, __builtin_fmaf(-ay, bx, __builtin_fmaf(ax, by, aw * bz)));
}

// t = R(q)' v = R(conj q) v without forming R: with u = (x, y, z), R(conj q) v = v + 2 u x (u x v - w v)   (18 instructions)
__device__ __forceinline__ void quat_rot_inv(const float *q, const float *v, float *t) {
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float c0 = __builtin_fmaf(-w, v[0], __builtin_fmaf(y, v[2], -(z * v[1])));
    const float c1 = __builtin_fmaf(-w, v[1], __builtin_fmaf(z, v[0], -(x * v[2])));
    const float c2 = __builtin_fmaf(-w, v[2], __builtin_fmaf(x, v[1], -(y * v[0])));
    const float d0 = __builtin_fmaf(y, c2, -(z * c1)), d1 = __builtin_fmaf(z, c0, -(x * c2)), d2 = __builtin_fmaf(x, c1, -(y * c0));
    t[0] = __builtin_fmaf(2.0f, d0, v[0]); t[1] = __builtin_fmaf(2.0f, d1, v[1]); t[2] = __builtin_fmaf(2.0f, d2, v[2]);
}

// rotation matrix (column-major 3x3) of a unit quaternion, Float32
__device__ __forceinline__ void quat_to_R(const float *q, float *R) {
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float x2 = x + x, y2 = y + y, z2 = z + z;
    const float xx = x * x2, yy = y * y2, zz = z * z2, xy = x * y2, xz = x * z2, yz = y * z2;
    const float wx = w * x2, wy = w * y2, wz = w * z2;
    R[0] = 1.0f - (yy + zz); R[3] = xy - wz; R[6] = xz + wy;
    R[1] = xy + wz; R[4] = 1.0f - (xx + zz); R[7] = yz - wx;
    R[2] = xz - wy; R[5] = yz + wx; R[8] = 1.0f - (xx + yy);
}

// BB_BB_intersect(tt, a, b) (:2-12): dh_final = inv(dh_a) * dh_a_b * dh_b, then the 15-axis test.
__device__ __forceinline__ bool bb_bb_intersect(const NodeRec &a, const NodeRec &b, const double *Rab,
                                                const double *tab) {
    double R[9], aR[9], t[3];
    // tmp = i_dh_a * dh_a_b, with i_dh_a = [Ra' | (-Ra') ca]
    double T[9], tt[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        // row i of Ra' = column i of Ra
        double r0 = a.R[3 * i], r1 = a.R[3 * i + 1], r2 = a.R[3 * i + 2];
        double nt = ((-r0) * a.c[0] + (-r1) * a.c[1]) + (-r2) * a.c[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) T[i + 3 * j] = (r0 * Rab[3 * j] + r1 * Rab[3 * j + 1]) + r2 * Rab[3 * j + 2];
        tt[i] = ((r0 * tab[0] + r1 * tab[1]) + r2 * tab[2]) + nt;
    }
    // fin = tmp * dh_b, with dh_b = [Rb | cb]
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double r = (T[i] * b.R[3 * j] + T[i + 3] * b.R[3 * j + 1]) + T[i + 6] * b.R[3 * j + 2];
            R[i + 3 * j] = r;
            aR[i + 3 * j] = __builtin_fabs(r) + 1.0e-14;
        }
        t[i] = ((T[i] * b.c[0] + T[i + 3] * b.c[1]) + T[i + 6] * b.c[2]) + tt[i];
    }
    return sat15(a.e, b.e, t, R, aR);
}

// The same composition without the test: R_tot, |R_tot| + 1e-14 and t of dh_final (:7-10)
__device__ __forceinline__ void bb_compose(const NodeRec &a, const NodeRec &b, const double *Rab, const double *tab,
                                           double *R, double *aR, double *t) {
    double T[9], tt[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double r0 = a.R[3 * i], r1 = a.R[3 * i + 1], r2 = a.R[3 * i + 2];
        double nt = ((-r0) * a.c[0] + (-r1) * a.c[1]) + (-r2) * a.c[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) T[i + 3 * j] = (r0 * Rab[3 * j] + r1 * Rab[3 * j + 1]) + r2 * Rab[3 * j + 2];
        tt[i] = ((r0 * tab[0] + r1 * tab[1]) + r2 * tab[2]) + nt;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double r = (T[i] * b.R[3 * j] + T[i + 3] * b.R[3 * j + 1]) + T[i + 6] * b.R[3 * j + 2];
            R[i + 3 * j] = r;
            aR[i + 3 * j] = __builtin_fabs(r) + 1.0e-14;
        }
        t[i] = ((T[i] * b.c[0] + T[i + 3] * b.c[1]) + T[i + 6] * b.c[2]) + tt[i];
    }
}

// BB_BB_intersect for two axis-aligned boxes (R_a = R_b = I, every internal node).  The reference still composes the
// three 4x4 transforms (:3-7); with identity rotations that product is exactly R_tot = R_a_b and
// t = R_a_b c_b + (t_a_b - c_a) with the same roundings (products with the structural 0/1 entries are exact), so
// this shortcut returns the same boolean bit for bit.  aRab = |R_a_b| + 1e-14 is per item.
__device__ __forceinline__ bool bb_bb_intersect_aabb(const double *ca, const double *ea, const double *cb,
                                                     const double *eb, const double *Rab, const double *aRab,
                                                     const double *tab) {
    double t[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        t[i] = ((Rab[i] * cb[0] + Rab[i + 3] * cb[1]) + Rab[i + 6] * cb[2]) + (tab[i] - ca[i]);
    return sat15(ea, eb, t, Rab, aRab);
}

// ---------------------------------------------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ int prefix_count(unsigned long long mask) {  // set bits of mask below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Segmented wave reduction.  Work lists are grouped by item in runs (a seed's candidates, a polygon's traction
// points), so the 64 lanes of a wave hold a few runs of equal keys.  seg_setup() finds the runs with one ballot;
// seg_sum() is a segmented inclusive scan whose value on the last lane of each run is the run's total; that lane
// issues the run's single atomic.  key < 0 marks a lane without contribution.
//
// The scan runs on DPP (data-parallel primitives: VALU cross-lane moves, no LDS round trip): row_shr 1/2/4/8 inside
// the four 16-lane rows, then row_bcast:15 (last lane of a row into the next row; rows 1 and 3) and row_bcast:31
// (lane 31 into lanes 32..63).  A step's contribution is added only if its source lane lies inside the lane's run.
struct Seg {
    unsigned steps;  // bit k set: step k of the scan (shr1, shr2, shr4, shr8, bcast15, bcast31) stays inside the run
    bool tail;       // last lane of its run
    bool valid;      // key >= 0
    int tail_lane;   // lane index of the last lane of this lane's run
};
__device__ __forceinline__ Seg seg_setup(int key) {
    const int lane = lane_id();
    const int prev = __shfl_up(key, 1, 64);
    const bool head = (lane == 0) || (key != prev);
    const unsigned long long H = __ballot(head);
    const unsigned long long below = H & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int start = 63 - __builtin_clzll(below);
    Seg s;
    s.steps = 0;
    s.steps |= (lane - 1 >= start) ? 1u : 0u;
    s.steps |= (lane - 2 >= start) ? 2u : 0u;
    s.steps |= (lane - 4 >= start) ? 4u : 0u;
    s.steps |= (lane - 8 >= start) ? 8u : 0u;
    s.steps |= ((lane & 16) && ((lane & ~15) - 1 >= start)) ? 16u : 0u;
    s.steps |= (lane >= 32 && 31 >= start) ? 32u : 0u;
    s.tail = (lane == 63) || (((H >> (lane + 1)) & 1ull) != 0);
    s.valid = key >= 0;
    const unsigned long long above = (lane == 63) ? 0ull : (H >> (lane + 1));
    s.tail_lane = above ? lane + __builtin_ctzll(above) : 63;
    return s;
}
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_move(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, kRowMask, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, kRowMask, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int kCtrl, int kRowMask>
__device__ __forceinline__ int dpp_move(int v) {
    return __builtin_amdgcn_update_dpp(0, v, kCtrl, kRowMask, 0xF, true);
}
// plain (unsegmented) inclusive scan of an int over the wave, same DPP steps
__device__ __forceinline__ int seg_incl_scan(int v) {
    const int lane = lane_id();
    int t;
    t = dpp_move<0x111, 0xF>(v); v += t;
    t = dpp_move<0x112, 0xF>(v); v += t;
    t = dpp_move<0x114, 0xF>(v); v += t;
    t = dpp_move<0x118, 0xF>(v); v += t;
    t = dpp_move<0x142, 0xA>(v); v += (lane & 16) ? t : 0;
    t = dpp_move<0x143, 0xC>(v); v += (lane >= 32) ? t : 0;
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// Region-partitioned append lists.  k_narrow appends kept polygons and moment records once per wave round; with ONE
// counter that is 60 k returning atomics on a single address per C3 batch, which the L2 serves serially (~11 ns each):
// 0.21 ms of a 0.85 ms kernel.  The lists are therefore cut into kRgn regions with a counter each, on separate 128-byte
// lines (word 0: polygons, word 1: records, reserved together by one 64-bit atomic); workgroup b appends to region
// b % kRgn, region c owns slots [c cap, (c + 1) cap).  A consumer maps its flat wave index onto (region, offset) with
// a scan of the per-region wave counts held one region per lane.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kRgn = 64;
constexpr int kRgnStride = 32;   // ints between two regions' counters
struct RgnScan { int cnt, excl, total; };   // lane c: live slots of region c, waves before region c; total waves (uniform)
__device__ __forceinline__ RgnScan rgn_scan(const int *rgn, int word, int cap, int lane) {
    RgnScan r;
    r.cnt = rgn[lane * kRgnStride + word];
    if (r.cnt > cap) r.cnt = cap;
    const int nw = (r.cnt + 63) >> 6;
    int incl = nw;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    r.excl = incl - nw;
    r.total = __builtin_amdgcn_readlane(incl, 63);
    return r;
}
// wave w (uniform, < total) of the pass: first slot and number of live slots (1..64)
__device__ __forceinline__ void rgn_locate(const RgnScan &r, int w, int cap, int &slot0, int &n_live) {
    const unsigned long long m = __ballot(r.excl <= w);          // lane 0 always qualifies
    const int c = 63 - __builtin_clzll(m);                       // the last region that starts at or before w
    const int start = __shfl(r.excl, c, 64), cnt = __shfl(r.cnt, c, 64);
    const int off = (w - start) * 64;
    slot0 = c * cap + off;
    n_live = cnt - off < 64 ? cnt - off : 64;
}
template <class T>
__device__ __forceinline__ T seg_sum(T v, const Seg &s) {
    T t;
    t = dpp_move<0x111, 0xF>(v); v += (s.steps & 1u) ? t : T(0);   // row_shr:1
    t = dpp_move<0x112, 0xF>(v); v += (s.steps & 2u) ? t : T(0);   // row_shr:2
    t = dpp_move<0x114, 0xF>(v); v += (s.steps & 4u) ? t : T(0);   // row_shr:4
    t = dpp_move<0x118, 0xF>(v); v += (s.steps & 8u) ? t : T(0);   // row_shr:8
    t = dpp_move<0x142, 0xA>(v); v += (s.steps & 16u) ? t : T(0);  // row_bcast:15 into rows 1, 3
    t = dpp_move<0x143, 0xC>(v); v += (s.steps & 32u) ? t : T(0);  // row_bcast:31 into rows 2, 3
    return v;
}

}  // namespace pfc
