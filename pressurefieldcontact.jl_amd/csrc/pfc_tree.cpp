// pfc_tree.cpp — host-side OBB-tree construction for pfc_add_mesh.
//
// Restates eMesh_to_tree (src/geometry/blob_types.jl:136-173): leaf AABBs, face/edge adjacency, bottom-up merging of
// neighbouring blobs in order of marginal cost, a median-split top-down pass over whatever blobs remain, and tight
// OBBs on the leaves (src/obb/obb_construction.jl:13-41).  The reference's merge order among equal costs follows the
// internal heap order of DataStructures.PriorityQueue and the iteration order of Dict/Set; here ties are broken by
// the (cost, key) order, so trees are quality-equivalent, not node-for-node identical, to Julia-built ones (hosts
// that need identical candidate sets pass their own flattened tree to pfc_add_mesh instead).
//
// Plain C++ (no device code); compiled into libpfc_hip.so next to pfc_hip.hip.
#include "../../include/pfc.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <numeric>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

constexpr int kInternal = -9999;  // src/obb/tree_types.jl:11

thread_local std::string g_tree_err;

struct V3 {
    double x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 vmin(V3 a, V3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }
inline double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 normalize(V3 a) {
    const double s = 1.0 / std::sqrt(dot(a, a));
    return a * s;
}
inline double comp(V3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

struct Box {  // axis-aligned while the tree is being built (R = I)
    V3 c, e;
};
// calc_obb(min, max) / minMaxToCenterExtent (src/obb/util.jl:10-14,47-51)
inline Box box_from_min_max(V3 lo, V3 hi) { return {(hi + lo) * 0.5, (hi - lo) * 0.5}; }
// OBB(a, b) (src/obb/box_types.jl:11-15) for axis-aligned boxes: calc_min_max(a) = c -/+ |I| e
inline Box box_union(const Box &a, const Box &b) {
    const V3 lo = vmin(vmin(a.c - a.e, a.c + a.e), vmin(b.c - b.e, b.c + b.e));
    const V3 hi = vmax(vmax(a.c - a.e, a.c + a.e), vmax(b.c - b.e, b.c + b.e));
    return box_from_min_max(lo, hi);
}
inline double box_area(const Box &b) { return 8 * dot(b.e, V3{b.e.y, b.e.z, b.e.x}); }  // box_types.jl:17
inline double box_volume(const Box &b) { return 8 * ((b.e.x * b.e.y) * b.e.z); }        // :18

struct Node {
    Box box;
    int child[2];
    int leaf;
};

struct Builder {
    std::vector<Node> node;
    int new_leaf(int id, const Box &b) {
        node.push_back({b, {-1, -1}, id});
        return (int)node.size() - 1;
    }
    int join(int a, int b) {  // bin_BB_Tree(node_1, node_2), src/obb/tree_types.jl:10-13
        node.push_back({box_union(node[a].box, node[b].box), {a, b}, kInternal});
        return (int)node.size() - 1;
    }
    // recursive_top_down (src/geometry/top_down.jl:10-32)
    int top_down(const std::vector<int> &t) {
        const size_t n = t.size();
        if (n == 1) return t[0];
        if (n == 2) return join(t[0], t[1]);
        Box all = node[t[0]].box;
        for (int k : t) all = box_union(all, node[k].box);
        int ax = 0;  // findmax: first maximum
        if (all.e.y > comp(all.e, ax)) ax = 1;
        if (all.e.z > comp(all.e, ax)) ax = 2;
        std::vector<int> perm(n);
        std::iota(perm.begin(), perm.end(), 0);
        std::stable_sort(perm.begin(), perm.end(),
                         [&](int i, int j) { return comp(node[t[i]].box.c, ax) < comp(node[t[j]].box.c, ax); });
        const size_t n_mid = (n + 1) / 2;
        std::vector<int> a, b;
        for (size_t i = 0; i + 1 < n_mid; ++i) a.push_back(t[perm[i]]);
        for (size_t i = n_mid - 1; i < n; ++i) b.push_back(t[perm[i]]);
        const int ta = top_down(a);
        const int tb = top_down(b);
        return join(ta, tb);
    }
};

// blobCost (src/geometry/blob_types.jl:74-82)
inline double blob_cost(const Box &b, long long n_below, double scale) {
    double v = 0.0;
    v += (double)n_below * std::log2((double)(2 * n_below));
    v += 1.0 * box_area(b) / (scale * scale);
    v += 1.0 * box_volume(b) / (scale * scale * scale);
    return v;
}

struct Blob {
    long long n_below = 0;
    double cost = 0;
    std::set<int> nb;
    int tree = -1;
    bool alive = false;
};

struct FaceKey {
    int v[3];
    bool operator==(const FaceKey &o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2]; }
};
struct FaceHash {
    size_t operator()(const FaceKey &k) const {
        size_t h = 1469598103934665603ull;
        for (int i = 0; i < 3; ++i) h = (h ^ (size_t)(unsigned)k.v[i]) * 1099511628211ull;
        return h;
    }
};

// make_obb (src/obb/obb_construction.jl:13-26); i_start 0-based
struct Obb {
    V3 c, e;
    double R[9];  // column-major
};
Obb make_obb(const V3 *p, int n, int i_start) {
    const V3 e1 = normalize(p[(i_start + 1) % 3] - p[i_start]);
    const V3 e3 = normalize(cross(p[1] - p[0], p[2] - p[1]) * 0.5);  // triangleNormal (geometry_kernel.jl:5,10)
    const V3 e2 = cross(e3, e1);
    V3 lo{INFINITY, INFINITY, INFINITY}, hi{-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < n; ++k) {
        const V3 pr{dot(p[k], e1), dot(p[k], e2), dot(p[k], e3)};
        lo = vmin(lo, pr);
        hi = vmax(hi, pr);
    }
    const Box b = box_from_min_max(lo, hi);
    Obb o;
    o.c = {(e1.x * b.c.x + e2.x * b.c.y) + e3.x * b.c.z, (e1.y * b.c.x + e2.y * b.c.y) + e3.y * b.c.z,
           (e1.z * b.c.x + e2.z * b.c.y) + e3.z * b.c.z};
    o.e = b.e;
    const double R[9] = {e1.x, e1.y, e1.z, e2.x, e2.y, e2.z, e3.x, e3.y, e3.z};
    std::memcpy(o.R, R, sizeof(R));
    return o;
}
inline double obb_area(const Obb &o) { return 8 * dot(o.e, V3{o.e.y, o.e.z, o.e.x}); }

// tet_perm_by_num / sort_so_big_eps_last (src/obb/obb_construction.jl:1-7, src/obb/util.jl:54-60), 0-based
const int kTetPerm[4][4] = {{1, 3, 2, 0}, {3, 0, 2, 1}, {0, 3, 1, 2}, {0, 1, 2, 3}};

int fail(int code, const char *msg) {
    g_tree_err = msg;
    return -code;
}

}  // namespace

extern "C" {

const char *pfc_tree_last_error(void) { return g_tree_err.c_str(); }

int pfc_build_tree(int n_pt, const double *pt, int n_elem, int arity, const int *elem, const double *eps, int method,
                   double *node_c, double *node_e, double *node_R, int *node_child, int *node_leaf) {
    if (n_pt <= 0 || !pt || n_elem <= 0 || !elem || (arity != 3 && arity != 4) || !node_c || !node_e || !node_R ||
        !node_child || !node_leaf || (method != PFC_TREE_BLOB && method != PFC_TREE_MEDIAN))
        return fail(PFC_ERR_BAD_ARG, "pfc_build_tree: bad argument");
    if (arity == 4 && !eps) return fail(PFC_ERR_BAD_ARG, "pfc_build_tree: a tet mesh needs eps (the leaf boxes use it)");
    for (long long i = 0; i < (long long)n_elem * arity; ++i)
        if (elem[i] < 0 || elem[i] >= n_pt) return fail(PFC_ERR_BAD_ARG, "pfc_build_tree: vertex index out of range");
    for (long long i = 0; i < 3ll * n_pt; ++i)
        if (!std::isfinite(pt[i])) return fail(PFC_ERR_NONFINITE, "pfc_build_tree: non-finite point");
    auto P = [&](int i) { return V3{pt[3 * (size_t)i], pt[3 * (size_t)i + 1], pt[3 * (size_t)i + 2]}; };

    Builder B;
    B.node.reserve(2 * (size_t)n_elem);
    std::vector<int> roots;
    // leaf AABBs: calc_obb(point[element]) (blob_types.jl:150-154), findMin/MaxSVSV pairing (src/obb/util.jl:2-8)
    for (int k = 0; k < n_elem; ++k) {
        const int *v = elem + (size_t)arity * k;
        V3 lo, hi;
        if (arity == 3) {
            lo = vmin(vmin(P(v[0]), P(v[1])), P(v[2]));
            hi = vmax(vmax(P(v[0]), P(v[1])), P(v[2]));
        } else {
            lo = vmin(vmin(P(v[0]), P(v[1])), vmin(P(v[2]), P(v[3])));
            hi = vmax(vmax(P(v[0]), P(v[1])), vmax(P(v[2]), P(v[3])));
        }
        roots.push_back(B.new_leaf(k, box_from_min_max(lo, hi)));
    }

    if (n_elem > 1 && method == PFC_TREE_BLOB) {
        // scale = sum(calc_obb(point).e) / 3 (:148)
        V3 lo{INFINITY, INFINITY, INFINITY}, hi{-INFINITY, -INFINITY, -INFINITY};
        for (int i = 0; i < n_pt; ++i) {
            lo = vmin(lo, P(i));
            hi = vmax(hi, P(i));
        }
        const Box all = box_from_min_max(lo, hi);
        const double scale = ((all.e.x + all.e.y) + all.e.z) / 3;
        if (!(scale > 0.0)) return fail(PFC_ERR_BAD_ARG, "pfc_build_tree: degenerate mesh (zero extent)");

        // createSharedEdgeFaceDict / extractTriTetNeighborInformation (:28-72)
        std::unordered_map<FaceKey, std::array<int, 2>, FaceHash> shared;
        shared.reserve((size_t)n_elem * arity);
        for (int k = 0; k < n_elem; ++k) {
            const int *v = elem + (size_t)arity * k;
            for (int j = 0; j < arity; ++j) {
                FaceKey f{{-1, -1, -1}};
                int m = 0;
                for (int q = 1; q < arity; ++q) f.v[m++] = v[(j + q) % arity];  // sortEdgeFace (geometry/util.jl:2-20)
                std::sort(f.v, f.v + (arity - 1));
                auto it = shared.find(f);
                if (it == shared.end()) {
                    shared.emplace(f, std::array<int, 2>{k, kInternal});
                } else {
                    if (it->second[1] != kInternal)
                        return fail(PFC_ERR_BAD_ARG, arity == 3
                                                         ? "three triangles share the same edge something is wrong"
                                                         : "three tetrahedrons share the same face something is wrong");
                    it->second[1] = k;
                }
            }
        }
        std::vector<Blob> blob((size_t)2 * n_elem);
        for (auto &kv : shared) {
            const int a = kv.second[0], b = kv.second[1];
            if (b == kInternal) {
                if (arity == 3)  // every edge needs a partner (:62-69) -> is_abort -> error (:156)
                    return fail(PFC_ERR_BAD_ARG, "not implemented error: disconnected mesh");
                continue;  // boundary face of a tet mesh
            }
            blob[a].nb.insert(b);
            blob[b].nb.insert(a);
        }
        auto self_cost = [&](const Box &b, long long n) { return blob_cost(box_union(b, b), n, scale); };  // :7-11
        for (int k = 0; k < n_elem; ++k) {
            blob[k].n_below = 1;
            blob[k].tree = roots[k];
            blob[k].cost = self_cost(B.node[roots[k]].box, 1);
            blob[k].alive = true;
        }
        auto marginal = [&](const Blob &a, const Blob &b) {  // calcMarginalCost (:95-98)
            return blob_cost(box_union(B.node[a.tree].box, B.node[b.tree].box), a.n_below + b.n_below, scale) - a.cost -
                   b.cost;
        };
        using Key = std::pair<int, int>;
        std::set<std::pair<double, Key>> pq;       // ordered by (delta cost, key)
        std::map<Key, double> pq_val;
        auto pq_set = [&](Key k, double v) {
            auto it = pq_val.find(k);
            if (it != pq_val.end()) {
                pq.erase({it->second, k});
                it->second = v;
            } else {
                pq_val.emplace(k, v);
            }
            pq.insert({v, k});
        };
        auto pq_del = [&](Key k) {
            auto it = pq_val.find(k);
            if (it == pq_val.end()) return;
            pq.erase({it->second, k});
            pq_val.erase(it);
        };
        auto mm = [](int a, int b) { return Key{std::min(a, b), std::max(a, b)}; };
        for (int a = 0; a < n_elem; ++a)  // createBlobPriorityQueue (:112-121)
            for (int b : blob[a].nb)
                if (a < b) pq_set({a, b}, marginal(blob[a], blob[b]));
        int k_next = n_elem;
        while (!pq.empty()) {  // bottomUp! (:123-134)
            const Key key = pq.begin()->second;
            pq_del(key);
            const int a = key.first, b = key.second;
            blob[a].nb.erase(b);  // doCombineBlob (:84-93)
            blob[b].nb.erase(a);
            const int c = k_next++;
            Blob &C = blob[c];
            C.nb = blob[a].nb;
            C.nb.insert(blob[b].nb.begin(), blob[b].nb.end());
            C.tree = B.join(blob[a].tree, blob[b].tree);
            C.n_below = blob[a].n_below + blob[b].n_below;
            C.cost = self_cost(B.node[C.tree].box, C.n_below);
            C.alive = true;
            for (int side = 0; side < 2; ++side) {  // refreshCostQueue! (:100-110)
                const int a_k = side == 0 ? a : b;
                for (int b_k : blob[a_k].nb) {
                    pq_del(mm(b_k, a_k));
                    pq_set(mm(b_k, c), marginal(blob[b_k], C));
                    blob[b_k].nb.erase(a_k);
                    blob[b_k].nb.insert(c);
                }
                blob[a_k].alive = false;
                blob[a_k].nb.clear();
            }
        }
        roots.clear();  // collect(values(dict_blob)) (:160-162), here in ascending blob key
        for (int k = 0; k < k_next; ++k)
            if (blob[k].alive) roots.push_back(blob[k].tree);
    }
    const int root = B.top_down(roots);

    // flatten in preorder (node 0 = root; parents precede children) and fit the leaves tight (:170,175-190)
    const size_t n_node = B.node.size();
    std::vector<int> order;
    order.reserve(n_node);
    std::vector<int> new_id(n_node, -1), stack{root};
    while (!stack.empty()) {
        const int k = stack.back();
        stack.pop_back();
        new_id[k] = (int)order.size();
        order.push_back(k);
        if (B.node[k].leaf == kInternal) {
            stack.push_back(B.node[k].child[1]);
            stack.push_back(B.node[k].child[0]);
        }
    }
    if (order.size() != 2 * (size_t)n_elem - 1) return fail(PFC_ERR_STATE, "pfc_build_tree: internal node count mismatch");
    for (size_t i = 0; i < order.size(); ++i) {
        const Node &nd = B.node[order[i]];
        double *c = node_c + 3 * i, *e = node_e + 3 * i, *R = node_R + 9 * i;
        c[0] = nd.box.c.x; c[1] = nd.box.c.y; c[2] = nd.box.c.z;
        e[0] = nd.box.e.x; e[1] = nd.box.e.y; e[2] = nd.box.e.z;
        for (int q = 0; q < 9; ++q) R[q] = (q % 4 == 0) ? 1.0 : 0.0;
        node_leaf[i] = nd.leaf;
        if (nd.leaf == kInternal) {
            node_child[2 * i] = new_id[nd.child[0]];
            node_child[2 * i + 1] = new_id[nd.child[1]];
            continue;
        }
        node_child[2 * i] = node_child[2 * i + 1] = -1;
        if (n_elem == 1) continue;  // a single-element mesh keeps its AABB (:139-146)
        const int *v = elem + (size_t)arity * nd.leaf;
        Obb o;
        if (arity == 3) {
            const V3 p[3] = {P(v[0]), P(v[1]), P(v[2])};
            o = make_obb(p, 3, 0);  // fit_tri_obb (obb_construction.jl:28)
        } else {
            V3 q[4] = {P(v[0]), P(v[1]), P(v[2]), P(v[3])};
            {  // volume (src/math_kernel/geometry_kernel.jl:25-38): the reference refuses inverted tets (:30)
                const V3 a = q[0], b = q[1], c4 = q[2], d = q[3];
                double vol = (b.x - a.x) * (c4.y * d.z - c4.z * d.y);
                vol = (b.y - a.y) * (c4.z * d.x - c4.x * d.z) + vol;
                vol = (b.z - a.z) * (c4.x * d.y - c4.y * d.x) + vol;
                vol = (c4.x - d.x) * (a.z * b.y - a.y * b.z) + vol;
                vol = (c4.y - d.y) * (a.x * b.z - a.z * b.x) + vol;
                vol = (c4.z - d.z) * (a.y * b.x - a.x * b.y) + vol;
                if (!(0.0 < vol * (1.0 / 6.0))) return fail(PFC_ERR_INVERTED_TET, "inverted tet");
            }
            int big = 0;  // findmax(abs.(eps)): first maximum
            for (int m = 1; m < 4; ++m)
                if (std::fabs(eps[v[m]]) > std::fabs(eps[v[big]])) big = m;
            const V3 p[4] = {q[kTetPerm[big][0]], q[kTetPerm[big][1]], q[kTetPerm[big][2]], q[kTetPerm[big][3]]};
            const Obb o1 = make_obb(p, 4, 0), o2 = make_obb(p, 4, 1), o3 = make_obb(p, 4, 2);
            const double a1 = obb_area(o1), a2 = obb_area(o2), a3 = obb_area(o3);
            if (std::max(a2, a3) <= a1) o = o1;          // keeps the LARGEST area (sic, :35-40)
            else if (std::max(a1, a3) <= a2) o = o2;
            else o = o3;
        }
        c[0] = o.c.x; c[1] = o.c.y; c[2] = o.c.z;
        e[0] = o.e.x; e[1] = o.e.y; e[2] = o.e.z;
        std::memcpy(R, o.R, sizeof(o.R));
    }
    return (int)order.size();
}

}  // extern "C"
