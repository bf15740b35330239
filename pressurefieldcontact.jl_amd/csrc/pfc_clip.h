// pfc_clip.h -- the library's ONE statement of clip_in_tet_coordinates (src/clip/static_clip.jl:7-23: Sutherland-Hodgman
// against the half-spaces zeta_i >= 0, i = 1..4, `clip` :34-128, `cut_clip` :135-195, `clip_node` :197-201, weightPoly
// src/math_kernel/utility.jl:21-26), shared by every kernel that clips: the batched narrowphase (k_narrow, k_clip_queue:
// pfc_np.h), the small-scene kernel (k_fused, value and Dual rounds: pfc_fused.h) and the Dual narrowphase (k_narrow_dual:
// pfc_dual.h).  Rounds 1-2 kept three hand-maintained copies.  Included by pfc_hip.hip inside namespace pfc (device code).
//
// The polygon lives in an 8-slot ring of 4-vectors (tet coordinates) that the caller owns -- LDS, one column per owner,
// [slot][coord][column] layout so that per-lane dynamic indexing is conflict-free -- and is clipped IN PLACE: the
// reference's arity-unrolled recursion (clip(z1..zN) -> cut_clip -> clip(z1'..zM')) becomes one loop over the four planes
// with the sign pattern of a plane held in two bit masks.  A Ring type supplies
//     typedef scalar                     double, or a (value, partial) pair (pfc_dual.h: Du)
//     double val(k, c)                   the VALUE of coordinate c of logical vertex k (every predicate compares values,
//                                        as ForwardDiff's comparisons do)
//     scalar get(k, c); set(k, c, x)     the coordinate itself
//     move(src, dst, c)                  slot-to-slot copy of one coordinate
//     rotate(st)                         logical vertex st becomes logical vertex 0
// Logical indices are taken modulo 8 by the ring.  Per plane i (:34-128): s_k = z_k[i]; all s <= 0 -> empty (:44); all
// 0 <= s -> next plane (:45-46); else rotate to the first k with s_k <= 0 < s_{k+1} (:48-50; none: "Non-finite vertex
// likely", :52), drop trailing vertices while z_{m-1} is non-positive (cut_clip :135-195), and replace the cut corner by
// z_start = clip_node(z1, z2) and z_end = clip_node(z1, z_m) or clip_node(z_m, z_{m-1}) depending on the inside test of the
// last vertex -- STRICT 0 < z for arities 3..5 (:140,150,162), NON-strict 0 <= z for 6..7 (:176,188); the 7-vertex method
// returns its polygon directly (:185-195).
#pragma once

// a column of doubles in an LDS ring with a compile-time column count
template <int STRIDE>
struct RingCol {
    typedef double scalar;
    double *base;
    int col, rbase;
    __device__ __forceinline__ double &at(int k, int c) const { return base[((((rbase + k) & 7) * 4 + c) * STRIDE) + col]; }
    __device__ __forceinline__ double val(int k, int c) const { return at(k, c); }
    __device__ __forceinline__ double get(int k, int c) const { return at(k, c); }
    __device__ __forceinline__ void set(int k, int c, double x) const { at(k, c) = x; }
    __device__ __forceinline__ void move(int src, int dst, int c) const { const double t = at(src, c); at(dst, c) = t; }
    __device__ __forceinline__ void rotate(int st) { rbase = (rbase + st) & 7; }
};

// Clips the n_in-gon (3 or 4 vertices, already in the ring at logical 0..n_in-1) against the four planes; returns the
// vertex count of the result (0, 3..8) with the polygon at logical 0..n-1.  err: "Non-finite vertex likely" (:52).
template <class Ring>
__device__ __forceinline__ int clip_ring_in_tet_coordinates(Ring &R, int n_in, bool &err) {
    typedef typename Ring::scalar T;
    int n = n_in;
    for (int i = 0; i < 4 && n > 0; ++i) {
        unsigned nonpos = 0, nonneg = 0;
        for (int k = 0; k < n; ++k) {
            const double sv = R.val(k, i);
            nonpos |= (unsigned)(sv <= 0.0) << k;
            nonneg |= (unsigned)(0.0 <= sv) << k;
        }
        const unsigned full = (1u << n) - 1u;
        if (nonpos == full) { n = 0; break; }       // :44
        if (nonneg == full) continue;               // :45-46
        // first k with is_non_pos[k] && !is_non_pos[k+1] (cyclic) (:48-50)
        const unsigned nxt = ((nonpos >> 1) | ((nonpos & 1u) << (n - 1))) & full;
        const unsigned cand_start = nonpos & ~nxt & full;
        if (cand_start == 0) { err = true; n = 0; break; }  // "Non-finite vertex likely" (:52)
        const int st = __builtin_ctz(cand_start);
        // cut_clip (:135-195): drop trailing vertices while z_{m-1} is non-positive
        int m = n;
        while (m > 3) {
            int k2 = st + m - 2; if (k2 >= n) k2 -= n;
            if ((nonpos >> k2) & 1u) --m; else break;
        }
        int k1 = st + 1; if (k1 >= n) k1 -= n;
        int kl = st + m - 1; if (kl >= n) kl -= n;   // z_m (last)
        int kp = st + m - 2; if (kp >= n) kp -= n;   // z_{m-1}
        // inside test of the last vertex: 0 < z for arity 3..5 (:140,150,162), 0 <= z for 6..7 (:176,188)
        const bool inside = (m <= 5) ? (((nonpos >> kl) & 1u) == 0) : (((nonneg >> kl) & 1u) != 0);
        // z_start = clip_node(z1, z2); z_end = clip_node(z1, z_m) or clip_node(z_m, z_{m-1}); both are formed in
        // registers before the ring is touched
        T zs[4], ze[4];
        {
            const T w1 = R.get(st, i), w2 = R.get(k1, i);
            const T sw = w1 - w2, c1 = w1 / sw, c2 = w2 / sw;
#pragma unroll
            for (int c = 0; c < 4; ++c) zs[c] = c1 * R.get(k1, c) - c2 * R.get(st, c);
        }
        {
            const int kn = inside ? st : kl, kq = inside ? kl : kp;
            const T w1 = R.get(kn, i), w2 = R.get(kq, i);
            const T sw = w1 - w2, c1 = w1 / sw, c2 = w2 / sw;
#pragma unroll
            for (int c = 0; c < 4; ++c) ze[c] = c1 * R.get(kq, c) - c2 * R.get(kn, c);
        }
        const int ncopy = inside ? (m - 1) : (m - 2);   // z2 .. z_m  or  z2 .. z_{m-1} stay in the polygon
        // In place: the new polygon starts at old logical st.  Kept vertices st+1 .. n-1 do not move; kept vertices that
        // wrapped around (old logical 0 .. ) move up by n slots, in increasing order (a destination is either a free
        // slot or the source of an earlier move).
        for (int q = n - st - 1; q < ncopy; ++q) {
            const int src = st + 1 + q - n, dst = st + 1 + q;
#pragma unroll
            for (int c = 0; c < 4; ++c) R.move(src, dst, c);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { R.set(st, c, zs[c]); R.set(st + ncopy + 1, c, ze[c]); }
        R.rotate(st);
        n = ncopy + 2;
        if (m == 7) break;  // the 7-vertex method returns the polygon directly (:185-195)
    }
    return n;
}
