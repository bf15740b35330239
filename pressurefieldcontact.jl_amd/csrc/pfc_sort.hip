// pfc_sort.hip -- canonical order of the candidate list (option "fixed_order").  A translation unit of its own: rocPRIM's radix
// sort is the only library code in the product and its headers are kept out of the kernels' compile.
//
// The broadphase appends candidate pairs in runs, in the order its workgroups finish (atomic reservation): the SET of candidates
// of an evaluation is fixed (tree_tree_intersect, src/obb/tree_types.jl:88-111), its ORDER is not.  Every sum downstream
// (the per-item wrench, the cop, the 27 patch-stiffness moments of calc_patch_spatial_stiffness!,
// src/contact_algorithms_friction.jl:147-169) is taken in list order, lane by lane and chunk by chunk, so the last bits of those
// sums differ between two identical evaluations -- harmless, except where decompose_K! (friction.jl:104-117) clamps an eigenvalue
// that is zero in exact arithmetic (a flat patch) at 1e-16 sigma_max (:92): there the rounding noise decides the branch.  With
// the list sorted by (item, element of mesh 1, element of mesh 2) the chunks, the lanes and with them every partial sum are
// the same in every run; pfc_np.h / pfc_br.h then add the per-(chunk, item) records of an item in chunk order.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "pfc_sort.h"

namespace {

struct alignas(16) Rec16 { int item, a, b, pad; };      // = pfc::WorkRec (pfc_kernels.h)

__global__ void __launch_bounds__(256) k_sort_pack(const Rec16 *cand, const int *ccount, size_t cap, unsigned long long *keys,
                                                   int n_items, int bits_a, int bits_b, unsigned *status, unsigned cover_bit) {
    size_t n_c = (size_t)(*ccount < 0 ? 0 : *ccount);
    if (n_c > cap) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && status) atomicOr(status, cover_bit);
        n_c = cap;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long k;
        if (i < n_c) {
            const Rec16 c = cand[i];
            k = ((unsigned long long)(unsigned)c.item << (bits_a + bits_b)) | ((unsigned long long)(unsigned)c.a << bits_b) |
                (unsigned long long)(unsigned)c.b;
        } else {
            k = (unsigned long long)(unsigned)n_items << (bits_a + bits_b);      // behind every entry of the list
        }
        keys[i] = k;
    }
}

__global__ void __launch_bounds__(256) k_sort_unpack(Rec16 *cand, const int *ccount, size_t cap, const unsigned long long *keys,
                                                     int bits_a, int bits_b) {
    size_t n_c = (size_t)(*ccount < 0 ? 0 : *ccount);
    if (n_c > cap) n_c = cap;
    const unsigned long long ma = (1ull << bits_a) - 1ull, mb = (1ull << bits_b) - 1ull;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_c; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long k = keys[i];
        Rec16 c;
        c.item = (int)(k >> (bits_a + bits_b)); c.a = (int)((k >> bits_b) & ma); c.b = (int)(k & mb); c.pad = 0;
        cand[i] = c;
    }
}

__global__ void __launch_bounds__(256) k_sort_pack32(const int *list, const int *count, size_t cap, unsigned *keys) {
    size_t n_c = (size_t)(*count < 0 ? 0 : *count);
    if (n_c > cap) n_c = cap;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (size_t)gridDim.x * blockDim.x)
        keys[i] = i < n_c ? (unsigned)list[i] : 0xFFFFFFFFu;
}
__global__ void __launch_bounds__(256) k_sort_unpack32(int *list, const int *count, size_t cap, const unsigned *keys) {
    size_t n_c = (size_t)(*count < 0 ? 0 : *count);
    if (n_c > cap) n_c = cap;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_c; i += (size_t)gridDim.x * blockDim.x) list[i] = (int)keys[i];
}

// ---- the same order without a sort of the whole list: the broadphase has counted every item's candidates (icnt), so the list can be
// cut into one segment per item -- offsets by a scan, every candidate dropped into its item's segment -- and only the segments need
// sorting, by (a, b): one workgroup per item, a bitonic network in LDS (segments of up to kCanonSeg keys; a longer one is reported
// and the handle goes back to the sort of the whole list).  Four launches whatever the list's length, nothing read back by the host --
// rocPRIM's segmented sort copies its segment-size classes to the host and sizes its launches from them, which a captured graph
// would replay with the sizes of the day it was captured.
// (Offsets are CLAMPED to the covered part of the list: the counters hold what the broadphase FOUND, which for a list that overflowed
// -- the first evaluation of a handle -- is far more than the list holds; the sort must not be led beyond the buffers.)
__global__ void __launch_bounds__(1024) k_canon_offsets(const int *icnt, int n_items, int *off, int *fill, int cap) {
    __shared__ long long s_part[16], s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n_items; i0 += 1024) {
        const int i = i0 + tid;
        const long long c = i < n_items ? (long long)icnt[4 * (size_t)i + 1] : 0ll;
        long long x = c;      // inclusive scan over the wave (shuffles), then over the 16 waves
        for (int d = 1; d < 64; d <<= 1) { const long long y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        if (lane == 63) s_part[wave] = x;
        __syncthreads();
        long long before = s_base;
        for (int w = 0; w < wave; ++w) before += s_part[w];
        if (i < n_items) { const long long o = before + x - c; off[i] = o < cap ? (int)o : cap; fill[i] = 0; }
        __syncthreads();
        if (tid == 1023) s_base = before + x;
        __syncthreads();
    }
    if (tid == 0) off[n_items] = s_base < cap ? (int)s_base : cap;
}
__global__ void __launch_bounds__(256) k_canon_scatter(const Rec16 *cand, const int *ccount, size_t cap, const int *off, int *fill,
                                                        int n_items, unsigned long long *keys, int *item_of, int bits_b,
                                                        unsigned *status, unsigned cover_bit) {
    // A list longer than the covered part (or than its own capacity: an overflowing evaluation) is left as it is: the per-item counts
    // say what the broadphase FOUND, the list holds what fitted, and segments cut for the one would be filled from the other -- slots
    // nobody writes, i.e. whatever the buffers held, as candidates.  The evaluation is re-issued either way.
    const size_t n_c = (size_t)(*ccount < 0 ? 0 : *ccount);
    if (n_c > cap) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && status) atomicOr(status, cover_bit);
        return;
    }
    // One returning atomic per (wave, item): the list is made of runs of one item's candidates, so the 64 entries of a wave belong to
    // one item, two at a run boundary -- a counter per candidate would put a thousand atomics on the address of every big item
    // (2 048 full-size C3 poses: 1.1 ms of this kernel).
    const int lane = threadIdx.x & 63;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j0 = (size_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); j0 < n_c; j0 += stride) {      // (uniform over the wave)
        const size_t j = j0 + lane;
        Rec16 c;
        c.item = -1; c.a = 0; c.b = 0; c.pad = 0;
        if (j < n_c) c = cand[j];
        const bool ok = (unsigned)c.item < (unsigned)n_items;
        int pos = -1;
        unsigned long long todo = __ballot(ok);
        while (todo) {
            const int f = __builtin_ctzll(todo);
            const int it = __builtin_amdgcn_readlane(c.item, f);
            const unsigned long long m = __ballot(ok && c.item == it);
            int base = 0;
            if (lane == f) base = atomicAdd(&fill[it], __popcll(m));
            base = __builtin_amdgcn_readlane(base, f);
            if (ok && c.item == it) pos = off[it] + base + __popcll(m & ((1ull << lane) - 1ull));
            todo &= ~m;
        }
        if (ok && pos >= 0 && pos < off[c.item + 1] && (size_t)pos < cap) {
            keys[pos] = ((unsigned long long)(unsigned)c.a << bits_b) | (unsigned long long)(unsigned)c.b;
            item_of[pos] = c.item;
        }
    }
}
__global__ void __launch_bounds__(256) k_canon_unpack(Rec16 *cand, const int *ccount, size_t cap, const unsigned long long *keys,
                                                       const int *item_of, int bits_b) {
    const size_t n_c = (size_t)(*ccount < 0 ? 0 : *ccount);
    if (n_c > cap) return;      // (see k_canon_scatter)
    const unsigned long long mb = (1ull << bits_b) - 1ull;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_c; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long k = keys[i];
        Rec16 c;
        c.item = item_of[i]; c.a = (int)(k >> bits_b); c.b = (int)(k & mb); c.pad = 0;
        cand[i] = c;
    }
}

constexpr int kCanonSeg = 4096;
__global__ void __launch_bounds__(256) k_canon_segsort(const int *ccount, size_t cap, const int *off, int n_items,
                                                        const unsigned long long *keys_in, unsigned long long *keys_out, unsigned *status,
                                                        unsigned big_bit) {
    __shared__ unsigned long long s_k[kCanonSeg];
    const int tid = threadIdx.x;
    if ((size_t)(*ccount < 0 ? 0 : *ccount) > cap) return;      // (see k_canon_scatter)
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {      // (uniform over the workgroup)
        const int o0 = off[item], n = off[item + 1] - o0;
        if (n <= 0) continue;
        if (n > kCanonSeg) {      // reported; its keys pass through unsorted (the passes behind must see candidates, not whatever keys_out held)
            if (tid == 0) atomicOr(status, big_bit);
            for (int k = tid; k < n; k += 256) keys_out[o0 + k] = keys_in[o0 + k];
            continue;
        }
        if (n == 1) { if (tid == 0) keys_out[o0] = keys_in[o0]; continue; }
        int N = 2;
        while (N < n) N <<= 1;
        for (int k = tid; k < N; k += 256) s_k[k] = k < n ? keys_in[o0 + k] : ~0ull;
        __syncthreads();
        for (int k = 2; k <= N; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (N >> 1); t += 256) {
                    const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;      // the t-th pair at distance j
                    const bool up = (lo & k) == 0;
                    const unsigned long long a = s_k[lo], b = s_k[hi];
                    if ((a > b) == up) { s_k[lo] = b; s_k[hi] = a; }
                }
                __syncthreads();
            }
        }
        for (int k = tid; k < n; k += 256) keys_out[o0 + k] = s_k[k];
        __syncthreads();
    }
}

int bits_for(unsigned long long n_values) {      // bits that hold 0 .. n_values - 1
    int b = 1;
    while (b < 63 && (1ull << b) < n_values) ++b;
    return b;
}

}  // namespace

int pfc_sort_key_bits(int n_items, int max_elem_1, int max_elem_2, int *bits_a, int *bits_b) {
    const int bi = bits_for((unsigned long long)n_items + 1ull);      // (+ 1: the key of the unused tail)
    *bits_a = bits_for((unsigned long long)(max_elem_1 > 0 ? max_elem_1 : 1));
    *bits_b = bits_for((unsigned long long)(max_elem_2 > 0 ? max_elem_2 : 1));
    return bi + *bits_a + *bits_b;
}

hipError_t pfc_sort_temp_bytes(size_t cap, int bits, size_t *bytes) {
    size_t b64 = 0, b32 = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, b64, (unsigned long long *)nullptr, (unsigned long long *)nullptr, cap, 0u,
                                            (unsigned)bits, (hipStream_t) nullptr);
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_keys(nullptr, b32, (unsigned *)nullptr, (unsigned *)nullptr, cap, 0u, 32u, (hipStream_t) nullptr);
    *bytes = b64 > b32 ? b64 : b32;
    return e;
}

hipError_t pfc_sort_indices(int *list, const int *count, size_t cap, unsigned *keys_in, unsigned *keys_out, void *temp, size_t temp_bytes,
                            hipStream_t st) {
    if (cap == 0) return hipSuccess;
    size_t g = (cap + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    hipLaunchKernelGGL(k_sort_pack32, dim3((unsigned)g), dim3(256), 0, st, (const int *)list, count, cap, keys_in);
    hipError_t e = rocprim::radix_sort_keys(temp, temp_bytes, keys_in, keys_out, cap, 0u, 32u, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_sort_unpack32, dim3((unsigned)g), dim3(256), 0, st, list, count, cap, (const unsigned *)keys_out);
    return hipGetLastError();
}

hipError_t pfc_sort_candidates(void *cand, const int *ccount, size_t cap, unsigned long long *keys_in, unsigned long long *keys_out,
                               void *temp, size_t temp_bytes, int n_items, int bits_a, int bits_b, int bits, unsigned *status,
                               unsigned cover_bit, hipStream_t st) {
    if (cap == 0) return hipSuccess;
    size_t g = (cap + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    hipLaunchKernelGGL(k_sort_pack, dim3((unsigned)g), dim3(256), 0, st, (const Rec16 *)cand, ccount, cap, keys_in, n_items, bits_a,
                       bits_b, status, cover_bit);
    hipError_t e = rocprim::radix_sort_keys(temp, temp_bytes, keys_in, keys_out, cap, 0u, (unsigned)bits, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_sort_unpack, dim3((unsigned)g), dim3(256), 0, st, (Rec16 *)cand, ccount, cap, keys_out, bits_a, bits_b);
    return hipGetLastError();
}

hipError_t pfc_canon_candidates(void *cand, const int *ccount, size_t cap, const int *icnt, int n_items, unsigned long long *keys_in,
                                unsigned long long *keys_out, int *off, int *fill, int *item_of, int bits_b, unsigned *status,
                                unsigned cover_bit, unsigned big_bit, hipStream_t st) {
    if (cap == 0 || n_items <= 0) return hipSuccess;
    size_t g = (cap + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    hipLaunchKernelGGL(k_canon_offsets, dim3(1), dim3(1024), 0, st, icnt, n_items, off, fill, (int)(cap > 0x7FFFFFFF ? 0x7FFFFFFF : cap));
    hipLaunchKernelGGL(k_canon_scatter, dim3((unsigned)g), dim3(256), 0, st, (const Rec16 *)cand, ccount, cap, (const int *)off, fill,
                       n_items, keys_in, item_of, bits_b, status, cover_bit);
    hipLaunchKernelGGL(k_canon_segsort, dim3((unsigned)(n_items < 256 * 8 ? n_items : 256 * 8)), dim3(256), 0, st, ccount, cap, (const int *)off, n_items,
                       (const unsigned long long *)keys_in, keys_out, status, big_bit);
    hipLaunchKernelGGL(k_canon_unpack, dim3((unsigned)g), dim3(256), 0, st, (Rec16 *)cand, ccount, cap, (const unsigned long long *)keys_out,
                       (const int *)item_of, bits_b);
    return hipGetLastError();
}
