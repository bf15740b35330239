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
