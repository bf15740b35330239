// pfc_sort.h -- interface of pfc_sort.hip (canonical order of the candidate list, option "fixed_order").
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

// Key layout item | element of mesh 1 | element of mesh 2; returns the number of key bits (the caller refuses more than 64).
int pfc_sort_key_bits(int n_items, int max_elem_1, int max_elem_2, int *bits_a, int *bits_b);
hipError_t pfc_sort_temp_bytes(size_t cap, int bits, size_t *bytes);
// Sorts cand[0 .. min(*ccount, cap)) -- 16-byte records (item, a, b, 0) -- by key, on stream st; keys_in / keys_out: cap words each.
// cap may be less than the list's capacity (the slots the caller expects to be in use): *ccount > cap sets cover_bit in *status.
hipError_t pfc_sort_candidates(void *cand, const int *ccount, size_t cap, unsigned long long *keys_in, unsigned long long *keys_out,
                               void *temp, size_t temp_bytes, int n_items, int bits_a, int bits_b, int bits, unsigned *status,
                               unsigned cover_bit, hipStream_t st);
// Sorts the non-negative indices list[0 .. min(*count, cap)) ascending (negative entries go last), on stream st; keys_in / keys_out:
// cap 32-bit words each.  The temporary storage of pfc_sort_temp_bytes for the same cap is large enough.
hipError_t pfc_sort_indices(int *list, const int *count, size_t cap, unsigned *keys_in, unsigned *keys_out, void *temp, size_t temp_bytes,
                            hipStream_t st);
// The same order by segments: offsets from the per-item candidate counts icnt[4 i + 1] (off: n_items + 1 ints), every candidate
// dropped into its item's segment (fill: n_items ints, item_of: cap ints), the segments sorted by (a, b) in LDS.  An item with more
// than 4 096 candidates sets big_bit in *status (nothing is sorted then: the caller falls back to pfc_sort_candidates).
hipError_t pfc_canon_candidates(void *cand, const int *ccount, size_t cap, const int *icnt, int n_items, unsigned long long *keys_in,
                                unsigned long long *keys_out, int *off, int *fill, int *item_of, int bits_b, unsigned *status,
                                unsigned cover_bit, unsigned big_bit, hipStream_t st);
