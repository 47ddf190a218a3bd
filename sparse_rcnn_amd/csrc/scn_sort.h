// Internal: stable LSD radix sort of (uint32 key, int32 value) pairs (scn_sort.hip).
#pragma once
#include "scn_common.h"

namespace scn {
// bytes of scratch sort_pairs needs for n pairs
int64_t sort_pairs_scratch_bytes(int64_t n);
// keys_out / vals_out = the pairs in ascending order of the low `bits` key bits, equal keys in input order.
// vals == nullptr: the values are the input positions 0..n-1.  Inputs are left untouched; outputs must not alias them.
int sort_pairs(const uint32_t* keys, const int32_t* vals, int64_t n, int bits, uint32_t* keys_out, int32_t* vals_out,
               void* scratch, hipStream_t st);
}  // namespace scn
