// Hand-written stable LSD radix sort of (uint64 key, int32 value) pairs (csrc/radix.hip): internal entry points behind
// cdseg_sort_pairs / cdseg_sort_curves (csrc/serialize.hip).
#pragma once
#include "../../cdsegnet_amd/csrc/common.h"

size_t radix_sort_ws_bytes(size_t n);
// sorts by key bits [0, end_bit); vin == nullptr: the value of a pair is its input position.  kin / kout and vin / vout must
// not alias; ws >= radix_sort_ws_bytes(n).
int radix_sort_pairs(const uint64_t* kin, uint64_t* kout, const int32_t* vin, int32_t* vout, size_t n, int end_bit, void* ws,
                     size_t ws_bytes, hipStream_t s);
