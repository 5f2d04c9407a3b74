// Space-filling-curve keys, usable from device code and from the host-side checker
// (tests/host_check/curves_check.cpp builds this header with g++ and compares it with the
// numpy oracle, so the bit arithmetic is validated without a GPU).
//   z-order : pointcept/models/utils/serialization/z_order.py:40-50 (bit interleave x->3i+2, y->3i+1, z->3i)
//   hilbert : pointcept/models/utils/serialization/hilbert.py:143-198 (Skilling transform, Gray->binary)
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define CDSEG_HD __host__ __device__ __forceinline__
#else
#define CDSEG_HD static inline
#endif
#ifndef CDSEG_ORDER_Z
#define CDSEG_ORDER_Z 0
#define CDSEG_ORDER_Z_TRANS 1
#define CDSEG_ORDER_HILBERT 2
#define CDSEG_ORDER_HILBERT_TRANS 3
#endif

// spread the low 16 bits of v so that bit i lands on bit 3i
CDSEG_HD uint64_t part1by2(uint64_t v) {
  v &= 0xffffull;
  v = (v | (v << 32)) & 0x001f00000000ffffull;
  v = (v | (v << 16)) & 0x001f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

CDSEG_HD uint64_t z_key(uint32_t x, uint32_t y, uint32_t z, int depth) {
  const uint32_t m = (depth >= 32) ? 0xffffffffu : ((1u << depth) - 1u);
  return (part1by2(x & m) << 2) | (part1by2(y & m) << 1) | part1by2(z & m);
}

// Skilling's axes->transpose, then Gray->binary of the interleaved string (prefix XOR)
CDSEG_HD uint64_t hilbert_key(uint32_t x, uint32_t y, uint32_t z, int depth) {
  const uint32_t m = (1u << depth) - 1u;
  uint32_t X0 = x & m, X1 = y & m, X2 = z & m;
  for (int bit = 0; bit < depth; ++bit) {
    const uint32_t q = 1u << (depth - 1 - bit);
    const uint32_t low = q - 1u;
    // dim 0
    if (X0 & q) X0 ^= low;
    // dim 1
    if (X1 & q) {
      X0 ^= low;
    } else {
      const uint32_t t = (X0 ^ X1) & low;
      X1 ^= t;
      X0 ^= t;
    }
    // dim 2
    if (X2 & q) {
      X0 ^= low;
    } else {
      const uint32_t t = (X0 ^ X2) & low;
      X2 ^= t;
      X0 ^= t;
    }
  }
  uint64_t h = (part1by2(X0) << 2) | (part1by2(X1) << 1) | part1by2(X2);
  for (int s = 1; s < 3 * depth; s <<= 1) h ^= h >> s;
  return h;
}

CDSEG_HD uint64_t curve_key(int order_id, uint32_t x, uint32_t y, uint32_t z, int depth) {
  switch (order_id) {
    case CDSEG_ORDER_Z: return z_key(x, y, z, depth);
    case CDSEG_ORDER_Z_TRANS: return z_key(y, x, z, depth);
    case CDSEG_ORDER_HILBERT: return hilbert_key(x, y, z, depth);
    default: return hilbert_key(y, x, z, depth);
  }
}

