// Submanifold 3x3x3 sparse convolution for the wide, bandwidth-bound stages (C = 32 / 64, bf16) on gfx950.
//
//   y[i, :] = bias + sum_o W_o x[nbr[o, i], :]          (spconv.SubMConv3d; ref call sites ptv3.py:356-362 - the CPE
//                                                         conv of every Block; nbr = the offset-major kernel map)
//
// The tiled gathered-A GEMM (gemm.hip) stages the gathered rows AND the weight columns of every 128-row tile through
// LDS: per launch ~6x the compulsory bytes move L2 -> LDS, and a tile is a chain of barrier-separated
// index -> row -> LDS round trips (PMC, profiles/r01l: VALU 11 %, matrix pipe 2 %, 60 % of the wave cycles idle).
// This kernel turns the roles around:
//   * W is STATIONARY: the whole 27 x C x C kernel lives in LDS for the lifetime of a persistent block (C = 32:
//     55 KB, 2 blocks / CU; C = 64: the 19 face / edge / centre offsets and one corner offset = 160 KB, the other 7
//     corner offsets are read from L2), laid out in MFMA A-fragment order (conflict-free b128 reads);
//   * gathered rows go STRAIGHT into MFMA B-fragment registers: lane (j, c) of v_mfma_f32_16x16x32_bf16 holds
//     channels 8c..8c+7 of point j, which is one 16-byte BUFFER load from row nbr[o, j] - no LDS round trip, no
//     barrier; a missing neighbour (index -1) becomes an out-of-range buffer offset: zeros, no memory access;
//   * waves are independent (no __syncthreads in the main loop);
//   * the product is computed transposed (D^T = W X^T) with the output channels permuted inside the A operand so
//     that a lane ends up holding C/4 CONSECUTIVE channels of one point: the epilogue is bias + one (two) 16-byte
//     store(s) per lane, 16 complete rows = 1 KB contiguous per store instruction - no LDS transpose;
//   * LIVE LIST: per wave and 32-row tile the 27 x 32 block of the kernel map is loaded ONCE, coalesced and without
//     duplicates (14 loads; the first version issued 54 index loads and 27 x 2 KS row loads per tile, one dependent
//     index -> row round trip per offset), a ballot per register gives the offsets any of the 32 rows has, and only
//     those are visited (z-ordered surface points: 14-16 of 27 per 32 rows on the benchmark scenes), the lane's row index coming from registers.
// HBM traffic = features once + kernel map once + output once (PMC: profiles/r02_pmc_conv*.txt); W never leaves the
// CU after the first tile.  History and measurements of the variants: DESIGN.md 4.2.
#include <cstdlib>
#include <type_traits>

#include <atomic>

#include "common.h"
#include "prof.h"

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4_t;

struct ConvP {
  const bf16_t* x;      // (n, ldx) bf16
  const float* bias;    // (C) or nullptr
  const int32_t* nbr;   // (27, n) offset-major kernel map, -1 = no neighbour
  bf16_t* y;            // (n, ldy) bf16
  long n;
  int ldx, ldy;
  int row_shift;        // log2(ldx * 2): byte stride of a feature row (a power of two)
  int tiles;            // ceil(n / rows per block)
  // fp32 output form (cdseg_subm_conv3_f32, the operand-pair convs of precision "fp32x3"): yf (n, ldyf) fp32,
  // yf = (accumulate ? yf : 0) + (acc + bias) * out_scale; y is then unused
  float* yf;
  int ldyf, accumulate;
  float out_scale;
};

// offsets in LDS-residency priority order (C = 64 keeps the first CONV64_LDS_OFFSETS = 20 in LDS): centre, 6 faces, 12 edges, 8 corners
__device__ __constant__ int8_t c_slot_of_offset[27] = {
    // o = a*9 + b*3 + c, (a,b,c) in {0,1,2}^3; #ones = number of coordinates equal to 1 (centre = 3, face = 2, edge = 1)
    19, 7, 20, 8, 1, 9, 21, 10, 22,   // a = 0
    11, 2, 12, 3, 0, 4, 13, 5, 14,    // a = 1
    23, 15, 24, 16, 6, 17, 25, 18, 26 // a = 2
};

// the same table packed 5 bits per entry (offsets 0-11, 12-23, 24-26) for scalar-register lookups
struct SlotBits {
  static constexpr unsigned long long t0 = 0x6097655521450f3ull, t1 = 0x89a187ddc569003ull, t2 = 0x6a59ull;
};

// C = 64: how many of the 27 offsets' weights (8 KB each) stay in LDS; 19 = centre + faces + edges (152 KB), 20 = the whole
// 160 KB of a CU: one corner offset fewer to fetch per tile (a third of the kernel's vector-memory instructions are corner
// weights) - 125.3 -> 123.5 us at 446 k rows (profiles/r05_conv64_lds20.txt)
#ifndef CONV64_LDS_OFFSETS
#define CONV64_LDS_OFFSETS 20
#endif

// Lane -> address map of the gathered-row loads.  0: the MFMA operand layout itself (lane (j, c) reads chunk c of row j: the
// four lanes of a quad read 16 bytes of four different rows).  1: quad-contiguous (lane 4 j + c reads chunk c of row j: a
// quad = 64 contiguous bytes of one row) and a ds_bpermute per dword moves the data to the operand layout before the MFMAs
// (tools/ubench/gather_map.hip times the two maps on bare loads).  A mask: bit 0 = the C = 32 kernel, bit 1 = the C = 64 kernel.
#ifndef CONV_QUAD_GATHER
#define CONV_QUAD_GATHER 0
#endif
#ifndef CONV_CORNER_FIRST
#define CONV_CORNER_FIRST 0
#endif

// row buffers in rotation per wave (loads of NB - 1 list entries in flight behind the MFMAs of one)
#ifndef CONV32_ROW_BUFFERS
#define CONV32_ROW_BUFFERS 3
#endif
#ifndef CONV64_ROW_BUFFERS
#define CONV64_ROW_BUFFERS 4
#endif

template <int C>
struct ConvCfg {
  static constexpr int CT = C / 16;       // 16-channel output tiles
  static constexpr int KS = C / 32;       // MFMA k steps per offset
  static constexpr int KC = C / 8;        // 16-byte channel chunks per row
  static constexpr int OFF_BYTES = C * C * 2;                // one offset's weights
  static constexpr int LDS_OFFSETS = C <= 32 ? 27 : CONV64_LDS_OFFSETS;  // offsets resident in LDS
  static constexpr int LDS_BYTES = LDS_OFFSETS * OFF_BYTES;  // 55,296 / 163,840 (= the CU's whole LDS)
  static constexpr int WAVES = 8;                            // pack kernel block size
  static constexpr int RG = 2;                               // 16-row groups per wave
  static constexpr int ROWS_PER_WAVE = RG * 16;
  static constexpr bool QUAD = ((CONV_QUAD_GATHER) >> (C == 64 ? 1 : 0)) & 1;  // lane map of the row loads
};

// Fragment-order weight image.  16-byte unit u = ((slot * CT + ct) * KC + kc) * 16 + i holds
// W[channel(ct, i)][o * C + kc * 8 .. + 8] with channel(ct, i) = (i >> 2) * (C / 4) + ct * 4 + (i & 3):
// lane l of the MFMA reads unit (.., kc = 4 * ks + (l >> 4), i = l & 15) -> a wave reads 1 KB contiguous.
template <int C>
__device__ __forceinline__ int frag_channel(int ct, int i) {
  return (i >> 2) * (C / 4) + ct * 4 + (i & 3);
}

template <int C>
__global__ __launch_bounds__(ConvCfg<C>::WAVES * 64) void conv_pack_w_kernel(const bf16_t* w, uint4* img) {
  // one-off repack (per weight tensor, cached by the caller): global image in the same fragment order, slot-major, so
  // that the resident part is one contiguous copy and a corner offset is a 1 KB-per-wave contiguous read from L2
  using K = ConvCfg<C>;
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= 27 * K::CT * K::KC * 16) return;
  const int i = u & 15, kc = (u >> 4) % K::KC, ct = (u >> 4) / K::KC % K::CT, slot = (u >> 4) / (K::KC * K::CT);
  int o = slot;  // C = 32: everything is resident, slots in offset order
  if (K::LDS_OFFSETS != 27)
    for (int q = 0; q < 27; ++q)
      if (c_slot_of_offset[q] == slot) o = q;
  img[u] = *reinterpret_cast<const uint4*>(w + (long)frag_channel<C>(ct, i) * (27 * C) + o * C + kc * 8);
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

typedef int i32x16_t __attribute__((ext_vector_type(16)));

// Per wave and 32-row tile:
//   phase 1  the 27 x 32 block of the kernel map is loaded ONCE, coalesced and without duplicates (lane l holds
//            offsets 2t + (l >> 5) of row l & 31 in register t: 14 loads instead of 54), and a ballot per register
//            gives the set of offsets any of the 32 rows has;
//   phase 2  only those offsets are visited (z-ordered surface points: 14-16 of 27 per 32 rows on the benchmark scenes): a lane fetches its row index
//            from the register block (ds_bpermute, register picked by the wave-uniform offset), issues the row
//            loads - no index -> row memory round trip - and the MFMAs of the offset two places back in the list
//            run meanwhile (NB row buffers in rotation).  The list is padded with offset 27 (index -1, loads
//            return zeros without touching memory) so that every step issues the same number of loads.
// Texture-address work per tile drops from 27 * (2 + 2 KS) to 14 + live * 2 KS wave instructions.
template <int C, int WAVES, int NB, int OCC>
__global__ __launch_bounds__(WAVES * 64, OCC) void conv_ll_kernel(ConvP p, const uint4* __restrict__ wimg) {
  using K = ConvCfg<C>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int jrow = lane & 15;
  const int cgrp = lane >> 4;
  {
    uint4* dst = reinterpret_cast<uint4*>(smem);
    for (int u = tid; u < K::LDS_BYTES / 16; u += WAVES * 64) dst[u] = wimg[u];
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t x_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(p.n * p.ldx * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t nbr_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.nbr, 0, (int)(p.n * 27 * 4), 0x00020000);
  const int prow = lane & 31, phalf = lane >> 5;
  const int plane_bytes = (int)(p.n * 4);
  const int lrow = K::QUAD ? lane >> 2 : jrow, lchunk = K::QUAD ? lane & 3 : cgrp;  // load map: lane 4 j + c | the operand's
  const int to_frag = (4 * jrow + cgrp) << 2;  // ds_bpermute address: operand lane (j, c) takes from load lane 4 j + c

  const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, nbx = (gridDim.x + 7 - xcd) >> 3;
  const int per = (p.tiles + 7) >> 3;
  const int t_end = min(p.tiles, (xcd + 1) * per);
  for (int tile = xcd * per + bx; tile < t_end; tile += nbx) {
    const long row0 = (long)tile * (WAVES * K::ROWS_PER_WAVE) + wave * K::ROWS_PER_WAVE;
    if (row0 >= p.n) continue;  // no barrier below: waves are independent
    // ---- phase 1: the wave's block of the kernel map
    i32x16_t I;
    {
      long pr = row0 + prow;
      if (pr >= p.n) pr = p.n - 1;  // rows past the end read the last row's map: computed, never stored
      const unsigned voff = (unsigned)(pr * 4) + (phalf ? (unsigned)plane_bytes : 0u);
#pragma unroll
      for (int t = 0; t < 14; ++t) I[t] = __builtin_amdgcn_raw_buffer_load_b32(nbr_rsrc, voff, 2 * t * plane_bytes, 0);
      if (phalf) I[13] = -1;  // "offset 27": the padding entry of the live list
      I[14] = -1;
      I[15] = -1;
    }
    unsigned live = 0;
#pragma unroll
    for (int t = 0; t < 14; ++t) {
      const unsigned long long m = __builtin_amdgcn_ballot_w64(I[t] >= 0);
      if ((unsigned)m != 0u) live |= 1u << (2 * t);
      if ((unsigned)(m >> 32) != 0u) live |= 1u << (2 * t + 1);
    }
    f32x4_t acc[K::RG][K::CT];
#pragma unroll
    for (int g = 0; g < K::RG; ++g)
#pragma unroll
      for (int ct = 0; ct < K::CT; ++ct) acc[g][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t xb[NB][K::RG][K::KS];
    auto pop = [&]() {
      int o = 27;
      if (live) {
        o = __builtin_ctz(live);
        live &= live - 1;
      }
      return o;
    };
    // one list entry = four pieces:
    //   perm   the lane's row indices of offset o out of the register block (ds_bpermute)
    //   rows   the row loads (KS per 16-row group)
    //   wload  the offset's KS * CT weight fragments, ALL requested at once (one LDS round trip per step, not one per
    //          fragment: 160 -> 122 us at C = 64 together with the two fixes noted below; requesting them a step ahead
    //          into a second fragment set, and loading the next tile's map block a tile ahead, changed nothing)
    //   mfmas  the products
    auto perm = [&](int o, int (&id)[K::RG]) {
      const int v = I[__builtin_amdgcn_readfirstlane(o >> 1)];
#pragma unroll
      for (int g = 0; g < K::RG; ++g) id[g] = __builtin_amdgcn_ds_bpermute((((o & 1) << 5) + 16 * g + lrow) << 2, v);
    };
    auto rows = [&](const int (&id)[K::RG], auto buf) {
      constexpr int B = decltype(buf)::value;
#pragma unroll
      for (int g = 0; g < K::RG; ++g)
#pragma unroll
        for (int ks = 0; ks < K::KS; ++ks) {
          // -1 << row_shift wraps beyond the end of x: zeros, no memory access
          const unsigned off = ((unsigned)id[g] << p.row_shift) + (unsigned)((ks * 4 + lchunk) * 16);
          xb[B][g][ks] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, off, 0, 0));
        }
    };
    int lane_unit = cgrp * 16 + jrow;
    asm volatile("" : "+v"(lane_unit));  // opaque per tile: keeps the fragment address arithmetic inside the tile loop
    auto wload = [&](int o, bf16x8_t (&wf)[K::KS][K::CT]) {
      if (o >= 27) return;
      // slot of the offset from a 5-bit-per-entry table in scalar registers (an indexed read of the __constant__ table
      // compiles to a VECTOR load + vmcnt(0): it drained the row loads in flight at every step)
      int slot = o;
      if (K::LDS_OFFSETS != 27) {
        const unsigned long long tab = o < 12 ? SlotBits::t0 : (o < 24 ? SlotBits::t1 : SlotBits::t2);
        const int sh = 5 * (o < 12 ? o : (o < 24 ? o - 12 : o - 24));
        slot = (int)((tab >> sh) & 31ull);
      }
      if (K::LDS_OFFSETS == 27 || slot < K::LDS_OFFSETS) {
        const char* base = smem + slot * K::OFF_BYTES + lane_unit * 16;
#pragma unroll
        for (int ks = 0; ks < K::KS; ++ks)
#pragma unroll
          for (int ct = 0; ct < K::CT; ++ct)
            wf[ks][ct] = *reinterpret_cast<const bf16x8_t*>(base + (ct * K::KC + ks * 4) * 256);
      } else {
        // corner offset of the C = 64 kernel: from L2.  The explicit vmcnt(0) keeps the fragment registers from counting
        // as "pending on vmcnt" after the join, where EVERY step would then wait for the row loads in flight (and a
        // per-lane LDS / global address select would become FLAT loads)
        const uint4* base = wimg + slot * (K::OFF_BYTES / 16) + lane_unit;
#pragma unroll
        for (int ks = 0; ks < K::KS; ++ks)
#pragma unroll
          for (int ct = 0; ct < K::CT; ++ct)
            wf[ks][ct] = *reinterpret_cast<const bf16x8_t*>(base + (ct * K::KC + ks * 4) * 16);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
      }
    };
    auto mfmas = [&](int o, const bf16x8_t (&wf)[K::KS][K::CT], auto buf) {
      constexpr int B = decltype(buf)::value;
      if (o >= 27) return;
      bf16x8_t xq[K::RG][K::KS];
#pragma unroll
      for (int g = 0; g < K::RG; ++g)
#pragma unroll
        for (int ks = 0; ks < K::KS; ++ks) {
          if constexpr (K::QUAD) {
            i32x4_t t = __builtin_bit_cast(i32x4_t, xb[B][g][ks]);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = __builtin_amdgcn_ds_bpermute(to_frag, t[e]);
            xq[g][ks] = __builtin_bit_cast(bf16x8_t, t);
          } else {
            xq[g][ks] = xb[B][g][ks];
          }
        }
#pragma unroll
      for (int ks = 0; ks < K::KS; ++ks)
#pragma unroll
        for (int ct = 0; ct < K::CT; ++ct)
#pragma unroll
          for (int g = 0; g < K::RG; ++g)
            acc[g][ct] = mfma_16x16x32_bf16(wf[ks][ct], xq[g][ks], acc[g][ct]);
    };
#if CONV_CORNER_FIRST
    // (experiment) a corner offset's weights are requested BEFORE the step's row loads and waited for with vmcnt(row loads of
    // the step): the newest rows stay in flight instead of being drained with everything else
    auto slot_of = [&](int o) {
      const unsigned long long tab = o < 12 ? SlotBits::t0 : (o < 24 ? SlotBits::t1 : SlotBits::t2);
      const int sh = 5 * (o < 12 ? o : (o < 24 ? o - 12 : o - 24));
      return (int)((tab >> sh) & 31ull);
    };
    auto wload_l2 = [&](int slot, bf16x8_t (&wf)[K::KS][K::CT]) {
      const uint4* base = wimg + slot * (K::OFF_BYTES / 16) + lane_unit;
#pragma unroll
      for (int ks = 0; ks < K::KS; ++ks)
#pragma unroll
        for (int ct = 0; ct < K::CT; ++ct)
          wf[ks][ct] = *reinterpret_cast<const bf16x8_t*>(base + (ct * K::KC + ks * 4) * 16);
    };
    auto wload_lds = [&](int slot, bf16x8_t (&wf)[K::KS][K::CT]) {
      const char* base = smem + slot * K::OFF_BYTES + lane_unit * 16;
#pragma unroll
      for (int ks = 0; ks < K::KS; ++ks)
#pragma unroll
        for (int ct = 0; ct < K::CT; ++ct)
          wf[ks][ct] = *reinterpret_cast<const bf16x8_t*>(base + (ct * K::KC + ks * 4) * 256);
    };
#endif
    int o[NB];
    static_for<0, NB - 1>([&](auto J) {
      o[J.value] = pop();
      int id[K::RG];
      perm(o[J.value], id);
      rows(id, J);
    });
#pragma unroll 1
    do {  // NB list entries per trip, no exits in between: at most NB - 1 padding steps per tile
      static_for<0, NB>([&](auto J) {
        constexpr int j = J.value, nb = (j + NB - 1) % NB;
        o[nb] = pop();
        int id[K::RG];
        perm(o[nb], id);
        bf16x8_t wf[K::KS][K::CT];
#if CONV_CORNER_FIRST
        if constexpr (K::LDS_OFFSETS != 27) {
          const int slot = o[j] < 27 ? slot_of(o[j]) : 0;
          const bool from_l2 = slot >= K::LDS_OFFSETS;
          if (from_l2) wload_l2(slot, wf);
          rows(id, std::integral_constant<int, nb>{});
          if (from_l2) __builtin_amdgcn_s_waitcnt(0x0F70 | (K::RG * K::KS));  // vmcnt(this step's row loads)
          else if (o[j] < 27) wload_lds(slot, wf);
        } else {
          rows(id, std::integral_constant<int, nb>{});
          wload(o[j], wf);
        }
#else
        rows(id, std::integral_constant<int, nb>{});
        wload(o[j], wf);
#endif
        mfmas(o[j], wf, J);
      });
    } while (o[0] < 27);
    // ---- epilogue: lane (j, cgrp) holds channels cgrp * C/4 + [0, C/4) of point j
    if (p.yf) {  // fp32 output, optionally accumulating (wave-uniform)
#pragma unroll
      for (int g = 0; g < K::RG; ++g) {
        const long myrow = row0 + g * 16 + jrow;
        if (myrow >= p.n) continue;
        float* dst = p.yf + myrow * p.ldyf + cgrp * (C / 4);
#pragma unroll
        for (int ct = 0; ct < K::CT; ++ct) {
          float4 b = make_float4(0.f, 0.f, 0.f, 0.f), o = b;
          if (p.bias) b = *reinterpret_cast<const float4*>(p.bias + cgrp * (C / 4) + 4 * ct);
          if (p.accumulate) o = *reinterpret_cast<const float4*>(dst + 4 * ct);
          o.x += (acc[g][ct][0] + b.x) * p.out_scale; o.y += (acc[g][ct][1] + b.y) * p.out_scale;
          o.z += (acc[g][ct][2] + b.z) * p.out_scale; o.w += (acc[g][ct][3] + b.w) * p.out_scale;
          *reinterpret_cast<float4*>(dst + 4 * ct) = o;
        }
      }
      continue;
    }
#pragma unroll
    for (int g = 0; g < K::RG; ++g) {
      const long myrow = row0 + g * 16 + jrow;
      if (myrow >= p.n) continue;
      bf16_t* dst = p.y + myrow * p.ldy + cgrp * (C / 4);
#pragma unroll
      for (int h = 0; h < K::CT / 2; ++h) {
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
        if (p.bias) {
          b0 = *reinterpret_cast<const float4*>(p.bias + cgrp * (C / 4) + 8 * h);
          b1 = *reinterpret_cast<const float4*>(p.bias + cgrp * (C / 4) + 8 * h + 4);
        }
        uint4 u;
        u.x = pack_bf16x2(acc[g][2 * h][0] + b0.x, acc[g][2 * h][1] + b0.y);
        u.y = pack_bf16x2(acc[g][2 * h][2] + b0.z, acc[g][2 * h][3] + b0.w);
        u.z = pack_bf16x2(acc[g][2 * h + 1][0] + b1.x, acc[g][2 * h + 1][1] + b1.y);
        u.w = pack_bf16x2(acc[g][2 * h + 1][2] + b1.z, acc[g][2 * h + 1][3] + b1.w);
        *reinterpret_cast<uint4*>(dst + 8 * h) = u;
      }
    }
  }
}

template <int C, int WAVES, int NB, int OCC>
int launch_conv_ll(const ConvP& p0, int per_cu, const void* wimg, hipStream_t s) {
  using K = ConvCfg<C>;
  ConvP p = p0;
  p.tiles = (int)((p.n + WAVES * K::ROWS_PER_WAVE - 1) / (WAVES * K::ROWS_PER_WAVE));
  int grid = 256 * per_cu;
  if (grid > p.tiles) grid = (p.tiles + 7) / 8 * 8;
  static std::atomic<bool> attr_done{false};  // (a concurrent first call sets the attribute twice: harmless)
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)conv_ll_kernel<C, WAVES, NB, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            K::LDS_BYTES) != hipSuccess)
      return CDSEG_ERR_LAUNCH;
    attr_done = true;
  }
  CdsegProfToken tok;
  const bool prof = cdseg_prof_begin(CDSEG_PROF_CONV, s, &tok);
  hipLaunchKernelGGL((conv_ll_kernel<C, WAVES, NB, OCC>), dim3(grid), dim3(WAVES * 64), K::LDS_BYTES, s, p, (const uint4*)wimg);
  if (prof) cdseg_prof_end(tok, s);
  if (hipGetLastError() != hipSuccess) return CDSEG_ERR_LAUNCH;
  return CDSEG_OK;
}

template <int C>
int launch_pack(const bf16_t* w, void* wimg, hipStream_t s) {
  using K = ConvCfg<C>;
  const int units = 27 * K::CT * K::KC * 16;
  hipLaunchKernelGGL((conv_pack_w_kernel<C>), dim3((units + K::WAVES * 64 - 1) / (K::WAVES * 64)), dim3(K::WAVES * 64), 0, s,
                     w, (uint4*)wimg);
  return hipGetLastError() == hipSuccess ? CDSEG_OK : CDSEG_ERR_LAUNCH;
}

template <int C>
int launch_conv(const ConvP& p0, const void* wimg, hipStream_t s) {
  using K = ConvCfg<C>;
  // persistent blocks: as many as are co-resident (LDS: 2 per CU at C = 32, 1 at C = 64), never more than tiles
  static const int blocks_per_cu = cdseg_knob("CDSEG_CONV_BLOCKS", 0);
  int per_cu = K::LDS_BYTES > 80 * 1024 ? 1 : 2;
  if (blocks_per_cu > 0) per_cu = blocks_per_cu;
  // C = 32: 8 waves, 3 row buffers, 4 waves / SIMD; C = 64: 8 waves on the 256-register budget of 2 waves / SIMD,
  // 4 row buffers (16 waves x 2 buffers spills and is 1.8x slower; 6 buffers, 12-wave blocks at 3 waves / SIMD and, at
  // C = 32, 12-wave blocks or 4 buffers: no gain - profiles/r02_conv_livelist_sweep.txt)
  if constexpr (C == 32) return launch_conv_ll<32, 8, CONV32_ROW_BUFFERS, 4>(p0, per_cu, wimg, s);
  else return launch_conv_ll<64, 8, CONV64_ROW_BUFFERS, 2>(p0, per_cu, wimg, s);
}

}  // namespace

extern "C" size_t cdseg_subm_conv3_wimg_bytes(int channels) {
  return (channels == 32 || channels == 64) ? (size_t)27 * channels * channels * 2 : 0;
}

extern "C" int cdseg_subm_conv3_pack(const void* w, int channels, void* wimg, void* stream) {
  if (!w || !wimg || (((uintptr_t)w | (uintptr_t)wimg) & 15)) return CDSEG_ERR_ARG;
  if (channels == 32) return launch_pack<32>((const bf16_t*)w, wimg, (hipStream_t)stream);
  if (channels == 64) return launch_pack<64>((const bf16_t*)w, wimg, (hipStream_t)stream);
  return CDSEG_ERR_UNSUPPORTED;
}

extern "C" int cdseg_subm_conv3(const void* x, int ldx, const void* wimg, const float* bias, const int32_t* nbr_kmajor,
                                long n, int channels, void* y, int ldy, void* stream) {
  if (!x || !nbr_kmajor || !y || !wimg) return CDSEG_ERR_ARG;
  if (n <= 0) return CDSEG_OK;
  if (channels != 32 && channels != 64) return CDSEG_ERR_UNSUPPORTED;
  if (n * 27 * 4 >= (1l << 31) || n * (long)ldx * 2 >= (1l << 31) - 65536) return CDSEG_ERR_UNSUPPORTED;  // 32-bit buffer offsets
  if ((ldx & 7) || (ldy & 7) || (((uintptr_t)x | (uintptr_t)y | (uintptr_t)wimg) & 15)) return CDSEG_ERR_ARG;
  if (bias && (((uintptr_t)bias) & 15)) return CDSEG_ERR_ARG;
  ConvP p;
  p.x = (const bf16_t*)x; p.bias = bias; p.nbr = nbr_kmajor; p.y = (bf16_t*)y;
  p.n = n; p.ldx = ldx; p.ldy = ldy; p.tiles = 0;
  p.yf = nullptr; p.ldyf = 0; p.accumulate = 0; p.out_scale = 1.f;
  p.row_shift = 0;
  while ((1 << p.row_shift) < ldx * 2) ++p.row_shift;
  if ((1 << p.row_shift) != ldx * 2) return CDSEG_ERR_UNSUPPORTED;  // feature rows with a power-of-two stride only
  hipStream_t s = (hipStream_t)stream;
  return channels == 32 ? launch_conv<32>(p, wimg, s) : launch_conv<64>(p, wimg, s);
}

// The same kernel with an fp32 output: yf = (accumulate ? yf : 0) + (sum_o W_o x[nbr] + bias) * out_scale.  Precision
// "fp32x3" runs it three times per conv on IEEE-half operand pairs (x_hi W_hi; x_hi W_lo and x_lo W_hi with out_scale = 2^-11)
extern "C" int cdseg_subm_conv3_f32(const void* x, int ldx, const void* wimg, const float* bias, const int32_t* nbr_kmajor,
                                    long n, int channels, float* yf, int ldyf, float out_scale, int accumulate, void* stream) {
  if (!x || !nbr_kmajor || !yf || !wimg) return CDSEG_ERR_ARG;
  if (n <= 0) return CDSEG_OK;
  if (channels != 32 && channels != 64) return CDSEG_ERR_UNSUPPORTED;
  if (n * 27 * 4 >= (1l << 31) || n * (long)ldx * 2 >= (1l << 31) - 65536) return CDSEG_ERR_UNSUPPORTED;
  if ((ldx & 7) || (ldyf & 3) || (((uintptr_t)x | (uintptr_t)yf | (uintptr_t)wimg) & 15)) return CDSEG_ERR_ARG;
  if (bias && (((uintptr_t)bias) & 15)) return CDSEG_ERR_ARG;
  ConvP p;
  p.x = (const bf16_t*)x; p.bias = bias; p.nbr = nbr_kmajor; p.y = nullptr;
  p.n = n; p.ldx = ldx; p.ldy = 0; p.tiles = 0;
  p.yf = yf; p.ldyf = ldyf; p.accumulate = accumulate ? 1 : 0; p.out_scale = out_scale;
  p.row_shift = 0;
  while ((1 << p.row_shift) < ldx * 2) ++p.row_shift;
  if ((1 << p.row_shift) != ldx * 2) return CDSEG_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  return channels == 32 ? launch_conv<32>(p, wimg, s) : launch_conv<64>(p, wimg, s);
}
