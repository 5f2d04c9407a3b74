// Block head and tail of the DEEP stages (C = 128 / 256, 16-bit trunk) in one launch each (gfx950).
//
//   head (ref: ptv3.py:401-414)   x += LN_cpe(y Wl^T + bl) [+ t bias] ;  h = LN1(x) ;  qkv = h Wqkv^T + bqkv
//   tail (ref: ptv3.py:416-427)   x += proj(o) ;  h = LN2(x) ;  x += fc2(GELU(fc1(h))) ;  xc = T(x)
//
// The wide stages (C = 32 / 64, blockrr.hip) keep every weight of the kernel in LDS and the activations in registers.
// Here the weights are too large for that (tail: 1.2 MB at C = 256) and the rows too few to amortise a weight tile per
// workgroup through LDS: as separate GEMM launches these products ran at 5 - 19 % of the MFMA peak (a 128 x 128 tile
// waits ~1.7 us for every 64-deep K step of A + W, then spends 2 us in its epilogue; LayerNorm over rows wider than a
// column tile needed a second pass).  So the roles are swapped:
//   * the ACTIVATIONS of a BM-row tile live in LDS (16-bit, row-major, 16-byte chunks XOR-swizzled with the row so that
//     both the MFMA B-fragment reads and the producers' 16-byte writes are bank-conflict free) for the whole chain of
//     products - y / o, h, the hidden chunk - and never touch HBM;
//   * the WEIGHTS stream L2 -> registers: a wave owns the same 32 output channels of every C-wide product for ALL rows
//     of the tile (C / 32 waves per workgroup), so a weight fragment is needed by exactly one wave of the workgroup
//     and goes straight into its MFMA A operand - no LDS staging, no barrier per K step.  The images are packed
//     per wave in consumption order (cdseg_block_rr_pack), i.e. a wave reads ONE linear stream of 2 KB per K step
//     through a 4-step register ring that runs across the products and their epilogues;
//   * products are computed transposed (D^T = W X^T, v_mfma_f32_16x16x32): a lane ends up with 8 CONSECUTIVE channels
//     of one point (the rows of the weight image are permuted accordingly), so every LDS / global access of the
//     epilogues is 16 bytes; LayerNorm statistics are a per-lane sum, two cross-lane adds and one exchange of per-wave
//     partial sums through LDS;
//   * the MLP runs in hidden chunks of C units: fc1 chunk -> bias + GELU -> 16-bit -> LDS -> fc2 partial product into
//     accumulators that were initialised with the residual row.
// Per workgroup and weight byte the tile does BM (128) FLOP pairs instead of the 64 of a 128 x 128 GEMM tile, the
// weight stream of all workgroups is the same sequence (L2 hits), and the launches per Block drop from 8 to 2.
#include <atomic>

#include "deep.h"

namespace {

constexpr int DEEP_D = 4;  // weight ring depth in K steps (2 fragments = 8 VGPRs per step)

#ifdef CDSEG_DEEP_TIMING
// experimental builds only (tools/deep_timing.py): cycle stamps of wave 0 of the first 2048 blocks at the phase boundaries
__device__ unsigned long long g_deep_t[2048 * 16];
#define DT_STAMP(k) if (wave == 0 && lane == 0 && blockIdx.x < 2048) g_deep_t[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter()
#else
#define DT_STAMP(k)
#endif

template <int NCH>
__device__ __forceinline__ int act_off(int row, int chunk) {  // byte offset of 16-byte chunk `chunk` of activation row `row`
  return row * (NCH * 16) + ((chunk ^ (row & 15)) << 4);
}

__device__ __forceinline__ void lds_barrier() {  // LDS hazards only: the weight ring's global loads stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ uint4 pack8(const f32x4_t& a, const f32x4_t& b) {
  uint4 r;
  r.x = pack_bf16x2(a[0], a[1]); r.y = pack_bf16x2(a[2], a[3]);
  r.z = pack_bf16x2(b[0], b[1]); r.w = pack_bf16x2(b[2], b[3]);
  return r;
}
__device__ __forceinline__ uint4 pack8_truebf16(const f32x4_t& a, const f32x4_t& b) {
  uint4 r;
  r.x = pack_truebf16x2(a[0], a[1]); r.y = pack_truebf16x2(a[2], a[3]);
  r.z = pack_truebf16x2(b[0], b[1]); r.w = pack_truebf16x2(b[2], b[3]);
  return r;
}

// BM x C tile of 16-bit rows, global -> LDS (swizzled).  Rows past the end re-read the last row (computed, never stored).
template <int C, int BM>
__device__ __forceinline__ void load_tile(const bf16_t* src, int ld, long m0, long n, char* buf, int tid) {
  constexpr int NCH = C / 8, NT = 2 * C, PT = BM * NCH / NT;
  uint4 r[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int id = i * NT + tid, row = id / NCH, c = id % NCH;
    long m = m0 + row;
    if (m >= n) m = n - 1;
    r[i] = *reinterpret_cast<const uint4*>(src + m * ld + c * 8);
  }
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int id = i * NT + tid, row = id / NCH, c = id % NCH;
    *reinterpret_cast<uint4*>(buf + act_off<NCH>(row, c)) = r[i];
  }
}

// The same tile from `splits` raw split-K partial planes (splits, n, C) fp32 of the conv that produced y: slice sum in slice order,
// + bias, rounded with the conversion the conv's own second pass uses (gemm.hip splitk_epilogue_kernel / epilogue4 / store_vec4):
// bit-identical rows, one launch fewer.
template <int C, int BM>
__device__ __forceinline__ void load_tile_partials(const float* part, int splits, const float* bias, long m0, long n, char* buf,
                                                   int tid) {
  constexpr int NCH = C / 8, NT = 2 * C, PT = BM * NCH / NT;
  const long plane = n * (long)C;
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int id = i * NT + tid, row = id / NCH, c = id % NCH;
    long m = m0 + row;
    if (m >= n) m = n - 1;
    const float* q = part + m * C + c * 8;
    f32x4_t a = *reinterpret_cast<const f32x4_t*>(q), b = *reinterpret_cast<const f32x4_t*>(q + 4);
    for (int s = 1; s < splits; ++s) {
      a = a + *reinterpret_cast<const f32x4_t*>(q + s * plane);
      b = b + *reinterpret_cast<const f32x4_t*>(q + s * plane + 4);
    }
    a = a + *reinterpret_cast<const f32x4_t*>(bias + c * 8);
    b = b + *reinterpret_cast<const f32x4_t*>(bias + c * 8 + 4);
    *reinterpret_cast<uint4*>(buf + act_off<NCH>(row, c)) = pack8(a, b);
  }
}

// acc[pt][f] (channels 8 g + 4 f + r of the wave's 32, point 16 pt + p) += W X^T over K = C: KS steps of 32, the two
// weight fragments of a step from the ring, the next ring slot requested right after.  `step0`: position of the
// product's first step in the wave's weight stream; requests past the end re-read the last step (no branch around a
// load: hipcc would wait vmcnt(0) there).
template <int C, int BM>
__device__ __forceinline__ void product(f32x4_t (&acc)[BM / 16][2], const char* bufX, uint4 (&ring)[DEEP_D][2],
                                        const uint4* wp, int step0, int last_step, int p, int g) {
  constexpr int KS = C / 32, NCH = C / 8, PTS = BM / 16;
  const char* xrow = bufX + p * (NCH * 16);
  // the B fragments (activation rows) of step s + 1 are requested before the MFMAs of step s: with two waves per SIMD
  // and only two fragments in flight per wave (what the scheduler chose on its own) every pair of reads was a
  // full LDS round trip in front of four MFMAs
  bf16x8_t b[2][PTS];
  auto fetch = [&](int s, bf16x8_t (&dst)[PTS]) {
    const char* xs = xrow + (((4 * s + g) ^ p) << 4);
#pragma unroll
    for (int pt = 0; pt < PTS; ++pt) dst[pt] = *reinterpret_cast<const bf16x8_t*>(xs + pt * 16 * (NCH * 16));
  };
  fetch(0, b[0]);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const bf16x8_t a0 = __builtin_bit_cast(bf16x8_t, ring[s % DEEP_D][0]);
    const bf16x8_t a1 = __builtin_bit_cast(bf16x8_t, ring[s % DEEP_D][1]);
    if (s + 1 < KS) fetch(s + 1, b[(s + 1) & 1]);
#pragma unroll
    for (int pt = 0; pt < PTS; ++pt) {
      acc[pt][0] = mfma_16x16x32_bf16(a0, b[s & 1][pt], acc[pt][0]);
      acc[pt][1] = mfma_16x16x32_bf16(a1, b[s & 1][pt], acc[pt][1]);
    }
    {
      int nx = step0 + s + DEEP_D;
      nx = nx < last_step ? nx : last_step;
      ring[s % DEEP_D][0] = wp[(nx * 2 + 0) * 64];
      ring[s % DEEP_D][1] = wp[(nx * 2 + 1) * 64];
    }
    // one request per step, DEEP_D steps ahead of its use: left to itself the scheduler sinks the requests towards their
    // uses and bunches them at the end of the product (s_waitcnt vmcnt(0) right behind a request in the middle of it)
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ void prime(uint4 (&ring)[DEEP_D][2], const uint4* wp, int step0) {
#pragma unroll
  for (int d = 0; d < DEEP_D; ++d) {
    ring[d][0] = wp[((step0 + d) * 2 + 0) * 64];
    ring[d][1] = wp[((step0 + d) * 2 + 1) * 64];
  }
}

// LayerNorm statistics of the tile's rows: a row's C channels are spread over 4 lanes (g) x NW waves.
template <int PTS, int NW>
__device__ __forceinline__ void row_stats(const f32x4_t (&v)[PTS][2], float* st1, float* st2, int wave, int p, int g,
                                          float inv_c, float eps, float (&mean)[PTS], float (&rstd)[PTS]) {
  auto across = [&](float* st, int pt) {
    float t = 0.f;
    if constexpr (NW == 16) {
      const float* s0 = st + (16 * pt + p) * NW;
      const f32x4_t a = *reinterpret_cast<const f32x4_t*>(s0), b = *reinterpret_cast<const f32x4_t*>(s0 + 4);
      const f32x4_t c = *reinterpret_cast<const f32x4_t*>(s0 + 8), d = *reinterpret_cast<const f32x4_t*>(s0 + 12);
      t = (((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]))) +
          (((c[0] + c[1]) + (c[2] + c[3])) + ((d[0] + d[1]) + (d[2] + d[3])));
    } else if constexpr (NW == 8) {
      const f32x4_t a = *reinterpret_cast<const f32x4_t*>(st + (16 * pt + p) * NW);
      const f32x4_t b = *reinterpret_cast<const f32x4_t*>(st + (16 * pt + p) * NW + 4);
      t = ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
    } else {
      const f32x4_t a = *reinterpret_cast<const f32x4_t*>(st + (16 * pt + p) * NW);
      t = (a[0] + a[1]) + (a[2] + a[3]);
    }
    return t;
  };
#pragma unroll
  for (int pt = 0; pt < PTS; ++pt) {
    float s = ((v[pt][0][0] + v[pt][0][1]) + (v[pt][0][2] + v[pt][0][3])) +
              ((v[pt][1][0] + v[pt][1][1]) + (v[pt][1][2] + v[pt][1][3]));
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (g == 0) st1[(16 * pt + p) * NW + wave] = s;
  }
  lds_barrier();
#pragma unroll
  for (int pt = 0; pt < PTS; ++pt) mean[pt] = across(st1, pt) * inv_c;
#pragma unroll
  for (int pt = 0; pt < PTS; ++pt) {
    float q = 0.f;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = v[pt][f][r] - mean[pt];
        q += d * d;
      }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    if (g == 0) st2[(16 * pt + p) * NW + wave] = q;
  }
  lds_barrier();
#pragma unroll
  for (int pt = 0; pt < PTS; ++pt) rstd[pt] = 1.0f / sqrtf(across(st2, pt) * inv_c + eps);
}

template <int C, int BM>
struct DeepCfg {
  static constexpr int NW = C / 32, KS = C / 32, NT = 64 * NW;
  static constexpr int ACT = BM * C * 2;                // one activation buffer
  static constexpr int STAT = 2 * BM * NW * 4;          // two arrays of per-wave partial sums
  static constexpr int HEAD_LDS = ACT + STAT + 9 * C * 4;       // bl lnp_g lnp_b colbias ln1_g ln1_b bqkv(3C)
  static constexpr int TAIL_LDS = 2 * ACT + STAT + 8 * C * 4;   // bp ln_g ln_b b2 b1(4C)
  static constexpr int HEAD_STEPS = 4 * KS, TAIL_STEPS = 9 * KS;
};

struct DeepHeadP {
  const bf16_t* y; const uint4* wimg; const float* bl; const float* lnp_g; const float* lnp_b; const float* x;
  float* x_out;  // the updated residual rows (= x for the in-place form; a split launch needs x_out != x: three workgroups read a row)
  const float* colbias; const float* ln1_g; const float* ln1_b; const float* bqkv; bf16_t* qkv;
  long n; int ldy, ldx, ldxo, ldqkv; float eps;
  int v_bf16;  // IEEE-half build: write the v third as bfloat16 (CDSEG_ATTN_V_BF16)
  int nsplit;  // 1: a workgroup writes q, k and v of its tile; 3: workgroup (tile, c) writes column block c only (few-row launches)
  const float* ypart; const float* ybias; int ysplits;  // ysplits > 1: y as raw split-K partial planes + bias (load_tile_partials)
};

template <int C, int BM>
__global__ __launch_bounds__(2 * C, 2) void deep_head_kernel(DeepHeadP P) {
  using K = DeepCfg<C, BM>;
  constexpr int NW = K::NW, KS = K::KS, NCH = C / 8, NT = K::NT, PTS = BM / 16, S = K::HEAD_STEPS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* bufA = smem;
  float* st1 = reinterpret_cast<float*>(smem + K::ACT);
  float* st2 = st1 + BM * NW;
  float* pr = st2 + BM * NW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, g = lane >> 4;
  // few-row launches (a single scene's deep stages): three workgroups per tile, one per q / k / v column block - each repeats
  // the cpe linear + the two LayerNorms (a quarter of the weight stream) and streams a third of Wqkv, so the chain a
  // workgroup waits for is half as long and three times as many CUs stream weights; x is written by block 0 alone
  const int c_lo = P.nsplit == 3 ? (int)(blockIdx.x % 3u) : 0, c_hi = P.nsplit == 3 ? c_lo + 1 : 3;
  const long m0 = (long)(P.nsplit == 3 ? blockIdx.x / 3u : blockIdx.x) * BM;
  const uint4* wp = P.wimg + (size_t)wave * S * 128 + lane;
  DT_STAMP(0);
  uint4 ring[DEEP_D][2];
  prime(ring, wp, 0);
  for (int c = tid; c < C; c += NT) {
    pr[c] = P.bl[c]; pr[C + c] = P.lnp_g[c]; pr[2 * C + c] = P.lnp_b[c]; pr[3 * C + c] = P.colbias ? P.colbias[c] : 0.f;
    pr[4 * C + c] = P.ln1_g[c]; pr[5 * C + c] = P.ln1_b[c];
  }
  for (int c = tid; c < 3 * C; c += NT) pr[6 * C + c] = P.bqkv[c];
  if (P.ysplits > 1) load_tile_partials<C, BM>(P.ypart, P.ysplits, P.ybias, m0, P.n, bufA, tid);
  else load_tile<C, BM>(P.y, P.ldy, m0, P.n, bufA, tid);
  const int ch0 = 32 * wave + 8 * g;  // the lane's channels: ch0 + 4 f + r
  lds_barrier();  // tile + parameters visible
  DT_STAMP(1);

  f32x4_t v[PTS][2];
  {
    const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(pr + ch0), b1 = *reinterpret_cast<const f32x4_t*>(pr + ch0 + 4);
#pragma unroll
    for (int pt = 0; pt < PTS; ++pt) { v[pt][0] = b0; v[pt][1] = b1; }
  }
  // y Wl^T + bl.  The ring is NOT kept running across the two LayerNorms that follow (32 more live VGPRs next to the
  // product, the residual rows and the statistics: the compiler spilled the prefetched fragments with vmcnt(0) waits
  // inside the product); the qkv stream is primed again once the residual rows are dead
  product<C, BM>(v, bufA, ring, wp, 0, KS - 1, p, g);
  DT_STAMP(2);
  // residual rows: needed after the first LayerNorm's statistics, requested before them (live across the product
  // - where the scheduler hoists them unless fenced - they cost 190 spilled VGPRs)
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  f32x4_t xr[PTS][2];
#pragma unroll
  for (int pt = 0; pt < PTS; ++pt) {
    long m = m0 + 16 * pt + p;
    if (m >= P.n) m = P.n - 1;
#pragma unroll
    for (int f = 0; f < 2; ++f) xr[pt][f] = *reinterpret_cast<const f32x4_t*>(P.x + m * P.ldx + ch0 + 4 * f);
  }
  float mean[PTS], rstd[PTS];
  const float inv_c = 1.0f / C;
  row_stats<PTS, NW>(v, st1, st2, wave, p, g, inv_c, P.eps, mean, rstd);
  DT_STAMP(3);
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const f32x4_t ga = *reinterpret_cast<const f32x4_t*>(pr + C + ch0 + 4 * f);
    const f32x4_t be = *reinterpret_cast<const f32x4_t*>(pr + 2 * C + ch0 + 4 * f);
    const f32x4_t tb = *reinterpret_cast<const f32x4_t*>(pr + 3 * C + ch0 + 4 * f);
#pragma unroll
    for (int pt = 0; pt < PTS; ++pt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[pt][f][r] = ((v[pt][f][r] - mean[pt]) * rstd[pt] * ga[r] + be[r]) + xr[pt][f][r] + tb[r];
      const long m = m0 + 16 * pt + p;
      if (m < P.n && c_lo == 0) *reinterpret_cast<f32x4_t*>(P.x_out + m * P.ldxo + ch0 + 4 * f) = v[pt][f];
    }
  }
  DT_STAMP(4);
  row_stats<PTS, NW>(v, st1, st2, wave, p, g, inv_c, P.eps, mean, rstd);
  DT_STAMP(5);
  prime(ring, wp, KS * (1 + c_lo));
  {
    // h = LN1(x) over the tile, in place of y (every wave is past its y reads: the statistics barriers above)
    const f32x4_t ga0 = *reinterpret_cast<const f32x4_t*>(pr + 4 * C + ch0), ga1 = *reinterpret_cast<const f32x4_t*>(pr + 4 * C + ch0 + 4);
    const f32x4_t be0 = *reinterpret_cast<const f32x4_t*>(pr + 5 * C + ch0), be1 = *reinterpret_cast<const f32x4_t*>(pr + 5 * C + ch0 + 4);
#pragma unroll
    for (int pt = 0; pt < PTS; ++pt) {
      f32x4_t h0, h1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        h0[r] = (v[pt][0][r] - mean[pt]) * rstd[pt] * ga0[r] + be0[r];
        h1[r] = (v[pt][1][r] - mean[pt]) * rstd[pt] * ga1[r] + be1[r];
      }
      *reinterpret_cast<uint4*>(bufA + act_off<NCH>(16 * pt + p, 4 * wave + g)) = pack8(h0, h1);
    }
  }
  lds_barrier();
  DT_STAMP(6);
#pragma unroll 1
  for (int c = c_lo; c < c_hi; ++c) {  // q, k, v column blocks
    f32x4_t a[PTS][2];
    {
      const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(pr + 6 * C + c * C + ch0);
      const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(pr + 6 * C + c * C + ch0 + 4);
#pragma unroll
      for (int pt = 0; pt < PTS; ++pt) { a[pt][0] = b0; a[pt][1] = b1; }
    }
    int po = p;  // opaque per iteration: h is loop invariant, and its 64 fragment reads (256 VGPRs) would be hoisted
    asm volatile("" : "+v"(po));
    product<C, BM>(a, bufA, ring, wp, KS * (1 + c), S - 1, po, g);
    DT_STAMP(7 + 2 * c);
#pragma unroll
    for (int pt = 0; pt < PTS; ++pt) {
      const long m = m0 + 16 * pt + p;
      if (m < P.n)
        *reinterpret_cast<uint4*>(P.qkv + m * P.ldqkv + c * C + ch0) =
            (LP_IS_F16 && c == 2 && P.v_bf16) ? pack8_truebf16(a[pt][0], a[pt][1]) : pack8(a[pt][0], a[pt][1]);
    }
    DT_STAMP(8 + 2 * c);
  }
}

struct DeepTailP {
  const bf16_t* o; const uint4* wimg; const float* bp; const float* ln_g; const float* ln_b; const float* b1;
  const float* b2; const float* x_in; float* x; bf16_t* xc;  // x_in: residual rows (= x for the in-place form)
  long n; int ldo, ldx, ldxi, ldxc; float eps;
  int nsplit;   // 1 | 2 | 4: workgroup (tile, js) runs hidden chunks [4 js / nsplit, 4 (js + 1) / nsplit) of the MLP
  float* part;  // nsplit > 1: (nsplit, n, C) fp32 partial rows; deep_tail_reduce_kernel adds them up in a fixed order
};

template <int C, int BM>
__global__ __launch_bounds__(2 * C, 2) void deep_tail_kernel(DeepTailP P) {
  using K = DeepCfg<C, BM>;
  constexpr int NW = K::NW, KS = K::KS, NCH = C / 8, NT = K::NT, PTS = BM / 16, S = K::TAIL_STEPS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* bufA = smem;             // o, then h
  char* bufU = smem + K::ACT;    // hidden chunk (C units)
  float* st1 = reinterpret_cast<float*>(smem + 2 * K::ACT);
  float* st2 = st1 + BM * NW;
  float* pr = st2 + BM * NW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, g = lane >> 4;
  // few-row launches (round 6): a tile's MLP is cut by hidden chunks over nsplit workgroups - each repeats proj + LayerNorm
  // (one ninth of the weight stream) and streams its own chunks' fc1 / fc2 weights, so the serial weight chain a workgroup
  // waits for shrinks from 9 to 3 phases at nsplit = 4 and four times as many CUs stream; the partial rows (split 0's carry
  // the residual) meet in deep_tail_reduce_kernel.  At 778 rows x C = 512 the unsplit tile took ~56 us whatever the row count
  const int nsp = P.nsplit, js = nsp > 1 ? (int)(blockIdx.x % (unsigned)nsp) : 0;
  const int j_lo = js * (4 / nsp), j_hi = j_lo + 4 / nsp;
  const long m0 = (long)(nsp > 1 ? blockIdx.x / (unsigned)nsp : blockIdx.x) * BM;
  const uint4* wp = P.wimg + (size_t)wave * S * 128 + lane;
  DT_STAMP(0);
  uint4 ring[DEEP_D][2];
  prime(ring, wp, 0);
  for (int c = tid; c < C; c += NT) {
    pr[c] = P.bp[c]; pr[C + c] = P.ln_g[c]; pr[2 * C + c] = P.ln_b[c]; pr[3 * C + c] = P.b2[c];
  }
  for (int c = tid; c < 4 * C; c += NT) pr[4 * C + c] = P.b1[c];
  load_tile<C, BM>(P.o, P.ldo, m0, P.n, bufA, tid);
  const int ch0 = 32 * wave + 8 * g;
  f32x4_t acc2[PTS][2];  // proj + bp, then x' = that + x, then x' + fc2 partial sums
  {
    f32x4_t xr[PTS][2];
#pragma unroll
    for (int pt = 0; pt < PTS; ++pt) {
      long m = m0 + 16 * pt + p;
      if (m >= P.n) m = P.n - 1;
#pragma unroll
      for (int f = 0; f < 2; ++f) xr[pt][f] = *reinterpret_cast<const f32x4_t*>(P.x_in + m * P.ldxi + ch0 + 4 * f);
    }
    lds_barrier();
    DT_STAMP(1);
    const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(pr + ch0), b1 = *reinterpret_cast<const f32x4_t*>(pr + ch0 + 4);
#pragma unroll
    for (int pt = 0; pt < PTS; ++pt) { acc2[pt][0] = b0; acc2[pt][1] = b1; }
    product<C, BM>(acc2, bufA, ring, wp, 0, S - 1, p, g);
    DT_STAMP(2);
#pragma unroll
    for (int pt = 0; pt < PTS; ++pt) { acc2[pt][0] += xr[pt][0]; acc2[pt][1] += xr[pt][1]; }
  }
  {
    float mean[PTS], rstd[PTS];
    DT_STAMP(3);
    row_stats<PTS, NW>(acc2, st1, st2, wave, p, g, 1.0f / C, P.eps, mean, rstd);
    DT_STAMP(4);
    const f32x4_t ga0 = *reinterpret_cast<const f32x4_t*>(pr + C + ch0), ga1 = *reinterpret_cast<const f32x4_t*>(pr + C + ch0 + 4);
    const f32x4_t be0 = *reinterpret_cast<const f32x4_t*>(pr + 2 * C + ch0), be1 = *reinterpret_cast<const f32x4_t*>(pr + 2 * C + ch0 + 4);
#pragma unroll
    for (int pt = 0; pt < PTS; ++pt) {
      f32x4_t h0, h1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        h0[r] = (acc2[pt][0][r] - mean[pt]) * rstd[pt] * ga0[r] + be0[r];
        h1[r] = (acc2[pt][1][r] - mean[pt]) * rstd[pt] * ga1[r] + be1[r];
      }
      *reinterpret_cast<uint4*>(bufA + act_off<NCH>(16 * pt + p, 4 * wave + g)) = pack8(h0, h1);
    }
  }
  if (js) {  // the ring holds chunk 0's first steps (requested behind proj): this workgroup's chunks start elsewhere, and its
    prime(ring, wp, KS * (1 + 2 * j_lo));  // partial rows carry no residual
    const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pt = 0; pt < PTS; ++pt) { acc2[pt][0] = z; acc2[pt][1] = z; }
  }
  lds_barrier();
  DT_STAMP(5);
#ifdef CDSEG_DEEP_TIMING
  unsigned long long dt_fc1 = 0, dt_gelu = 0, dt_fc2 = 0;
#endif
#pragma unroll 1
  for (int j = j_lo; j < j_hi; ++j) {  // hidden units C j .. C j + C - 1
#ifdef CDSEG_DEEP_TIMING
    const unsigned long long q0 = __builtin_readcyclecounter();
#endif
    f32x4_t acc1[PTS][2];
    {
      const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(pr + 4 * C + j * C + ch0);
      const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(pr + 4 * C + j * C + ch0 + 4);
#pragma unroll
      for (int pt = 0; pt < PTS; ++pt) { acc1[pt][0] = b0; acc1[pt][1] = b1; }
    }
    int po = p;  // opaque per iteration: h is loop invariant (see deep_head_kernel)
    asm volatile("" : "+v"(po));
    product<C, BM>(acc1, bufA, ring, wp, KS * (1 + 2 * j), S - 1, po, g);
#ifdef CDSEG_DEEP_TIMING
    const unsigned long long q1 = __builtin_readcyclecounter();
#endif
    if (j > j_lo) lds_barrier();  // every wave is done with the previous chunk's fc2 reads of bufU
#pragma unroll
    for (int pt = 0; pt < PTS; ++pt) {
      gelu_lp4(acc1[pt][0]);
      gelu_lp4(acc1[pt][1]);
      *reinterpret_cast<uint4*>(bufU + act_off<NCH>(16 * pt + p, 4 * wave + g)) = pack8(acc1[pt][0], acc1[pt][1]);
    }
    lds_barrier();
#ifdef CDSEG_DEEP_TIMING
    const unsigned long long q2 = __builtin_readcyclecounter();
#endif
    product<C, BM>(acc2, bufU, ring, wp, KS * (2 + 2 * j), S - 1, p, g);
#ifdef CDSEG_DEEP_TIMING
    dt_fc1 += q1 - q0; dt_gelu += q2 - q1; dt_fc2 += __builtin_readcyclecounter() - q2;
#endif
  }
#ifdef CDSEG_DEEP_TIMING
  DT_STAMP(6);
  if (wave == 0 && lane == 0 && blockIdx.x < 2048) {
    g_deep_t[blockIdx.x * 16 + 8] = dt_fc1; g_deep_t[blockIdx.x * 16 + 9] = dt_gelu; g_deep_t[blockIdx.x * 16 + 10] = dt_fc2;
  }
#endif
  if (nsp > 1) {
#pragma unroll
    for (int pt = 0; pt < PTS; ++pt) {
      const long m = m0 + 16 * pt + p;
      if (m < P.n) {
        float* dst = P.part + ((size_t)js * P.n + m) * C + ch0;
        *reinterpret_cast<f32x4_t*>(dst) = acc2[pt][0];
        *reinterpret_cast<f32x4_t*>(dst + 4) = acc2[pt][1];
      }
    }
  } else {
    const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(pr + 3 * C + ch0), b1 = *reinterpret_cast<const f32x4_t*>(pr + 3 * C + ch0 + 4);
#pragma unroll
    for (int pt = 0; pt < PTS; ++pt) {
      const long m = m0 + 16 * pt + p;
      if (m < P.n) {
        const f32x4_t o0 = acc2[pt][0] + b0, o1 = acc2[pt][1] + b1;
        *reinterpret_cast<f32x4_t*>(P.x + m * P.ldx + ch0) = o0;
        *reinterpret_cast<f32x4_t*>(P.x + m * P.ldx + ch0 + 4) = o1;
        if (P.xc) *reinterpret_cast<uint4*>(P.xc + m * P.ldxc + ch0) = pack8(o0, o1);
      }
    }
  }
  DT_STAMP(7);
}

// x = sum_js part[js] + b2 (js in ascending order: the result does not depend on which workgroup finished first), xc = T(x)
__global__ __launch_bounds__(256) void deep_tail_reduce_kernel(const float* __restrict__ part, int nsplit, long n, int C,
                                                               const float* __restrict__ b2, float* x, int ldx, bf16_t* xc,
                                                               int ldxc) {
  const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one 8-channel piece of a row
  const int per_row = C / 8;
  if (u >= n * per_row) return;
  const long m = u / per_row;
  const int ch = (int)(u % per_row) * 8;
  f32x4_t a = *reinterpret_cast<const f32x4_t*>(part + m * C + ch), b = *reinterpret_cast<const f32x4_t*>(part + m * C + ch + 4);
  for (int js = 1; js < nsplit; ++js) {
    const float* src = part + ((size_t)js * n + m) * C + ch;
    a += *reinterpret_cast<const f32x4_t*>(src);
    b += *reinterpret_cast<const f32x4_t*>(src + 4);
  }
  a += *reinterpret_cast<const f32x4_t*>(b2 + ch);
  b += *reinterpret_cast<const f32x4_t*>(b2 + ch + 4);
  *reinterpret_cast<f32x4_t*>(x + m * ldx + ch) = a;
  *reinterpret_cast<f32x4_t*>(x + m * ldx + ch + 4) = b;
  if (xc) *reinterpret_cast<uint4*>(xc + m * ldxc + ch) = pack8(a, b);
}

// ---- weight images.  One product phase: KS steps x 2 fragments per wave.  16-byte unit
// ((w * S + step0 + s) * 2 + f) * 64 + lane,  lane = 16 q + i:
//     W[row_base + 32 w + 8 (i >> 2) + 4 f + (i & 3)][col_base + 32 s + 8 q + 0..7]
// (MFMA A operand: lane (q, i) = row i, k slots 8 q .. 8 q + 7; output row i lands in lane group i >> 2, register i & 3)
struct DeepPackP {
  const bf16_t* w;
  uint4* img;
  int ld, C, S, step0, row_base, col_base;
};

__global__ void deep_pack_kernel(DeepPackP p) {
  const int KS = p.C / 32, NW = p.C / 32;
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= NW * KS * 128) return;
  const int lane = u & 63, f = (u >> 6) & 1, s = (u >> 7) % KS, w = (u >> 7) / KS;
  const int q = lane >> 4, i = lane & 15;
  const long row = p.row_base + 32 * w + 8 * (i >> 2) + 4 * f + (i & 3);
  const long col = p.col_base + 32 * s + 8 * q;
  p.img[((size_t)(w * p.S + p.step0 + s) * 2 + f) * 64 + lane] = *reinterpret_cast<const uint4*>(p.w + row * p.ld + col);
}

int pack_phase(const void* w, int ld, int C, int S, int step0, int row_base, int col_base, void* img, hipStream_t s) {
  DeepPackP p;
  p.w = (const bf16_t*)w; p.img = (uint4*)img; p.ld = ld; p.C = C; p.S = S; p.step0 = step0; p.row_base = row_base;
  p.col_base = col_base;
  const int units = (C / 32) * (C / 32) * 128;
  hipLaunchKernelGGL(deep_pack_kernel, dim3((units + 255) / 256), dim3(256), 0, s, p);
  return hipGetLastError() == hipSuccess ? CDSEG_OK : CDSEG_ERR_LAUNCH;
}

// rows per workgroup: 128 when that still gives most CUs a workgroup, else 32 (single scenes, the deepest levels)
inline int pick_bm(long n) { return (n + 127) / 128 >= 160 ? 128 : 32; }

// Few-row launches (32-row tiles that leave most of the 256 CUs without a workgroup): cut a tile's weight stream over several
// workgroups - the head by q / k / v column block (3), the tail by hidden chunks (4, or 2) - while the grid stays within one
// workgroup per CU, and only at C = 512, whose 16-wave workgroup streams 2 / 4.7 MB per tile.  Measured (profiles/
// r06_deep_split.txt): 778 rows x 512: head 27.7 (separate launches) / 28.1 (one workgroup per tile) -> 19.9 us, tail 40.7 /
// 55.3 -> 28.1 us incl. the reduce launch; at 6224 rows (195 tiles: 585 / 780 workgroups) the split LOSES (34 -> 62, 58 ->
// 82 us), and so it does at C = 256 with 106 tiles (12.3 -> 14.1, 17.8 -> 24.6 us): hence the cap and the channel count.
inline int pick_head_split(long n, int C, int bm) {
  const long tiles = (n + bm - 1) / bm;
  return (bm == 32 && C >= cdseg_knob("CDSEG_DEEP_SPLIT_MIN_C", 512) && tiles * 3 <= cdseg_knob("CDSEG_DEEP_SPLIT_MAX_WGS", 256)) ? 3 : 1;
}
inline int pick_tail_split(long n, int C, int bm, size_t ws_bytes) {
  const long tiles = (n + bm - 1) / bm;
  if (bm != 32 || C < cdseg_knob("CDSEG_DEEP_SPLIT_MIN_C", 512)) return 1;
  const long cap = cdseg_knob("CDSEG_DEEP_SPLIT_MAX_WGS", 256);
  for (int k = 4; k >= 2; k >>= 1)
    if (tiles * k <= cap && (size_t)k * n * C * 4 <= ws_bytes) return k;
  return 1;
}

template <int C, int BM>
int launch_head(const DeepHeadP& p, hipStream_t s) {
  constexpr int lds = DeepCfg<C, BM>::HEAD_LDS;
  static std::atomic<bool> attr_done{false};  // (a concurrent first call sets the attribute twice: harmless)
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)deep_head_kernel<C, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return CDSEG_ERR_LAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL((deep_head_kernel<C, BM>), dim3((unsigned)((p.n + BM - 1) / BM) * (unsigned)p.nsplit), dim3(2 * C), lds, s, p);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

template <int C, int BM>
int launch_tail(const DeepTailP& p, hipStream_t s) {
  constexpr int lds = DeepCfg<C, BM>::TAIL_LDS;
  static std::atomic<bool> attr_done{false};  // (a concurrent first call sets the attribute twice: harmless)
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)deep_tail_kernel<C, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return CDSEG_ERR_LAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL((deep_tail_kernel<C, BM>), dim3((unsigned)((p.n + BM - 1) / BM) * (unsigned)p.nsplit), dim3(2 * C), lds, s, p);
  CDSEG_CHECK_LAUNCH();
  if (p.nsplit > 1) {
    const long units = p.n * (C / 8);
    hipLaunchKernelGGL(deep_tail_reduce_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, p.part, p.nsplit, p.n, C,
                       p.b2, p.x, p.ldx, p.xc, p.ldxc);
    CDSEG_CHECK_LAUNCH();
  }
  return CDSEG_OK;
}

}  // namespace

#ifdef CDSEG_DEEP_TIMING
extern "C" int cdseg_debug_deep_timing(unsigned long long* host_dst, size_t count) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_deep_t), count * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

// C = 512 (round 5: the deepest stage, 16 waves per workgroup, 32-row tiles only - the 1024-thread workgroup has 128 VGPRs
// per wave): a tile streams 2 / 4.7 MB of head / tail weights from L2, which 6 k rows (8 collated scenes: 194 tiles) or 800
// rows (one scene) amortise better than the 13 GEMM / split-K / row-finish launches per Block they replace
bool deep_supported(int channels) { return channels == 128 || channels == 256 || channels == 512; }

int deep_pack(int C, const void* wl, const void* wqkv, void* head_img, const void* wp, const void* w1, const void* w2,
              void* tail_img, hipStream_t s) {
  if (!deep_supported(C)) return CDSEG_ERR_UNSUPPORTED;
  const int KS = C / 32;
  int rc;
  if (head_img) {
    if (!wl || !wqkv) return CDSEG_ERR_ARG;
    if ((rc = pack_phase(wl, C, C, 4 * KS, 0, 0, 0, head_img, s)) != CDSEG_OK) return rc;
    for (int c = 0; c < 3; ++c)
      if ((rc = pack_phase(wqkv, C, C, 4 * KS, KS * (1 + c), c * C, 0, head_img, s)) != CDSEG_OK) return rc;
  }
  if (tail_img) {
    if (!wp || !w1 || !w2) return CDSEG_ERR_ARG;
    if ((rc = pack_phase(wp, C, C, 9 * KS, 0, 0, 0, tail_img, s)) != CDSEG_OK) return rc;
    for (int j = 0; j < 4; ++j) {
      if ((rc = pack_phase(w1, C, C, 9 * KS, KS * (1 + 2 * j), j * C, 0, tail_img, s)) != CDSEG_OK) return rc;
      if ((rc = pack_phase(w2, 4 * C, C, 9 * KS, KS * (2 + 2 * j), 0, j * C, tail_img, s)) != CDSEG_OK) return rc;
    }
  }
  return CDSEG_OK;
}

int deep_head(const void* y, int ldy, const void* head_img, const float* bl, const float* lnp_g, const float* lnp_b,
              const float* x, int ldx, float* x_out, int ldxo, const float* colbias, const float* ln1_g, const float* ln1_b, float eps, const float* bqkv, void* qkv,
              int ldqkv, long n, int channels, int qkv_flags, hipStream_t s, const float* ypart, int ysplits, const float* ybias) {
  DeepHeadP p;
  p.ypart = ypart; p.ybias = ybias; p.ysplits = (ypart && ybias && ysplits > 1) ? ysplits : 1;
  if (p.ysplits > 1 && ((((uintptr_t)ypart) | ((uintptr_t)ybias)) & 15)) return CDSEG_ERR_ARG;
  p.v_bf16 = (qkv_flags & CDSEG_ATTN_V_BF16) ? 1 : 0;
  p.y = (const bf16_t*)y; p.wimg = (const uint4*)head_img; p.bl = bl; p.lnp_g = lnp_g; p.lnp_b = lnp_b; p.x = x;
  p.x_out = x_out; p.ldxo = ldxo;
  p.colbias = colbias; p.ln1_g = ln1_g; p.ln1_b = ln1_b; p.bqkv = bqkv; p.qkv = (bf16_t*)qkv;
  p.n = n; p.ldy = ldy; p.ldx = ldx; p.ldqkv = ldqkv; p.eps = eps;
  const int bm = channels == 512 ? 32 : cdseg_knob("CDSEG_DEEP_BM", pick_bm(n));
  // (in place - x_out == x - a split launch would race: the three workgroups of a tile read the rows one of them rewrites)
  p.nsplit = (cdseg_knob("CDSEG_DEEP_SPLIT", 1) && (const float*)x_out != x) ? pick_head_split(n, channels, bm) : 1;
  if (channels == 128) return bm == 128 ? launch_head<128, 128>(p, s) : launch_head<128, 32>(p, s);
  if (channels == 256) return bm == 128 ? launch_head<256, 128>(p, s) : launch_head<256, 32>(p, s);
  if (channels == 512) return launch_head<512, 32>(p, s);
  return CDSEG_ERR_UNSUPPORTED;
}

int deep_tail(const void* o, int ldo, const void* tail_img, const float* bp, const float* ln_g, const float* ln_b, float eps,
              const float* b1, const float* b2, const float* x_in, int ldxi, float* x, int ldx, void* xc, int ldxc, long n,
              int channels, void* ws, size_t ws_bytes, hipStream_t s) {
  DeepTailP p;
  p.o = (const bf16_t*)o; p.wimg = (const uint4*)tail_img; p.bp = bp; p.ln_g = ln_g; p.ln_b = ln_b; p.b1 = b1; p.b2 = b2;
  p.x_in = x_in; p.ldxi = ldxi;
  p.x = x; p.xc = (bf16_t*)xc; p.n = n; p.ldo = ldo; p.ldx = ldx; p.ldxc = ldxc; p.eps = eps;
  const int bm = channels == 512 ? 32 : cdseg_knob("CDSEG_DEEP_BM", pick_bm(n));
  p.nsplit = (ws && cdseg_knob("CDSEG_DEEP_SPLIT", 1) && !(((uintptr_t)ws) & 15)) ? pick_tail_split(n, channels, bm, ws_bytes) : 1;
  p.part = (float*)ws;
  if (channels == 128) return bm == 128 ? launch_tail<128, 128>(p, s) : launch_tail<128, 32>(p, s);
  if (channels == 256) return bm == 128 ? launch_tail<256, 128>(p, s) : launch_tail<256, 32>(p, s);
  if (channels == 512) return launch_tail<512, 32>(p, s);
  return CDSEG_ERR_UNSUPPORTED;
}
