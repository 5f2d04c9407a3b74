// Serialized window attention for CDSegNet / PTv3 on gfx950 (head dim 16, patch <= 1024).
//
// ref: pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py
//      :246-296 SerializedAttention (flash_attn_varlen_qkvpacked_func :282-288, CPU branch :264-280)
//      :988-1055 SerializedCrossAttention (flash_attn_varlen_kvpacked_func :1038-1047)
//
// One workgroup (8 waves) = one (patch, head) x one slice of its queries.
//  * the gather by serialized order is fused into the K/V staging (bf16: LDS-DMA with per-lane source addresses, no
//    staging registers, no transposing writes) and the Q fragment loads (row indices come from the slot plan,
//    cdseg_pad_plan); the scatter by the inverse order and the dropping of the padding duplicates are fused into
//    the store;
//  * the whole K and V tile of the patch-head live in LDS for the lifetime of the block
//    (bf16: 32 KB + 32 KB row-major -> 2 blocks / CU; f32: 64 KB + 65 KB, V transposed);
//  * scores are computed TRANSPOSED (S^T = K Q^T) so that a query's scores stay inside one lane
//    pair: row statistics need no cross-lane traffic in the key loop;
//  * bf16: ONE pass over the keys, and since round 6 without any shift: softmax is shift invariant and fp32 / bfloat16
//    carry 8 exponent bits, so P~ = exp2(s') (scale * log2(e) folded into Q' by the producer) needs neither the row max nor a
//    bound as long as nothing leaves fp32's range; the MFMA's C operand is the inline constant 0, a score costs ONE v_exp_f32
//    and the query tile's set-up is a row fetch.  The denominator (row 16 of the PV product) tells whether that was
//    legitimate: rows whose denominator left [1e-30, 1e30] are redone with the exact row max (rounds 2 - 5 subtracted the
//    Cauchy-Schwarz bound |q'_i| max_j |k_j|: a norm + sqrt + sixteen shift registers per query tile and a max-|k| pass over
//    the staged keys per block - 4.5 - 5 % of the kernel, profiles/r06_attention_unshifted_sweep.txt);
//  * the kernel is bound by the VALU / transcendental issue, not by the matrix pipe: 16 v_exp_f32 + 8
//    v_cvt_pk_bf16_f32 per 32x32 score tile against 3 MFMAs of 32 cycles (tools/ubench/pipes.hip,
//    profiles/r03_ubench_pipes.txt: 128 cycles per tile and SIMD at 4 waves per SIMD = half the MFMA-only rate - and with
//    both pipes busy the chip holds only ~1.5 GHz: 82-87 ns per tile in wall time, with or without the real data
//    dependences, interleaved or grouped; the key loop alone runs at 91 ns, the whole kernel at 134-145 ns);
//  * bf16: v_mfma_f32_32x32x16_bf16 for both products; the MFMA k-slot <-> key assignment of
//    the PV product is chosen so the exponentiated scores feed it straight from the
//    accumulator registers (no permute); the matching V^T operand is two ds_read_b64_tr_b16 of the row-major V;
//    a row of ones (constant LDS page) makes the softmax denominator fall out of the same MFMA (row 16 of the result);
//  * f32 (the 1e-3 parity mode): v_mfma_f32_16x16x4_f32 for both products, exact fp32.
// Measured and dropped in round 3 (tools/attn_timing.py stamps, tools/ubench/attn_loop.hip): a persistent
// 16-wave workgroup per CU with double-buffered K/V and one barrier per patch-head (the SIMD issues its waves oldest
// first: the four waves of a SIMD advance at 1 : 0.7 : 0.5 : 0.45 and 30 % of a wave's life went into the barrier),
// the same with progress-ranked s_setprio (in step, but 216 instead of ~140 cycles per tile and SIMD), and a
// barrier-free dataflow form (tasks claimed by LDS compare-and-swap, stage recycled by the last finisher: 261 us
// against 233 us for this form at 960k points x 2 heads).
// Round 5: the producer-side flags of cdseg_attention_ex (q pre-scaled by folded weights, v written as bfloat16 by the fused
// qkv epilogues), one 16-byte output store per lane, the graded launch schedule (decode_block / plan_zones below), and the
// block shapes that lost to this one (10 / 12 waves, one score tile in flight: profiles/r05_attention_waves.txt).
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "common.h"
#include "prof.h"

namespace {

constexpr int ATTN_ZONES = 5;
constexpr int ATTN_STORE8 = 1 << 30;  // internal flag (AttnP::flags): output rows are 8- but not 16-byte aligned

struct AttnP {
  const void* q;
  const void* k;
  const void* v;
  const int32_t* q_gidx;
  const int32_t* kv_gidx;
  const int32_t* widx;
  const int32_t* patch_start;
  void* out;
  int ldq, ldk, ldv, ldo;
  int num_heads;
  int num_patches;
  float scale_log2e;
  int flags;  // CDSEG_ATTN_Q_PRESCALED | CDSEG_ATTN_V_BF16
  // schedule: the (patch, head) units of an XCD, in launch order, fall into zones - zcnt[k] units (-1: whatever the XCD
  // has left = the bulk zone) cut into zsplit[k] query slices each
  int nzones;
  int zcnt[ATTN_ZONES];
  int zsplit[ATTN_ZONES];
  int full_patches;  // host only: the launch's longest patch has all 32 query tiles (zones are for those launches)
  int zfixed;    // sum of the fixed zone sizes
  int bulk_max;  // largest bulk zone over the XCDs (the grid's extent of that zone)
};

// XCD-aware block -> (patch, head, query slice) map.  Workgroups are dispatched round-robin over the 8 XCDs (block b ->
// XCD b % 8, a performance-only assumption).  The U = P * H (patch, head) units are numbered patch-major and XCD x owns
// the CONTIGUOUS run [U x / 8, U (x + 1) / 8): the heads of a patch read the same gathered rows (a 192-byte qkv row at
// C = 32 holds both heads), and patches that follow each other on the curve are neighbours in space whose rows interleave
// in memory, so neighbouring units belong on the same L2, adjacent in time - with the patches dealt round-robin the
// shared lines were fetched by two XCDs (1.27x the algorithmic HBM bytes at C = 32).  At most 7 patches per launch
// straddle two XCDs.  Surplus block ids exit.
//
// Graded schedule (round 5).  A launch of equal blocks on S block slots ends with a round in which the slots drain: the
// in-kernel stamps show a 4096-wave chip 80 % occupied over a 1876-block launch and 9 % over its last tenth
// (profiles/r03_attention_timing.txt, r05_attention_sched.txt).  So the blocks an XCD runs LAST (and, optionally, some of
// its first ones) cover fewer queries: zones, in launch order, of (units, slices per unit); a slice stages the whole
// K / V of its patch-head and takes every `split`-th run of 8 query tiles.  The host sizes the zones (plan_zones);
// a block's `qsplit` is zone-dependent and wave-uniform.  Results do not depend on the schedule (every output row is
// written by exactly one lane pair, with the same arithmetic).
__host__ __device__ __forceinline__ bool decode_block(const AttnP& p, int B, int& patch, int& head, int& qslice, int& qsplit) {
  const int U = p.num_patches * p.num_heads;
  const int xcd = B & 7;
  int t = B >> 3;
  const int u0 = (int)(((long)U * xcd) >> 3), u1 = (int)(((long)U * (xcd + 1)) >> 3);
  int j0 = 0;
  for (int z = 0; z < p.nzones; ++z) {
    const int fixed = p.zcnt[z];
    const int cnt = fixed >= 0 ? fixed : (u1 - u0) - p.zfixed;  // this XCD's units in the zone
    const int ext = fixed >= 0 ? fixed : p.bulk_max;            // the grid's extent of the zone
    const int split = p.zsplit[z];
    const int nb = ext * split;
    if (t < nb) {
      const int jl = t / split;
      if (jl >= cnt) return false;
      const int u = u0 + j0 + jl;
      patch = u / p.num_heads;
      head = u - patch * p.num_heads;
      qslice = t - jl * split;
      qsplit = split;
      return true;
    }
    t -= nb;
    j0 += cnt;
  }
  return false;
}

constexpr int VT_STRIDE_F32 = 4128;  // 1024 f32 + 32 B pad
constexpr int KS_BYTES_F32 = 1024 * 64;
constexpr int SMEM_F32 = KS_BYTES_F32 + 16 * VT_STRIDE_F32;

// ------------------------------------------------------------------------------------ bf16
// LDS image of one patch-head: K and V both ROW-MAJOR [key][16 dims] bf16 (32 B per key), exactly the bytes of the
// gathered qkv rows - so the staging is LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 32 keys per instruction,
// the gather is the per-lane SOURCE address, the LDS side is lane-linear) and costs the wave ~3 VALU per 32 keys
// instead of the load -> transpose -> ds_write prologue the kernel used to spend a third of its time in.
//  * K: the two 16-B halves of a key are swapped for keys with bit 3 set (source-side swizzle): the QK^T A-operand
//    read (lane = key, 16 B) is then bank-conflict free;
//  * V: the PV A operand is V^T, read with ds_read_b64_tr_b16 (gfx950 transposing LDS read: a 16-lane group reads a
//    [4 keys][16 dims] block and lane i gets column i = 4 keys of head dim i).  The 32x32x16 MFMA wants rows
//    (= head dims) in lanes 0..31 of each half wave, so lane groups 0 / 2 (head dims 0..15 of k-slot halves 0 / 1)
//    read V and lane groups 1 / 3 (MFMA rows 16..31) read a constant "ones page": row 16 comes out as all ones, so
//    row 16 of O^T is the softmax denominator, accumulated from the SAME bf16-rounded probabilities as the numerator.
#ifndef ATTN_FIRST_STATIC
#define ATTN_FIRST_STATIC 1
#endif
#ifndef ATTN_PRIO_OUTSIDE
#define ATTN_PRIO_OUTSIDE 1
#endif
#ifndef ATTN_STORE16
#define ATTN_STORE16 1
#endif
// graded schedule: blocks per XCD in the lead / tail zones (plan_zones; 0 = zone off)
#ifndef ATTN_LEAD_BLOCKS
#define ATTN_LEAD_BLOCKS 0
#endif
#ifndef ATTN_TAIL1_BLOCKS
#define ATTN_TAIL1_BLOCKS 64
#endif
#ifndef ATTN_TAIL2_BLOCKS
#define ATTN_TAIL2_BLOCKS 64
#endif
// ... and only for launches of at least this many (patch, head) units per XCD (1.5 rounds of its 64 block slots): below,
// the chip is not full to begin with and every extra slice is a K / V staging that buys nothing (measured -10 .. -25 % on
// the deep stages and on single scenes, profiles/r05_attention_sched.txt)
#ifndef ATTN_ZONE_MIN_UNITS
#define ATTN_ZONE_MIN_UNITS 96
#endif
constexpr int ATTN_THREADS = 512;  // fp32 kernel: 8 waves, one block per CU (LDS 130 KB)
constexpr int ATTN_WAVES = ATTN_THREADS / 64;
constexpr int ATTN_RUN = 8;        // query tiles per run: the unit slices of a patch-head are dealt in, and the number of
                                   // waves that get a pre-assigned first tile
// 16-bit kernel: waves per block (2 blocks per CU, LDS 66 KB each) and score tiles in flight per wave.  8 x 2: four waves
// per SIMD, 116 VGPRs.  10 x 1 (round 5): five waves per SIMD on 96 VGPRs - when one wave of a SIMD stages, normalises or
// claims, four are still in their key loops, and one tile in flight per wave is all the instruction mix needs from four
// waves up (tools/ubench/pipes.hip: 128 cycles per tile and SIMD with the real dependences at W = 4)
#ifndef ATTN_BF16_WAVES
#define ATTN_BF16_WAVES 8
#endif
#ifndef ATTN_TILES_IN_FLIGHT
#define ATTN_TILES_IN_FLIGHT 2
#endif
constexpr int BF_WAVES = ATTN_BF16_WAVES, BF_THREADS = 64 * BF_WAVES;
static_assert(BF_WAVES >= ATTN_RUN && BF_WAVES <= 12, "the first run of a block's tiles is pre-assigned to waves 0 .. 7");
constexpr int BF_WAVES_WIDE = 16;  // the one-block-per-CU shape of under-filled launches (attn_bf16_kernel<16>)
constexpr int KV_STAGE = 1024 * 32;  // K (or V) of one patch-head
constexpr int ONES_BYTES = 2048;     // 8-byte words {1.0bf16, 0, 0, 0}: covers every immediate offset of a tile pair
constexpr int SMEM_BF16 = 2 * KV_STAGE + ONES_BYTES + 64 + 4096;  // + slot -> query row table

__device__ uint4 g_attn_zero[2];  // DMA source of the key slots past the end of a ragged patch

#ifdef CDSEG_ATTN_TIMING
// experimental builds only (tools/_ab): per wave {realtime at entry, realtime at exit, cycles entry -> staging barrier,
// cycles in the key loops, cycles entry -> exit, 0, 0, 0}
__device__ unsigned long long g_attn_t[8 * 8 * 4096];
#define ATTN_STAMP(x) const unsigned long long x = __builtin_readcyclecounter()
#else
#define ATTN_STAMP(x)
#endif

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

// one 32-key x 32-query tile of S^T = K Q'^T + C  (Q' = Q * scale * log2 e, C = -bound: see below)
__device__ __forceinline__ f32x16_t qk_tile(const char* k_lane, bf16x8_t qf, const f32x16_t& c0) {
  const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(k_lane);
  return mfma_32x32x16_bf16(kf, qf, c0);
}

__device__ __forceinline__ float tile_max(const f32x16_t& s, float m) {
  const float a = fmaxf(fmaxf(s[0], s[1]), s[2]);
  const float b = fmaxf(fmaxf(s[3], s[4]), s[5]);
  const float c = fmaxf(fmaxf(s[6], s[7]), s[8]);
  const float d = fmaxf(fmaxf(s[9], s[10]), s[11]);
  const float e = fmaxf(fmaxf(s[12], s[13]), s[14]);
  return fmaxf(fmaxf(fmaxf(a, b), fmaxf(c, d)), fmaxf(fmaxf(e, s[15]), m));
}

// P = exp2(S') for one tile (S' already holds s*c - m*c: scale folded into Q', -m into the MFMA's C operand, so the
// softmax costs one v_exp_f32 per score and nothing else), rounded to bf16 (v_cvt_pk_bf16_f32, round to nearest
// even), then O^T += [V^T; 1; 0] P^T (two K=16 MFMAs).  The S^T accumulator registers of lane (q, h) hold keys
// 4h + (r & 3) + 8 (r >> 2): registers 8 mf .. 8 mf + 7 are the k-slots 8h .. 8h+7 of PV MFMA mf = keys
// 16 mf + 4h + {0..3} and 16 mf + 8 + 4h + {0..3} - two transposing reads of 4 consecutive keys each.
// `va`: the lane's LDS byte address for this tile (V rows of keys 4h + (lane & 15) / 4 .., or the ones page).
template <bool TAIL, bool FIRST = false>
__device__ __forceinline__ void pv_tile(const f32x16_t& s, int kt, int h, int L, unsigned va, f32x16_t& o) {
  float pr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) pr[r] = __builtin_amdgcn_exp2f(s[r]);
  if (TAIL) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= L) pr[r] = 0.f;
  }
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    union { bf16x8_t v; uint32_t u[4]; } pf;
#pragma unroll
    for (int j = 0; j < 4; ++j) pf.u[j] = pack_truebf16x2(pr[8 * mf + 2 * j], pr[8 * mf + 2 * j + 1]);
    const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4_t*>(va + 512 * mf));
    const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4_t*>(va + 512 * mf + 256));
    const bf16x8_t vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    if (FIRST && mf == 0) {  // the pass's first product: C = the inline constant 0, no accumulator set-up (16 v_mov per query tile)
      const f32x16_t z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      o = mfma_32x32x16_truebf16(vf, pf.v, z);
    } else {
      o = mfma_32x32x16_truebf16(vf, pf.v, o);
    }
  }
}

__device__ __forceinline__ float sq8_bf16(const uint4& a) {
  float t = dot2_bf16(a.x, a.x, 0.f);
  t = dot2_bf16(a.y, a.y, t);
  t = dot2_bf16(a.z, a.z, t);
  return dot2_bf16(a.w, a.w, t);
}

// WAVES = 8: two blocks per CU (the product's shape for every launch that fills the chip).  WAVES = 16 (round 6): ONE block per
// CU with four waves per SIMD behind a single K / V staging - for launches with at most one block per CU (a single scene's
// stages 0 - 2: 236 / 216 / 2 x 112 blocks), where the 8-wave block left every SIMD with two waves and a second slice of the same
// patch-head on the same CU paid a second staging (make_schedule picks; same arithmetic per output row, results identical).
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 4) void attn_bf16_kernel(AttnP p) {
  constexpr int BF_WAVES = WAVES, BF_THREADS = 64 * WAVES;
  constexpr int NPC = 32 / WAVES;  // 32-key pieces of K (and of V) a wave stages
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  unsigned* s_next = reinterpret_cast<unsigned*>(smem + 2 * KV_STAGE + ONES_BYTES + 48);  // query tiles handed out so far
  int* s_qidx = reinterpret_cast<int*>(smem + 2 * KV_STAGE + ONES_BYTES + 64);           // slot -> query row
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int patch, head, qslice, qsplit;
  if (!decode_block(p, (int)blockIdx.x, patch, head, qslice, qsplit)) return;
#ifdef CDSEG_ATTN_TIMING
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t_loop = 0, t_first = 0, t_epi = 0, n_tiles = 0;
#endif
  ATTN_STAMP(t0);
  // the staging is a few hundred instructions of a young block next to the key loops of an older one: at the default
  // priority it waits behind them for every issue slot (in-kernel stamps: 29k cycles from entry to the staging barrier,
  // 10k with the priority raised)
  __builtin_amdgcn_s_setprio(3);
  const int ps = p.patch_start[patch];
  const int L = p.patch_start[patch + 1] - ps;
  const int nkt = (L + 31) >> 5;  // 32-key tiles
  const bf16_t* kb = (const bf16_t*)p.k + head * 16;
  const bf16_t* vb = (const bf16_t*)p.v + head * 16;

  const int ql = lane & 31;  // query (B operand column) / key or head-dim row (A operand row)
  const int h = lane >> 5;
  const int nqt = nkt;
  // Query tiles are handed out dynamically (LDS counter): the SIMD issues its waves oldest first, so the waves of a
  // block advance at very different rates (2 : 1 and more); with a static split the fast waves exit early and their
  // slots stay empty until the block's slowest wave is done (the next block needs all eight).  A wave claims its next
  // tile when it starts the current one and fetches that tile's query rows (row index from LDS) behind the key loop.
  auto claim = [&]() {  // next query tile of this block's slice, -1: none left
    for (;;) {
      unsigned i = 0;
      if (lane == 0) i = atomicAdd(s_next, 1u);
      i = __builtin_amdgcn_readfirstlane(i);
      if (ATTN_FIRST_STATIC && qsplit == 1) {  // tiles 4w are pre-assigned (below): hand out the others, in order
        const int t = (int)(i + i / 3u + 1u);
        return t < nqt ? t : -1;
      }
      // a slice takes every qsplit-th run of 8 query tiles; the tiles of its first run are pre-assigned (run base + wave,
      // below: the counter starts at 8)
      const int base = (qslice + (int)(i >> 3) * qsplit) * ATTN_RUN;
      if (base >= nqt) return -1;
      const int t = base + (int)(i & 7);
      if (t < nqt) return t;
    }
  };
  auto load_qrow = [&](int t) {  // (unconditional: a wave without a next tile re-reads tile 0 and drops it)
    const int g = s_qidx[min(max(t, 0) * 32 + ql, L - 1)];
    return *reinterpret_cast<const uint4*>((const bf16_t*)p.q + (long)g * p.ldq + head * 16 + h * 8);
  };

  int qt_first = -1;
  uint4 q_first = make_uint4(0, 0, 0, 0);
  // ---- stage K and V by LDS-DMA.  Wave w moves the 32-key pieces w, w + 8, ..: lane -> (key = lane / 2, 16-B half).
  {
    const int kip = lane >> 1, hs = lane & 1;
    int gk[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int s = (wave + BF_WAVES * i) * 32 + kip;
      gk[i] = s < L ? p.kv_gidx[ps + s] : -1;
    }
    // slot -> query row table (this wave's 128 slots), for the query-row fetches of the tile loop
    int gq[2] = {0, 0};
    if (wave < ATTN_RUN) {  // (1024 slots: 128 per wave of the first eight)
#pragma unroll
      for (int i = 0; i < 2; ++i) gq[i] = p.q_gidx[ps + min(wave * 128 + i * 64 + lane, L - 1)];
    }
    // the wave's FIRST query tile is fixed: tile 4 w of an unsliced patch-head (its rows are the first 32 of the indices
    // just loaded), tile `first run of the slice` + w of a slice (its row indices are fetched here, with the others)
    const int qt_slice0 = qslice * ATTN_RUN + wave;
    int gfirst = 0;
    if (ATTN_FIRST_STATIC && qsplit > 1 && wave < ATTN_RUN) gfirst = p.q_gidx[ps + min(qt_slice0 * 32 + ql, L - 1)];
    if (tid == 0) *s_next = (ATTN_FIRST_STATIC && qsplit > 1) ? 8u : 0u;
    // (the compiler waits for its own loads above; the DMAs are invisible to it and are waited for by hand below)
    // all source addresses first (pinned by the empty asm): the compiler's vmcnt(0) for an index load must not sit
    // between two DMAs, where it would wait for the DMA before it as well
    const void* ksrc[NPC];
    const void* vsrc[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int s = (wave + BF_WAVES * i) * 32 + kip;
      ksrc[i] = vsrc[i] = (const char*)g_attn_zero + hs * 16;
      if (gk[i] >= 0) {
        ksrc[i] = kb + (long)gk[i] * p.ldk + ((hs ^ ((s >> 3) & 1)) << 3);
        vsrc[i] = vb + (long)gk[i] * p.ldv + (hs << 3);
      }
      asm volatile("" : "+v"(ksrc[i]), "+v"(vsrc[i]));
    }
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int pc = wave + BF_WAVES * i;
      if (pc < nkt) {
        dma16(ksrc[i], lds_base + pc * 1024);
        dma16(vsrc[i], lds_base + KV_STAGE + pc * 1024);
      }
    }
    // the wave's FIRST query tile is fixed (tile 4 w: its rows are the first 32 of the indices just loaded) and its
    // query rows are fetched here, next to the K / V DMAs: claimed and fetched after the barrier, they were an exposed
    // index -> row load chain issued at key-loop priority behind the older waves (in-kernel stamps: 14.5 % of a wave's
    // life between the barrier and its first key loop)
    if (ATTN_FIRST_STATIC && qsplit == 1 && 4 * wave < nqt) {
      qt_first = 4 * wave;
      const int g = __shfl(gq[0], lane & 31, 64);
      q_first = *reinterpret_cast<const uint4*>((const bf16_t*)p.q + (long)g * p.ldq + head * 16 + h * 8);
    } else if (ATTN_FIRST_STATIC && qsplit > 1 && wave < ATTN_RUN && qt_slice0 < nqt) {
      qt_first = qt_slice0;
      q_first = *reinterpret_cast<const uint4*>((const bf16_t*)p.q + (long)gfirst * p.ldq + head * 16 + h * 8);
    }
    if (wave < ATTN_RUN) {
#pragma unroll
      for (int i = 0; i < 2; ++i) s_qidx[wave * 128 + i * 64 + lane] = gq[i];
    }
    for (int w = tid; w < ONES_BYTES / 8; w += BF_THREADS)
      *reinterpret_cast<uint2*>(smem + 2 * KV_STAGE + w * 8) = make_uint2(0x3F80u, 0u);
    // half build, V in half (a producer that writes V as bfloat16 sets CDSEG_ATTN_V_BF16 and this pass is skipped): Q and K
    // stay half (the scores keep 11-bit operands), the P V product runs in bfloat16 - P = exp2(s) needs fp32's exponent range
    // (half underflows at 2^-24) - so the lane rewrites the 8 V values it fetched itself as bfloat16, in place.  Its own
    // vmcnt(0) is all the ordering its own pieces need
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (LP_IS_F16 && !(p.flags & CDSEG_ATTN_V_BF16)) {
#pragma unroll
      for (int i = 0; i < NPC; ++i) {
        const int pc = wave + BF_WAVES * i;
        if (pc < nkt) {
          uint4* vp = reinterpret_cast<uint4*>(smem + KV_STAGE + pc * 1024 + lane * 16);
          uint4 u = *vp;
          float a0, a1;
          unpack_bf16x2(u.x, a0, a1); u.x = pack_truebf16x2(a0, a1);
          unpack_bf16x2(u.y, a0, a1); u.y = pack_truebf16x2(a0, a1);
          unpack_bf16x2(u.z, a0, a1); u.z = pack_truebf16x2(a0, a1);
          unpack_bf16x2(u.w, a0, a1); u.w = pack_truebf16x2(a0, a1);
          *vp = u;
        }
      }
    }
    __syncthreads();  // everybody's DMA has landed, the ones page is written
  }
  if (!ATTN_PRIO_OUTSIDE) __builtin_amdgcn_s_setprio(0);
  ATTN_STAMP(t1);
  // lane constants of the key loop
  const char* k_lane = Ks + ql * 32 + ((h ^ ((ql >> 3) & 1)) << 4);
  const bool v_lane = (lane & 16) == 0;  // lane groups 0 / 2 read V, 1 / 3 the ones page
  // half wave 0: V rows on banks 0..31 -> its ones word on banks 32..63, and the other way round for half wave 1
  const unsigned va0 = v_lane ? lds_base + KV_STAGE + (4 * h + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8
                              : lds_base + 2 * KV_STAGE + (h ? 0 : 128);
  const unsigned vstep = v_lane ? 1024u : 0u;
  const float c = p.scale_log2e;
  const bool tail = (nkt << 5) != L;
  const int nfull = tail ? nkt - 1 : nkt;  // key tiles that need no masking

  int qt = qt_first;
  uint4 q_cur = q_first;
  if (qt < 0) {
    qt = claim();
    q_cur = load_qrow(qt);
  }
  while (qt >= 0) {
    // Q' = Q * (softmax scale * log2 e), rounded to bf16 once: scores come out of the MFMA in exp2 units (consumed
    // BEFORE the next loads are issued: the compiler's wait for q_cur then has nothing younger in the queue)
    // (CDSEG_ATTN_Q_PRESCALED: the producer's weights already carry that factor - Engine.prepare folds it into Wq / bq -
    // and the fetched row IS the operand)
    bf16x8_t qf;
    {
      union { bf16x8_t v; uint32_t u[4]; } qs;
      const uint32_t qr[4] = {q_cur.x, q_cur.y, q_cur.z, q_cur.w};
      if (p.flags & CDSEG_ATTN_Q_PRESCALED) {
#pragma unroll
        for (int j = 0; j < 4; ++j) qs.u[j] = qr[j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float lo, hi;
          unpack_bf16x2(qr[j], lo, hi);
          qs.u[j] = pack_bf16x2(lo * c, hi * c);
        }
      }
      qf = qs.v;
      asm volatile("" : "+v"(qf));
    }
    const int qt_nxt = claim();
    const uint4 q_nxt = load_qrow(qt_nxt);
    const int qslot = qt * 32 + ql;
    const bool qvalid = qslot < L;
    int w = p.widx[ps + min(qslot, L - 1)];
    if (!qvalid) w = -1;
    const f32x16_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // P = exp2(S' - m), O^T (+ row sums in row 16) = [V^T; 1; 0] P^T, with S' - m straight out of the MFMA (C operand = -m,
    // loop invariant; SHIFT = false: C is the inline constant 0 and needs no registers).  The pass's first product starts
    // the accumulator from the constant 0 as well.
    auto exp_pv_pass = [&](auto shift_tag, float mrow) {
      constexpr bool SHIFT = decltype(shift_tag)::value;
      const float nm = SHIFT ? -mrow : 0.f;
      const f32x16_t negm = {nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm};
      const f32x16_t& c0 = SHIFT ? negm : zero16;
      f32x16_t acc;
      const char* kp = k_lane;
      unsigned va = va0;
      int kt;
      if (ATTN_TILES_IN_FLIGHT == 2 && nfull >= 2) {
        const f32x16_t sa = qk_tile(kp, qf, c0);
        const f32x16_t sb = qk_tile(kp + 1024, qf, c0);
        pv_tile<false, true>(sa, 0, h, L, va, acc);
        pv_tile<false>(sb, 1, h, L, va + 1024, acc);
        kp += 2048;
        va += 2 * vstep;
        kt = 2;
      } else {
        const f32x16_t s0 = qk_tile(kp, qf, c0);
        if (nfull == 0) pv_tile<true, true>(s0, 0, h, L, va, acc);
        else pv_tile<false, true>(s0, 0, h, L, va, acc);
        kp += 1024;
        va += vstep;
        kt = 1;
      }
      for (; ATTN_TILES_IN_FLIGHT == 2 && kt + 1 < nfull; kt += 2) {  // two independent tiles in flight
        const f32x16_t sa = qk_tile(kp, qf, c0);
        const f32x16_t sb = qk_tile(kp + 1024, qf, c0);
        pv_tile<false>(sa, kt, h, L, va, acc);
        pv_tile<false>(sb, kt + 1, h, L, va + 1024, acc);
        kp += 2048;
        va += 2 * vstep;
      }
      for (; kt < nkt; ++kt) {
        const f32x16_t s = qk_tile(kp, qf, c0);
        if (kt >= nfull) pv_tile<true>(s, kt, h, L, va, acc);
        else pv_tile<false>(s, kt, h, L, va, acc);
        kp += 1024;
        va += vstep;
      }
      return acc;
    };
    // ---- single pass WITHOUT a shift (round 6).  Softmax is shift invariant and fp32 / bfloat16 carry 8 exponent bits:
    // P~ = exp2(s') needs no row maximum and no bound as long as nothing leaves fp32's range - scores (in exp2 units) up to
    // +-100, i.e. +-69 in the reference's natural-log units, which no softmax that is not already one-hot produces.  The pass
    // therefore runs with C = 0 (an inline constant: no |q| norm, no sqrt, no max |k| pass over the staged keys, no sixteen
    // shift registers per query tile - rounds 2 - 5 subtracted the Cauchy-Schwarz bound |q'| max |k|), and the denominator
    // that falls out of the same MFMA says whether it was legitimate: a row whose denominator left [1e-30, 1e30] (an exp2 that
    // overflowed shows up as inf / NaN there, a row of underflows as 0) is redone with the exact row maximum below
    // (wave-uniform branch; tests/test_gpu_ops.py::test_attention_softmax_is_shift_safe drives it).  Relative precision is
    // that of the shifted form: the terms only differ by a power of two.
    // everything outside the key loops (prologue, normalise + store, the next tile's set-up) runs at raised priority:
    // it is a few hundred instructions that otherwise queue behind the older waves' key loops on the same SIMD
    if (ATTN_PRIO_OUTSIDE) __builtin_amdgcn_s_setprio(0);
    ATTN_STAMP(tl0);
    f32x16_t o = exp_pv_pass(std::false_type{}, 0.f);
    if (ATTN_PRIO_OUTSIDE) __builtin_amdgcn_s_setprio(2);
#ifdef CDSEG_ATTN_TIMING
    asm volatile("" :: "v"(o[0]), "v"(o[8]));
    const unsigned long long tl1 = __builtin_readcyclecounter();
    t_loop += tl1 - tl0;
    if (n_tiles == 0) t_first = tl0 - t1;
    ++n_tiles;
#endif
    {
      const float den = __shfl(o[8], ql, 64);
      const bool redo = qvalid && !(den >= 1e-30f && den <= 1e30f);
      if (__any(redo)) {
        // ---- exact pass 1: row max of S'^T = K Q'^T (lane (q,h) sees keys (r&3) + 8*(r>>2) + 4h of each tile)
        float m0 = -INFINITY;
        const char* kp = k_lane;
        for (int kt = 0; kt < nkt; ++kt, kp += 1024) {
          f32x16_t s = qk_tile(kp, qf, zero16);
          if (kt >= nfull) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= L) s[r] = -INFINITY;
          }
          m0 = tile_max(s, m0);
        }
        m0 = fmaxf(m0, __shfl_xor(m0, 32, 64));
        o = exp_pv_pass(std::true_type{}, m0);
      }
    }
    // ---- epilogue: O^T rows (r&3) + 8*(r>>2) + 4h; row 16 (lane h=0, r=8) is the denominator
    const float lsum = __shfl(o[8], ql, 64);
    const float inv = 1.0f / lsum;
    {
      uint2 a, b;  // a: d = 4h .. 4h+3, b: d = 8+4h .. 8+4h+3 of query ql
      a.x = pack_bf16x2(o[0] * inv, o[1] * inv);
      a.y = pack_bf16x2(o[2] * inv, o[3] * inv);
      b.x = pack_bf16x2(o[4] * inv, o[5] * inv);
      b.y = pack_bf16x2(o[6] * inv, o[7] * inv);
      if (ATTN_STORE16 && !(p.flags & ATTN_STORE8)) {
        // one 16-byte store per lane instead of two 8-byte ones: lanes q and q + 32 hold the two halves of each 8-dim run
        // of query q; v_permlane32_swap exchanges a[32..63] with b[0..31], after which lane q holds d = 0..7 and lane
        // q + 32 holds d = 8..15 (a = first, b = second half of the run)
        const auto sx = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
        if (w >= 0) {
          bf16_t* orow = (bf16_t*)p.out + (long)w * p.ldo + head * 16 + 8 * h;
          *reinterpret_cast<uint4*>(orow) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
      } else if (w >= 0) {  // output rows that are only 8-byte aligned (ldo % 8 != 0): the two 8-byte pieces the lane holds
        bf16_t* orow = (bf16_t*)p.out + (long)w * p.ldo + head * 16 + 4 * h;
        *reinterpret_cast<uint2*>(orow) = a;      // d = 4h .. 4h+3
        *reinterpret_cast<uint2*>(orow + 8) = b;  // d = 8+4h .. 8+4h+3
      }
    }
    qt = qt_nxt;
    q_cur = q_nxt;
#ifdef CDSEG_ATTN_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (timing builds: the stores and the next rows inside the stamp)
    t_epi += __builtin_readcyclecounter() - tl1;
#endif
  }
#ifdef CDSEG_ATTN_TIMING
  if (lane == 0 && blockIdx.x < 4096) {
    unsigned long long* d = g_attn_t + ((size_t)blockIdx.x * 8 + wave) * 8;
    d[0] = rt0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = t1 - t0; d[3] = t_loop;
    d[4] = __builtin_readcyclecounter() - t0;
    d[5] = t_first; d[6] = t_epi; d[7] = n_tiles;
  }
#endif
}

// ------------------------------------------------------------------------------------ f32
__global__ __launch_bounds__(ATTN_THREADS) void attn_f32_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vt = smem + KS_BYTES_F32;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  int patch, head, qslice, qsplit;
  if (!decode_block(p, (int)blockIdx.x, patch, head, qslice, qsplit)) return;
  const int ps = p.patch_start[patch];
  const int L = p.patch_start[patch + 1] - ps;
  const int nkt = (L + 15) >> 4;  // 16-key tiles
  const int Lp = nkt << 4;
  const float* kb = (const float*)p.k + head * 16;
  const float* vb = (const float*)p.v + head * 16;

  for (int s = tid; s < Lp; s += ATTN_THREADS) {
    uint4 kk[4], vv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) kk[i] = vv[i] = make_uint4(0, 0, 0, 0);
    if (s < L) {
      const long g = p.kv_gidx[ps + s];
      const uint4* kr = reinterpret_cast<const uint4*>(kb + g * p.ldk);
      const uint4* vr = reinterpret_cast<const uint4*>(vb + g * p.ldv);
#pragma unroll
      for (int i = 0; i < 4; ++i) { kk[i] = kr[i]; vv[i] = vr[i]; }
    }
    const int sw = (s >> 1) & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(Ks + s * 64 + ((i ^ sw) << 4)) = kk[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint32_t*>(Vt + (4 * i + 0) * VT_STRIDE_F32 + s * 4) = vv[i].x;
      *reinterpret_cast<uint32_t*>(Vt + (4 * i + 1) * VT_STRIDE_F32 + s * 4) = vv[i].y;
      *reinterpret_cast<uint32_t*>(Vt + (4 * i + 2) * VT_STRIDE_F32 + s * 4) = vv[i].z;
      *reinterpret_cast<uint32_t*>(Vt + (4 * i + 3) * VT_STRIDE_F32 + s * 4) = vv[i].w;
    }
  }
  __syncthreads();

  const int ql = lane & 15;
  const int g4 = lane >> 4;
  const float c = p.scale_log2e;
  const int nqt = (L + 15) >> 4;

  for (int qt = qslice * ATTN_WAVES + wave; qt < nqt; qt += qsplit * ATTN_WAVES) {
    const int qslot = qt * 16 + ql;
    const bool qvalid = qslot < L;
    f32x4_t qf = {0.f, 0.f, 0.f, 0.f};
    if (qvalid) {
      const long g = p.q_gidx[ps + qslot];
      qf = *reinterpret_cast<const f32x4_t*>((const float*)p.q + g * p.ldq + head * 16 + 4 * g4);
    }
    // S^T tile (16 keys x 16 queries): k-slot g4 of step ss <-> head dim 4*g4 + ss.
    // C layout: col = query = lane & 15, row = key offset = 4*g4 + r.
    float mloc = -INFINITY;
    for (int kt = 0; kt < nkt; ++kt) {
      const int key = kt * 16 + ql;
      const f32x4_t kf = *reinterpret_cast<const f32x4_t*>(Ks + key * 64 + ((g4 ^ ((key >> 1) & 3)) << 4));
      f32x4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[ss], qf[ss], s, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sv = s[r];
        if (kt * 16 + 4 * g4 + r >= L) sv = -INFINITY;
        mloc = fmaxf(mloc, sv);
      }
    }
    float m = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float mc = m * c;
    f32x4_t o = {0.f, 0.f, 0.f, 0.f};
    float lloc = 0.f;
    for (int kt = 0; kt < nkt; ++kt) {
      const int key = kt * 16 + ql;
      const f32x4_t kf = *reinterpret_cast<const f32x4_t*>(Ks + key * 64 + ((g4 ^ ((key >> 1) & 3)) << 4));
      f32x4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[ss], qf[ss], s, 0, 0, 0);
      // V^T fragment: row = head dim (lane & 15), keys kt*16 + 4*g4 + r
      const f32x4_t vf = *reinterpret_cast<const f32x4_t*>(Vt + ql * VT_STRIDE_F32 + (kt * 16 + 4 * g4) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pv = __builtin_amdgcn_exp2f(s[r] * c - mc);
        if (kt * 16 + 4 * g4 + r >= L) pv = 0.f;
        lloc += pv;
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r], pv, o, 0, 0, 0);
      }
    }
    float l = lloc + __shfl_xor(lloc, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    // O^T C layout: col = query, row = head dim 4*g4 + r -> this lane owns O[q][4*g4 .. 4*g4+3]
    if (qvalid) {
      const int w = p.widx[ps + qslot];
      if (w >= 0) {
        float4 ov = make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
        *reinterpret_cast<float4*>((float*)p.out + (long)w * p.ldo + head * 16 + 4 * g4) = ov;
      }
    }
  }
}


// ------------------------------------------------------------------------------------ fp32 x3 (CDSEG_F32X3, round 6)
// The parity mode's attention at a multiple of the exact-fp32 MFMA rate: q, k, v are fp32 in memory; on their way into LDS /
// the B operand every value is split into a 16-bit PAIR and a product runs as three 16-bit MFMAs with fp32 accumulation
// (csrc/gemm.hip "fp32 x3" does the same for the GEMMs):
//   scores   q' = q * scale * log2 e and k as IEEE-half pairs x ~= hi + lo (22 significant bits; |q'|, |k| <= 65504):
//            S^T = K_hi Q_hi^T + K_hi Q_lo^T + K_lo Q_hi^T, one accumulator (v_mfma_f32_32x32x16_f16)
//   P V      P~ = exp2(s) (unshifted single pass, see attn_bf16_kernel) and v as BFLOAT16 pairs (fp32's exponent range; 16
//            significant bits): O^T = V1^T P1^T + V2^T P1^T + V1^T P2^T; the row of ones that yields the denominator sits
//            behind V1 only (behind V2: zeros), so the denominator is sum(p1 + p2)
// Rows whose denominator left [1e-30, 1e30] are redone with the exact row max.  Same block -> (patch, head, slice) map and the
// same LDS images as the 16-bit kernel (K planes with the source-side half swap, V planes row-major for ds_read_b64_tr_b16);
// four 32 KB planes = one block of 8 waves per CU, two score tiles in flight per wave.
constexpr int X3_KH = 0, X3_KL = KV_STAGE, X3_V1 = 2 * KV_STAGE, X3_V2 = 3 * KV_STAGE;
constexpr int X3_ONES = 4 * KV_STAGE, X3_ZERO = X3_ONES + ONES_BYTES;
constexpr int SMEM_X3 = X3_ZERO + ONES_BYTES;
#ifndef ATTN_X3_WAVES
#define ATTN_X3_WAVES 8
#endif
#ifndef ATTN_X3_TILES
#define ATTN_X3_TILES 2
#endif
constexpr int X3_WAVES = ATTN_X3_WAVES, X3_THREADS = 64 * X3_WAVES, X3_TILES = ATTN_X3_TILES;  // 8 x 2 | 16 x 1 (128 VGPRs)
static_assert(X3_WAVES % ATTN_RUN == 0, "whole runs of query tiles per block");
typedef _Float16 x3h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 x3h8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void x3_split_f16(float a, float b, uint32_t& hi, uint32_t& lo) {
  const hw_f32x2_t v = {__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f)};
  const x3h2_t hh = __builtin_convertvector(v, x3h2_t);
  const hw_f32x2_t r = v - __builtin_convertvector(hh, hw_f32x2_t);
  hi = __builtin_bit_cast(uint32_t, hh);
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, x3h2_t));
}
__device__ __forceinline__ void x3_split_bf16(float a, float b, uint32_t& p1, uint32_t& p2) {
  p1 = pack_truebf16x2(a, b);
  p2 = pack_truebf16x2(a - __uint_as_float(p1 << 16), b - __uint_as_float(p1 & 0xffff0000u));
}
__device__ __forceinline__ f32x16_t x3_mfma_f16(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(x3h8_t, a), __builtin_bit_cast(x3h8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ f32x16_t x3_qk_tile(const char* k_lane, bf16x8_t qh, bf16x8_t ql, const f32x16_t& c0) {
  const bf16x8_t kh = *reinterpret_cast<const bf16x8_t*>(k_lane + X3_KH);
  const bf16x8_t kl = *reinterpret_cast<const bf16x8_t*>(k_lane + X3_KL);
  f32x16_t s = x3_mfma_f16(kh, qh, c0);
  s = x3_mfma_f16(kh, ql, s);
  return x3_mfma_f16(kl, qh, s);
}

template <bool TAIL, bool FIRST = false>
__device__ __forceinline__ void x3_pv_tile(const f32x16_t& s, int kt, int h, int L, unsigned va1, unsigned va2, f32x16_t& o) {
  float pr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) pr[r] = __builtin_amdgcn_exp2f(s[r]);
  if (TAIL) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= L) pr[r] = 0.f;
  }
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    union { bf16x8_t v; uint32_t u[4]; } p1, p2;
#pragma unroll
    for (int j = 0; j < 4; ++j) x3_split_bf16(pr[8 * mf + 2 * j], pr[8 * mf + 2 * j + 1], p1.u[j], p2.u[j]);
    const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4_t*>(va1 + 512 * mf));
    const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4_t*>(va1 + 512 * mf + 256));
    const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4_t*>(va2 + 512 * mf));
    const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4_t*>(va2 + 512 * mf + 256));
    const bf16x8_t v1 = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
    const bf16x8_t v2 = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    if (FIRST && mf == 0) {
      const f32x16_t z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      o = mfma_32x32x16_truebf16(v1, p1.v, z);
    } else {
      o = mfma_32x32x16_truebf16(v1, p1.v, o);
    }
    o = mfma_32x32x16_truebf16(v2, p1.v, o);
    o = mfma_32x32x16_truebf16(v1, p2.v, o);
  }
}

__global__ __launch_bounds__(X3_THREADS) void attn_x3_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int patch, head, qslice, qsplit;
  if (!decode_block(p, (int)blockIdx.x, patch, head, qslice, qsplit)) return;
  const int ps = p.patch_start[patch];
  const int L = p.patch_start[patch + 1] - ps;
  const int nkt = (L + 31) >> 5;
  const float* kb = (const float*)p.k + head * 16;
  const float* vb = (const float*)p.v + head * 16;
  // ---- stage K and V: a thread converts whole keys (16 + 16 floats -> two half planes, two bfloat16 planes)
  for (int s = tid; s < nkt * 32; s += X3_THREADS) {
    uint4 kk[4], vv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) kk[i] = vv[i] = make_uint4(0, 0, 0, 0);
    if (s < L) {
      const long g = p.kv_gidx[ps + s];
      const uint4* kr = reinterpret_cast<const uint4*>(kb + g * p.ldk);
      const uint4* vr = reinterpret_cast<const uint4*>(vb + g * p.ldv);
#pragma unroll
      for (int i = 0; i < 4; ++i) { kk[i] = kr[i]; vv[i] = vr[i]; }
    }
    const int sw = (s >> 3) & 1;  // keys with bit 3 set store their two 16-byte halves swapped (conflict-free A reads)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {  // dims 8 hf .. 8 hf + 7
      uint4 kh, kl, v1, v2;
      x3_split_f16(__uint_as_float(kk[2 * hf].x), __uint_as_float(kk[2 * hf].y), kh.x, kl.x);
      x3_split_f16(__uint_as_float(kk[2 * hf].z), __uint_as_float(kk[2 * hf].w), kh.y, kl.y);
      x3_split_f16(__uint_as_float(kk[2 * hf + 1].x), __uint_as_float(kk[2 * hf + 1].y), kh.z, kl.z);
      x3_split_f16(__uint_as_float(kk[2 * hf + 1].z), __uint_as_float(kk[2 * hf + 1].w), kh.w, kl.w);
      x3_split_bf16(__uint_as_float(vv[2 * hf].x), __uint_as_float(vv[2 * hf].y), v1.x, v2.x);
      x3_split_bf16(__uint_as_float(vv[2 * hf].z), __uint_as_float(vv[2 * hf].w), v1.y, v2.y);
      x3_split_bf16(__uint_as_float(vv[2 * hf + 1].x), __uint_as_float(vv[2 * hf + 1].y), v1.z, v2.z);
      x3_split_bf16(__uint_as_float(vv[2 * hf + 1].z), __uint_as_float(vv[2 * hf + 1].w), v1.w, v2.w);
      *reinterpret_cast<uint4*>(smem + X3_KH + s * 32 + ((hf ^ sw) << 4)) = kh;
      *reinterpret_cast<uint4*>(smem + X3_KL + s * 32 + ((hf ^ sw) << 4)) = kl;
      *reinterpret_cast<uint4*>(smem + X3_V1 + s * 32 + (hf << 4)) = v1;
      *reinterpret_cast<uint4*>(smem + X3_V2 + s * 32 + (hf << 4)) = v2;
    }
  }
  for (int w = tid; w < ONES_BYTES / 8; w += X3_THREADS) {
    *reinterpret_cast<uint2*>(smem + X3_ONES + w * 8) = make_uint2(0x3F80u, 0u);
    *reinterpret_cast<uint2*>(smem + X3_ZERO + w * 8) = make_uint2(0u, 0u);
  }
  __syncthreads();

  const int ql = lane & 31, h = lane >> 5;
  const char* k_lane = smem + ql * 32 + ((h ^ ((ql >> 3) & 1)) << 4);
  const bool v_lane = (lane & 16) == 0;  // lane groups 0 / 2 read V, 1 / 3 the ones (V1) / zero (V2) page
  const unsigned vrow = (4 * h + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
  const unsigned va1_0 = v_lane ? lds_base + X3_V1 + vrow : lds_base + X3_ONES + (h ? 0 : 128);
  const unsigned va2_0 = v_lane ? lds_base + X3_V2 + vrow : lds_base + X3_ZERO + (h ? 0 : 128);
  const unsigned vstep = v_lane ? 1024u : 0u;
  const float c = p.scale_log2e;
  const bool tail = (nkt << 5) != L;
  const int nfull = tail ? nkt - 1 : nkt;
  const f32x16_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  // a slice takes every qsplit-th run of 8 query tiles (decode_block); a block of 16 waves works on two of its runs at a time
  for (int run = qslice + (wave / ATTN_RUN) * qsplit; run * ATTN_RUN + (wave % ATTN_RUN) < nkt;
       run += (X3_WAVES / ATTN_RUN) * qsplit) {
    const int qt = run * ATTN_RUN + (wave % ATTN_RUN);
    const int qslot = qt * 32 + ql;
    const bool qvalid = qslot < L;
    const long g = p.q_gidx[ps + min(qslot, L - 1)];
    const uint4* qrow = reinterpret_cast<const uint4*>((const float*)p.q + g * p.ldq + head * 16 + h * 8);
    const uint4 qa = qrow[0], qb = qrow[1];
    int w = p.widx[ps + min(qslot, L - 1)];
    if (!qvalid) w = -1;
    union { bf16x8_t v; uint32_t u[4]; } qh, qlo;
    x3_split_f16(__uint_as_float(qa.x) * c, __uint_as_float(qa.y) * c, qh.u[0], qlo.u[0]);
    x3_split_f16(__uint_as_float(qa.z) * c, __uint_as_float(qa.w) * c, qh.u[1], qlo.u[1]);
    x3_split_f16(__uint_as_float(qb.x) * c, __uint_as_float(qb.y) * c, qh.u[2], qlo.u[2]);
    x3_split_f16(__uint_as_float(qb.z) * c, __uint_as_float(qb.w) * c, qh.u[3], qlo.u[3]);
    auto exp_pv_pass = [&](auto shift_tag, float mrow) {
      constexpr bool SHIFT = decltype(shift_tag)::value;
      const float nm = SHIFT ? -mrow : 0.f;
      const f32x16_t negm = {nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm};
      const f32x16_t& c0 = SHIFT ? negm : zero16;
      f32x16_t acc;
      const char* kp = k_lane;
      unsigned va1 = va1_0, va2 = va2_0;
      int kt;
      if (X3_TILES == 2 && nfull >= 2) {
        const f32x16_t sa = x3_qk_tile(kp, qh.v, qlo.v, c0);
        const f32x16_t sb = x3_qk_tile(kp + 1024, qh.v, qlo.v, c0);
        x3_pv_tile<false, true>(sa, 0, h, L, va1, va2, acc);
        x3_pv_tile<false>(sb, 1, h, L, va1 + 1024, va2 + 1024, acc);
        kp += 2048; va1 += 2 * vstep; va2 += 2 * vstep;
        kt = 2;
      } else {
        const f32x16_t s0 = x3_qk_tile(kp, qh.v, qlo.v, c0);
        if (nfull == 0) x3_pv_tile<true, true>(s0, 0, h, L, va1, va2, acc);
        else x3_pv_tile<false, true>(s0, 0, h, L, va1, va2, acc);
        kp += 1024; va1 += vstep; va2 += vstep;
        kt = 1;
      }
      for (; X3_TILES == 2 && kt + 1 < nfull; kt += 2) {
        const f32x16_t sa = x3_qk_tile(kp, qh.v, qlo.v, c0);
        const f32x16_t sb = x3_qk_tile(kp + 1024, qh.v, qlo.v, c0);
        x3_pv_tile<false>(sa, kt, h, L, va1, va2, acc);
        x3_pv_tile<false>(sb, kt + 1, h, L, va1 + 1024, va2 + 1024, acc);
        kp += 2048; va1 += 2 * vstep; va2 += 2 * vstep;
      }
      for (; kt < nkt; ++kt) {
        const f32x16_t s = x3_qk_tile(kp, qh.v, qlo.v, c0);
        if (kt >= nfull) x3_pv_tile<true>(s, kt, h, L, va1, va2, acc);
        else x3_pv_tile<false>(s, kt, h, L, va1, va2, acc);
        kp += 1024; va1 += vstep; va2 += vstep;
      }
      return acc;
    };
    f32x16_t o = exp_pv_pass(std::false_type{}, 0.f);
    {
      const float den = __shfl(o[8], ql, 64);
      const bool redo = qvalid && !(den >= 1e-30f && den <= 1e30f);
      if (__any(redo)) {  // exact row max, then the shifted pass (wave-uniform, rare)
        float m0 = -INFINITY;
        const char* kp = k_lane;
        for (int kt = 0; kt < nkt; ++kt, kp += 1024) {
          f32x16_t s = x3_qk_tile(kp, qh.v, qlo.v, zero16);
          if (kt >= nfull) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= L) s[r] = -INFINITY;
          }
          m0 = tile_max(s, m0);
        }
        m0 = fmaxf(m0, __shfl_xor(m0, 32, 64));
        o = exp_pv_pass(std::true_type{}, m0);
      }
    }
    // O^T rows (r & 3) + 8 (r >> 2) + 4 h: registers 0..3 = dims 4h .. 4h+3, 4..7 = dims 8+4h .. 8+4h+3; row 16 = denominator
    const float inv = 1.0f / __shfl(o[8], ql, 64);
    if (w >= 0) {
      float* orow = (float*)p.out + (long)w * p.ldo + head * 16 + 4 * h;
      *reinterpret_cast<float4*>(orow) = make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
      *reinterpret_cast<float4*>(orow + 8) = make_float4(o[4] * inv, o[5] * inv, o[6] * inv, o[7] * inv);
    }
  }
}

}  // namespace

#ifdef CDSEG_ATTN_TIMING
extern "C" int cdseg_debug_attn_timing(unsigned long long* host_dst, size_t count) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_attn_t), count * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

// Zones of the graded schedule (see decode_block), sized in BLOCKS per XCD (an XCD has 32 CUs x 2 resident blocks = 64
// slots) and converted to whole (patch, head) units here.  tail1 / tail2: the last blocks of an XCD are half / quarter
// slices, so the launch's last round is short; lead: the first 32 blocks stay whole and the next `lead` are half slices -
// the two blocks a CU starts with then end at different times and later K / V stagings on that CU run next to the other
// block's key loop.  s0 = slices of the bulk (1 when the launch has enough patch-heads to fill the chip).  Defaults from the
// interleaved same-process sweep in profiles/r05_attention_sched.txt (tools/attn_sweep.py): tails of 64 + 64 blocks give
// +1 .. +2 % on the 1700 - 3400-unit launches of stages 0 / 1 and +7 .. +12 % on the 800 - 1600-unit ones; the lead zone
// measured +-0 and stays off.
static unsigned plan_zones(AttnP& p, int s0, int max_split, int lead, int tail1, int tail2) {
  const int U = p.num_patches * p.num_heads;
  const int n_min = U / 8, n_max = (U + 7) / 8;
  const int s1 = 2 * s0 <= max_split ? 2 * s0 : 0, s2 = 4 * s0 <= max_split ? 4 * s0 : 0;
  auto units = [&](int blocks, int split) { return split && blocks > 0 ? (blocks + split - 1) / split : 0; };
  if (n_min < cdseg_knob("CDSEG_ATTN_ZONE_MIN", ATTN_ZONE_MIN_UNITS)) lead = tail1 = tail2 = 0;
  // ... and only for full-length patches (32 query tiles): the 768-unit / 775-row launches of 24 collated scenes' deepest stage
  // run 12 % SLOWER with their last units cut into slices of 6 - 13 tiles (profiles/r06_attention_zones24.txt)
  if (max_split < cdseg_knob("CDSEG_ATTN_ZONE_MIN_SPLIT", 4) || !p.full_patches) lead = tail1 = tail2 = 0;
  int u_lead = units(lead, s1), u_head = u_lead ? units(32, s0) : 0;
  int u_t1 = units(tail1, s1), u_t2 = units(tail2, s2);
  // too few units per XCD for all the zones: drop the lead, then shrink the tails (at least a third stays bulk)
  if (u_head + u_lead + u_t1 + u_t2 > n_min - n_min / 3) u_head = u_lead = 0;
  while (u_t1 + u_t2 > n_min - n_min / 3 && (u_t1 | u_t2)) {
    u_t1 -= u_t1 > 0;
    if (u_t1 + u_t2 > n_min - n_min / 3) u_t2 -= u_t2 > 0;
  }
  int z = 0;
  auto add = [&](int cnt, int split) {
    if (cnt == 0) return;
    p.zcnt[z] = cnt; p.zsplit[z] = split; ++z;
  };
  add(u_head, s0);
  add(u_lead, s1);
  add(-1, s0);
  add(u_t1, s1);
  add(u_t2, s2);
  p.nzones = z;
  p.zfixed = u_head + u_lead + u_t1 + u_t2;
  p.bulk_max = n_max - p.zfixed;
  long per_xcd = 0;
  for (int i = 0; i < z; ++i) per_xcd += (long)(p.zcnt[i] >= 0 ? p.zcnt[i] : p.bulk_max) * p.zsplit[i];
  return (unsigned)(per_xcd * 8);
}

// Fills the schedule fields of p (and num_patches / num_heads); returns the grid size.
static unsigned make_schedule(AttnP& p, int num_patches, int num_heads, int max_len, int dtype) {
  p.num_heads = num_heads;
  // K/V staging is per block, so split a patch-head's queries over as few blocks as still fill
  // the chip (2 resident blocks per CU -> ~512 block slots)
  const int tile = dtype == CDSEG_F32 ? 16 : 32;  // (CDSEG_F32X3: 32-query tiles like the 16-bit kernel)
  const int nqt = (max_len + tile - 1) / tile;
  const int ph = num_patches * num_heads;
  // (sweep at the end of round 3, tools/bench_attention.py: 202 and 224 patch-heads run 2 - 6 % faster unsplit than in two
  // slices; 112 and fewer want 2 - 4 slices)
  int qsplit = ph >= 192 ? 1 : (ph >= 96 ? 2 : 4);
  qsplit = cdseg_knob("CDSEG_ATTN_QSPLIT", qsplit);  // (power of two)
  const int max_split = (nqt + ATTN_RUN - 1) / ATTN_RUN;
  while (qsplit > 1 && qsplit > max_split) qsplit >>= 1;
  p.num_patches = num_patches;
  p.full_patches = nqt * tile >= CDSEG_MAX_PATCH;
  for (int z = 0; z < ATTN_ZONES; ++z) p.zcnt[z] = p.zsplit[z] = 0;
  // (the fp32 parity kernel keeps the uniform schedule: one bulk zone)
  const bool graded = dtype == CDSEG_BF16;
  const unsigned nblocks =
      plan_zones(p, qsplit, max_split, graded ? cdseg_knob("CDSEG_ATTN_LEAD", ATTN_LEAD_BLOCKS) : 0,
                 graded ? cdseg_knob("CDSEG_ATTN_TAIL1", ATTN_TAIL1_BLOCKS) : 0,
                 graded ? cdseg_knob("CDSEG_ATTN_TAIL2", ATTN_TAIL2_BLOCKS) : 0);
  return nblocks;
}

// One block per CU at most, every block a whole patch-head or a slice with >= 16 query tiles (so that 16 waves have a tile each):
// the launches of a single scene's wide stages.  CDSEG_ATTN_W16 = 0 switches the shape off (tools A/B).
static bool attention_wide16(const AttnP& p, unsigned nblocks, int max_len) {
  const int on = cdseg_knob("CDSEG_ATTN_W16", 1);
  const int max_blocks = cdseg_knob("CDSEG_ATTN_W16_MAX_BLOCKS", 256);
  if (!on || nblocks > (unsigned)max_blocks || p.nzones != 1) return false;
  const int nqt = (max_len + 31) / 32;
  return nqt / p.zsplit[0] >= cdseg_knob("CDSEG_ATTN_W16_MIN_TILES", 16);
}

extern "C" int cdseg_attention_ex(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv,
                                  const int32_t* q_gidx, const int32_t* kv_gidx, const int32_t* widx,
                                  const int32_t* patch_start, int num_patches, int num_heads, int max_len, float scale,
                                  void* out, int ldo, int dtype, int flags, void* stream) {
  if (num_patches <= 0 || num_heads <= 0) return CDSEG_OK;
  if (max_len <= 0 || max_len > CDSEG_MAX_PATCH) return CDSEG_ERR_UNSUPPORTED;
  if (dtype != CDSEG_BF16 && dtype != CDSEG_F32 && dtype != CDSEG_F32X3) return CDSEG_ERR_ARG;
  if (flags & ~(CDSEG_ATTN_Q_PRESCALED | CDSEG_ATTN_V_BF16)) return CDSEG_ERR_ARG;
  if (dtype != CDSEG_BF16 && (flags & CDSEG_ATTN_V_BF16)) return CDSEG_ERR_ARG;
  const int esz = dtype == CDSEG_BF16 ? 2 : 4;
  // 16-byte alignment of every gathered row slice; output pieces: 16 bytes (fp32: always; 16-bit: the one-store epilogue),
  // or 8 bytes for 16-bit outputs with ldo % 8 != 0 (such calls were valid before the 16-byte store existed)
  if (((long)ldq * esz) & 15 || ((long)ldk * esz) & 15 || ((long)ldv * esz) & 15) return CDSEG_ERR_ARG;
  if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v)) & 15) return CDSEG_ERR_ARG;
  const bool out16 = !((((long)ldo * esz) | (long)(uintptr_t)out) & 15);
  if (!out16 && (dtype != CDSEG_BF16 || ((((long)ldo * esz) | (long)(uintptr_t)out) & 7))) return CDSEG_ERR_ARG;
  AttnP p;
  p.q = q; p.k = k; p.v = v; p.q_gidx = q_gidx; p.kv_gidx = kv_gidx; p.widx = widx; p.patch_start = patch_start;
  p.out = out; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.num_heads = num_heads;
  // Q pre-scaled: the producer's weights carry softmax scale * log2(e) (`scale` is then ignored)
  p.scale_log2e = (flags & CDSEG_ATTN_Q_PRESCALED) ? 1.0f : scale * 1.44269504088896340736f;
  p.flags = flags | (out16 ? 0 : ATTN_STORE8);
  hipStream_t s = (hipStream_t)stream;
  const unsigned nblocks = make_schedule(p, num_patches, num_heads, max_len, dtype);
  // 16-bit kernel, at most one block per CU and at least 16 query tiles per block: the 16-wave block shape
  const bool wide16 = dtype == CDSEG_BF16 && attention_wide16(p, nblocks, max_len);
  dim3 grid(nblocks), block(dtype == CDSEG_BF16 ? (wide16 ? 64 * BF_WAVES_WIDE : BF_THREADS) : ATTN_THREADS);
  static std::once_flag attr_once;
  static bool attr_ok = false;
  std::call_once(attr_once, [] {
    attr_ok = hipFuncSetAttribute((const void*)attn_bf16_kernel<BF_WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  SMEM_BF16) == hipSuccess &&
              hipFuncSetAttribute((const void*)attn_bf16_kernel<BF_WAVES_WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  SMEM_BF16) == hipSuccess &&
              hipFuncSetAttribute((const void*)attn_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_F32) ==
                  hipSuccess &&
              hipFuncSetAttribute((const void*)attn_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_X3) ==
                  hipSuccess;
  });
  if (!attr_ok) return CDSEG_ERR_LAUNCH;
  CdsegProfToken tok;
  const bool prof = cdseg_prof_begin(CDSEG_PROF_ATTENTION, s, &tok);
  if (dtype == CDSEG_BF16) {
    if (wide16) hipLaunchKernelGGL(attn_bf16_kernel<BF_WAVES_WIDE>, grid, block, SMEM_BF16, s, p);
    else hipLaunchKernelGGL(attn_bf16_kernel<BF_WAVES>, grid, block, SMEM_BF16, s, p);
  } else if (dtype == CDSEG_F32X3) {
    hipLaunchKernelGGL(attn_x3_kernel, grid, dim3(X3_THREADS), SMEM_X3, s, p);
  } else {
    hipLaunchKernelGGL(attn_f32_kernel, grid, block, SMEM_F32, s, p);
  }
  if (prof) cdseg_prof_end(tok, s);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

extern "C" int cdseg_attention(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv,
                               const int32_t* q_gidx, const int32_t* kv_gidx, const int32_t* widx,
                               const int32_t* patch_start, int num_patches, int num_heads, int max_len, float scale,
                               void* out, int ldo, int dtype, void* stream) {
  return cdseg_attention_ex(q, k, v, ldq, ldk, ldv, q_gidx, kv_gidx, widx, patch_start, num_patches, num_heads, max_len,
                            scale, out, ldo, dtype, 0, stream);
}

// Diagnostic (host only, no device work): the block -> (patch, head, query slice, slices) table cdseg_attention_ex would
// launch for this shape, 4 ints per block id (-1 x 4 for ids that exit).  Returns the number of block ids, or a negative
// status; writes at most `capacity` rows.  tests/test_attention_schedule.py checks that every (patch, head, slice) appears
// exactly once for the shapes of the model.
extern "C" long cdseg_attention_schedule(int num_patches, int num_heads, int max_len, int dtype, int32_t* table,
                                         long capacity) {
  if (num_patches <= 0 || num_heads <= 0 || max_len <= 0 || max_len > CDSEG_MAX_PATCH) return CDSEG_ERR_ARG;
  if (dtype != CDSEG_BF16 && dtype != CDSEG_F32 && dtype != CDSEG_F32X3) return CDSEG_ERR_ARG;
  AttnP p{};
  const unsigned nb = make_schedule(p, num_patches, num_heads, max_len, dtype);
  for (long b = 0; b < (long)nb && b < capacity && table; ++b) {
    int patch = -1, head = -1, qslice = -1, qsplit = -1;
    if (!decode_block(p, (int)b, patch, head, qslice, qsplit)) patch = head = qslice = qsplit = -1;
    table[4 * b] = patch; table[4 * b + 1] = head; table[4 * b + 2] = qslice; table[4 * b + 3] = qsplit;
  }
  return (long)nb;
}
