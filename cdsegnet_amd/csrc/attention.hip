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
//  * bf16: ONE pass over the keys.  Softmax is shift invariant, so instead of the row max the kernel subtracts the
//    Cauchy-Schwarz bound |q'_i| max_j |k_j| (one norm per query, one max over the keys after staging K): no max
//    sweep, no second QK^T.  Scale * log2(e) is folded into Q', the shift rides in the MFMA's C operand, so a score
//    costs ONE v_exp_f32.  Query tiles whose bound is too loose (> 2^60) are redone with the exact row max;
//  * the kernel is bound by the VALU / transcendental issue, not by the matrix pipe: 16 v_exp_f32 + 8
//    v_cvt_pk_bf16_f32 per 32x32 score tile against 3 MFMAs of 32 cycles (tools/ubench/pipes.hip,
//    profiles/r03_ubench_pipes.txt: 128 cycles per tile and SIMD at 4 waves per SIMD = half the MFMA-only rate);
//  * bf16: v_mfma_f32_32x32x16_bf16 for both products; the MFMA k-slot <-> key assignment of
//    the PV product is chosen so the exponentiated scores feed it straight from the
//    accumulator registers (no permute); the matching V^T operand is two ds_read_b64_tr_b16 of the row-major V;
//    a row of ones (constant LDS page) makes the softmax denominator fall out of the same MFMA (row 16 of the result);
//  * f32 (the 1e-3 parity mode): v_mfma_f32_16x16x4_f32 for both products, exact fp32.
#include <cstdlib>

#include "common.h"
#include "prof.h"

namespace {

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
  int qsplit;
  int hgroups;  // head groups per patch: the unit pinned to one XCD is (patch, head group)
  float scale_log2e;
  int dbg;  // experimental builds only: 1 = K / V rows taken in slot order (no gather), 2 = no K / V staging at all
};

// XCD-aware block -> (patch, head, query-slice) map.  Workgroups are dispatched round-robin over the 8 XCDs
// (block b -> XCD b % 8, a performance-only assumption).  The unit pinned to one XCD is a (patch, head group): all its
// blocks (heads and query slices read the same gathered rows) get ids that are equal mod 8 and adjacent in time, so
// the rows are fetched into ONE L2 once.  XCD x owns the CONTIGUOUS run of groups [G x / 8, G (x + 1) / 8): patches
// that follow each other on the curve are neighbours in space and their rows interleave in memory (a 192-byte qkv
// row shares a 128-byte line with the next row), so neighbouring patches belong on the same L2 - with the patches
// dealt round-robin the shared lines were fetched by two XCDs (1.27x the algorithmic HBM bytes at C = 32).  With
// fewer than 8 patches (deep stages) the heads of a patch are split into `hgroups` groups so that all XCDs still get
// work.  Grid = ceil(G / 8) * 8 * (H / hgroups) * Q blocks with G = P * hgroups; surplus ids exit.
__device__ __forceinline__ bool decode_block(const AttnP& p, int& patch, int& head, int& qslice) {
  const int hpg = p.num_heads / p.hgroups;  // heads per group
  const int per_group = hpg * p.qsplit;
  const int G = p.num_patches * p.hgroups;
  const int B = blockIdx.x;
  const int xcd = B & 7;
  const int t = B >> 3;
  const int r = t % per_group;
  const int g = (int)(((long)G * xcd) >> 3) + t / per_group;
  if (g >= (int)(((long)G * (xcd + 1)) >> 3)) return false;
  patch = g / p.hgroups;
  const int hg = g - patch * p.hgroups;
  head = hg * hpg + r / p.qsplit;
  qslice = r % p.qsplit;
  return true;
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
constexpr int ATTN_THREADS = 512;  // 8 waves: 2 per SIMD per block, 2 blocks per CU (LDS 66 KB each)
constexpr int ATTN_WAVES = ATTN_THREADS / 64;
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
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf, c0, 0, 0, 0);
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
template <bool TAIL>
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
    for (int j = 0; j < 4; ++j) pf.u[j] = pack_bf16x2(pr[8 * mf + 2 * j], pr[8 * mf + 2 * j + 1]);
    const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4_t*>(va + 512 * mf));
    const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4_t*>(va + 512 * mf + 256));
    const bf16x8_t vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf.v, o, 0, 0, 0);
  }
}

__device__ __forceinline__ float sq8_bf16(const uint4& a) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  float t = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.x), __builtin_bit_cast(bf2, a.x), 0.f, false);
  t = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.y), __builtin_bit_cast(bf2, a.y), t, false);
  t = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.z), __builtin_bit_cast(bf2, a.z), t, false);
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.w), __builtin_bit_cast(bf2, a.w), t, false);
}

__global__ __launch_bounds__(ATTN_THREADS, 4) void attn_bf16_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  float* s_kn2 = reinterpret_cast<float*>(smem + 2 * KV_STAGE + ONES_BYTES);  // per-wave max |k|^2
  unsigned* s_next = reinterpret_cast<unsigned*>(smem + 2 * KV_STAGE + ONES_BYTES + 32);  // query tiles handed out so far
  int* s_qidx = reinterpret_cast<int*>(smem + 2 * KV_STAGE + ONES_BYTES + 64);           // slot -> query row
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int patch, head, qslice;
  if (!decode_block(p, patch, head, qslice)) return;
#ifdef CDSEG_ATTN_TIMING
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t_loop = 0;
#endif
  ATTN_STAMP(t0);
  if (p.dbg & 8) __builtin_amdgcn_s_setprio(3);
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
      const int base = (qslice + (int)(i >> 3) * p.qsplit) * ATTN_WAVES;
      if (base >= nqt) return -1;
      const int t = base + (int)(i & 7);
      if (t < nqt) return t;
    }
  };
  auto load_qrow = [&](int t) {  // (unconditional: a wave without a next tile re-reads tile 0 and drops it)
    const int g = s_qidx[min(max(t, 0) * 32 + ql, L - 1)];
    return *reinterpret_cast<const uint4*>((const bf16_t*)p.q + (long)g * p.ldq + head * 16 + h * 8);
  };

  // ---- stage K and V by LDS-DMA.  Wave w moves the 32-key pieces w, w + 8, ..: lane -> (key = lane / 2, 16-B half).
  {
    const int kip = lane >> 1, hs = lane & 1;
    int gk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = (wave + ATTN_WAVES * i) * 32 + kip;
      gk[i] = s < L ? p.kv_gidx[ps + s] : -1;
    }
    // slot -> query row table (this wave's 128 slots), for the query-row fetches of the tile loop
    int gq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) gq[i] = p.q_gidx[ps + min(wave * 128 + i * 64 + lane, L - 1)];
    if (tid == 0) *s_next = 0u;
    // (the compiler waits for its own loads above; the DMAs are invisible to it and are waited for by hand below)
    // all source addresses first (pinned by the empty asm): the compiler's vmcnt(0) for an index load must not sit
    // between two DMAs, where it would wait for the DMA before it as well
    const void* ksrc[4];
    const void* vsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = (wave + ATTN_WAVES * i) * 32 + kip;
      ksrc[i] = vsrc[i] = (const char*)g_attn_zero + hs * 16;
      if (gk[i] >= 0) {
        ksrc[i] = kb + (long)gk[i] * p.ldk + ((hs ^ ((s >> 3) & 1)) << 3);
        vsrc[i] = vb + (long)gk[i] * p.ldv + (hs << 3);
      }
      asm volatile("" : "+v"(ksrc[i]), "+v"(vsrc[i]));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pc = wave + ATTN_WAVES * i;
      if (pc < nkt) {
        dma16(ksrc[i], lds_base + pc * 1024);
        dma16(vsrc[i], lds_base + KV_STAGE + pc * 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) s_qidx[wave * 128 + i * 64 + lane] = gq[i];
    for (int w = tid; w < ONES_BYTES / 8; w += ATTN_THREADS)
      *reinterpret_cast<uint2*>(smem + 2 * KV_STAGE + w * 8) = make_uint2(0x3F80u, 0u);
    // largest squared key norm of the patch-head (for the score bound of the single-pass softmax below), from the
    // wave's own pieces: its own vmcnt(0) is all the ordering they need
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float kn2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pc = wave + ATTN_WAVES * i;
      if (pc < nkt) {
        float t = sq8_bf16(*reinterpret_cast<const uint4*>(Ks + pc * 1024 + lane * 16));
        t += __shfl_xor(t, 1, 64);  // the key's other half
        kn2 = fmaxf(kn2, t);
      }
    }
    kn2 = wave_max(kn2);
    if (lane == 0) s_kn2[wave] = kn2;
    __syncthreads();  // everybody's DMA has landed, the ones page and the norms are written
  }
  if (p.dbg & 8) __builtin_amdgcn_s_setprio(0);
  ATTN_STAMP(t1);
  float kmax2;
  {
    const float4 a = *reinterpret_cast<const float4*>(s_kn2), b = *reinterpret_cast<const float4*>(s_kn2 + 4);
    kmax2 = fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
  }

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

  int qt = claim();
  uint4 q_cur = load_qrow(qt);
  while (qt >= 0) {
    // Q' = Q * (softmax scale * log2 e), rounded to bf16 once: scores come out of the MFMA in exp2 units (consumed
    // BEFORE the next loads are issued: the compiler's wait for q_cur then has nothing younger in the queue)
    bf16x8_t qf;
    {
      union { bf16x8_t v; uint32_t u[4]; } qs;
      const uint32_t qr[4] = {q_cur.x, q_cur.y, q_cur.z, q_cur.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        qs.u[j] = pack_bf16x2(__uint_as_float(qr[j] << 16) * c, __uint_as_float(qr[j] & 0xffff0000u) * c);
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
    // P = exp2(S' - m), O^T (+ row sums in row 16) += [V^T; 1; 0] P^T, with S' - m straight out of the MFMA
    // (C operand = -m, loop invariant)
    auto exp_pv_pass = [&](float mrow) {
      const float nm = -mrow;
      const f32x16_t negm = {nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm};
      f32x16_t acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      const char* kp = k_lane;
      unsigned va = va0;
      int kt = 0;
      for (; kt + 1 < nfull; kt += 2) {  // two independent tiles in flight
        const f32x16_t sa = qk_tile(kp, qf, negm);
        const f32x16_t sb = qk_tile(kp + 1024, qf, negm);
        pv_tile<false>(sa, kt, h, L, va, acc);
        pv_tile<false>(sb, kt + 1, h, L, va + 1024, acc);
        kp += 2048;
        va += 2 * vstep;
      }
      for (; kt < nkt; ++kt) {
        const f32x16_t s = qk_tile(kp, qf, negm);
        if (kt >= nfull) pv_tile<true>(s, kt, h, L, va, acc);
        else pv_tile<false>(s, kt, h, L, va, acc);
        kp += 1024;
        va += vstep;
      }
      return acc;
    };
    // ---- single pass: softmax is shift invariant, so any m >= max_j s_ij that does not underflow the row works.
    // Cauchy-Schwarz gives one for free: s_ij <= |q'_i| * max_j |k_j|.  It replaces the row-max pass (a second QK^T
    // MFMA sweep + a v_max3 per score pair; the kernel is VALU-issue bound).  P keeps its relative precision at
    // any magnitude (fp32 / bf16 share the exponent range, the denominator comes from the same rounded values).
    float qn2 = sq8_bf16(__builtin_bit_cast(uint4, qf));
    qn2 += __shfl_xor(qn2, 32, 64);
    ATTN_STAMP(tl0);
    f32x16_t o = exp_pv_pass(sqrtf(qn2 * kmax2) * 1.0005f);
#ifdef CDSEG_ATTN_TIMING
    asm volatile("" :: "v"(o[0]), "v"(o[8]));
    t_loop += __builtin_readcyclecounter() - tl0;
#endif
    // rows whose bound is looser than 2^60 (the largest term could sink towards the denormal range) are redone
    // with the exact row max; wave-uniform branch, never taken for ordinary logits
    const bool loose = qvalid && !(__shfl(o[8], ql, 64) >= 8.6736174e-19f);
    if (__any(loose)) {
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
      o = exp_pv_pass(m0);
    }
    // ---- epilogue: O^T rows (r&3) + 8*(r>>2) + 4h; row 16 (lane h=0, r=8) is the denominator
    const float lsum = __shfl(o[8], ql, 64);
    const float inv = 1.0f / lsum;
    if (w >= 0) {
      bf16_t* orow = (bf16_t*)p.out + (long)w * p.ldo + head * 16 + 4 * h;
      uint2 a, b;
      a.x = pack_bf16x2(o[0] * inv, o[1] * inv);
      a.y = pack_bf16x2(o[2] * inv, o[3] * inv);
      b.x = pack_bf16x2(o[4] * inv, o[5] * inv);
      b.y = pack_bf16x2(o[6] * inv, o[7] * inv);
      *reinterpret_cast<uint2*>(orow) = a;      // d = 4h .. 4h+3
      *reinterpret_cast<uint2*>(orow + 8) = b;  // d = 8+4h .. 8+4h+3
    }
    qt = qt_nxt;
    q_cur = q_nxt;
  }
#ifdef CDSEG_ATTN_TIMING
  if (lane == 0 && blockIdx.x < 4096) {
    unsigned long long* d = g_attn_t + ((size_t)blockIdx.x * 8 + wave) * 8;
    d[0] = rt0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = t1 - t0; d[3] = t_loop;
    d[4] = __builtin_readcyclecounter() - t0;
  }
#endif
}

// ------------------------------------------------------------------------------------ bf16, persistent dataflow form
// What the in-kernel stamps (tools/attn_timing.py) and tools/ubench/attn_loop.hip say about the block form above:
//   * the SIMD arbitrates its waves by age.  Four co-resident waves in the key loop do not advance in step - the
//     oldest runs ~2.2x as fast as the youngest - and that is the EFFICIENT regime (together ~130-150 cycles per 32x32
//     tile and SIMD, the floor of the instruction mix); forcing them into step (progress-ranked s_setprio: measured)
//     costs 40 % of the aggregate rate;
//   * so every synchronisation of a group of waves wastes the lead of its fast members: the block form loses it at each
//     block's end (a block's LDS is released when its slowest wave exits; with the staging waves starved by the older
//     block and a quarter of the chip empty while the last round drains, the launch runs at ~256 cycles per tile and
//     SIMD), a persistent workgroup with one barrier per patch-head loses 30 % of every wave's life at the barrier.
// This form never synchronises the workgroup.  ONE 16-wave workgroup per CU walks a contiguous run of patch-heads;
//   * a TASK is one 32-query tile of the resident patch-head; a wave claims the next unclaimed task with a
//     compare-and-swap on an LDS word (position | tasks claimed): fast waves simply take more tasks, a wave claims its
//     next task when it starts the current one and fetches that task's query rows behind the current key loop;
//   * two LDS stages (K, V, row indices of a patch-head each).  The wave that finishes the LAST task of a patch-head owns
//     its stage: it stages the patch-head two positions ahead into it by LDS-DMA (row indices, then all 64 K / V
//     pieces), waits for its own DMAs, computes the key-norm bound, and publishes the stage by storing the new
//     position into the stage's word - while the other 15 waves are working on the other stage;
//   * a wave whose next patch-head is not published yet spins on that word (s_sleep, bounded; it holds no claimed
//     task while it spins, so the wait-for graph has no cycle: a stage is recycled by the finisher of its own tasks);
//   * the CUs of XCD x take the contiguous range [U x / 8, U (x + 1) / 8) of half-patch-heads in equal shares (a share
//     boundary may cut a patch-head between its two halves of query tiles): no partial last round.
constexpr int FL_THREADS = 1024, FL_WAVES = 16;
constexpr int FL_STAGE = 2 * KV_STAGE;        // K then V of one patch-head
constexpr int FL_ONES = 2 * FL_STAGE;         // ones page behind the two stages
constexpr int FL_IDX = FL_ONES + ONES_BYTES;  // per stage: slot -> row index of the resident patch (1024 x int32)
constexpr int FL_CTL = FL_IDX + 2 * 4096;
constexpr int SMEM_FL = FL_CTL + 64 + 2 * FL_WAVES * 4;
constexpr unsigned FL_SPIN_LIMIT = 1u << 22;  // x ~0.3 us: a lost wake-up ends the launch instead of hanging it

struct FlCtl {
  unsigned word[2];  // (position + 1) << 8 | 0x80 while the stage is being loaded | tasks claimed
  unsigned done[2];  // tasks finished
  float kmax2[2];    // largest squared key norm of the resident patch-head
  unsigned err;
  unsigned pad;
};

struct FlItem {  // wave-uniform description of one patch-head as far as THIS workgroup processes it
  int ps, L, nkt, head, qlo, ntask;
};
struct FlTask {
  int item, pos, qt;  // item < 0: none
  FlItem I;
};

__device__ unsigned g_attn_err;  // set if a wave gave up waiting for a stage (cdseg_attention_status)

__global__ __launch_bounds__(FL_THREADS) void attn_bf16_flow_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  FlCtl* ctl = reinterpret_cast<FlCtl*>(smem + FL_CTL);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.num_heads, QC = p.qsplit;  // QC = halves per patch-head (2 when a patch has more than 16 query tiles)
  const int U = p.num_patches * H * QC;
  int u0, u1;
  {
    const int nb = gridDim.x, b = blockIdx.x;
    if (nb < 8) {
      u0 = (int)(((long)U * b) / nb); u1 = (int)(((long)U * (b + 1)) / nb);
    } else {
      const int x = b & 7, j = b >> 3, nbx = (nb - x + 7) >> 3;
      const int r0 = (int)(((long)U * x) >> 3), r1 = (int)(((long)U * (x + 1)) >> 3);
      u0 = r0 + (int)(((long)(r1 - r0) * j) / nbx);
      u1 = r0 + (int)(((long)(r1 - r0) * (j + 1)) / nbx);
    }
    if (u0 >= u1) return;
  }
#ifdef CDSEG_ATTN_TIMING
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t_loop = 0, t_pub = 0, t_spin = 0, n_task = 0;
#endif
  ATTN_STAMP(t0);
  const int ql = lane & 31, h = lane >> 5;
  const int kip = lane >> 1, hs = lane & 1;
  const float c = p.scale_log2e;
  // the patch table through the CONSTANT address space: uniform loads from it are scalar loads the compiler never
  // waits for with vmcnt (as plain global loads they turn into vector loads behind the kernel's own stores, and their
  // vmcnt(0) would wait for the LDS-DMAs in flight)
  typedef const __attribute__((address_space(4))) int32_t* const_i32_t;
  const const_i32_t pstart = (const_i32_t)(uintptr_t)p.patch_start;
  const int it_last = (u1 - 1) / QC;

  auto item_info = [&](int it) {
    const int patch = it / H;
    FlItem r;
    r.ps = pstart[patch];
    r.L = pstart[patch + 1] - r.ps;
    r.nkt = (r.L + 31) >> 5;
    r.head = it - patch * H;
    const int lo = max(u0, it * QC) - it * QC, hi = min(u1, (it + 1) * QC) - it * QC;
    r.qlo = lo * FL_WAVES;
    r.ntask = max(0, min(hi * FL_WAVES, r.nkt) - r.qlo);
    return r;
  };
  auto next_item = [&](int it) {  // the next patch-head this workgroup has tasks of (it_last + 1: none)
    do ++it; while (it <= it_last && item_info(it).ntask == 0);
    return it;
  };

  // ---- stage a patch-head: pieces first, first + step, .. (32 keys each): row indices -> LDS-DMA of K and V
  auto stage_pieces = [&](int it, int st, int first, int step) {
    const FlItem I = item_info(it);
    const bf16_t* kb = (const bf16_t*)p.k + I.head * 16;
    const bf16_t* vb = (const bf16_t*)p.v + I.head * 16;
    int* idx = reinterpret_cast<int*>(smem + FL_IDX + st * 4096);
    int g[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int pc = first + i * step;
      g[i] = 0;
      if (pc < I.nkt) g[i] = p.kv_gidx[I.ps + min(pc * 32 + kip, I.L - 1)];  // (wave-uniform branch)
    }
    // every index is waited for HERE (the empty asm reads them all): a compiler-placed vmcnt between two DMAs would
    // wait for the DMA before it as well
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(g[i]));
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int pc = first + i * step;
      if (pc < I.nkt) {
        const int s = pc * 32 + kip;
        if (hs == 0) idx[s] = g[i];
        const void* ksrc = (const char*)g_attn_zero + hs * 16;
        const void* vsrc = ksrc;
        if (s < I.L) {
          ksrc = kb + (long)g[i] * p.ldk + ((hs ^ ((s >> 3) & 1)) << 3);
          vsrc = vb + (long)g[i] * p.ldv + (hs << 3);
        }
        dma16(ksrc, lds_base + st * FL_STAGE + pc * 1024);
        dma16(vsrc, lds_base + st * FL_STAGE + KV_STAGE + pc * 1024);
      }
    }
  };
  auto piece_norms = [&](int it, int st, int first, int step) {  // max |k|^2 over the named pieces (after their DMAs landed)
    const int nkt = item_info(it).nkt;
    float kn2 = 0.f;
    for (int pc = first; pc < nkt; pc += step) {
      float t = sq8_bf16(*reinterpret_cast<const uint4*>(smem + st * FL_STAGE + pc * 1024 + lane * 16));
      t += __shfl_xor(t, 1, 64);  // the key's other half
      kn2 = fmaxf(kn2, t);
    }
    return wave_max(kn2);
  };
  auto lds_load = [](const unsigned* a) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); };
  auto lds_store = [&](unsigned* a, unsigned v) {
    if (lane == 0) __hip_atomic_store(a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  // ---- prologue: the first two patch-heads, staged by all 16 waves (the only workgroup barrier of the launch)
  const int item_a = item_info(u0 / QC).ntask > 0 ? u0 / QC : next_item(u0 / QC);
  if (item_a > it_last) return;
  const int item_b = next_item(item_a);
  {
    if (tid < (int)(sizeof(FlCtl) / 4)) reinterpret_cast<unsigned*>(ctl)[tid] = 0u;
    for (int w = tid; w < ONES_BYTES / 8; w += FL_THREADS)
      *reinterpret_cast<uint2*>(smem + FL_ONES + w * 8) = make_uint2(0x3F80u, 0u);
  }
  // wave w stages pieces w and w + 16 of both patch-heads; the per-wave norms meet in LDS behind the barrier
  float (*s_kn2)[FL_WAVES] = reinterpret_cast<float (*)[FL_WAVES]>(smem + FL_CTL + 64);
  stage_pieces(item_a, 0, wave, FL_WAVES);
  if (item_b <= it_last) stage_pieces(item_b, 1, wave, FL_WAVES);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  {
    const float na = piece_norms(item_a, 0, wave, FL_WAVES);
    const float nb2 = item_b <= it_last ? piece_norms(item_b, 1, wave, FL_WAVES) : 0.f;
    if (lane == 0) { s_kn2[0][wave] = na; s_kn2[1][wave] = nb2; }
  }
  __syncthreads();
  if (wave == 0) {
    float a = lane < FL_WAVES ? s_kn2[0][lane] : 0.f, b = lane < FL_WAVES ? s_kn2[1][lane] : 0.f;
    a = wave_max(a); b = wave_max(b);
    if (lane == 0) { ctl->kmax2[0] = a; ctl->kmax2[1] = b; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    lds_store(&ctl->word[0], 1u << 8);
    if (item_b <= it_last) lds_store(&ctl->word[1], 2u << 8);
  }

  // ---- task claiming.  Cursor = (item, position): the patch-head this wave claims from next.
  int c_item = item_a, c_pos = 0;
  FlItem c_info = item_info(item_a);
  const int rank = wave >> 2;  // age order of the four waves of a SIMD (waves w, w + 4, w + 8, w + 12): 0 issues first
  // slow (young) waves leave the last tasks of a patch-head to the fast ones and move on to the next patch-head: a
  // task started late by a wave that gets a third of the issue slots is what the stage's recycling ends up waiting for
  const int late = (p.dbg & 128) ? 0 : (rank == 3 ? 12 : rank == 2 ? 5 : rank == 1 ? 2 : 0);
  auto advance = [&]() {
    c_item = next_item(c_item);
    ++c_pos;
    if (c_item <= it_last) c_info = item_info(c_item);
  };
  // 0: claimed (t valid), 1: would block (stage not published yet), 2: no work left
  auto try_claim = [&](FlTask& t) {
    while (c_item <= it_last) {
      const int st = c_pos & 1;
      const unsigned w = lds_load(&ctl->word[st]);
      const unsigned id = w >> 8;
      if (id == (unsigned)c_pos + 1u) {
        if (w & 0x80u) return 1;
        const FlItem& I = c_info;
        const int cnt = (int)(w & 0x7fu);
        if (cnt >= I.ntask) { advance(); continue; }
        if (I.ntask - cnt <= late && c_item < it_last) {
          // the next patch-head is published and has unclaimed tasks?  then take one of those instead
          const unsigned w2 = lds_load(&ctl->word[st ^ 1]);
          if ((w2 >> 8) == (unsigned)c_pos + 2u && !(w2 & 0x80u)) { advance(); continue; }
        }
        unsigned old = 0u;
        if (lane == 0) {
          old = w;
          __hip_atomic_compare_exchange_strong(&ctl->word[st], &old, w + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == w) { t.item = c_item; t.pos = c_pos; t.qt = I.qlo + cnt; t.I = I; return 0; }
      } else if (id > (unsigned)c_pos + 1u) {  // this patch-head is finished and its stage already recycled
        advance();
      } else {
        return 1;
      }
    }
    return 2;
  };
  auto claim_blocking = [&](FlTask& t) {
    ATTN_STAMP(ts0);
    unsigned spins = 0;
    int r;
    while ((r = try_claim(t)) == 1) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > FL_SPIN_LIMIT) {
        if (lane == 0) { ctl->err = 1u; atomicOr(&g_attn_err, 1u); }
        r = 2;
        break;
      }
    }
#ifdef CDSEG_ATTN_TIMING
    t_spin += __builtin_readcyclecounter() - ts0;
#endif
    return r == 0;
  };
  // the task's query rows: row index from the stage's LDS index slot, one global round trip
  auto load_qrow = [&](const FlTask& t) {
    const FlItem& I = t.I;
    const int* idx = reinterpret_cast<const int*>(smem + FL_IDX + (t.pos & 1) * 4096);
    const int g = idx[min(t.qt * 32 + ql, I.L - 1)];
    return *reinterpret_cast<const uint4*>((const bf16_t*)p.q + (long)g * p.ldq + I.head * 16 + h * 8);
  };

  const bool v_lane = (lane & 16) == 0;  // lane groups 0 / 2 read V, 1 / 3 the ones page
  const unsigned vstep = v_lane ? 1024u : 0u;
  __builtin_amdgcn_s_setprio(2);
  FlTask cur, nxt;
  cur.item = -1;
  if (!claim_blocking(cur)) cur.item = -1;
  uint4 q_cur = make_uint4(0, 0, 0, 0);
  if (cur.item >= 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    q_cur = load_qrow(cur);
  }
  while (cur.item >= 0) {
    const FlItem I = cur.I;
    const int st = cur.pos & 1;
    const int L = I.L, nkt = I.nkt;
    // ---- Q' = Q * (softmax scale * log2 e), rounded to bf16 once (consumed BEFORE the loads below are issued: the
    // compiler's wait for q_cur then has nothing younger in the queue)
    bf16x8_t qf;
    {
      union { bf16x8_t v; uint32_t u[4]; } qs;
      const uint32_t qr[4] = {q_cur.x, q_cur.y, q_cur.z, q_cur.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        qs.u[j] = pack_bf16x2(__uint_as_float(qr[j] << 16) * c, __uint_as_float(qr[j] & 0xffff0000u) * c);
      qf = qs.v;
      asm volatile("" : "+v"(qf));
    }
    // ---- claim the next task now and put its query rows in flight behind this task's key loop
    nxt.item = -1;
    uint4 q_nxt = make_uint4(0, 0, 0, 0);
    if (((p.dbg & 256) || rank < 2) && try_claim(nxt) == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      q_nxt = load_qrow(nxt);
    } else {
      nxt.item = -1;
    }
    const int qslot = cur.qt * 32 + ql;
    const bool qvalid = qslot < L;
    int w = p.widx[I.ps + min(qslot, L - 1)];
    if (!qvalid) w = -1;
    const float kmax2 = ctl->kmax2[st];

    const char* k_lane = smem + st * FL_STAGE + ql * 32 + ((h ^ ((ql >> 3) & 1)) << 4);
    // half wave 0: V rows on banks 0..31 -> its ones word on banks 32..63, and the other way round for half wave 1
    const unsigned va0 = v_lane ? lds_base + st * FL_STAGE + KV_STAGE + (4 * h + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8
                                : lds_base + FL_ONES + (h ? 0 : 128);
    const bool tail = (nkt << 5) != L;
    const int nfull = tail ? nkt - 1 : nkt;  // key tiles that need no masking
    const f32x16_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool boosted = (p.dbg & 64) != 0;  // (experimental builds: 64 = no straggler boost)
    auto exp_pv_pass = [&](float mrow) {
      const float nm = -mrow;
      const f32x16_t negm = {nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm};
      f32x16_t acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      const char* kp = k_lane;
      unsigned va = va0;
      int kt = 0;
      for (; kt + 1 < nfull; kt += 2) {  // two independent tiles in flight
        // a patch-head with no unclaimed task left is waited for (its stage cannot be recycled before its last task
        // ends): the waves still working on it go to the front of the SIMD's issue order
        if (!boosted && (kt & 15) == 0 && (int)(lds_load(&ctl->word[st]) & 0x7fu) >= I.ntask) {
          __builtin_amdgcn_s_setprio(3);
          boosted = true;
        }
        const f32x16_t sa = qk_tile(kp, qf, negm);
        const f32x16_t sb = qk_tile(kp + 1024, qf, negm);
        pv_tile<false>(sa, kt, h, L, va, acc);
        pv_tile<false>(sb, kt + 1, h, L, va + 1024, acc);
        kp += 2048;
        va += 2 * vstep;
      }
      for (; kt < nkt; ++kt) {
        const f32x16_t s = qk_tile(kp, qf, negm);
        if (kt >= nfull) pv_tile<true>(s, kt, h, L, va, acc);
        else pv_tile<false>(s, kt, h, L, va, acc);
        kp += 1024;
        va += vstep;
      }
      return acc;
    };
    float qn2 = sq8_bf16(__builtin_bit_cast(uint4, qf));
    qn2 += __shfl_xor(qn2, 32, 64);
    ATTN_STAMP(tl0);
    __builtin_amdgcn_s_setprio(0);
    f32x16_t o = exp_pv_pass(sqrtf(qn2 * kmax2) * 1.0005f);  // single pass, Cauchy-Schwarz shift (see the block form)
    // everything between two key loops (epilogue, claims, staging) is short and latency-bound: it goes to the front of
    // the issue order, or a young wave spends as long there as in its key loop
    __builtin_amdgcn_s_setprio(2);
#ifdef CDSEG_ATTN_TIMING
    asm volatile("" ::"v"(o[0]), "v"(o[8]));
    t_loop += __builtin_readcyclecounter() - tl0;
    ++n_task;
#endif
    const bool loose = qvalid && !(__shfl(o[8], ql, 64) >= 8.6736174e-19f);
    if (__any(loose)) {  // bound looser than 2^60: redo with the exact row max (never taken for ordinary logits)
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
      o = exp_pv_pass(m0);
    }
    const float lsum = __shfl(o[8], ql, 64);
    const float inv = 1.0f / lsum;
    if (w >= 0) {
      bf16_t* orow = (bf16_t*)p.out + (long)w * p.ldo + I.head * 16 + 4 * h;
      uint2 a, b;
      a.x = pack_bf16x2(o[0] * inv, o[1] * inv);
      a.y = pack_bf16x2(o[2] * inv, o[3] * inv);
      b.x = pack_bf16x2(o[4] * inv, o[5] * inv);
      b.y = pack_bf16x2(o[6] * inv, o[7] * inv);
      *reinterpret_cast<uint2*>(orow) = a;      // d = 4h .. 4h+3
      *reinterpret_cast<uint2*>(orow + 8) = b;  // d = 8+4h .. 8+4h+3
    }
    // ---- this task is finished: its LDS reads are done (the accumulator they fed has been consumed above).  The
    // finisher of a patch-head's last task recycles the stage for the patch-head two positions ahead.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    unsigned fin = 0u;
    if (lane == 0) fin = __hip_atomic_fetch_add(&ctl->done[st], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    fin = __builtin_amdgcn_readfirstlane(fin);
    if ((int)fin == I.ntask - 1) {
      const int target = next_item(next_item(cur.item));
      if (target <= it_last) {
        ATTN_STAMP(tp0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // every finisher's reads are behind its release
        lds_store(&ctl->done[st], 0u);
        lds_store(&ctl->word[st], ((unsigned)cur.pos + 3u) << 8 | 0x80u);
        stage_pieces(target, st, 0, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float kn2 = piece_norms(target, st, 0, 1);
        if (lane == 0) ctl->kmax2[st] = kn2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        lds_store(&ctl->word[st], ((unsigned)cur.pos + 3u) << 8);
#ifdef CDSEG_ATTN_TIMING
        t_pub += __builtin_readcyclecounter() - tp0;
#endif
      }
    }
    if (nxt.item < 0) {
      if (!claim_blocking(nxt)) break;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      q_nxt = load_qrow(nxt);
    }
    cur = nxt;
    q_cur = q_nxt;
  }
#ifdef CDSEG_ATTN_TIMING
  if (lane == 0 && blockIdx.x < 2048) {
    unsigned long long* d = g_attn_t + ((size_t)blockIdx.x * 16 + wave) * 8;
    d[0] = rt0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = t_pub; d[3] = t_loop;
    d[4] = __builtin_readcyclecounter() - t0; d[5] = n_task; d[6] = t_spin; d[7] = 0;
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
  int patch, head, qslice;
  if (!decode_block(p, patch, head, qslice)) return;
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

  for (int qt = qslice * ATTN_WAVES + wave; qt < nqt; qt += p.qsplit * ATTN_WAVES) {
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

}  // namespace

// 0 if no attention launch so far gave up waiting inside the kernel (synchronises the device; tests and benchmarks call it)
extern "C" int cdseg_attention_status(void) {
  unsigned v = 0;
  if (hipDeviceSynchronize() != hipSuccess) return CDSEG_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_attn_err), sizeof(v)) != hipSuccess) return CDSEG_ERR_LAUNCH;
  return v ? CDSEG_ERR_LAUNCH : CDSEG_OK;
}

#ifdef CDSEG_ATTN_TIMING
extern "C" int cdseg_debug_attn_timing(unsigned long long* host_dst, size_t count) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_attn_t), count * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int cdseg_attention(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv,
                               const int32_t* q_gidx, const int32_t* kv_gidx, const int32_t* widx,
                               const int32_t* patch_start, int num_patches, int num_heads, int max_len, float scale,
                               void* out, int ldo, int dtype, void* stream) {
  if (num_patches <= 0 || num_heads <= 0) return CDSEG_OK;
  if (max_len <= 0 || max_len > CDSEG_MAX_PATCH) return CDSEG_ERR_UNSUPPORTED;
  const int esz = dtype == CDSEG_F32 ? 4 : 2;
  // 16-byte alignment of every gathered row slice
  if (((long)ldq * esz) & 15 || ((long)ldk * esz) & 15 || ((long)ldv * esz) & 15) return CDSEG_ERR_ARG;
  if (dtype == CDSEG_F32 ? (ldo & 3) : (ldo & 3)) return CDSEG_ERR_ARG;
  AttnP p;
  p.q = q; p.k = k; p.v = v; p.q_gidx = q_gidx; p.kv_gidx = kv_gidx; p.widx = widx; p.patch_start = patch_start;
  p.out = out; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.num_heads = num_heads;
  p.scale_log2e = scale * 1.44269504088896340736f;
  p.dbg = 0;
#ifdef CDSEG_EXPERIMENTS
  if (const char* e = getenv("CDSEG_ATTN_DBG")) p.dbg = atoi(e);
#endif
  hipStream_t s = (hipStream_t)stream;
  // K/V staging is per block, so split a patch-head's queries over as few blocks as still fill
  // the chip (2 resident blocks per CU -> ~512 block slots)
  const int tile = dtype == CDSEG_F32 ? 16 : 32;
  const int nqt = (max_len + tile - 1) / tile;
  // Powers of two so that every wave of every block gets the same number of query tiles.
  const int ph = num_patches * num_heads;
  int qsplit = ph >= 384 ? 1 : (ph >= 160 ? 2 : 4);
#ifdef CDSEG_EXPERIMENTS
  if (const char* e = getenv("CDSEG_ATTN_QSPLIT")) qsplit = atoi(e);  // tuning knob (power of two)
#endif
  const int max_split = (nqt + ATTN_WAVES - 1) / ATTN_WAVES;
  while (qsplit > 1 && qsplit > max_split) qsplit >>= 1;
  p.num_patches = num_patches;
  p.qsplit = qsplit;
  int hgroups = 1;  // smallest divisor of H giving every XCD a group (or H itself)
  while (hgroups < num_heads && (num_patches * hgroups < 8 || num_heads % hgroups)) ++hgroups;
  p.hgroups = hgroups;
  const int groups = num_patches * hgroups;
  dim3 grid((unsigned)(((groups + 7) / 8) * 8 * (num_heads / hgroups) * qsplit)), block(ATTN_THREADS);
  static int num_cus = 0;
  if (!num_cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return CDSEG_ERR_LAUNCH;
    if (hipFuncSetAttribute((const void*)attn_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BF16) !=
            hipSuccess ||
        hipFuncSetAttribute((const void*)attn_bf16_flow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_FL) !=
            hipSuccess ||
        hipFuncSetAttribute((const void*)attn_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_F32) !=
            hipSuccess)
      return CDSEG_ERR_LAUNCH;
    num_cus = prop.multiProcessorCount;
  }
  if (dtype != CDSEG_BF16 && dtype != CDSEG_F32) return CDSEG_ERR_ARG;
  CdsegProfToken tok;
  const bool prof = cdseg_prof_begin(CDSEG_PROF_ATTENTION, s, &tok);
  // bf16 self-attention: the persistent dataflow form (one 16-wave workgroup per CU walks a run of patch-heads); cross
  // attention (separate q / kv slot plans: one small launch per forward) and f32: one block per (patch, head, slice)
  int form = q_gidx == kv_gidx ? 1 : 0;
#ifdef CDSEG_EXPERIMENTS
  if (const char* e = getenv("CDSEG_ATTN_FORM")) form = atoi(e) && q_gidx == kv_gidx;
#endif
  if (dtype == CDSEG_BF16 && form == 1) {
    const int qc = nqt > FL_WAVES ? 2 : 1;  // halves per patch-head (the unit the launch is balanced in)
    p.qsplit = qc;
    const long units = (long)ph * qc;
    const unsigned blocks = (unsigned)(units < num_cus ? units : num_cus);
    hipLaunchKernelGGL(attn_bf16_flow_kernel, dim3(blocks), dim3(FL_THREADS), SMEM_FL, s, p);
  } else if (dtype == CDSEG_BF16) {
    hipLaunchKernelGGL(attn_bf16_kernel, grid, block, SMEM_BF16, s, p);
  } else {
    hipLaunchKernelGGL(attn_f32_kernel, grid, block, SMEM_F32, s, p);
  }
  if (prof) cdseg_prof_end(tok, s);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}
