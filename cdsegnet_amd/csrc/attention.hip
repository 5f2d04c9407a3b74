// Serialized window attention for CDSegNet / PTv3 on gfx950 (head dim 16, patch <= 1024).
//
// ref: pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py
//      :246-296 SerializedAttention (flash_attn_varlen_qkvpacked_func :282-288, CPU branch :264-280)
//      :988-1055 SerializedCrossAttention (flash_attn_varlen_kvpacked_func :1038-1047)
//
// One workgroup (8 waves) = one (patch, head) x one slice of its queries.
//  * the gather by serialized order is fused into the K/V staging loads and the Q fragment
//    loads (row indices come from the slot plan, cdseg_pad_plan); the scatter by the inverse
//    order and the dropping of the padding duplicates are fused into the store;
//  * the whole K tile and V^T tile of the patch-head live in LDS for the lifetime of the block
//    (bf16: 32 KB + 34 KB -> 2 blocks / CU; f32: 64 KB + 65 KB);
//  * scores are computed TRANSPOSED (S^T = K Q^T) so that a query's scores stay inside one lane
//    pair: row statistics need no cross-lane traffic in the key loop;
//  * bf16: ONE pass over the keys.  Softmax is shift invariant, so instead of the row max the kernel subtracts the
//    Cauchy-Schwarz bound |q'_i| max_j |k_j| (one norm per query, one max over the keys while staging K): no max
//    sweep, no second QK^T.  Scale * log2(e) is folded into Q', the shift rides in the MFMA's C operand, so a score
//    costs ONE v_exp_f32.  Query tiles whose bound is too loose (> 2^60) are redone with the exact row max;
//  * the kernel is bound by the VALU, not by the matrix pipe: 16 v_exp_f32 (quarter rate, ~8.7 cycles each with 4
//    waves per SIMD) + 8 packs per 32x32 score tile against 3 MFMAs of 32 cycles - tools/ubench/pipes.hip: the
//    transcendental unit shares the VALU issue (exp + fma times ADD), the matrix pipe overlaps both, and the
//    instruction mix alone tops out at ~30 % of the bf16 MFMA peak;
//  * bf16: v_mfma_f32_32x32x16_bf16 for both products; the MFMA k-slot <-> key assignment of
//    the PV product is chosen so the exponentiated scores feed it straight from the
//    accumulator registers (no permute); V^T is STORED in that key order, so a PV operand is one ds_read_b128;
//    a row of ones appended to V^T makes the softmax denominator fall out of the same MFMA (row 16 of the result);
//  * f32 (the 1e-3 parity mode): v_mfma_f32_16x16x4_f32 for both products, exact fp32.
// LDS layouts are bank-conflict free for every fragment read (tools/lds_conflicts.py).
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
};

// XCD-aware block -> (patch, head, query-slice) map.  Workgroups are dispatched round-robin over the 8 XCDs
// (block b -> XCD b % 8, a performance-only assumption); all blocks of one patch (its heads and query slices
// read the same gathered rows) get ids that are equal mod 8 and adjacent in time, so the rows are fetched into
// ONE L2 once instead of up to 8 times.  With fewer than 8 patches (deep stages) the heads of a patch are
// split into `hgroups` groups so that all XCDs still get work.  Grid = ceil(G / 8) * 8 * (H / hgroups) * Q
// blocks with G = P * hgroups; surplus ids exit.
__device__ __forceinline__ bool decode_block(const AttnP& p, int& patch, int& head, int& qslice) {
  const int hpg = p.num_heads / p.hgroups;  // heads per group
  const int per_group = hpg * p.qsplit;
  const int L = blockIdx.x;
  const int xcd = L & 7;
  const int t = L >> 3;
  const int r = t % per_group;
  const int g = (t / per_group) * 8 + xcd;
  patch = g / p.hgroups;
  const int hg = g - patch * p.hgroups;
  head = hg * hpg + r / p.qsplit;
  qslice = r % p.qsplit;
  return patch < p.num_patches;
}

// V^T rows: 1024 bf16 + 16 B pad.  Row stride = 516 dwords = 4 mod 64 banks: the 16 rows a ds_read_b128 lane group
// touches sit on 16 distinct 4-bank slots -> conflict free.  Inside every 32-key tile the keys are stored in the
// order the PV MFMA's k-slots want them (vt_pos): a lane's 8 keys are ONE 16-byte read (was two ds_read2_b64 halves:
// 36 -> 12 LDS cycles per tile; the tile loop was 77 % LDS-busy next to the VALU-bound softmax).
constexpr int VT_STRIDE_BF16 = 2064;
constexpr int VT_ROWS_BF16 = 17;      // 16 head dims + ones row (MFMA rows 17..31 are never read: their lanes load row 16)
constexpr int KS_BYTES_BF16 = 1024 * 32;
constexpr int SMEM_BF16 = KS_BYTES_BF16 + VT_ROWS_BF16 * VT_STRIDE_BF16;

constexpr int VT_STRIDE_F32 = 4128;  // 1024 f32 + 32 B pad
constexpr int KS_BYTES_F32 = 1024 * 64;
constexpr int SMEM_F32 = KS_BYTES_F32 + 16 * VT_STRIDE_F32;

// ------------------------------------------------------------------------------------ bf16
constexpr int ATTN_THREADS = 512;  // 8 waves: 2 per SIMD per block, 2 blocks per CU (LDS 69 KB each)
constexpr int ATTN_WAVES = ATTN_THREADS / 64;

// one 32-key x 32-query tile of S^T = K Q'^T + C  (Q' = Q * scale * log2 e, C = 0 or -max: see below)
__device__ __forceinline__ f32x16_t qk_tile(const char* Ks, int kt, int ql, int h, bf16x8_t qf, const f32x16_t& c0) {
  const int key = kt * 32 + ql;
  const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + key * 32 + ((h ^ ((key >> 3) & 1)) << 4));
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

// position of key slot s inside its V^T row: within a 32-key tile the 4-key blocks go 0,2,1,3,4,6,5,7, so that the
// k-slots of PV MFMA mf for lane half h (keys 16mf + 4h + {0..3} and 16mf + 8 + 4h + {0..3}: what the exponentiated
// S^T accumulator registers hold, see pv_tile) are the 8 consecutive positions 16mf + 8h ..
__device__ __forceinline__ int vt_pos(int s) {
  const int b = (s >> 2) & 7;
  return (s & ~31) | ((b & 4) | ((b & 1) << 1) | ((b >> 1) & 1)) << 2 | (s & 3);
}

// two fp32 -> packed bf16 by truncation: ONE v_perm_b32 (measured 2.8x cheaper than v_cvt_pk_bf16_f32 on
// gfx950, tools/ubench/valu_rates.hip).  The softmax denominator is accumulated from the SAME truncated
// values (row of ones in V^T), so the truncation bias cancels in the normalisation.
__device__ __forceinline__ uint32_t pack_bf16x2_trunc(float lo, float hi) {
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}

// P = exp2(S') for one tile (S' already holds s*c - m*c: scale folded into Q', -max into the MFMA's C
// operand, so the softmax costs one v_exp_f32 per score and nothing else), then
// O^T += [V^T; 1; 0] P^T (two K=16 MFMAs)
template <bool TAIL>
__device__ __forceinline__ void pv_tile(const f32x16_t& s, int kt, int h, int L, const char* vt_lane, f32x16_t& o) {
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
    union { bf16x8_t v; uint32_t u[4]; } pf, vf;
#pragma unroll
    for (int j = 0; j < 4; ++j) pf.u[j] = pack_bf16x2_trunc(pr[8 * mf + 2 * j], pr[8 * mf + 2 * j + 1]);
    // k-slots 8h+j (j<4) <-> keys kbase + 4h + j ; (j>=4) <-> keys kbase + 8 + 4h + (j-4): stored contiguously (vt_pos)
    vf.v = *reinterpret_cast<const bf16x8_t*>(vt_lane + (kt * 32 + 16 * mf) * 2);
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, o, 0, 0, 0);
  }
}

__global__ __launch_bounds__(ATTN_THREADS) void attn_bf16_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vt = smem + KS_BYTES_BF16;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  int patch, head, qslice;
  if (!decode_block(p, patch, head, qslice)) return;
  const int ps = p.patch_start[patch];
  const int L = p.patch_start[patch + 1] - ps;
  const int nkt = (L + 31) >> 5;  // 32-key tiles
  const int Lp = nkt << 5;
  const bf16_t* kb = (const bf16_t*)p.k + head * 16;
  const bf16_t* vb = (const bf16_t*)p.v + head * 16;

  // largest squared key norm of the patch-head (for the score bound of the single-pass softmax below)
  __shared__ unsigned s_kmax2;
  if (tid == 0) s_kmax2 = 0u;
  __syncthreads();
  float kn2max = 0.f;
  auto sq8 = [](const uint4& a) {
    float t = 0.f;
    const uint32_t u[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = __uint_as_float(u[j] << 16), hi = __uint_as_float(u[j] & 0xffff0000u);
      t = fmaf(lo, lo, fmaf(hi, hi, t));
    }
    return t;
  };

  // ---- stage K (row-major, 16-B halves swizzled) and V^T (+ ones row, zero row).
  // One thread per key PAIR: 8 independent 16-B gathers in flight, V^T written as packed dwords.
  for (int pr = tid; pr < (Lp >> 1); pr += ATTN_THREADS) {
    const int s0 = 2 * pr, s1 = s0 + 1;
    uint4 k0[2], k1[2], v0[2], v1[2];
    k0[0] = k0[1] = k1[0] = k1[1] = v0[0] = v0[1] = v1[0] = v1[1] = make_uint4(0, 0, 0, 0);
    if (s0 < L) {
      const long g = p.kv_gidx[ps + s0];
      const uint4* kr = reinterpret_cast<const uint4*>(kb + g * p.ldk);
      const uint4* vr = reinterpret_cast<const uint4*>(vb + g * p.ldv);
      k0[0] = kr[0]; k0[1] = kr[1]; v0[0] = vr[0]; v0[1] = vr[1];
    }
    if (s1 < L) {
      const long g = p.kv_gidx[ps + s1];
      const uint4* kr = reinterpret_cast<const uint4*>(kb + g * p.ldk);
      const uint4* vr = reinterpret_cast<const uint4*>(vb + g * p.ldv);
      k1[0] = kr[0]; k1[1] = kr[1]; v1[0] = vr[0]; v1[1] = vr[1];
    }
    kn2max = fmaxf(kn2max, fmaxf(sq8(k0[0]) + sq8(k0[1]), sq8(k1[0]) + sq8(k1[1])));
    const int sw = (s0 >> 3) & 1;  // same for s1 (s0 even)
    *reinterpret_cast<uint4*>(Ks + s0 * 32 + ((0 ^ sw) << 4)) = k0[0];
    *reinterpret_cast<uint4*>(Ks + s0 * 32 + ((1 ^ sw) << 4)) = k0[1];
    *reinterpret_cast<uint4*>(Ks + s1 * 32 + ((0 ^ sw) << 4)) = k1[0];
    *reinterpret_cast<uint4*>(Ks + s1 * 32 + ((1 ^ sw) << 4)) = k1[1];
    const uint32_t a[8] = {v0[0].x, v0[0].y, v0[0].z, v0[0].w, v0[1].x, v0[1].y, v0[1].z, v0[1].w};
    const uint32_t b[8] = {v1[0].x, v1[0].y, v1[0].z, v1[0].w, v1[1].x, v1[1].y, v1[1].z, v1[1].w};
    const int vp = vt_pos(s0) * 2;  // s0 is even: s0 + 1 is the next position too
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      *reinterpret_cast<uint32_t*>(Vt + (2 * d) * VT_STRIDE_BF16 + vp) = (a[d] & 0xffffu) | (b[d] << 16);
      *reinterpret_cast<uint32_t*>(Vt + (2 * d + 1) * VT_STRIDE_BF16 + vp) = (a[d] >> 16) | (b[d] & 0xffff0000u);
    }
    *reinterpret_cast<uint32_t*>(Vt + 16 * VT_STRIDE_BF16 + vp) = (s0 < L ? 0x3F80u : 0u) | (s1 < L ? 0x3F800000u : 0u);
  }
  kn2max = wave_max(kn2max);
  if (lane == 0) atomicMax(&s_kmax2, __float_as_uint(kn2max));  // non-negative floats order like their bit patterns
  __syncthreads();
  const float kmax2 = __uint_as_float(s_kmax2);

  const int ql = lane & 31;  // query (B operand column) / key or head-dim row (A operand row)
  const int h = lane >> 5;
  const int vrow = ql < 16 ? ql : 16;  // rows 17..31 of O^T are never read: their lanes reload the ones row (broadcast)
  const char* vt_lane = Vt + vrow * VT_STRIDE_BF16 + h * 16;
  const float c = p.scale_log2e;
  const int nqt = (L + 31) >> 5;
  const bool tail = (nkt << 5) != L;
  const int nfull = tail ? nkt - 1 : nkt;  // key tiles that need no masking

  for (int qt = qslice * ATTN_WAVES + wave; qt < nqt; qt += p.qsplit * ATTN_WAVES) {
    const int qslot = qt * 32 + ql;
    const bool qvalid = qslot < L;
    bf16x8_t qf = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qvalid) {
      // Q' = Q * (softmax scale * log2 e), rounded to bf16 once: scores come out of the MFMA in exp2 units
      const long g = p.q_gidx[ps + qslot];
      union { bf16x8_t v; uint32_t u[4]; } qr, qs;
      qr.v = *reinterpret_cast<const bf16x8_t*>((const bf16_t*)p.q + g * p.ldq + head * 16 + h * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        qs.u[j] = pack_bf16x2(__uint_as_float(qr.u[j] << 16) * c, __uint_as_float(qr.u[j] & 0xffff0000u) * c);
      qf = qs.v;
    }
    const f32x16_t zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // P = exp2(S' - m), O^T (+ row sums in row 16) += [V^T; 1; 0] P^T, with S' - m straight out of the MFMA
    // (C operand = -m, loop invariant)
    auto exp_pv_pass = [&](float mrow) {
      const float nm = -mrow;
      const f32x16_t negm = {nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm};
      f32x16_t acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      int kt = 0;
      for (; kt + 1 < nfull; kt += 2) {  // two independent tiles in flight
        const f32x16_t sa = qk_tile(Ks, kt, ql, h, qf, negm);
        const f32x16_t sb = qk_tile(Ks, kt + 1, ql, h, qf, negm);
        pv_tile<false>(sa, kt, h, L, vt_lane, acc);
        pv_tile<false>(sb, kt + 1, h, L, vt_lane, acc);
      }
      for (; kt < nkt; ++kt) {
        const f32x16_t s = qk_tile(Ks, kt, ql, h, qf, negm);
        if (kt >= nfull) pv_tile<true>(s, kt, h, L, vt_lane, acc);
        else pv_tile<false>(s, kt, h, L, vt_lane, acc);
      }
      return acc;
    };
    // ---- single pass: softmax is shift invariant, so any m >= max_j s_ij that does not underflow the row works.
    // Cauchy-Schwarz gives one for free: s_ij <= |q'_i| * max_j |k_j|.  It replaces the row-max pass (a second QK^T
    // MFMA sweep + a v_max3 per score pair; the kernel is VALU-issue bound).  P keeps its relative precision at
    // any magnitude (fp32 / bf16 share the exponent range, the denominator comes from the same truncated values).
    float qn2 = 0.f;
    {
      union { bf16x8_t v; uint32_t u[4]; } qq;
      qq.v = qf;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = __uint_as_float(qq.u[j] << 16), hi = __uint_as_float(qq.u[j] & 0xffff0000u);
        qn2 = fmaf(lo, lo, fmaf(hi, hi, qn2));
      }
      qn2 += __shfl_xor(qn2, 32, 64);
    }
    f32x16_t o = exp_pv_pass(sqrtf(qn2 * kmax2) * 1.0005f);
    // rows whose bound is looser than 2^60 (the largest term could sink towards the denormal range) are redone
    // with the exact row max; wave-uniform branch, never taken for ordinary logits
    const bool loose = qvalid && !(__shfl(o[8], ql, 64) >= 8.6736174e-19f);
    if (__any(loose)) {
      // ---- exact pass 1: row max of S'^T = K Q'^T (lane (q,h) sees keys (r&3) + 8*(r>>2) + 4h of each tile)
      float m0 = -INFINITY, m1 = -INFINITY;
      int kt = 0;
      for (; kt + 1 < nfull; kt += 2) {
        const f32x16_t sa = qk_tile(Ks, kt, ql, h, qf, zero16);
        const f32x16_t sb = qk_tile(Ks, kt + 1, ql, h, qf, zero16);
        m0 = tile_max(sa, m0);
        m1 = tile_max(sb, m1);
      }
      for (; kt < nkt; ++kt) {
        f32x16_t s = qk_tile(Ks, kt, ql, h, qf, zero16);
        if (kt >= nfull) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= L) s[r] = -INFINITY;
        }
        m0 = tile_max(s, m0);
      }
      float m = fmaxf(m0, m1);
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      o = exp_pv_pass(m);
    }
    // ---- epilogue: O^T rows (r&3) + 8*(r>>2) + 4h; row 16 (lane h=0, r=8) is the denominator
    const float lsum = __shfl(o[8], ql, 64);
    const float inv = 1.0f / lsum;
    if (qvalid) {
      const int w = p.widx[ps + qslot];
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
    }
  }
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
  hipStream_t s = (hipStream_t)stream;
  // K/V staging is per block, so split a patch-head's queries over as few blocks as still fill
  // the chip (2 resident blocks per CU -> ~512 block slots)
  const int tile = dtype == CDSEG_F32 ? 16 : 32;
  const int nqt = (max_len + tile - 1) / tile;
  // Powers of two so that every wave of every block gets the same number of query tiles.
  const int ph = num_patches * num_heads;
  int qsplit = ph >= 384 ? 1 : (ph >= 160 ? 2 : 4);
  if (const char* e = getenv("CDSEG_ATTN_QSPLIT")) qsplit = atoi(e);  // tuning knob (power of two)
  const int max_split = (nqt + ATTN_WAVES - 1) / ATTN_WAVES;
  while (qsplit > 1 && qsplit > max_split) qsplit >>= 1;
  p.num_patches = num_patches;
  p.qsplit = qsplit;
  int hgroups = 1;  // smallest divisor of H giving every XCD a group (or H itself)
  while (hgroups < num_heads && (num_patches * hgroups < 8 || num_heads % hgroups)) ++hgroups;
  p.hgroups = hgroups;
  const int groups = num_patches * hgroups;
  dim3 grid((unsigned)(((groups + 7) / 8) * 8 * (num_heads / hgroups) * qsplit)), block(ATTN_THREADS);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)attn_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BF16) !=
            hipSuccess ||
        hipFuncSetAttribute((const void*)attn_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_F32) !=
            hipSuccess)
      return CDSEG_ERR_LAUNCH;
    attr_done = true;
  }
  if (dtype != CDSEG_BF16 && dtype != CDSEG_F32) return CDSEG_ERR_ARG;
  CdsegProfToken tok;
  const bool prof = cdseg_prof_begin(CDSEG_PROF_ATTENTION, s, &tok);
  if (dtype == CDSEG_BF16)
    hipLaunchKernelGGL(attn_bf16_kernel, grid, block, SMEM_BF16, s, p);
  else
    hipLaunchKernelGGL(attn_f32_kernel, grid, block, SMEM_F32, s, p);
  if (prof) cdseg_prof_end(tok, s);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}
