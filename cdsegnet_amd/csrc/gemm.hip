// MFMA GEMM with fused epilogues for the CDSegNet hot path (gfx950).
//
//   out = epilogue(A @ W^T)        nn.Linear             (ref: ptv3.py:170-171, 310-313, 463, 597-599, 1562)
//   out = epilogue(sum_o A[nbr[:,o]] @ W[:,o,:]^T)       spconv.SubMConv3d as a gathered-A GEMM
//                                                        (ref call sites: ptv3.py:356-362, 1106-1124)
//
// Workgroup = 4 waves (2 x 2) -> one 64 x BN output tile.
//   bf16: v_mfma_f32_16x16x32_bf16;   f32 (parity mode): v_mfma_f32_16x16x4_f32 (exact fp32).
// K runs in steps of one LDS row (RB bytes = 64 / 128 / 256: 32..128 bf16 or 32..64 f32 per step);
// the wide step is what makes the deep sparse-conv reductions (K = 27 * C up to 13824) and the
// C >= 128 linears run from a handful of barrier-separated iterations instead of hundreds.
// A / W K-tiles are staged through LDS in 16-byte chunks with an XOR swizzle that makes every
// ds_read_b128 of an MFMA fragment bank-conflict free (tools/lds_conflicts.py); global loads of
// step k+1 are in flight during the MFMAs of step k (register double buffering).
// (Measured on the 480k-point stage-0 conv, profiles/r01l_pmc_conv_attention.txt: HBM traffic = the compulsory
// 82 MB, VALU 11 %, ~60 % of the wave cycles idle: a tile is a chain of dependent index -> row round trips.  Issuing
// the loads of 2-4 K steps at once shortened the chain (196 -> 180 us) but the extra registers cost more overlap
// with the other streams than that gained (37.5 -> 34.3 M points/s end to end), so the loop keeps one step in flight.
// What did help: 128-row tiles (8 waves, same per-wave tile), see launch_bn - but only once the LayerNorm epilogue
// was compiled out of that variant: with it the kernel spilled 165 VGPRs and lost 8 % end to end.)
// Sparse conv: the block first compacts the kernel offsets that ANY of its 64 rows has a
// neighbour at (on z-ordered points: ~10-15 of 27) and reduces over those only; A chunks are
// gathered through the neighbour table per 16-byte chunk, so one K step can span several offsets
// (C = 32: four offsets per 128-wide step).
// Epilogue: the accumulator tile goes through LDS so that bias / folded BatchNorm / GELU /
// residual / un-pooling gather-add / second typed copy / row scatter run on row-contiguous
// float4 groups and every store is a full 16-byte (8-byte for bf16) coalesced access.
#include <cstdlib>

#include <atomic>

#include "common.h"
#include "prof.h"

// row-count window in which a C = 128 sparse conv takes the 8-wave 256 x 128 tile (MIN 0 = never).  Measured
// (tools/bench_conv.py 2 <scenes>, profiles/r05_conv128.txt): 14 k rows (one scene) 42.2 -> 37.2 us, but 114 k rows (8 scenes)
// 118 -> 141 us and 342 k rows 315 -> 365 us - with enough 128 x 128 tiles to give every CU two blocks the smaller tile wins
#ifndef CDSEG_CONV_SQ128_MIN_M
#define CDSEG_CONV_SQ128_MIN_M 8000
#endif
#ifndef CDSEG_CONV_SQ128_MAX_M
#define CDSEG_CONV_SQ128_MAX_M 32768
#endif

namespace {

struct GemmP {
  const void* A;
  const void* W;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* res;
  const float* add_src;
  const int32_t* add_idx;
  const int32_t* nbr;
  const int32_t* out_idx;
  void* out;
  void* out2;
  long M;
  int N, K, kvol;
  int lda, ldo, ldo2, ldres, ldadd;
  int out_dtype, out2_dtype;
  int act, out2_pre_add;
  int vec_ok;  // all row strides / N multiples of 4 -> float4 epilogue
  float* ws;   // split-K partial sums (splits, M, N) fp32, or nullptr
  int splits;
  int kshift;
  long nbr_sm, nbr_so;  // neighbour table strides (row, offset): (kvol, 1) row-major or (1, M) offset-major
  int gm, gn;  // output tiles
  int xmode;   // block -> (tile, split) map: 0 plain, 1 W-slice per XCD, 2 contiguous row range per XCD
  int fix;     // 0: epilogue straight from the accumulators; partial tiles -> ws, then 1: splitk_epilogue_kernel,
               // 2: row_finish_kernel (LayerNorm over rows wider than a column tile, with or without split-K)
  // row-wise LayerNorm fusion (needs complete rows: one block when N <= BN and no split-K, else fix == 2)
  const float* colbias;   // (N) added with the residual (timestep-embedding bias)
  const float* ln_pre_g;  // LayerNorm of the GEMM result BEFORE the residual add (CPE: x += LN(Linear(conv)))
  const float* ln_pre_b;
  const float* ln_post_g;  // LayerNorm of the final row -> ln_out (the pre-norm of the next sub-block)
  const float* ln_post_b;
  void* ln_out;
  int ldln, ln_out_dtype;
  float ln_eps;
  int alt;  // LDS-DMA loop: half of the waves multiply before they issue the next step's DMAs
#ifdef CDSEG_EXPERIMENTS
  int dbg;  // timing experiments (tools/_ab builds; results are wrong): 1 = no MFMAs, 2 = no DMA, 4 = no epilogue
#endif
};

// ---- "fp32 x3" compute type (round 6; CDSEG_F32X3): A and W are fp32 in memory; on their way into LDS every value is split
// into an IEEE-half pair, x ~= hi + lo' / 2048 (hi = rn(x), lo' = rn((x - hi) * 2048): 22 significant bits, the scaled low
// part stays clear of half's subnormals), and a product runs as THREE half MFMAs - hi hi' into one accumulator, hi lo' +
// lo' hi' into a second one that is folded in with 2^-11 at the end (the lo' lo'' term, <= 2^-22 of the product, is dropped).
// fp32 accumulation as before.  3 x v_mfma_f32_16x16x32_f16 (48 cycles) replace 8 x v_mfma_f32_16x16x4_f32 (256 cycles) per
// 16 x 16 x 32 block: the parity mode's 1e-3 logit bound at a multiple of the exact-fp32 rate.  Range: |x| <= 65504
// (saturating, as in the half trunk).
struct F32X3 { float v; };
template <typename CT> constexpr bool kIsX3 = false;
template <> constexpr bool kIsX3<F32X3> = true;
typedef _Float16 x3_f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 x3_f16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void x3_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  const hw_f32x2_t v = {__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f)};
  const x3_f16x2_t h = __builtin_convertvector(v, x3_f16x2_t);
  const hw_f32x2_t hf = __builtin_convertvector(h, hw_f32x2_t);
  const hw_f32x2_t r = (v - hf) * 2048.f;
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, x3_f16x2_t));
}
__device__ __forceinline__ f32x4_t x3_mfma(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(x3_f16x8_t, a), __builtin_bit_cast(x3_f16x8_t, b), c, 0, 0, 0);
}

template <int NCH>
__device__ __forceinline__ int lds_off(int row, int chunk) {
  constexpr int RB = NCH * 16;
  const int sw = NCH == 16 ? (row & 15) : ((row >> 1) & (NCH - 1));
  return row * RB + ((chunk ^ sw) << 4);
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == CDSEG_ACT_GELU) return gelu_erf(v);
  if (act == CDSEG_ACT_SWISH) return v / (1.0f + expf(-v));
  return v;
}

__device__ __forceinline__ void store_val(void* p, int dtype, long idx, float v) {
  if (dtype == CDSEG_F32) ((float*)p)[idx] = v;
  else ((bf16_t*)p)[idx] = f32_to_bf16(v);
}

__device__ __forceinline__ void store_vec4(void* p, int dtype, long idx, float4 v) {
  if (dtype == CDSEG_F32) {
    *reinterpret_cast<float4*>((float*)p + idx) = v;
  } else {
    uint2 u;
    u.x = pack_bf16x2(v.x, v.y);
    u.y = pack_bf16x2(v.z, v.w);
    *reinterpret_cast<uint2*>((bf16_t*)p + idx) = u;
  }
}

// fused epilogue on one group of 4 consecutive columns (row m, columns n..n+3)
__device__ __forceinline__ void epilogue4(const GemmP& g, long m, int n, float4 v) {
  long orow = m;
  if (g.out_idx) {
    orow = g.out_idx[m];
    if (orow < 0) return;
  }
  if (g.vec_ok) {
    if (g.bias) {
      const float4 t = *reinterpret_cast<const float4*>(g.bias + n);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (g.scale) {
      const float4 sc = *reinterpret_cast<const float4*>(g.scale + n);
      const float4 sh = *reinterpret_cast<const float4*>(g.shift + n);
      v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
    }
    if (g.act != CDSEG_ACT_NONE) {
      v.x = apply_act(v.x, g.act); v.y = apply_act(v.y, g.act);
      v.z = apply_act(v.z, g.act); v.w = apply_act(v.w, g.act);
    }
    if (g.out2 && g.out2_pre_add) store_vec4(g.out2, g.out2_dtype, m * g.ldo2 + n, v);
    if (g.res) {
      const float4 t = *reinterpret_cast<const float4*>(g.res + m * g.ldres + n);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (g.colbias) {
      const float4 t = *reinterpret_cast<const float4*>(g.colbias + n);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (g.add_src) {
      const float4 t = *reinterpret_cast<const float4*>(g.add_src + (long)g.add_idx[m] * g.ldadd + n);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    store_vec4(g.out, g.out_dtype, orow * g.ldo + n, v);
    if (g.out2 && !g.out2_pre_add) store_vec4(g.out2, g.out2_dtype, m * g.ldo2 + n, v);
  } else {
    const float vv[4] = {v.x, v.y, v.z, v.w};
    for (int e = 0; e < 4 && n + e < g.N; ++e) {
      const int ne = n + e;
      float x = vv[e];
      if (g.bias) x += g.bias[ne];
      if (g.scale) x = x * g.scale[ne] + g.shift[ne];
      x = apply_act(x, g.act);
      if (g.out2 && g.out2_pre_add) store_val(g.out2, g.out2_dtype, m * g.ldo2 + ne, x);
      if (g.res) x += g.res[m * g.ldres + ne];
      if (g.colbias) x += g.colbias[ne];
      if (g.add_src) x += g.add_src[(long)g.add_idx[m] * g.ldadd + ne];
      store_val(g.out, g.out_dtype, orow * g.ldo + ne, x);
      if (g.out2 && !g.out2_pre_add) store_val(g.out2, g.out2_dtype, m * g.ldo2 + ne, x);
    }
  }
}

// Plain bf16 outputs (qkv, fc1, the conv outputs: bias / folded BN / activation only, no residual, no second output, no
// scatter): 8 consecutive columns per lane = ONE 16-byte store.  The 8-byte stores of epilogue4 made these launches
// store-issue bound: the qkv / fc1 GEMMs of the deep stages spent 70 % of their time in the epilogue
// (profiles/r03_gemm_phase_ablation.txt: 66 us with, 19 us without it at 114k x 128 -> 384).
__device__ __forceinline__ bool plain_bf16_out(const GemmP& g) {
  return g.out_dtype == CDSEG_BF16 && g.vec_ok && !g.out2 && !g.res && !g.add_src && !g.colbias && !g.out_idx && !g.ln_pre_g &&
         !g.ln_post_g && (g.N & 7) == 0 && (g.ldo & 7) == 0;
}

struct Cols8 {  // per-column epilogue constants of a lane's 8 columns (loaded once per tile, ahead of the C staging)
  float4 b0, b1, sc0, sc1, sh0, sh1;
};

__device__ __forceinline__ Cols8 load_cols8(const GemmP& g, int n) {
  Cols8 c;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f), one = make_float4(1.f, 1.f, 1.f, 1.f);
  c.b0 = c.b1 = c.sh0 = c.sh1 = z;
  c.sc0 = c.sc1 = one;
  if (n < g.N) {
    if (g.bias) { c.b0 = *reinterpret_cast<const float4*>(g.bias + n); c.b1 = *reinterpret_cast<const float4*>(g.bias + n + 4); }
    if (g.scale) {
      c.sc0 = *reinterpret_cast<const float4*>(g.scale + n); c.sc1 = *reinterpret_cast<const float4*>(g.scale + n + 4);
      c.sh0 = *reinterpret_cast<const float4*>(g.shift + n); c.sh1 = *reinterpret_cast<const float4*>(g.shift + n + 4);
    }
  }
  return c;
}

__device__ __forceinline__ void epilogue8_bf16(const GemmP& g, long m, int n, float4 a, float4 b, const Cols8& c) {
  a.x += c.b0.x; a.y += c.b0.y; a.z += c.b0.z; a.w += c.b0.w; b.x += c.b1.x; b.y += c.b1.y; b.z += c.b1.z; b.w += c.b1.w;
  if (g.scale) {
    a.x = a.x * c.sc0.x + c.sh0.x; a.y = a.y * c.sc0.y + c.sh0.y; a.z = a.z * c.sc0.z + c.sh0.z; a.w = a.w * c.sc0.w + c.sh0.w;
    b.x = b.x * c.sc1.x + c.sh1.x; b.y = b.y * c.sc1.y + c.sh1.y; b.z = b.z * c.sc1.z + c.sh1.z; b.w = b.w * c.sc1.w + c.sh1.w;
  }
  if (g.act == CDSEG_ACT_GELU) {  // 16-bit output: the packed polynomial form (common.h), like the fused MLP kernels
    f32x2_t t0 = gelu_lp2(f32x2_t{a.x, a.y}), t1 = gelu_lp2(f32x2_t{a.z, a.w});
    a.x = t0[0]; a.y = t0[1]; a.z = t1[0]; a.w = t1[1];
    t0 = gelu_lp2(f32x2_t{b.x, b.y}); t1 = gelu_lp2(f32x2_t{b.z, b.w});
    b.x = t0[0]; b.y = t0[1]; b.z = t1[0]; b.w = t1[1];
  } else if (g.act != CDSEG_ACT_NONE) {
    a.x = apply_act(a.x, g.act); a.y = apply_act(a.y, g.act); a.z = apply_act(a.z, g.act); a.w = apply_act(a.w, g.act);
    b.x = apply_act(b.x, g.act); b.y = apply_act(b.y, g.act); b.z = apply_act(b.z, g.act); b.w = apply_act(b.w, g.act);
  }
  uint4 o;
  o.x = pack_bf16x2(a.x, a.y); o.y = pack_bf16x2(a.z, a.w); o.z = pack_bf16x2(b.x, b.y); o.w = pack_bf16x2(b.z, b.w);
#ifdef CDSEG_EXPERIMENTS
  if ((g.dbg & 8) && o.x != 0x12345678u) return;   // everything but the store
#endif
  *reinterpret_cast<uint4*>((bf16_t*)g.out + m * g.ldo + n) = o;
}

// ---- row-wise epilogue on a lane's float4 groups (columns c0 + 4 * (part + L * i)) of row m.
// L lanes share a row; with LayerNorm the lanes of a row reduce through shuffles, so ALL lanes of a row group
// must call this (inactive rows pass act_row = false).
template <int MAXG>
__device__ __forceinline__ void finish_row(const GemmP& g, long m, bool act_row, int part, int L, int groups, int c0,
                                           float4 (&v)[MAXG]) {
  const bool ln = g.ln_pre_g || g.ln_post_g;
  if (!ln) {
    if (act_row) {
#pragma unroll
      for (int i = 0; i < MAXG; ++i)
        if (part + L * i < groups) epilogue4(g, m, c0 + 4 * (part + L * i), v[i]);
    }
    return;
  }
  const float inv_n = 1.0f / (float)g.N;
#pragma unroll
  for (int i = 0; i < MAXG; ++i) {
    const int cg = part + L * i;
    if (cg < groups) {
      const int n = c0 + 4 * cg;
      if (g.bias) {
        const float4 t = *reinterpret_cast<const float4*>(g.bias + n);
        v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
      }
      if (g.scale) {
        const float4 sc = *reinterpret_cast<const float4*>(g.scale + n);
        const float4 sh = *reinterpret_cast<const float4*>(g.shift + n);
        v[i].x = v[i].x * sc.x + sh.x; v[i].y = v[i].y * sc.y + sh.y;
        v[i].z = v[i].z * sc.z + sh.z; v[i].w = v[i].w * sc.w + sh.w;
      }
      if (g.act != CDSEG_ACT_NONE) {
        v[i].x = apply_act(v[i].x, g.act); v[i].y = apply_act(v[i].y, g.act);
        v[i].z = apply_act(v[i].z, g.act); v[i].w = apply_act(v[i].w, g.act);
      }
    } else {
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  auto row_layernorm = [&](const float* gam, const float* bet, float4 (&o)[MAXG]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXG; ++i)
      if (part + L * i < groups) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    for (int o2 = 1; o2 < L; o2 <<= 1) s += __shfl_xor(s, o2, 64);
    const float mean = s * inv_n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXG; ++i)
      if (part + L * i < groups) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
      }
    for (int o2 = 1; o2 < L; o2 <<= 1) q += __shfl_xor(q, o2, 64);
    const float rstd = 1.0f / sqrtf(q * inv_n + g.ln_eps);
#pragma unroll
    for (int i = 0; i < MAXG; ++i)
      if (part + L * i < groups) {
        const int n = c0 + 4 * (part + L * i);
        const float4 ga = *reinterpret_cast<const float4*>(gam + n);
        const float4 be = *reinterpret_cast<const float4*>(bet + n);
        o[i].x = (v[i].x - mean) * rstd * ga.x + be.x;
        o[i].y = (v[i].y - mean) * rstd * ga.y + be.y;
        o[i].z = (v[i].z - mean) * rstd * ga.z + be.z;
        o[i].w = (v[i].w - mean) * rstd * ga.w + be.w;
      }
  };
  if (g.ln_pre_g) row_layernorm(g.ln_pre_g, g.ln_pre_b, v);
  if (act_row) {
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
      const int cg = part + L * i;
      if (cg >= groups) continue;
      const int n = c0 + 4 * cg;
      if (g.out2 && g.out2_pre_add) store_vec4(g.out2, g.out2_dtype, m * g.ldo2 + n, v[i]);
      if (g.res) {
        const float4 t = *reinterpret_cast<const float4*>(g.res + m * g.ldres + n);
        v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
      }
      if (g.colbias) {
        const float4 t = *reinterpret_cast<const float4*>(g.colbias + n);
        v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
      }
      if (g.add_src) {
        const float4 t = *reinterpret_cast<const float4*>(g.add_src + (long)g.add_idx[m] * g.ldadd + n);
        v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
      }
      store_vec4(g.out, g.out_dtype, m * g.ldo + n, v[i]);
      if (g.out2 && !g.out2_pre_add) store_vec4(g.out2, g.out2_dtype, m * g.ldo2 + n, v[i]);
    }
  } else {
    // keep the shuffles of inactive rows well defined
#pragma unroll
    for (int i = 0; i < MAXG; ++i) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (g.ln_post_g) {
    float4 h[MAXG];
    row_layernorm(g.ln_post_g, g.ln_post_b, h);
    if (act_row) {
#pragma unroll
      for (int i = 0; i < MAXG; ++i) {
        const int cg = part + L * i;
        if (cg < groups) store_vec4(g.ln_out, g.ln_out_dtype, m * g.ldln + c0 + 4 * cg, h[i]);
      }
    }
  }
}

// second pass of a split-K GEMM without LayerNorm: sum the partial tiles in split order, then the fused epilogue.
// (An in-kernel "last block finishes the tile" variant was measured and dropped: an agent-scope fence flushes and
// invalidates the XCD's L2 (+55 us per launch); write-through partials avoid that but the serial tail of the
// finishing block costs more than this launch.)
__global__ void splitk_epilogue_kernel(GemmP g) {
  const long groups = g.M * (long)(g.N >> 2);
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= groups) return;
  const long m = t / (g.N >> 2);
  const int n = (int)(t - m * (g.N >> 2)) << 2;
  const long stride = g.M * (long)g.N;
  const float* p = g.ws + m * g.N + n;
  float4 v = *reinterpret_cast<const float4*>(p);
  for (int s = 1; s < g.splits; ++s) {
    const float4 u = *reinterpret_cast<const float4*>(p + s * stride);
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  epilogue4(g, m, n, v);
}

// second pass for LayerNorm-fused GEMMs whose rows span several column tiles (and / or K splits): one wave per
// row, the whole row in registers (N <= 512: two float4 per lane), split-K sum + the full row epilogue.
__global__ __launch_bounds__(256) void row_finish_kernel(GemmP g) {
  const int lane = threadIdx.x & 63;
  const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= g.M) return;
  const int groups = g.N >> 2;
  const long sstride = g.M * (long)g.N;
  float4 v[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane + 64 * i < groups) {
      const float* p = g.ws + m * g.N + 4 * (lane + 64 * i);
      v[i] = *reinterpret_cast<const float4*>(p);
      for (int s2 = 1; s2 < g.splits; ++s2) {
        const float4 u = *reinterpret_cast<const float4*>(p + s2 * sstride);
        v[i].x += u.x; v[i].y += u.y; v[i].z += u.z; v[i].w += u.w;
      }
    }
  }
  finish_row<2>(g, m, true, lane, 64, groups, 0, v);
}

#ifdef CDSEG_GEMM_TIMING
__device__ unsigned long long g_gemm_t[8 * 16384];  // per block: realtime in / out, cycles of the main loop / epilogue
extern "C" int cdseg_debug_gemm_timing(unsigned long long* host_dst, size_t count) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_gemm_t), count * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
// K-loop phases of gemm_dma_kernel, per block and wave (first 4096 blocks, 16 waves): cycles waiting for the wave's own
// DMA, in the barrier, issuing the next step's DMAs, in the fragment reads + MFMAs; steps; prologue cycles
__device__ unsigned long long g_gemm_kt[4096 * 16 * 8];
extern "C" int cdseg_debug_gemm_ktiming(unsigned long long* host_dst, size_t count) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_gemm_kt), count * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#define KT_STAMP(x) const unsigned long long x = __builtin_readcyclecounter()
#else
#define KT_STAMP(x)
#endif

// ---- accumulators -> LDS C tile -> fused epilogue, 64 rows at a time (shared by both main loops).
// Wave (wm, wn) owns rows wm * 32 .. + 32, columns wn * BN/2 .. of the BM x BN tile; smem is free for reuse.
// SQ: 8 waves as 4 x 2, a wave owns 64 rows x BN / 2 columns (acc[4][BN / 32]); else a wave owns 32 rows x BN / 2 columns
// (acc[2][BN / 32])
template <int BN, int BM, bool SQ = false, int AI = 2, int AJ = BN / 32>
__device__ __forceinline__ void tile_epilogue(const GemmP& g, f32x4_t (&acc)[AI][AJ], char* smem, int tid, int lane,
                                              int wm, int wn, long m0, int n0, int zs) {
  constexpr int NT = SQ ? 512 : 4 * BM;
  constexpr int TN = BN / 32;
  constexpr int CLD = BN + 4;
  const int fr = lane & 15, fg = lane >> 4;
  // ---- accumulators -> LDS C tile -> epilogue, 64 rows at a time.  MFMA C layout: col = lane & 15,
  // row = (lane >> 4) * 4 + r
  float* Cs = reinterpret_cast<float*>(smem);
  const bool plain = !g.fix && plain_bf16_out(g);
  Cols8 cols;
  if (plain) cols = load_cols8(g, n0 + 8 * (tid % (BN / 8)));  // in flight behind the C staging: was a dependent L2 round trip per item
  // (barriers of the C staging: LDS hazards only.  __syncthreads() also waits for vmcnt(0), i.e. for the GLOBAL STORES of
  // the previous 64 rows to be acknowledged - ~1.5 us per tile in which the block did nothing, in-kernel stamps
  // tools/gemm_timing.py: the epilogue took 5.3k cycles, as long as two K steps)
  auto lds_barrier = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
#pragma unroll 1
  for (int hh = 0; hh < BM / 64; ++hh) {
#ifdef CDSEG_GEMM_TIMING
    if (tid == 0 && blockIdx.x < 16384) g_gemm_t[(size_t)blockIdx.x * 8 + 4 + 2 * hh] = __builtin_readcyclecounter();
#endif
    if (hh) lds_barrier();  // the previous 64 rows have left the C tile
    int zcol = 0;  // opaque zero in the column index: keeps the per-column epilogue vectors from being hoisted out of
    if constexpr (BM > 64) asm volatile("v_mov_b32 %0, 0" : "=v"(zcol));  // the loop (and into 100+ extra VGPRs)
    if constexpr (SQ) {
      if (wm == hh) {
#pragma unroll
        for (int i = 0; i < AI; ++i)
#pragma unroll
          for (int j = 0; j < AJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(i * 16 + fg * 4 + r) * CLD + wn * (BN / 2) + j * 16 + fr] = acc[i][j][r];
      }
    } else if ((wm >> 1) == hh) {
      const int rbase = (wm & 1) * 32;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            Cs[(rbase + i * 16 + fg * 4 + r) * CLD + wn * (BN / 2) + j * 16 + fr] = acc[i][j][r];
    }
    lds_barrier();
#ifdef CDSEG_GEMM_TIMING
    if (tid == 0 && blockIdx.x < 16384) g_gemm_t[(size_t)blockIdx.x * 8 + 5 + 2 * hh] = __builtin_readcyclecounter();
#endif
    const long mb = m0 + 64 * hh;
    constexpr int GPR = BN / 4;  // float4 groups per row
    if (g.fix) {
      // raw partial tile -> workspace; a second launch sums the splits and runs the (row) epilogue
      for (int item = tid; item < 64 * GPR; item += NT) {
        const int row = item / GPR, cg = item % GPR;
        const long m = mb + row;
        const int n = n0 + 4 * cg;
        if (m >= g.M || n >= g.N) continue;
        *reinterpret_cast<float4*>(g.ws + ((long)zs * g.M + m) * g.N + n) =
            *reinterpret_cast<const float4*>(Cs + row * CLD + 4 * cg);
      }
    } else if (BM == 64 && (g.ln_pre_g || g.ln_post_g)) {
      // fused LayerNorm (64-row launches only): the block holds complete rows; 4 lanes own one row, values stay
      // in registers
      constexpr int MAXG = BN / 16;
      const int row = (tid >> 2) & 63, part = tid & 3;
      const long m = mb + row;
      const int ng = g.N >> 2;
      float4 v[MAXG];
#pragma unroll
      for (int i = 0; i < MAXG; ++i)
        v[i] = (part + 4 * i < ng) ? *reinterpret_cast<const float4*>(Cs + row * CLD + 4 * (part + 4 * i))
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
      finish_row<MAXG>(g, m, m < g.M, part, 4, ng, 0, v);
    } else if (plain) {
      // plain bf16 output: row-contiguous groups of 8 columns, one 16-byte store each (NT is a multiple of the groups per
      // row: a lane always has the same 8 columns, whose bias / folded-BN constants were fetched ahead of the staging)
      static_assert(NT % (GPR / 2) == 0, "a lane's columns must not change between items");
      for (int item = tid; item < 64 * (GPR / 2); item += NT) {
        const int row = item / (GPR / 2), cg = item % (GPR / 2);
        const long m = mb + row;
        const int n = n0 + 8 * cg;
        if (m >= g.M || n >= g.N) continue;
        epilogue8_bf16(g, m, n, *reinterpret_cast<const float4*>(Cs + row * CLD + 8 * cg),
                       *reinterpret_cast<const float4*>(Cs + row * CLD + 8 * cg + 4), cols);
      }
    } else {
      // epilogue on row-contiguous groups of 4 columns
      for (int item = tid; item < 64 * GPR; item += NT) {
        const int row = item / GPR, cg = item % GPR;
        const long m = mb + row;
        const int n = n0 + 4 * cg + zcol;
        if (m >= g.M || n >= g.N) continue;
        epilogue4(g, m, n, *reinterpret_cast<const float4*>(Cs + row * CLD + 4 * cg));
      }
    }
  }
}

// CT: compute/storage type of A and W.  BN: tile width.  NCH: 16-byte chunks per LDS row.
// BM: rows per tile = 64 (4 waves) or 128 (8 waves); a wave always owns 32 rows x BN/2 columns.  The 128-row form
// moves a third less A + W through L2 -> LDS per FLOP (the deep-stage linears are bound by that path, ~5 TB/s).
template <typename CT, int BN, int NCH, bool GATHER, int BM = 64>
__global__ __launch_bounds__(4 * BM, BM >= 128 ? 4 : 1) void gemm_kernel(GemmP g) {
  constexpr int NT = 4 * BM;   // threads
  constexpr int TN = BN / 32;  // 16-wide column tiles per wave
  constexpr int RB = NCH * 16;
  constexpr int EPC = 16 / (int)sizeof(CT);  // elements per chunk
  constexpr int BK = NCH * EPC;
  constexpr int A_CH = BM * NCH, B_CH = BN * NCH;
  constexpr int A_PT = (A_CH + NT - 1) / NT, B_PT = (B_CH + NT - 1) / NT;
  constexpr int CLD = BN + 4;  // padded fp32 C tile row (conflict-free MFMA-layout writes)
  constexpr int AB_BYTES = (BM + BN) * RB;
  constexpr int C_BYTES = 64 * CLD * 4;  // the epilogue goes through LDS 64 rows at a time
  constexpr int SM_BYTES = (AB_BYTES > C_BYTES ? AB_BYTES : C_BYTES) + 1024;

  char* smem;
  if constexpr (SM_BYTES <= 65536) {
    __shared__ __attribute__((aligned(16))) char static_smem[SM_BYTES];
    smem = static_smem;
  } else {  // launched with gemm_smem_bytes() of dynamic LDS
    extern __shared__ __attribute__((aligned(16))) char dynamic_smem[];
    smem = dynamic_smem;
  }
  char* As = smem;
  char* Bs = smem + BM * RB;
  constexpr int TAILB = 1024;
  int* live = reinterpret_cast<int*>(smem + SM_BYTES - TAILB);  // [0] = count, [1..] = live offsets (<= 128)
  unsigned long long* smask = reinterpret_cast<unsigned long long*>(smem + SM_BYTES - TAILB + 640);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // block -> (row tile, column tile, K split).  Blocks go round-robin over the 8 XCDs (own L2 each):
  //   xmode 1: all row tiles of one (column tile, split) = one W slice run on ONE XCD, so every XCD pulls 1/8 of W
  //            from HBM instead of all of it (deep stages: few rows, W of several MB);
  //   xmode 2: every XCD owns a contiguous range of row tiles (z-ordered rows: neighbour gathers stay in one L2).
  int mt, sl;
  {
    const int bid = blockIdx.x, slices = g.gn * g.splits;
    if (g.xmode == 1) {
      const int j = bid >> 3;
      mt = j % g.gm;
      sl = (bid & 7) + 8 * (j / g.gm);
      if (sl >= slices) return;
    } else if (g.xmode == 2) {
      const int j = bid >> 3, chunk = (g.gm + 7) >> 3;
      mt = (bid & 7) * chunk + j / slices;
      sl = j % slices;
      if (j / slices >= chunk || mt >= g.gm) return;
    } else {
      mt = bid % g.gm;
      sl = bid / g.gm;
    }
  }
  const int nt = sl % g.gn, zs = sl / g.gn;
  const long m0 = (long)mt * BM;
  const int n0 = nt * BN;
  const long Kw = (long)g.kvol * g.K;  // W row length

  int nlive = 1;
  if (GATHER) {
    if (tid < 2) smask[tid] = 0ull;
    __syncthreads();
    for (int r = tid >> 2; r < BM; r += NT / 4) {
      const long m = m0 + r;
      if (m < g.M) {
        unsigned long long lo = 0ull, hi = 0ull;
        for (int o = tid & 3; o < g.kvol; o += 4)
          if (g.nbr[m * g.nbr_sm + (long)o * g.nbr_so] >= 0) {
            if (o < 64) lo |= 1ull << o; else hi |= 1ull << (o - 64);
          }
        if (lo) atomicOr(&smask[0], lo);
        if (hi) atomicOr(&smask[1], hi);
      }
    }
    __syncthreads();
    if (tid < g.kvol) {  // compaction by population count: offset tid goes to slot (#live offsets below it)
      const unsigned long long m0 = smask[0], m1 = smask[1];
      const bool on = ((tid < 64 ? m0 >> tid : m1 >> (tid - 64)) & 1ull) != 0ull;
      const int below = tid < 64 ? __popcll(m0 & ((1ull << tid) - 1ull))
                                 : __popcll(m0) + __popcll(m1 & ((1ull << (tid - 64)) - 1ull));
      if (on) live[1 + below] = tid;
      if (tid == 0) live[0] = __popcll(m0) + __popcll(m1);
    }
    __syncthreads();
    nlive = __builtin_amdgcn_readfirstlane(live[0]);
  }
  const int KV = nlive * g.K;  // virtual (compacted) reduction length (< 2^31: kvol <= 128, K <= 2^16)
  const int nkc = (KV + BK - 1) / BK;
  const int kshift = g.kshift;  // log2(K) when K is a power of two (every shipped config), else -1

  uint4 a_reg[A_PT];
  uint4 b_reg[B_PT];

  auto load_tiles = [&](int kc) {
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      const int id = i * NT + tid;
      const int row = id / NCH, ch = id % NCH;
      const long m = m0 + row;
      const int kv = kc * BK + ch * EPC;
      a_reg[i] = make_uint4(0u, 0u, 0u, 0u);
      if (id < A_CH && m < g.M && kv < KV) {
        if (GATHER) {
          const int j = kshift >= 0 ? (kv >> kshift) : (int)((unsigned)kv / (unsigned)g.K);
          const int cc = kv - j * g.K;
          const int src = g.nbr[m * g.nbr_sm + (long)live[1 + j] * g.nbr_so];
          if (src >= 0) a_reg[i] = *reinterpret_cast<const uint4*>((const CT*)g.A + (long)src * g.lda + cc);
        } else {
          a_reg[i] = *reinterpret_cast<const uint4*>((const CT*)g.A + m * g.lda + kv);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
      const int id = i * NT + tid;
      const int row = id / NCH, ch = id % NCH;
      const int kv = kc * BK + ch * EPC;
      b_reg[i] = make_uint4(0u, 0u, 0u, 0u);
      if (id < B_CH && (n0 + row) < g.N && kv < KV) {
        int col = kv;
        if (GATHER) {
          const int j = kshift >= 0 ? (kv >> kshift) : (int)((unsigned)kv / (unsigned)g.K);
          col = live[1 + j] * g.K + (kv - j * g.K);
        }
        b_reg[i] = *reinterpret_cast<const uint4*>((const CT*)g.W + (long)(n0 + row) * Kw + col);
      }
    }
  };
  // x3: a row of the LDS tile holds the hi plane in chunks [0, NCH / 2) and the scaled lo plane in [NCH / 2, NCH); a thread's
  // four floats (float chunk ch) become 8 bytes of each plane
  auto store_x3 = [&](char* base, int row, int ch, const uint4& r) {
    uint2 hi, lo;
    x3_split2(__uint_as_float(r.x), __uint_as_float(r.y), hi.x, lo.x);
    x3_split2(__uint_as_float(r.z), __uint_as_float(r.w), hi.y, lo.y);
    *reinterpret_cast<uint2*>(base + lds_off<NCH>(row, ch >> 1) + ((ch & 1) << 3)) = hi;
    *reinterpret_cast<uint2*>(base + lds_off<NCH>(row, NCH / 2 + (ch >> 1)) + ((ch & 1) << 3)) = lo;
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      const int id = i * NT + tid;
      if (id < A_CH) {
        if constexpr (kIsX3<CT>) store_x3(As, id / NCH, id % NCH, a_reg[i]);
        else *reinterpret_cast<uint4*>(As + lds_off<NCH>(id / NCH, id % NCH)) = a_reg[i];
      }
    }
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
      const int id = i * NT + tid;
      if (id < B_CH) {
        if constexpr (kIsX3<CT>) store_x3(Bs, id / NCH, id % NCH, b_reg[i]);
        else *reinterpret_cast<uint4*>(Bs + lds_off<NCH>(id / NCH, id % NCH)) = b_reg[i];
      }
    }
  };

  f32x4_t acc[2][TN];
  f32x4_t accx[kIsX3<CT> ? 2 : 1][kIsX3<CT> ? TN : 1];  // x3: the cross terms hi lo' + lo' hi', scaled by 2^11
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if constexpr (kIsX3<CT>) accx[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

  const int fr = lane & 15, fg = lane >> 4;

  // split-K: this block reduces chunks [kc0, kc1) of the (compacted) K range
  const int kc0 = (int)(((long)nkc * zs) / g.splits);
  const int kc1 = (int)(((long)nkc * (zs + 1)) / g.splits);
  if (kc0 < kc1) {
    load_tiles(kc0);
    store_tiles();
  }
  __syncthreads();
  for (int kc = kc0; kc < kc1; ++kc) {
    if (kc + 1 < kc1) load_tiles(kc + 1);  // global loads in flight during the MFMAs below

    if constexpr (kIsX3<CT>) {
      constexpr int HC = NCH / 2;  // 16-byte chunks per plane
#pragma unroll
      for (int kk = 0; kk < HC / 4; ++kk) {
        bf16x8_t ah[2], al[2], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ah[i] = *reinterpret_cast<const bf16x8_t*>(As + lds_off<NCH>(wm * 32 + i * 16 + fr, 4 * kk + fg));
          al[i] = *reinterpret_cast<const bf16x8_t*>(As + lds_off<NCH>(wm * 32 + i * 16 + fr, HC + 4 * kk + fg));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bh[j] = *reinterpret_cast<const bf16x8_t*>(Bs + lds_off<NCH>(wn * (BN / 2) + j * 16 + fr, 4 * kk + fg));
          bl[j] = *reinterpret_cast<const bf16x8_t*>(Bs + lds_off<NCH>(wn * (BN / 2) + j * 16 + fr, HC + 4 * kk + fg));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = x3_mfma(ah[i], bh[j], acc[i][j]);
            accx[i][j] = x3_mfma(ah[i], bl[j], accx[i][j]);
            accx[i][j] = x3_mfma(al[i], bh[j], accx[i][j]);
          }
      }
    } else if constexpr (sizeof(CT) == 2) {
#pragma unroll
      for (int kk = 0; kk < NCH / 4; ++kk) {
        bf16x8_t a[2], b[TN];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a[i] = *reinterpret_cast<const bf16x8_t*>(As + lds_off<NCH>(wm * 32 + i * 16 + fr, 4 * kk + fg));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[j] = *reinterpret_cast<const bf16x8_t*>(Bs + lds_off<NCH>(wn * (BN / 2) + j * 16 + fr, 4 * kk + fg));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = mfma_16x16x32_bf16(a[i], b[j], acc[i][j]);
      }
    } else {
      // per 32-wide k block: k-slot g of MFMA step (h, ss) holds k = 16*h + 4*g + ss (same map for A and W)
#pragma unroll
      for (int kb = 0; kb < NCH / 8; ++kb) {
        f32x4_t a[2][2], b[TN][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
            a[i][h] = *reinterpret_cast<const f32x4_t*>(As + lds_off<NCH>(wm * 32 + i * 16 + fr, 8 * kb + 4 * h + fg));
#pragma unroll
          for (int j = 0; j < TN; ++j)
            b[j][h] = *reinterpret_cast<const f32x4_t*>(
                Bs + lds_off<NCH>(wn * (BN / 2) + j * 16 + fr, 8 * kb + 4 * h + fg));
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int ss = 0; ss < 4; ++ss)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][h][ss], b[j][h][ss], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
    if (kc + 1 < kc1) store_tiles();
    __syncthreads();
  }

  if constexpr (kIsX3<CT>) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] += accx[i][j] * (1.0f / 2048.f);
  }
  tile_epilogue<BN, BM>(g, acc, smem, tid, lane, wm, wn, m0, n0, zs);
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 main loop on LDS-DMA (global_load_lds_dwordx4): BM x 128 tile, K steps of 64, TWO LDS stages, ONE barrier per
// step.  The register-staged loop above waits for its global loads, writes them to LDS and barriers twice per step: on
// the deep stages (C >= 128: few rows, long reductions, a dependent index -> row chain per step of the sparse convs)
// that left the matrix pipe at ~10 %.  Here the tile of step k+1 is in flight - straight into LDS, no staging
// registers - while the MFMAs of step k run:
//     wait (own DMA of step k landed) -> barrier -> issue DMA of step k+1 -> ds_read + MFMA of step k
//   * the barrier both publishes every wave's DMA of step k and certifies that all waves are done READING the other
//     stage (which the DMA issued right after it overwrites);
//   * an LDS-DMA instruction writes 64 lanes x 16 B linearly; the XOR swizzle that keeps the fragment reads conflict
//     free is applied to the per-lane SOURCE address instead (lane (row, slot) fetches chunk slot ^ ((row >> 1) & 7));
//   * sparse convs: the tile's kernel-map entries (BM rows x live offsets) sit in LDS, a missing neighbour reads a
//     zero page - the index lookups are LDS reads one step ahead, not global loads in the dependency chain;
//   * the DMA is issued from inline asm (M0 = LDS base in the same statement) and waited for with counted s_waitcnt:
//     hipcc drains vmcnt(0) around compiler-visible LDS-DMA, and no other VMEM instruction lives in the loop.
__device__ uint4 g_zero_page[8];  // 128 zero bytes: the source of a missing neighbour's row chunk

template <int BM, bool GATHER = true, int BN = 128, int NST = 2, bool SQ = false>
struct DmaCfg {
  static constexpr int WAVES = SQ ? 8 : BM / 16, NT = WAVES * 64, BK = 64;
  static constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
  static constexpr int A_PW = (BM / 8) / WAVES, W_PW = (BN / 8) / WAVES;  // DMA instructions per wave and step (2, BN / 8 / WAVES)
  static constexpr int STAGES = NST * (A_BYTES + W_BYTES);
  static constexpr int ITAB = GATHER ? BM * 27 * 4 : 0;  // the plain Linears carry no kernel-map table: with it a 128-row
  static constexpr int LDS = STAGES + ITAB + 1024;         // block took 79.6 KB of LDS and the CU held ONE block, not two
};

// BN = 256, NST = 3 (deep sparse convs, C >= 256): the tile spans 256 output channels, so a gathered row is fetched once
// per 256 instead of once per 128 columns (48 KB of A + W per 4.2 MFLOP step instead of 64), and TWO steps are in flight
// behind the one being multiplied (144 KB of stages: one block per CU; these launches are bound by the L2 -> LDS fill
// rate, which grows with the bytes in flight: profiles/r03_ubench_dma_depth.txt).
// Sparse convs, all shapes: a (16-row group, kernel offset) pair none of whose rows has that neighbour is skipped - its
// A rows are not fetched (they would be 16 copies of the zero page) and its MFMAs not issued.  On z-ordered points a
// 16-row group has 17 - 24 of the 27 offsets where the 128-row tile has 23 - 27 (tools: DESIGN 4.2), i.e. 12 - 25 % of
// the tile's matrix work and gathered bytes go away.
// SQ (BM = 256 rows x BN = 256 columns, EIGHT waves as 4 x 2 with 64 x 128 wave tiles - acc[4][8] = 128 registers of a
// 256-register budget - split-K over the compacted offsets): the deep sparse convs (C >= 256) at 8+ scenes.  Per FLOP the
// tile moves half the bytes of the 128 x 128 one through the L2 -> LDS fill path these launches are bound by; a K step is
// 8 DMA pieces and 64 MFMAs per wave, all fragments of a K half are read ahead of its 32 MFMAs, and the two waves of a
// SIMD run in opposite phase (waves 0 - 3 issue the next step's DMAs and then multiply, waves 4 - 7 multiply first): one
// block per CU, so nothing else would use the matrix pipe while a wave issues.  (The first form of this tile - 16 waves
// as 4 x 4, 128 registers, fragment reads behind per-group branches, every wave in the same phase - spent 2.4x the MFMA
// time in its multiply section and only paid at C = 512: profiles/r04_conv_sq.txt.)
template <int BM, bool GATHER, int BN = 128, int NST = 2, bool SQ = false>
__global__ __launch_bounds__(SQ ? 512 : 4 * BM) void gemm_dma_kernel(GemmP g) {
#ifdef CDSEG_GEMM_TIMING
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), tc0 = __builtin_readcyclecounter();
#endif
  using D = DmaCfg<BM, GATHER, BN, NST, SQ>;
  constexpr int TN = BN / 32;
  constexpr int AI = SQ ? 4 : 2, AJ = TN;  // a wave owns 16 AI rows x BN / 2 columns
  static_assert(!SQ || (BM == 256 && NST == 2 && GATHER), "SQ: 256-row tiles of 8 waves, two stages, sparse convs");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tail = smem + D::STAGES + D::ITAB;
  int* live = reinterpret_cast<int*>(tail);  // [0] = count, [1..] = live offsets
  int* glive = reinterpret_cast<int*>(tail + 256);  // per live offset: bit r = 16-row group r has a neighbour there
  unsigned long long* smask = reinterpret_cast<unsigned long long*>(tail + 640);
  int* itab = reinterpret_cast<int*>(smem + D::STAGES);  // [row][live slot]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int mt, sl;
  {
    const int bid = blockIdx.x, slices = g.gn * g.splits;
    if (g.xmode == 1) {
      const int j = bid >> 3;
      mt = j % g.gm;
      sl = (bid & 7) + 8 * (j / g.gm);
      if (sl >= slices) return;
    } else if (g.xmode == 2) {
      const int j = bid >> 3, chunk = (g.gm + 7) >> 3;
      mt = (bid & 7) * chunk + j / slices;
      sl = j % slices;
      if (j / slices >= chunk || mt >= g.gm) return;
    } else {
      mt = bid % g.gm;
      sl = bid / g.gm;
    }
  }
  const int nt = sl % g.gn, zs = sl / g.gn;
  const long m0 = (long)mt * BM;
  const int n0 = nt * BN;
  const long Kw = (long)g.kvol * g.K;

  int nlive = 1;
  if (GATHER) {
    // The tile's slice of the kernel map, read ONCE and coalesced (offset-major tables: 64 consecutive rows of one offset
    // per wave instruction) into registers: thread -> row r = tid % BM, offsets og, og + OPT, ...  Its ballots give the
    // per-offset liveness of the tile and of each 16-row group; the compacted [row][live slot] table is written from the
    // registers.  (The first version walked the map three times with row-strided loads - 3456 scattered loads per block
    // for the table alone - and took 28k cycles per block, in-kernel stamps profiles/r04_conv_timing.txt: 13 % of a block's
    // life at 8 scenes, 63 % of it on single scenes.)
    constexpr int OPT = D::NT / BM, PASSES = (27 + OPT - 1) / OPT;
    int* glraw = reinterpret_cast<int*>(tail + 384);    // per offset: 16-row groups with a neighbour there
    int* slotmap = reinterpret_cast<int*>(tail + 512);  // offset -> live slot or -1
    const int r = tid % BM, og = tid / BM;
    if (tid < 32) glraw[tid] = 0;
    __syncthreads();
    int idx[PASSES];
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int o = og + OPT * i;
      idx[i] = (o < g.kvol && m0 + r < g.M) ? g.nbr[(m0 + r) * g.nbr_sm + (long)o * g.nbr_so] : -1;
    }
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int o = og + OPT * i;
      const unsigned long long bal = __ballot(idx[i] >= 0);  // rows (r & ~63) .. + 63 of offset o
      if (lane == 0 && bal && o < g.kvol) {
        const int bits = ((bal & 0xffffull) ? 1 : 0) | ((bal & 0xffff0000ull) ? 2 : 0) | ((bal & 0xffff00000000ull) ? 4 : 0) |
                         ((bal >> 48) ? 8 : 0);
        atomicOr(&glraw[o], bits << ((r & ~63) >> 4));
      }
    }
    __syncthreads();
    if (tid < 64) {
      const int gr = tid < g.kvol ? glraw[tid] : 0;
      const unsigned long long mk = __ballot(gr != 0);
      const int slot = __popcll(mk & ((1ull << tid) - 1ull));
      if (gr) { live[1 + slot] = tid; glive[slot] = gr; }
      if (tid < 32) slotmap[tid] = gr ? slot : -1;
      if (tid == 0) live[0] = __popcll(mk);
    }
    __syncthreads();
    nlive = __builtin_amdgcn_readfirstlane(live[0]);
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int o = og + OPT * i;
      if (o < g.kvol) {
        const int sl2 = slotmap[o];
        if (sl2 >= 0) itab[r * 27 + sl2] = idx[i];
      }
    }
    __syncthreads();
  }
  const int KV = nlive * g.K;
  const int nkc = KV / D::BK;  // K % 64 == 0 (launch condition)
  const int kshift = g.kshift;
  const int kc0 = (int)(((long)nkc * zs) / g.splits);
  const int kc1 = (int)(((long)nkc * (zs + 1)) / g.splits);

  // ---- per-lane DMA geometry (fixed for the whole tile)
  const int lrow = lane >> 3, slot = lane & 7;
  const bf16_t* a_src[D::A_PW];   // plain: row base pointers
  int a_row[D::A_PW];             // gather: tile row (index into itab)
  int a_chunk[D::A_PW];
  const bf16_t* w_src[D::W_PW];
  int w_chunk[D::W_PW];
#pragma unroll
  for (int i = 0; i < D::A_PW; ++i) {
    const int r = (wave * D::A_PW + i) * 8 + lrow;
    a_row[i] = r;
    a_chunk[i] = slot ^ ((r >> 1) & 7);
    long m = m0 + r;
    if (m >= g.M) m = g.M - 1;  // rows past the end: computed, never stored
    a_src[i] = (const bf16_t*)g.A + m * g.lda + a_chunk[i] * 8;
  }
#pragma unroll
  for (int i = 0; i < D::W_PW; ++i) {
    const int r = (wave * D::W_PW + i) * 8 + lrow;
    w_chunk[i] = slot ^ ((r >> 1) & 7);
    int n = n0 + r;
    if (n >= g.N) n = g.N - 1;
    w_src[i] = (const bf16_t*)g.W + (long)n * Kw + w_chunk[i] * 8;
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto a_stage = [&](int st) { return lds_base + st * D::A_BYTES; };
  auto w_stage = [&](int st) { return lds_base + NST * D::A_BYTES + st * D::W_BYTES; };

  // Which of the live offsets this wave needs, as 64-bit masks over the live slots (bit jl <-> live offset jl; scalar
  // registers, no LDS access in the K loop): m_dma - the 16-row group this wave STAGES (rows 16 wave ..) has a neighbour
  // there; m_on0 / m_on1 - the two groups it MULTIPLIES (rows 32 wm .. / 32 wm + 16 ..) have one.
  constexpr int DG = D::A_PW / 2;  // 16-row groups this wave stages (two 8-row pieces each): rows 16 DG wave ..
  static_assert(D::A_PW == 2 * DG, "a wave stages whole 16-row groups");
  unsigned long long m_dma[DG], m_on[AI];
#pragma unroll
  for (int d = 0; d < DG; ++d) m_dma[d] = ~0ull;
#pragma unroll
  for (int i = 0; i < AI; ++i) m_on[i] = ~0ull;
  if (GATHER) {
    const int gl = lane < nlive ? glive[lane] : 0;
#pragma unroll
    for (int d = 0; d < DG; ++d) m_dma[d] = __ballot((gl >> (DG * wave + d)) & 1);
#pragma unroll
    for (int i = 0; i < AI; ++i) m_on[i] = __ballot((gl >> (AI * wm + i)) & 1);
  }
  auto slot_of = [&](int kc) -> int { return GATHER ? ((kc * D::BK) >> kshift) : 0; };
  // gather indices of the step to be issued next (LDS reads, one step ahead of their DMA)
  int idx_n[D::A_PW];
  auto fetch_idx = [&](int kc) {
    if (GATHER) {
      const int jl = (kc * D::BK) >> kshift;
#pragma unroll
      for (int i = 0; i < D::A_PW; ++i) idx_n[i] = itab[a_row[i] * 27 + jl];
    }
  };
  const unsigned a_pitch = (unsigned)g.lda * 2u;  // bytes per A row (launch condition: M * lda * 2 < 2^32 for the gather form)
  const unsigned long long a_base = (unsigned long long)(uintptr_t)g.A;
  const unsigned long long z_page = (unsigned long long)(uintptr_t)g_zero_page + (unsigned)(slot * 16);
  // issues the DMAs of step kc into stage st; returns how many instructions this wave issued
  auto issue = [&](int kc, int st) -> int {
    const int kv = kc * D::BK;
    int cc = kv, wcol = kv;
    if (GATHER) {
      const int jl = kv >> kshift;
      cc = kv - (jl << kshift);
      wcol = live[1 + jl] * g.K + cc;
    }
#pragma unroll
    for (int i = 0; i < D::W_PW; ++i) dma16(w_src[i] + wcol, w_stage(st) + (wave * D::W_PW + i) * 1024);
    int issued = D::W_PW;
#pragma unroll
    for (int d = 0; d < DG; ++d) {
      if ((m_dma[d] >> slot_of(kc)) & 1ull) {  // the 16-row group has a neighbour at this offset (else: not read either)
        issued += 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int i = 2 * d + h;
          const void* src;
          if (GATHER) {
            // one 32 x 32 -> 64-bit multiply-add and two selects (written as `sidx >= 0 ? row address : zero page` this
            // compiled into a divergent branch around two 64-bit multiplies per piece).  NOT fetching the rows without a
            // neighbour at all (lanes off in the DMA instruction, the multiply side reading a zero row instead) was
            // measured slower: the fill path's cost is per DMA instruction, not per active lane (profiles/r04_conv_sq8.txt)
            const int sidx = idx_n[i];
            const unsigned long long pa = (unsigned long long)(unsigned)sidx * a_pitch + (a_base + (unsigned)((cc + a_chunk[i] * 8) * 2));
            const unsigned lo = sidx >= 0 ? (unsigned)pa : (unsigned)z_page, hi = sidx >= 0 ? (unsigned)(pa >> 32) : (unsigned)(z_page >> 32);
            src = (const void*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
          } else {
            src = a_src[i] + cc;
          }
          dma16(src, a_stage(st) + (wave * D::A_PW + i) * 1024);
        }
      }
    }
    return issued;
  };

  f32x4_t acc[AI][AJ];
#pragma unroll
  for (int i = 0; i < AI; ++i)
#pragma unroll
    for (int j = 0; j < AJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;

  // prologue: NST - 1 steps in flight
  int pending = 0;  // instructions of the youngest issued step (the only one allowed to be outstanding when NST == 3)
#pragma unroll
  for (int d = 0; d < NST - 1; ++d) {
    if (kc0 + d < kc1) {
      fetch_idx(kc0 + d);
      pending = issue(kc0 + d, d);
    }
  }
  if (kc0 + NST - 1 < kc1) fetch_idx(kc0 + NST - 1);
  int st = 0;
#ifdef CDSEG_GEMM_TIMING
  unsigned long long kt_wait = 0, kt_bar = 0, kt_issue = 0, kt_mma = 0;
  const unsigned long long kt_pro = __builtin_readcyclecounter() - tc0;
#endif
  if constexpr (SQ) {
    // ---- 256-row tiles.  ONE copy of the issue code and ONE of the multiply code in the loop (the K halves of a step are
    // a loop that is not unrolled; half 1's fragments sit 64 bytes from half 0's, an XOR on the swizzled offsets): with
    // the issue code inlined at two or three places next to 128 accumulator registers hipcc spilled inside the loop.
    // The early waves (0 - 3) issue the next step's DMAs ahead of K half 0, the late ones (4 - 7, g.alt != 0) ahead of K
    // half 1: a SIMD's two waves then use the fill path and the matrix pipe at the same time, and a late wave's DMAs
    // still have half a multiply section to land.
    const int my_slot = (g.alt && wave >= D::WAVES / 2) ? 1 : 0;
    const int a_off0 = lds_off<8>(wm * 64 + fr, fg), b_off0 = lds_off<8>(wn * (BN / 2) + fr, fg);  // + 2048 per 16 rows
#pragma unroll 1
    for (int kc = kc0; kc < kc1; ++kc) {
      KT_STAMP(k0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMAs of step kc have landed
      KT_STAMP(k1);
      __builtin_amdgcn_s_barrier();  // ... everybody's have, and nobody still reads the stage issued into next
      KT_STAMP(k2);
      const int jl = slot_of(kc);
      bool on[4], any = false;
#pragma unroll
      for (int i = 0; i < 4; ++i) { on[i] = (m_on[i] >> jl) & 1ull; any |= on[i]; }
      const bool more = kc + 1 < kc1;
      const char* As = smem + st * D::A_BYTES;
      const char* Bs = smem + NST * D::A_BYTES + st * D::W_BYTES;
#ifdef CDSEG_GEMM_TIMING
      unsigned long long ki = 0;
#endif
#pragma unroll 1
      for (int kk = 0; kk < 2; ++kk) {
        if (more && kk == my_slot) {
          KT_STAMP(i0);
          issue(kc + 1, st ^ 1);
          if (kc + 2 < kc1) fetch_idx(kc + 2);
#ifdef CDSEG_GEMM_TIMING
          ki = __builtin_readcyclecounter() - i0;
#endif
        }
        if (any) {
          // all twelve fragments of the K half are requested ahead of its MFMAs (a dead group's A rows are stale bytes,
          // never multiplied)
          const int x = kk << 6;
          bf16x8_t a[4], b[AJ];
#pragma unroll
          for (int j = 0; j < AJ; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(Bs + ((b_off0 ^ x) + 2048 * j));
#pragma unroll
          for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(As + ((a_off0 ^ x) + 2048 * i));
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (on[i]) {
#pragma unroll
              for (int j = 0; j < AJ; ++j) acc[i][j] = mfma_16x16x32_bf16(a[i], b[j], acc[i][j]);
            }
        }
      }
      st ^= 1;
#ifdef CDSEG_GEMM_TIMING
      {
        const unsigned long long k4 = __builtin_readcyclecounter();
        kt_wait += k1 - k0; kt_bar += k2 - k1; kt_issue += ki; kt_mma += k4 - k2 - ki;
      }
#endif
    }
  } else {
#pragma unroll 1
  for (int kc = kc0; kc < kc1; ++kc) {
    KT_STAMP(k0);
    // this wave's DMA of step kc has landed: everything but the youngest step (NST == 3) may still be in flight
    if (NST == 2 || kc + 1 >= kc1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (pending == D::W_PW) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D::W_PW) : "memory");
    } else {  // (three stages: 128-row tiles only, one staged group per wave)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D::W_PW + D::A_PW) : "memory");
    }
    KT_STAMP(k1);
    __builtin_amdgcn_s_barrier();  // ... everybody's has, and nobody still reads the stage issued into next
    KT_STAMP(k2);
    // The stage issued into (read last in step kc - 1) is free from the barrier on, so the order of "issue the next
    // step's DMAs" and "multiply this step" inside a step is free.  Waves w and w + 4 share a SIMD: the first four issue
    // first, the others multiply first - a block's two waves on a SIMD then use the memory path and the matrix pipe at the
    // same time instead of both queueing for the same one (stamps: issuing 4 pieces took 900 of a step's 2070 cycles).
    // Measured and left off (g.alt = 0): the step takes the same ~2100 cycles either way - the multiply section of the
    // late issuers grows by what their issue section shrinks (profiles/r04_conv_timing.txt): the step is bound by the
    // fill path itself (~31 B/clk and CU), not by how its users queue for it.
    auto issue_next = [&]() {
      if (kc + NST - 1 < kc1) {
        int sn = st + NST - 1;
        if (sn >= NST) sn -= NST;
        pending = issue(kc + NST - 1, sn);
        if (kc + NST < kc1) fetch_idx(kc + NST);
      }
    };
    const bool issue_first = !g.alt || wave < D::WAVES / 2;
#ifdef CDSEG_EXPERIMENTS
    if (!(g.dbg & 2))
#endif
    if (issue_first) issue_next();
#ifdef CDSEG_EXPERIMENTS
    if (g.dbg & 1) { if (++st == NST) st = 0; continue; }
#endif
    KT_STAMP(k3);
    const char* As = smem + st * D::A_BYTES;
    const char* Bs = smem + NST * D::A_BYTES + st * D::W_BYTES;
    const int jl = slot_of(kc);
    {
    const bool on0 = (m_on[0] >> jl) & 1ull, on1 = (m_on[1] >> jl) & 1ull;
    if (on0 || on1) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        // the fragments of a K half are requested before the branch (a dead group's A rows are stale bytes, never used)
        bf16x8_t a[2], b[TN];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a[i] = *reinterpret_cast<const bf16x8_t*>(As + lds_off<8>(wm * 32 + i * 16 + fr, 4 * kk + fg));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[j] = *reinterpret_cast<const bf16x8_t*>(Bs + lds_off<8>(wn * (BN / 2) + j * 16 + fr, 4 * kk + fg));
        if (on0 && on1) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mfma_16x16x32_bf16(a[i], b[j], acc[i][j]);
        } else if (on0) {
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[0][j] = mfma_16x16x32_bf16(a[0], b[j], acc[0][j]);
        } else {
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[1][j] = mfma_16x16x32_bf16(a[1], b[j], acc[1][j]);
        }
      }
    }
    }
    if (!issue_first) issue_next();
    if (++st == NST) st = 0;
#ifdef CDSEG_GEMM_TIMING
    {
      const unsigned long long k4 = __builtin_readcyclecounter();
      kt_wait += k1 - k0; kt_bar += k2 - k1; kt_issue += k3 - k2; kt_mma += k4 - k3;
    }
#endif
  }
  }  // !SQ
#ifdef CDSEG_GEMM_TIMING
  if (lane == 0 && blockIdx.x < 4096 && wave < 16) {
    unsigned long long* d = g_gemm_kt + ((size_t)blockIdx.x * 16 + wave) * 8;
    d[0] = kt_wait; d[1] = kt_bar; d[2] = kt_issue; d[3] = kt_mma; d[4] = (unsigned long long)(kc1 - kc0); d[5] = kt_pro;
    d[6] = (unsigned long long)nlive; d[7] = 1;
  }
#endif
  __syncthreads();  // all fragment reads done: the stages become the C tile
#ifdef CDSEG_EXPERIMENTS
  if (g.dbg & 4) { if (acc[0][0][0] == 12345.678f) ((float*)g.out)[0] = acc[1][1][1]; return; }
#endif
#ifdef CDSEG_GEMM_TIMING
  const unsigned long long tc1 = __builtin_readcyclecounter();
#endif
  tile_epilogue<BN, BM, SQ, AI, AJ>(g, acc, smem, tid, lane, wm, wn, m0, n0, zs);
#ifdef CDSEG_GEMM_TIMING
  if (tid == 0 && blockIdx.x < 16384) {
    unsigned long long* d = g_gemm_t + (size_t)blockIdx.x * 8;
    d[0] = rt0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = tc1 - tc0;
    const unsigned long long tend = __builtin_readcyclecounter();
    d[3] = tend - tc1;
    // d[4..7] hold absolute stamps (start of half 0, after its staging barrier, start of half 1, after its barrier)
    const unsigned long long a0 = d[4], a1 = d[5], b0 = d[6], b1 = d[7];
    d[4] = a1 - a0; d[5] = b0 - a1; d[6] = b1 - b0; d[7] = tend - b1;
  }
#endif
}

// gemm_leave_partials (bottom of the file): this thread's next split-K launch with a bias-only epilogue leaves its raw partial
// tiles in the workspace instead of launching splitk_epilogue_kernel
thread_local bool tl_leave_partials = false;
thread_local int tl_partial_splits = 1;

template <typename CT, int BN, int NCH, int BM>
constexpr int gemm_smem_bytes() {
  constexpr int ab = (BM + BN) * NCH * 16, c = 64 * (BN + 4) * 4;
  return (ab > c ? ab : c) + 1024;
}


// (A persistent weight-stationary variant for the short-K linears of the big stages - W slice resident in LDS /
// registers, the next row tile's A in flight during the epilogue - was measured and dropped: 41 us vs 29 us on the
// 120k x 32 -> 128 linear; four block barriers per tile at 3 workgroups / CU cost more than the reloads they save.
// What those launches were actually paying for was libm's erff in the GELU epilogue, see common.h.)
template <typename CT, int NCH, bool GATHER>
int launch_bn(GemmP p, size_t ws_bytes, hipStream_t s) {
  constexpr int BK = NCH * (16 / (int)sizeof(CT));
  // split-K of few-tile problems (a single scene's deep stages): blocks the split aims for, and the most slices.  320 / 16 (round
  // 6; 512 / 32 before): fewer, longer slices - less partial-sum traffic and a cheaper second pass - single scene 3.61 -> 3.52 ms,
  // flat between 192 and 384 blocks / 8 and 16 slices (profiles/r06_splitk_sweep.txt)
  static const int split_target = cdseg_knob("CDSEG_GEMM_SPLIT_TARGET", 320);
  static const int split_max = cdseg_knob("CDSEG_GEMM_SPLIT_MAX", 16);
  static const int xmode_env = cdseg_knob("CDSEG_GEMM_XMODE", -1);
  const bool ln = p.ln_pre_g || p.ln_post_g;
  // sparse convs: 128-row tiles (8 waves): half the W re-reads per row through L2 -> LDS, the path every conv level is
  // bound by (stage-0 conv of a 4-scene batch 196 -> 123 us, +2 % end to end).  The same tiles for the wide deep-stage
  // linears were neutral (-1 %) and are not instantiated.
  static const int conv_bm = cdseg_knob("CDSEG_CONV_BM", 128);
  const bool tall = GATHER && NCH == 16 && conv_bm >= 128 && p.M > 64 && !ln;
  // deep stages (C >= 256: few rows, 3.5 - 14 MB of weights per conv): 256-row tiles, 16 waves - per K step the block
  // moves 64 KB of A + 32 KB of W for 2048 MFMA cycles instead of 32 + 32 KB for 1024
  static const int deep_bm = cdseg_knob("CDSEG_CONV_DEEP_BM", 256);
  const bool deep = tall && sizeof(CT) == 2 && deep_bm == 256 && p.N >= 256 && p.M >= 512;
  int bm = deep ? 256 : (tall ? 128 : 64);
  // fp32 x3, plain products over many rows on 128-row tiles (the tile's W slice split into half pairs once per 128 rows instead
  // of once per 64): measured SLOWER - 45.1 -> 50.9 ms per 8-scene forward, the K > 64 products 74 -> 244 us per launch on two
  // 65 KB blocks per CU (profiles/r06_fp32x3.txt) - and off (0); experimental builds can switch it on
  bool tall_x3 = false;
  if constexpr (kIsX3<CT> && !GATHER) {
    static const long x3_tall_m = cdseg_knob("CDSEG_X3_TALL_MIN_M", 0);
    tall_x3 = x3_tall_m > 0 && !ln && p.M >= x3_tall_m;
    if (tall_x3) bm = 128;
  }
  // LDS-DMA main loop (bf16, K % 64 == 0, N > 64): 128-row tiles for plain linears too (CDSEG_GEMM_DMA_BM overrides)
  int dma_use_bm = -1;
  if constexpr (NCH == 16 && sizeof(CT) == 2) {
    static const int dma_on0 = cdseg_knob("CDSEG_GEMM_DMA", 1);
    static const int dma_bm0 = cdseg_knob("CDSEG_GEMM_DMA_BM", 0);
    if (dma_on0 && p.N > 64 && (p.K % 64) == 0 && (!GATHER || (p.kshift >= 6 && p.kvol <= 27)) && p.M >= 128) {
      // measured (tools/bench_gemm.py --scenes 8): 128-row tiles win for the sparse convs, for long row counts and for the
      // wide deep-stage linears (qkv / fc1 at C = 512: 34.9 / 39.0 us vs 48.9 / 50.3 with 64-row tiles)
      int want = dma_bm0 ? dma_bm0 : ((GATHER || p.M >= 16384 || (p.N >= 1024 && p.M >= 2048)) ? 128 : 64);
      if (ln && p.N <= 128) want = -1;  // complete rows in one block: the fused-LayerNorm epilogue lives in the 64-row loop
      if (want > 0) bm = dma_use_bm = want;
    }
  }
  // deep sparse convs (C >= 256) at 8+ scenes: 256 x 256 tiles, 8 waves as 4 x 2, split-K (gemm_dma_kernel<256, true, 256, 2, true>)
  bool sq = false;
  int sq_bn = 256;
  if constexpr (GATHER && NCH == 16 && sizeof(CT) == 2) {
    static const int sq_on = cdseg_knob("CDSEG_CONV_SQ", 1);
    static const int sq_min_m = cdseg_knob("CDSEG_CONV_SQ_MIN_M", 5000);
    static const int sq_min_n = cdseg_knob("CDSEG_CONV_SQ_MIN_N", 256);
    sq = sq_on && dma_use_bm == 128 && p.kvol == 27 && p.N >= sq_min_n && (p.N % 256) == 0 && !ln && p.kshift >= 6 &&
         p.M >= sq_min_m && p.ws && p.vec_ok && !p.out_idx;
    // C = 128 (round 5): the same 8-wave loop on a 256 x 128 tile (a wave owns 64 x 64: acc[4][4]) - 6 DMA pieces per 32
    // MFMAs instead of the 128 x 128 tile's 4 per 16.  Pays on single scenes only (see CDSEG_CONV_SQ128_MIN_M above)
    static const int sq128_min_m = cdseg_knob("CDSEG_CONV_SQ128_MIN_M", CDSEG_CONV_SQ128_MIN_M);
    static const int sq128_max_m = cdseg_knob("CDSEG_CONV_SQ128_MAX_M", CDSEG_CONV_SQ128_MAX_M);
    if (!sq && sq_on && sq128_min_m > 0 && dma_use_bm == 128 && p.kvol == 27 && p.N == 128 && !ln && p.kshift >= 6 &&
        p.M >= sq128_min_m && p.M < sq128_max_m && p.vec_ok && !p.out_idx) {
      sq = true;
      sq_bn = 128;
    }
    if (sq) {
      bm = 256;
      p.alt = cdseg_knob("CDSEG_CONV_SQ_ALT", 1);
    }
  }
  const int gm = (int)((p.M + bm - 1) / bm);
  // deep sparse convs (C >= 256): 256-column tiles on a three-stage ring, one block per CU (gemm_dma_kernel<128, true, 256, 3>)
  bool wide = false;
  if constexpr (GATHER && NCH == 16 && sizeof(CT) == 2) {
    static const int wide_on = cdseg_knob("CDSEG_CONV_WIDE", 0);  // measured slower (r04): see DESIGN 4.2
    wide = !sq && wide_on && dma_use_bm == 128 && p.kvol == 27 && p.N >= 256 && (p.N % 256) == 0 && !ln;
  }
  // wide tiles (better FLOP/byte against L2); few-tile problems get their parallelism from split-K instead
  const int bn = wide ? 256 : (sq ? sq_bn : (p.N <= 32 ? 32 : (p.N <= 64 ? 64 : 128)));
  const int gn = (p.N + bn - 1) / bn;
  // split-K when the output tiles alone cannot fill the chip (deep stages: few points, long reductions)
  int splits = 1;
  const long nkc = ((long)p.kvol * p.K + BK - 1) / BK;
  const long blocks = (long)gm * gn;
  const bool can_fix = p.ws && p.vec_ok && !p.out_idx;
  static const int plain_min_nkc = cdseg_knob("CDSEG_GEMM_SPLIT_MIN_NKC", 16);
  if (can_fix && blocks < 256 && nkc >= (GATHER ? 4 : plain_min_nkc)) {
    int smax = (int)(nkc / 2);
    if (smax > split_max) smax = split_max;
    while (smax > 1 && (size_t)smax * p.M * p.N * sizeof(float) > ws_bytes) --smax;
    if (wide || sq) {
      // one block per CU: as many K slices as keep the grid within one round of the chip
      splits = (int)(256 / blocks);
    } else {
      splits = (int)((split_target + blocks - 1) / blocks);
    }
    if (splits > smax) splits = smax;
    // prefer a slice count (column tiles x splits) that spreads evenly over the 8 XCDs
    if (!wide && !sq)
      for (int c = splits; c <= smax && c < splits + 4; ++c)
        if ((gn * c) % 8 == 0) { splits = c; break; }
    if (splits < 1) splits = 1;
  }
  p.fix = 0;
  if (splits > 1) p.fix = ln ? 2 : 1;
  else if (ln && gn > 1) p.fix = 2;
  if (p.fix) {
    if (!can_fix || (size_t)splits * p.M * p.N * sizeof(float) > ws_bytes) return CDSEG_ERR_WORKSPACE;
    if (p.fix == 2 && p.N > 512) return CDSEG_ERR_UNSUPPORTED;
  }
  p.splits = splits;
  p.gm = gm;
  p.gn = gn;
  const int slices = gn * splits;
  const long wbytes = (long)p.N * p.kvol * p.K * (long)sizeof(CT);
  p.xmode = 0;
  if (sq) p.xmode = gm >= 64 ? 2 : 0;
  else if (slices >= 8 && wbytes >= (1 << 20) && gm > 1) p.xmode = 1;
  else if (GATHER && gm >= 64) p.xmode = 2;
  if (xmode_env >= 0) p.xmode = (xmode_env == 1 && slices < 8) ? 0 : xmode_env;
  unsigned nblk;
  if (p.xmode == 1) nblk = 8u * (unsigned)((slices + 7) / 8) * (unsigned)gm;
  else if (p.xmode == 2) nblk = 8u * (unsigned)((gm + 7) / 8) * (unsigned)slices;
  else nblk = (unsigned)gm * (unsigned)slices;
  const dim3 grid(nblk);
  bool launched = false;
  // bf16, K a multiple of 64, wide outputs: the LDS-DMA pipelined main loop
  if constexpr (NCH == 16 && sizeof(CT) == 2) {
    static const int dma_on = cdseg_knob("CDSEG_GEMM_DMA", 1);
    static const int dma_bm = cdseg_knob("CDSEG_GEMM_DMA_BM", 0);
    const bool fused_ln_here = ln && gn == 1 && splits == 1;  // complete rows in one 64-row block: stays on the old loop
    if constexpr (GATHER) {
      if (sq && dma_on && sq_bn == 128) {
        launched = true;
        using Q = DmaCfg<256, true, 128, 2, true>;
        static std::atomic<bool> aq128{false};
        if (!aq128) {
          if (hipFuncSetAttribute((const void*)gemm_dma_kernel<256, true, 128, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  Q::LDS) != hipSuccess)
            return CDSEG_ERR_LAUNCH;
          aq128 = true;
        }
        hipLaunchKernelGGL((gemm_dma_kernel<256, true, 128, 2, true>), grid, dim3(Q::NT), Q::LDS, s, p);
      }
      if (!launched && sq && dma_on) {
        launched = true;
        using Q = DmaCfg<256, true, 256, 2, true>;
        static std::atomic<bool> aq{false};  // (a concurrent first call sets the attribute twice: harmless)
        if (!aq) {
          if (hipFuncSetAttribute((const void*)gemm_dma_kernel<256, true, 256, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  Q::LDS) != hipSuccess)
            return CDSEG_ERR_LAUNCH;
          aq = true;
        }
        hipLaunchKernelGGL((gemm_dma_kernel<256, true, 256, 2, true>), grid, dim3(Q::NT), Q::LDS, s, p);
      }
      if (!launched && wide && dma_on && p.kshift >= 6) {
        launched = true;
        using W = DmaCfg<128, true, 256, 3>;
        static std::atomic<bool> aw{false};  // (a concurrent first call sets the attribute twice: harmless)
        if (!aw) {
          if (hipFuncSetAttribute((const void*)gemm_dma_kernel<128, true, 256, 3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  W::LDS) != hipSuccess)
            return CDSEG_ERR_LAUNCH;
          aw = true;
        }
        hipLaunchKernelGGL((gemm_dma_kernel<128, true, 256, 3>), grid, dim3(512), W::LDS, s, p);
      }
    }
    if (!launched && dma_on && bn == 128 && (p.K % 64) == 0 && (!GATHER || (p.kshift >= 6 && p.kvol <= 27)) && !fused_ln_here &&
        p.M >= 128 && dma_use_bm == bm) {
      launched = true;
      (void)dma_bm;
      if (bm == 256) {
        static std::atomic<bool> a256{false};  // (a concurrent first call sets the attribute twice: harmless)
        if (!a256) {
          if (hipFuncSetAttribute((const void*)gemm_dma_kernel<256, GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  DmaCfg<256, GATHER>::LDS) != hipSuccess)
            return CDSEG_ERR_LAUNCH;
          a256 = true;
        }
        constexpr int lds256 = DmaCfg<256, GATHER>::LDS;
        hipLaunchKernelGGL((gemm_dma_kernel<256, GATHER>), grid, dim3(1024), lds256, s, p);
      } else if (bm == 128) {
        static std::atomic<bool> a128{false};  // (a concurrent first call sets the attribute twice: harmless)
        if (!a128) {
          if (hipFuncSetAttribute((const void*)gemm_dma_kernel<128, GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  DmaCfg<128, GATHER>::LDS) != hipSuccess)
            return CDSEG_ERR_LAUNCH;
          a128 = true;
        }
        constexpr int lds128 = DmaCfg<128, GATHER>::LDS;
        hipLaunchKernelGGL((gemm_dma_kernel<128, GATHER>), grid, dim3(512), lds128, s, p);
      } else {
        static std::atomic<bool> a64{false};  // (a concurrent first call sets the attribute twice: harmless)
        if (!a64) {
          if (hipFuncSetAttribute((const void*)gemm_dma_kernel<64, GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  DmaCfg<64, GATHER>::LDS) != hipSuccess)
            return CDSEG_ERR_LAUNCH;
          a64 = true;
        }
        constexpr int lds64 = DmaCfg<64, GATHER>::LDS;
        hipLaunchKernelGGL((gemm_dma_kernel<64, GATHER>), grid, dim3(256), lds64, s, p);
      }
    }
  }
  if constexpr (GATHER && NCH == 16 && sizeof(CT) == 2) {
    if (deep && !launched) {
      launched = true;
      constexpr int smem = gemm_smem_bytes<CT, 128, 16, 256>();  // 97 KB: 1 block / CU, 4 waves / SIMD
      static std::atomic<bool> attr_done256{false};  // (a concurrent first call sets the attribute twice: harmless)
      if (!attr_done256) {
        if (hipFuncSetAttribute((const void*)gemm_kernel<CT, 128, 16, GATHER, 256>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
          return CDSEG_ERR_LAUNCH;
        attr_done256 = true;
      }
      hipLaunchKernelGGL((gemm_kernel<CT, 128, 16, GATHER, 256>), grid, dim3(1024), smem, s, p);
    }
  }
  if constexpr (GATHER && NCH == 16) {
    if (tall && !launched) {
      launched = true;
      if (bn == 32) {
        hipLaunchKernelGGL((gemm_kernel<CT, 32, 16, GATHER, 128>), grid, dim3(512), 0, s, p);
      } else if (bn == 64) {
        hipLaunchKernelGGL((gemm_kernel<CT, 64, 16, GATHER, 128>), grid, dim3(512), 0, s, p);
      } else {
        constexpr int smem = gemm_smem_bytes<CT, 128, 16, 128>();  // 65 KB: above the static LDS limit
        static std::atomic<bool> attr_done{false};  // (a concurrent first call sets the attribute twice: harmless)
        if (!attr_done) {
          if (hipFuncSetAttribute((const void*)gemm_kernel<CT, 128, 16, GATHER, 128>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return CDSEG_ERR_LAUNCH;
          attr_done = true;
        }
        hipLaunchKernelGGL((gemm_kernel<CT, 128, 16, GATHER, 128>), grid, dim3(512), smem, s, p);
      }
    }
  }
  if constexpr (kIsX3<CT> && !GATHER) {
    if (tall_x3 && !launched) {
      launched = true;
      if (bn == 32) {
        hipLaunchKernelGGL((gemm_kernel<CT, 32, NCH, false, 128>), grid, dim3(512), 0, s, p);
      } else if (bn == 64) {
        hipLaunchKernelGGL((gemm_kernel<CT, 64, NCH, false, 128>), grid, dim3(512), 0, s, p);
      } else {
        constexpr int smem = gemm_smem_bytes<CT, 128, NCH, 128>();
        if constexpr (smem > 65536) {
          static std::atomic<bool> attr_x3{false};  // (a concurrent first call sets the attribute twice: harmless)
          if (!attr_x3) {
            if (hipFuncSetAttribute((const void*)gemm_kernel<CT, 128, NCH, false, 128>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
              return CDSEG_ERR_LAUNCH;
            attr_x3 = true;
          }
          hipLaunchKernelGGL((gemm_kernel<CT, 128, NCH, false, 128>), grid, dim3(512), smem, s, p);
        } else {
          hipLaunchKernelGGL((gemm_kernel<CT, 128, NCH, false, 128>), grid, dim3(512), 0, s, p);
        }
      }
    }
  }
  if (!launched) {
    if (bn == 32) hipLaunchKernelGGL((gemm_kernel<CT, 32, NCH, GATHER>), grid, dim3(256), 0, s, p);
    else if (bn == 64) hipLaunchKernelGGL((gemm_kernel<CT, 64, NCH, GATHER>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_kernel<CT, 128, NCH, GATHER>), grid, dim3(256), 0, s, p);
  }
  if (hipGetLastError() != hipSuccess) return CDSEG_ERR_LAUNCH;
  if (p.fix == 2) {
    hipLaunchKernelGGL(row_finish_kernel, dim3((unsigned)((p.M + 3) / 4)), dim3(256), 0, s, p);
    if (hipGetLastError() != hipSuccess) return CDSEG_ERR_LAUNCH;
  } else if (p.fix == 1) {
    // (gemm_leave_partials: the consumer - the deep Block head - sums the slices and adds the bias itself, in this order)
    if (tl_leave_partials && p.vec_ok && p.bias && !p.scale && p.act == CDSEG_ACT_NONE && !p.res && !p.add_src && !p.out2 &&
        !p.out_idx && !p.colbias) {
      tl_partial_splits = p.splits;
      return CDSEG_OK;
    }
    const long groups = p.M * (long)(p.N >> 2);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, p);
    if (hipGetLastError() != hipSuccess) return CDSEG_ERR_LAUNCH;
  }
  return CDSEG_OK;
}

// (A register-resident "streaming Linear" for the shallow products around the Blocks - K = 32 / 64, N = 64, whole weight
// in MFMA fragments, 16-row groups straight from global memory, no LDS - was built and measured in round 4: 63 instead
// of 90 us on the plain 960k x 32 -> 64 projection, but 335 us on the un-pooling form (gather-add + two outputs through
// 16-byte pieces) and -0.8 % end to end in a same-box A/B, profiles/r04_ab_stream_linear.txt: dropped.)
template <typename CT, bool GATHER>
int launch(const GemmP& p, size_t ws_bytes, hipStream_t s) {
  const long ktot = (long)p.kvol * p.K;
  // narrow K step (one 32-wide MFMA block) only when the whole reduction is that short
  if (ktot <= 64) {
    if constexpr (sizeof(CT) == 2) return launch_bn<CT, 4, GATHER>(p, ws_bytes, s);
    else return launch_bn<CT, 8, GATHER>(p, ws_bytes, s);
  }
  return launch_bn<CT, 16, GATHER>(p, ws_bytes, s);
}

}  // namespace

extern "C" int cdseg_gemm(const cdseg_gemm_args* a, void* stream) {
  if (!a || !a->A || !a->W || !a->out) return CDSEG_ERR_ARG;
  if (a->M <= 0 || a->N <= 0) return CDSEG_OK;
  if (a->K <= 0 || (a->K & 7) || a->K > 65536 || a->kvol <= 0 || a->kvol > 128) return CDSEG_ERR_ARG;
  // (CDSEG_F32X3: fp32 operands in memory, split-half arithmetic on the matrix pipe)
  if (a->a_dtype != a->compute_dtype && !(a->compute_dtype == CDSEG_F32X3 && a->a_dtype == CDSEG_F32)) return CDSEG_ERR_UNSUPPORTED;
  if (a->scale && !a->shift) return CDSEG_ERR_ARG;
  if (a->add_src && !a->add_idx) return CDSEG_ERR_ARG;
  if (!a->nbr && a->kvol != 1) return CDSEG_ERR_ARG;
  const int esz = a->compute_dtype == CDSEG_BF16 ? 2 : 4;
  if (((long)a->lda * esz) & 15) return CDSEG_ERR_ARG;  // 16-byte row alignment for the vector loads
  GemmP p;
#ifdef CDSEG_EXPERIMENTS
  p.dbg = cdseg_knob("CDSEG_GEMM_DBG", 0);
#endif
  p.A = a->A; p.W = a->W; p.bias = a->bias; p.scale = a->scale; p.shift = a->shift; p.res = a->res;
  p.add_src = a->add_src; p.add_idx = a->add_idx; p.nbr = a->nbr; p.out_idx = a->out_idx;
  p.out = a->out; p.out2 = a->out2; p.M = a->M; p.N = a->N; p.K = a->K; p.kvol = a->kvol;
  p.lda = a->lda; p.ldo = a->ldo; p.ldo2 = a->ldo2; p.ldres = a->ldres; p.ldadd = a->ldadd;
  p.out_dtype = a->out_dtype; p.out2_dtype = a->out2_dtype; p.act = a->act; p.out2_pre_add = a->out2_pre_add;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  p.vec_ok = (a->N % 4 == 0) && (a->ldo % 4 == 0) && al16(a->out) && (!a->out2 || (a->ldo2 % 4 == 0 && al16(a->out2))) &&
             (!a->res || (a->ldres % 4 == 0 && al16(a->res))) &&
             (!a->add_src || (a->ldadd % 4 == 0 && al16(a->add_src))) && (!a->bias || al16(a->bias)) &&
             (!a->scale || (al16(a->scale) && al16(a->shift)));
  p.nbr_sm = a->nbr_kmajor ? 1 : a->kvol;
  p.nbr_so = a->nbr_kmajor ? a->M : 1;
  p.colbias = a->colbias; p.ln_pre_g = a->ln_pre_g; p.ln_pre_b = a->ln_pre_b; p.ln_post_g = a->ln_post_g;
  p.ln_post_b = a->ln_post_b; p.ln_out = a->ln_out; p.ldln = a->ldln; p.ln_out_dtype = a->ln_out_dtype;
  p.ln_eps = a->ln_eps;
  if (p.ln_pre_g || p.ln_post_g) {
    // complete rows are finished by one block (N <= 128 directly, wider rows / split-K through the workspace);
    // rows are processed as float4 groups; no row scatter
    if (a->N > 512 || !p.vec_ok || a->out_idx || (p.ln_pre_g && !p.ln_pre_b) || (p.ln_post_g && (!p.ln_post_b || !a->ln_out)) ||
        (a->ln_out && (a->ldln % 4)))
      return CDSEG_ERR_UNSUPPORTED;
  }
  if (p.colbias && !al16(p.colbias)) p.vec_ok = 0;
  p.ws = (a->ws && ((((uintptr_t)a->ws) & 15) == 0)) ? (float*)a->ws : nullptr;
  p.splits = 1;
  p.fix = 0;
  p.alt = cdseg_knob("CDSEG_GEMM_ALT", 0);  // measured 0 ... +10 % step time (profiles/r04_conv_timing.txt)
  p.gm = p.gn = 1;
  p.xmode = 0;
  p.kshift = -1;
  if ((a->K & (a->K - 1)) == 0) {
    int sh = 0;
    while ((1 << sh) < a->K) ++sh;
    p.kshift = sh;
  }
  const size_t wsb = p.ws ? a->ws_bytes : 0;
  hipStream_t s = (hipStream_t)stream;
  CdsegProfToken tok;
  const bool prof = a->nbr && a->kvol == 27 && cdseg_prof_begin(CDSEG_PROF_CONV_DEEP, s, &tok);  // the k = 3 sparse convs on this kernel
  int rc = CDSEG_ERR_ARG;
  if (a->compute_dtype == CDSEG_BF16) rc = a->nbr ? launch<bf16_t, true>(p, wsb, s) : launch<bf16_t, false>(p, wsb, s);
  else if (a->compute_dtype == CDSEG_F32) rc = a->nbr ? launch<float, true>(p, wsb, s) : launch<float, false>(p, wsb, s);
  else if (a->compute_dtype == CDSEG_F32X3) rc = a->nbr ? launch<F32X3, true>(p, wsb, s) : launch<F32X3, false>(p, wsb, s);
  if (prof) cdseg_prof_end(tok, s);
  return rc;
}

// Internal (deep.h; csrc/runtime.hip): cdseg_gemm, except that a split-K launch whose epilogue is a bias add only does NOT run its
// second pass - *splits (> 1) raw partial planes (splits, M, N) fp32 stay in a->ws for the consumer, which sums them in slice
// order and adds the bias (what splitk_epilogue_kernel + epilogue4 do, bit for bit: csrc/deep.hip load_tile_partials).  *splits
// = 1: the launch did not split (or its epilogue is more than a bias) and a->out holds the result as usual.  Saves a launch per
// sparse conv of a few-row deep stage (a single scene: 14 of 242).
int gemm_leave_partials(const cdseg_gemm_args* a, int* splits, void* stream) {
  tl_leave_partials = true;
  tl_partial_splits = 1;
  const int rc = cdseg_gemm(a, stream);
  tl_leave_partials = false;
  *splits = tl_partial_splits;
  return rc;
}
