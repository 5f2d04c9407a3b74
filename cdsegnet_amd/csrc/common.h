// Shared device helpers for the CDSegNet hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cdseg.h"

// The 16-bit storage / MFMA operand type of the build.  The library is compiled twice from the same sources:
//   libcdseg_hip.so      bfloat16 (8-bit mantissa, fp32's exponent range)                      precision "bf16*"
//   libcdseg_hip_f16.so  IEEE half (-DCDSEG_LP_F16: 11-bit mantissa, |x| <= 65504, conversions SATURATE) precision "fp16*"
// Same MFMA rate, same bytes; every name below keeps "bf16" (the default build) and means "the build's 16-bit type";
// the ABI's CDSEG_BF16 dtype code likewise.  The reference's own GPU path computes its attention in half
// (point_transformer_v3m1_base.py:282 `qkv.half()`).
typedef uint16_t bf16_t;  // raw bits (storage type)

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA 16-bit A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator

#define CDSEG_WAVE 64

#define CDSEG_CHECK_LAUNCH()                                   \
  do {                                                         \
    hipError_t e__ = hipGetLastError();                        \
    if (e__ != hipSuccess) return CDSEG_ERR_LAUNCH;            \
  } while (0)

typedef float hw_f32x2_t __attribute__((ext_vector_type(2)));

// bfloat16 whatever the build's 16-bit type is (the attention kernel keeps its probabilities in bfloat16: they need
// fp32's exponent range, see attention.hip)
typedef __bf16 hw_truebf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_truebf16x2(float lo, float hi) {
  const hw_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_truebf16x2_t));
}
__device__ __forceinline__ f32x16_t mfma_32x32x16_truebf16(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

#ifdef CDSEG_LP_F16
typedef _Float16 hw_lp_t;
typedef _Float16 hw_lpx2_t __attribute__((ext_vector_type(2)));
typedef _Float16 hw_lpx8_t __attribute__((ext_vector_type(8)));
constexpr uint32_t LP_ONE_BITS = 0x3C00u;  // 1.0
constexpr bool LP_IS_F16 = true;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }

// both halves of a packed pair (v_cvt_f32_f16 + its SDWA form)
__device__ __forceinline__ void unpack_bf16x2(uint32_t u, float& lo, float& hi) {
  const hw_lpx2_t h = __builtin_bit_cast(hw_lpx2_t, u);
  lo = (float)h[0];
  hi = (float)h[1];
}

// float -> half, round-to-nearest-even, saturating at +-65504 (v_med3_f32 + v_cvt_pk_f16_f32): an activation outlier
// of a trained checkpoint clamps instead of turning the rest of the forward into inf / NaN
// (CDSEG_F16_NO_CLAMP: tools-only A/B switch that measures what the two clamps per conversion cost - never in the product)
#ifdef CDSEG_F16_NO_CLAMP
#define CDSEG_SAT(x) (x)
#else
#define CDSEG_SAT(x) __builtin_amdgcn_fmed3f((x), -65504.f, 65504.f)
#endif
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  const _Float16 b = (_Float16)CDSEG_SAT(f);
  return __builtin_bit_cast(uint16_t, b);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const hw_f32x2_t v = {CDSEG_SAT(lo), CDSEG_SAT(hi)};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_lpx2_t));
}

// values known to lie inside the range (probabilities, pre-scaled operands): no clamp
__device__ __forceinline__ uint32_t pack_bf16x2_inrange(float lo, float hi) {
  const hw_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_lpx2_t));
}

__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(hw_lpx2_t, a), __builtin_bit_cast(hw_lpx2_t, b), c, false);
}

__device__ __forceinline__ f32x4_t mfma_16x16x32_bf16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hw_lpx8_t, a), __builtin_bit_cast(hw_lpx8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16_t mfma_32x32x16_bf16(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hw_lpx8_t, a), __builtin_bit_cast(hw_lpx8_t, b), c, 0, 0, 0);
}
#else
typedef __bf16 hw_lp_t;
typedef __bf16 hw_lpx2_t __attribute__((ext_vector_type(2)));
constexpr uint32_t LP_ONE_BITS = 0x3F80u;  // 1.0
constexpr bool LP_IS_F16 = false;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}

__device__ __forceinline__ void unpack_bf16x2(uint32_t u, float& lo, float& hi) {
  lo = __uint_as_float(u << 16);
  hi = __uint_as_float(u & 0xffff0000u);
}

// float -> bfloat16, round-to-nearest-even (= torch's cast): the compiler lowers these to
// gfx950's v_cvt_pk_bf16_f32, one instruction per pair
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  const __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const hw_f32x2_t v = {lo, hi};
  const hw_lpx2_t b = __builtin_convertvector(v, hw_lpx2_t);
  return __builtin_bit_cast(uint32_t, b);
}

__device__ __forceinline__ uint32_t pack_bf16x2_inrange(float lo, float hi) { return pack_bf16x2(lo, hi); }

__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hw_lpx2_t, a), __builtin_bit_cast(hw_lpx2_t, b), c, false);
}

__device__ __forceinline__ f32x4_t mfma_16x16x32_bf16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16_t mfma_32x32x16_bf16(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
#endif

template <typename T> struct Cvt;
template <> struct Cvt<float> {
  static __device__ __forceinline__ float to_f32(float v) { return v; }
  static __device__ __forceinline__ float from_f32(float v) { return v; }
};
template <> struct Cvt<bf16_t> {
  static __device__ __forceinline__ float to_f32(bf16_t v) { return bf16_to_f32(v); }
  static __device__ __forceinline__ bf16_t from_f32(float v) { return f32_to_bf16(v); }
};

// erf-form GELU (torch.nn.GELU() default).  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, below fp32 GEMM
// rounding): one v_rcp + one v_exp + a 5-term Horner chain.  libm's erff costs ~3x as much and was 40 % of the
// fc1 (Linear -> GELU) launches on the 120k-point stages.
__device__ __forceinline__ float gelu_erf(float x) {
  // every rounding is written out (no implicit contraction): the same inputs give the same bits in every kernel that
  // inlines this - cdseg_pool_fused and cdseg_segment_max evaluate the same epilogue and are compared bit for bit
#pragma clang fp contract(off)
  const float ax = fabsf(x);
  const float z = ax * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float zz = z * z;
  const float e = __builtin_amdgcn_exp2f(zz * -1.44269504088896340736f);
  const float pt = p * t;
  const float erf_abs = __builtin_fmaf(-pt, e, 1.0f);  // erf(|x| / sqrt 2)
  const float hx = 0.5f * x, hax = 0.5f * ax;
  return __builtin_fmaf(hax, erf_abs, hx);  // 0.5 x (1 + sign(x) erf_abs)
}

// erf-form GELU where the result is rounded to the 16-bit type next (hidden activations of the MLPs): x Phi(x) with
// Phi(x) - 1/2 = x Q(x^2), Q a degree-7 weighted least-squares fit on |x| <= 4 rescaled so that Phi(+-4) = 1 / 0 exactly
// (the argument is clamped there).  |error| <= 6.5e-5 max(|x|, 1) ABSOLUTE (tools/fit_gelu.py): below the rounding of the
// 16-bit value it becomes wherever |GELU(x)| is of order 1 or larger, but NOT a relative bound - on the negative tail
// (x in [-4, -2], GELU between -4.5e-2 and -4e-3) it is tens of half ulps / > 10 bfloat16 ulps of the small result, and
// beyond |x| = 4 the result is exactly x or 0.  An absolute error of 6.5e-5 on a hidden unit is what the 16-bit rounding of
// the LARGER units of the same row already costs the fc2 dot product, which is why the end-to-end bounds did not move
// (profiles/r04_parity_measured.txt); it is a deviation from the reference's exact GELU all the same.  12 full-rate VALU
// operations on two values at a time (v_pk_mul_f32 / v_pk_fma_f32), no transcendental; gelu_erf above costs ~3x that and
// was the VALU half of the Block tails.  fp32 outputs keep gelu_erf.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_lp2(f32x2_t x) {
  f32x2_t xc;
  xc[0] = __builtin_amdgcn_fmed3f(x[0], -4.f, 4.f);
  xc[1] = __builtin_amdgcn_fmed3f(x[1], -4.f, 4.f);
  const f32x2_t t = xc * xc;
  f32x2_t q = {-1.2454853377e-09f, -1.2454853377e-09f};
  q = q * t + 1.0064627976e-07f;
  q = q * t + -3.5656154246e-06f;
  q = q * t + 7.3631240712e-05f;
  q = q * t + -9.9749399351e-04f;
  q = q * t + 9.4701653904e-03f;
  q = q * t + -6.5824832133e-02f;
  q = q * t + 3.9866018915e-01f;
  const f32x2_t phi = xc * q + 0.5f;
  return x * phi;
}
__device__ __forceinline__ void gelu_lp4(f32x4_t& v) {
  const f32x2_t a = gelu_lp2(f32x2_t{v[0], v[1]}), b = gelu_lp2(f32x2_t{v[2], v[3]});
  v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
}
__device__ __forceinline__ float gelu_lp(float x) { return gelu_lp2(f32x2_t{x, x})[0]; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// LDS-DMA (global_load_lds_dwordx4): 64 lanes x 16 B from per-lane global addresses to the lane-linear LDS range
// [lds_dst, lds_dst + 1024).  Issued from inline asm with M0 (the LDS base) written in the same statement: hipcc drains
// vmcnt(0) around compiler-visible LDS-DMA.  The compiler does not count it: wait with an explicit s_waitcnt vmcnt.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst /* wave-uniform LDS byte address */) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// Tuning knobs are constants in the product library.  Experimental builds (-DCDSEG_EXPERIMENTS, tools/build_ab.py ->
// tools/_ab/, loaded by the benchmark tools only) read them from the environment for A/B runs.
#ifdef CDSEG_EXPERIMENTS
#include <cstdlib>
static inline int cdseg_knob(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
#else
#define cdseg_knob(name, dflt) (dflt)
#endif

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
