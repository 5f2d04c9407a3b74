// Shared device helpers for the CDSegNet hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cdseg.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits (storage type)

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA bf16 A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator

#define CDSEG_WAVE 64

#define CDSEG_CHECK_LAUNCH()                                   \
  do {                                                         \
    hipError_t e__ = hipGetLastError();                        \
    if (e__ != hipSuccess) return CDSEG_ERR_LAUNCH;            \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}

// float -> bfloat16, round-to-nearest-even (= torch's cast): the compiler lowers these to
// gfx950's v_cvt_pk_bf16_f32, one instruction per pair
typedef __bf16 hw_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float hw_f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  const __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const hw_f32x2_t v = {lo, hi};
  const hw_bf16x2_t b = __builtin_convertvector(v, hw_bf16x2_t);
  return __builtin_bit_cast(uint32_t, b);
}

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
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
  const float erf_abs = 1.0f - p * t * e;  // erf(|x| / sqrt 2)
  return 0.5f * x + 0.5f * fabsf(x) * erf_abs;  // 0.5 x (1 + sign(x) erf_abs)
}

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
