// Micro-benchmark: issue rate of the VALU instructions the attention softmax is made of (gfx950).
// One wave per SIMD slot pattern: grid = 256 CUs * 4 SIMDs * W waves; each wave runs ITER iterations of 16
// independent ops of one kind.  Prints cycles per wave-instruction (per SIMD) for W = 1 and W = 4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 hw_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(float* out, int iters, float a, float b) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = a * (threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) v[i] = __builtin_fmaf(v[i], a, b);
      else if (KIND == 1) v[i] = __builtin_amdgcn_exp2f(v[i]);
      else if (KIND == 2) v[i] = fmaxf(fmaxf(v[i], a), v[(i + 1) & 15]);
      else if (KIND == 3) { f32x2_t t = {v[i], v[(i + 1) & 15]}; hw_bf16x2_t h = __builtin_convertvector(t, hw_bf16x2_t); v[i] += (float)h[0]; }
      else if (KIND == 4 && (i & 1) == 0) { f32x2_t t = {v[i], v[i + 1]}; f32x2_t aa = {a, a}, bb = {b, b}; t = __builtin_elementwise_fma(t, aa, bb); v[i] = t[0]; v[i + 1] = t[1]; }
      else if (KIND == 5) v[i] = __builtin_amdgcn_fractf(v[i]) + b;
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 12345.678f) out[0] = s;
}

template <int KIND>
void run(const char* name, int ops_per_iter) {
  float* d; hipMalloc(&d, 4);
  const int iters = 4096;
  for (int W : {1, 2, 4}) {
    dim3 grid(256 * 4 * W / 4), block(256);  // 4 waves per block -> one per SIMD; W blocks per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d, 16, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: W waves * iters * ops instructions in ms
    double inst_per_simd = (double)W * iters * ops_per_iter;
    printf("%-22s W=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (x clock GHz = cycles)\n", name, W, ms, ms * 1e6 / inst_per_simd);
  }
}
int main() {
  run<0>("v_fma_f32", 16);
  run<1>("v_exp_f32", 16);
  run<2>("v_max3_f32", 16);
  run<3>("v_cvt_pk_bf16_f32+add", 16);
  run<4>("v_pk_fma_f32 (8/iter)", 8);
  run<5>("v_fract_f32+add", 16);
  return 0;
}
