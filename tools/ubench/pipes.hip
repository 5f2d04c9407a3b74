// Micro-benchmark (gfx950): do the transcendental unit, the plain VALU and the matrix pipe of one SIMD overlap?
// Decides how the attention softmax is split between v_exp_f32 and a full-rate polynomial exp2.
// Each wave times its own loop with s_memtime (shader cycles); the host prints cycles per loop iteration for
// W = 1, 2, 4 waves per SIMD (grid = 256 CUs x W blocks of 256 threads, so every SIMD holds exactly W waves).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/pipes tools/ubench/pipes.hip ; run: tools/ubench/pipes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define EXP(i) asm volatile("v_exp_f32 %0, %1" : "=v"(e[i]) : "v"(x[i]))
#define FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(x[i]), "v"(c0))
#define PERM(i) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(p[i]) : "v"(e[2 * (i)]), "v"(e[2 * (i) + 1]), "s"(0x07060302u))
#define MFMA(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA16(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// KIND: which instruction mix one loop iteration contains
//  0: 16 exp                       1: 16 fma                      2: 16 exp + 16 fma (interleaved)
//  3: 16 exp + 48 fma              4: 3 mfma32                    5: 3 mfma32 + 16 exp
//  6: 3 mfma32 + 16 exp + 8 perm   7: 3 mfma32 + 10 exp + 42 fma + 8 perm (hybrid softmax)
//  8: 3 mfma32 + 64 fma            9: 8 exp + 56 fma             10: 3 mfma32 + 8 exp + 56 fma + 8 perm
// 11: 16 ldexp (v_ldexp_f32)      12: 16 v_cvt_pk_bf16_f32       13: 16 v_fract_f32
template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters, float seed) {
  float x[16], e[16], f[16];
  unsigned p[8];
  const float c0 = seed * 0.5f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { x[i] = seed * (float)(threadIdx.x + i) * 1e-3f - 1.0f; e[i] = 0.f; f[i] = x[i]; }
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = 0;
  bf16x8_t a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + i); b[i] = (short)(0x3f00 + threadIdx.x % 7); }
  f32x16_t acc0, acc1, acc2;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; }
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) { EXP(0); EXP(1); EXP(2); EXP(3); EXP(4); EXP(5); EXP(6); EXP(7); EXP(8); EXP(9); EXP(10); EXP(11); EXP(12); EXP(13); EXP(14); EXP(15); }
    if (KIND == 1) { FMA(0); FMA(1); FMA(2); FMA(3); FMA(4); FMA(5); FMA(6); FMA(7); FMA(8); FMA(9); FMA(10); FMA(11); FMA(12); FMA(13); FMA(14); FMA(15); }
    if (KIND == 2) {
#define EF(i) EXP(i); FMA(i)
      EF(0); EF(1); EF(2); EF(3); EF(4); EF(5); EF(6); EF(7); EF(8); EF(9); EF(10); EF(11); EF(12); EF(13); EF(14); EF(15);
    }
    if (KIND == 3) {
#define EF3(i) EXP(i); FMA(i); FMA((i + 5) & 15); FMA((i + 10) & 15)
      EF3(0); EF3(1); EF3(2); EF3(3); EF3(4); EF3(5); EF3(6); EF3(7); EF3(8); EF3(9); EF3(10); EF3(11); EF3(12); EF3(13); EF3(14); EF3(15);
    }
    if (KIND == 4) { MFMA(acc0); MFMA(acc1); MFMA(acc2); }
    if (KIND == 5) {
      MFMA(acc0); EXP(0); EXP(1); EXP(2); EXP(3); EXP(4);
      MFMA(acc1); EXP(5); EXP(6); EXP(7); EXP(8); EXP(9);
      MFMA(acc2); EXP(10); EXP(11); EXP(12); EXP(13); EXP(14); EXP(15);
    }
    if (KIND == 6) {
      MFMA(acc0); EXP(0); EXP(1); EXP(2); EXP(3); EXP(4); PERM(0); PERM(1);
      MFMA(acc1); EXP(5); EXP(6); EXP(7); EXP(8); EXP(9); PERM(2); PERM(3); PERM(4);
      MFMA(acc2); EXP(10); EXP(11); EXP(12); EXP(13); EXP(14); EXP(15); PERM(5); PERM(6); PERM(7);
    }
    if (KIND == 19) {  // the round-3 tile: RNE packs (v_cvt_pk_bf16_f32) instead of the truncating v_perm
#define CVTP(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[i]) : "v"(e[2 * (i)]), "v"(e[2 * (i) + 1]))
      MFMA(acc0); EXP(0); EXP(1); EXP(2); EXP(3); EXP(4); CVTP(0); CVTP(1);
      MFMA(acc1); EXP(5); EXP(6); EXP(7); EXP(8); EXP(9); CVTP(2); CVTP(3); CVTP(4);
      MFMA(acc2); EXP(10); EXP(11); EXP(12); EXP(13); EXP(14); EXP(15); CVTP(5); CVTP(6); CVTP(7);
    }
    if (KIND == 20) {  // the SAME instructions as 19 in the order the compiler emits them for the key loop: MFMAs back to back
      MFMA(acc0); MFMA(acc1); MFMA(acc2);
      EXP(0); EXP(1); EXP(2); EXP(3); EXP(4); EXP(5); EXP(6); EXP(7); EXP(8); EXP(9); EXP(10); EXP(11); EXP(12); EXP(13); EXP(14); EXP(15);
      CVTP(0); CVTP(1); CVTP(2); CVTP(3); CVTP(4); CVTP(5); CVTP(6); CVTP(7);
    }
    if (KIND == 21) {  // two tiles per iteration, grouped like the real loop body: 2 QK MFMAs, 32 exp, 16 packs, 4 PV MFMAs
      MFMA(acc0); MFMA(acc1);
      for (int r = 0; r < 2; ++r) { EXP(0); EXP(1); EXP(2); EXP(3); EXP(4); EXP(5); EXP(6); EXP(7); EXP(8); EXP(9); EXP(10); EXP(11); EXP(12); EXP(13); EXP(14); EXP(15); }
      for (int r = 0; r < 2; ++r) { CVTP(0); CVTP(1); CVTP(2); CVTP(3); CVTP(4); CVTP(5); CVTP(6); CVTP(7); }
      MFMA(acc2); MFMA(acc0); MFMA(acc1); MFMA(acc2);
    }
    if (KIND == 22) {  // KIND 19 with the real data dependences: exps read the FIRST MFMA's result, the other two MFMAs' B operand is the packs
#define EXPD(i) asm volatile("v_exp_f32 %0, %1" : "=v"(e[i]) : "v"(acc0[i]))
      union { bf16x8_t v; unsigned u[4]; } pb0, pb1;
      MFMA(acc0);
      EXPD(0); EXPD(1); EXPD(2); EXPD(3); EXPD(4); EXPD(5); EXPD(6); EXPD(7); CVTP(0); CVTP(1); CVTP(2); CVTP(3);
      pb0.u[0] = p[0]; pb0.u[1] = p[1]; pb0.u[2] = p[2]; pb0.u[3] = p[3];
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(pb0.v));
      EXPD(8); EXPD(9); EXPD(10); EXPD(11); EXPD(12); EXPD(13); EXPD(14); EXPD(15); CVTP(4); CVTP(5); CVTP(6); CVTP(7);
      pb1.u[0] = p[4]; pb1.u[1] = p[5]; pb1.u[2] = p[6]; pb1.u[3] = p[7];
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(pb1.v));
    }
    if (KIND == 25) {  // TWO tiles with their real dependences in the order the compiler emits the key loop: grouped
      union { bf16x8_t v; unsigned u[4]; } pb0, pb1;
      float e2[16];
      unsigned p2[8];
#define EXPD2(i) asm volatile("v_exp_f32 %0, %1" : "=v"(e2[i]) : "v"(acc2[i]))
#define CVTP2(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p2[i]) : "v"(e2[2 * (i)]), "v"(e2[2 * (i) + 1]))
      MFMA(acc0); MFMA(acc2);  // the two QK products
      EXPD(0); EXPD(1); EXPD(2); EXPD(3); EXPD(4); EXPD(5); EXPD(6); EXPD(7); EXPD(8); EXPD(9); EXPD(10); EXPD(11); EXPD(12); EXPD(13); EXPD(14); EXPD(15);
      EXPD2(0); EXPD2(1); EXPD2(2); EXPD2(3); EXPD2(4); EXPD2(5); EXPD2(6); EXPD2(7); EXPD2(8); EXPD2(9); EXPD2(10); EXPD2(11); EXPD2(12); EXPD2(13); EXPD2(14); EXPD2(15);
      CVTP(0); CVTP(1); CVTP(2); CVTP(3); CVTP(4); CVTP(5); CVTP(6); CVTP(7);
      CVTP2(0); CVTP2(1); CVTP2(2); CVTP2(3); CVTP2(4); CVTP2(5); CVTP2(6); CVTP2(7);
      pb0.u[0] = p[0]; pb0.u[1] = p[1]; pb0.u[2] = p[2]; pb0.u[3] = p[3];
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(pb0.v));
      pb1.u[0] = p[4]; pb1.u[1] = p[5]; pb1.u[2] = p[6]; pb1.u[3] = p[7];
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(pb1.v));
      pb0.u[0] = p2[0]; pb0.u[1] = p2[1]; pb0.u[2] = p2[2]; pb0.u[3] = p2[3];
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(pb0.v));
      pb1.u[0] = p2[4]; pb1.u[1] = p2[5]; pb1.u[2] = p2[6]; pb1.u[3] = p2[7];
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(pb1.v));
    }
    if (KIND == 23) {  // only the MFMA -> exp dependence
      MFMA(acc0);
      EXPD(0); EXPD(1); EXPD(2); EXPD(3); EXPD(4); EXPD(5); EXPD(6); EXPD(7); CVTP(0); CVTP(1); CVTP(2); CVTP(3);
      MFMA(acc1);
      EXPD(8); EXPD(9); EXPD(10); EXPD(11); EXPD(12); EXPD(13); EXPD(14); EXPD(15); CVTP(4); CVTP(5); CVTP(6); CVTP(7);
      MFMA(acc2);
    }
    if (KIND == 24) {  // only the pack -> MFMA dependence
      union { bf16x8_t v; unsigned u[4]; } pb0, pb1;
      MFMA(acc0);
      EXP(0); EXP(1); EXP(2); EXP(3); EXP(4); EXP(5); EXP(6); EXP(7); CVTP(0); CVTP(1); CVTP(2); CVTP(3);
      pb0.u[0] = p[0]; pb0.u[1] = p[1]; pb0.u[2] = p[2]; pb0.u[3] = p[3];
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(pb0.v));
      EXP(8); EXP(9); EXP(10); EXP(11); EXP(12); EXP(13); EXP(14); EXP(15); CVTP(4); CVTP(5); CVTP(6); CVTP(7);
      pb1.u[0] = p[4]; pb1.u[1] = p[5]; pb1.u[2] = p[6]; pb1.u[3] = p[7];
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(pb1.v));
    }
    if (KIND == 7) {  // 10 exps on the transcendental unit, 6 scores by a 7-op polynomial on the plain VALU
#define P7(i) FMA(i); FMA((i + 1) & 15); FMA((i + 2) & 15); FMA((i + 3) & 15); FMA((i + 4) & 15); FMA((i + 5) & 15); FMA((i + 6) & 15)
      MFMA(acc0); EXP(0); P7(0); EXP(1); P7(1); EXP(2); EXP(3); PERM(0); PERM(1);
      MFMA(acc1); EXP(4); P7(2); EXP(5); P7(3); EXP(6); PERM(2); PERM(3); PERM(4);
      MFMA(acc2); EXP(7); P7(4); EXP(8); P7(5); EXP(9); PERM(5); PERM(6); PERM(7);
    }
    if (KIND == 8) {
#define F8(i) FMA(i); FMA((i + 1) & 15); FMA((i + 2) & 15); FMA((i + 3) & 15); FMA((i + 4) & 15); FMA((i + 5) & 15); FMA((i + 6) & 15); FMA((i + 7) & 15)
      MFMA(acc0); F8(0); F8(8); F8(3);
      MFMA(acc1); F8(1); F8(9); F8(4);
      MFMA(acc2); F8(2); F8(10);
    }
    if (KIND == 9) {
      EXP(0); P7(0); EXP(1); P7(1); EXP(2); P7(2); EXP(3); P7(3); EXP(4); P7(4); EXP(5); P7(5); EXP(6); P7(6); EXP(7); P7(7);
    }
    if (KIND == 10) {
      MFMA(acc0); EXP(0); P7(0); EXP(1); P7(1); EXP(2); P7(2); PERM(0); PERM(1);
      MFMA(acc1); EXP(3); P7(3); EXP(4); P7(4); EXP(5); P7(5); PERM(2); PERM(3); PERM(4);
      MFMA(acc2); EXP(6); P7(6); EXP(7); P7(7); PERM(5); PERM(6); PERM(7);
    }
    if (KIND == 11) {
#define LDX(i) asm volatile("v_ldexp_f32 %0, %1, %2" : "=v"(e[i]) : "v"(x[i]), "v"(p[(i) & 7]))
      LDX(0); LDX(1); LDX(2); LDX(3); LDX(4); LDX(5); LDX(6); LDX(7); LDX(8); LDX(9); LDX(10); LDX(11); LDX(12); LDX(13); LDX(14); LDX(15);
    }
    if (KIND == 12) {
#define CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(e[i]) : "v"(x[i]), "v"(f[i]))
      CVT(0); CVT(1); CVT(2); CVT(3); CVT(4); CVT(5); CVT(6); CVT(7); CVT(8); CVT(9); CVT(10); CVT(11); CVT(12); CVT(13); CVT(14); CVT(15);
    }
    if (KIND == 13) {
#define FRA(i) asm volatile("v_fract_f32 %0, %1" : "=v"(e[i]) : "v"(x[i]))
      FRA(0); FRA(1); FRA(2); FRA(3); FRA(4); FRA(5); FRA(6); FRA(7); FRA(8); FRA(9); FRA(10); FRA(11); FRA(12); FRA(13); FRA(14); FRA(15);
    }
    if (KIND == 15) {
#define EXH(i) asm volatile("v_exp_f16 %0, %1" : "=v"(e[i]) : "v"(x[i]))
      EXH(0); EXH(1); EXH(2); EXH(3); EXH(4); EXH(5); EXH(6); EXH(7); EXH(8); EXH(9); EXH(10); EXH(11); EXH(12); EXH(13); EXH(14); EXH(15);
    }
    if (KIND == 16) {
#define EXL(i) asm volatile("v_exp_legacy_f32 %0, %1" : "=v"(e[i]) : "v"(x[i]))
      EXL(0); EXL(1); EXL(2); EXL(3); EXL(4); EXL(5); EXL(6); EXL(7); EXL(8); EXL(9); EXL(10); EXL(11); EXL(12); EXL(13); EXL(14); EXL(15);
    }
    if (KIND == 17) {
#define PERMX(i) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(e[i]) : "v"(x[i]), "v"(f[i]), "s"(0x07060302u))
      PERMX(0); PERMX(1); PERMX(2); PERMX(3); PERMX(4); PERMX(5); PERMX(6); PERMX(7); PERMX(8); PERMX(9); PERMX(10); PERMX(11); PERMX(12); PERMX(13); PERMX(14); PERMX(15);
    }
    if (KIND == 18) {  // bpermute (LDS crossbar, no LDS memory): cost next to VALU work
#define BPM(i) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(e[i]) : "v"(p[(i) & 7]), "v"(x[i]))
      BPM(0); BPM(1); BPM(2); BPM(3); BPM(4); BPM(5); BPM(6); BPM(7); BPM(8); BPM(9); BPM(10); BPM(11); BPM(12); BPM(13); BPM(14); BPM(15);
      asm volatile("s_waitcnt lgkmcnt(0)");
    }
    if (KIND == 14) {  // PV on 16x16x32 instead: 1 mfma32 (QK) + 4 mfma16 per tile
      f32x16_t& q = acc0;
      MFMA(q);
      typedef __attribute__((ext_vector_type(4))) float f32x4_t;
      f32x4_t r0 = {acc1[0], acc1[1], acc1[2], acc1[3]}, r1 = {acc1[4], acc1[5], acc1[6], acc1[7]};
      MFMA16(r0); MFMA16(r1); MFMA16(r0); MFMA16(r1);
      acc1[0] = r0[0]; acc1[4] = r1[0];
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += e[i] + f[i] + acc0[i] + acc1[i] + acc2[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (float)p[i];
  if (s == 12345.678f) out[1 << 20] = (unsigned long long)s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name) {
  unsigned long long* d;
  hipMalloc(&d, ((1 << 20) + 16) * sizeof(unsigned long long));
  const int iters = 2048;
  printf("%-46s", name);
  for (int W : {1, 2, 4}) {
    const int blocks = 256 * W;
    // warm-up with the full iteration count: a short first launch leaves the chip on its idle clock and the timed
    // launch then ramps up while it runs (round 2's ns columns of the first rows were taken at ~1.1 GHz)
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= h.size();
    // ticks = shader cycles (s_memtime); a SIMD runs its W waves concurrently, so cycles per iteration and SIMD
    // = ticks / W; the clock the chip held = cycles / wall time
    const double cyc = mean / iters / W, ns = ms * 1e6 / iters / W;
    printf("  W=%d: %6.1f cyc/iter/SIMD %6.2f ns (%4.2f GHz)", W, cyc, ns, cyc / ns);
  }
  printf("\n");
  hipFree(d);
}

int main() {
  printf("every SIMD of the chip runs W waves of the same loop; cyc/iter/SIMD = mean wave cycles per iteration / W\n"
         "(throughput cost of one iteration on one SIMD), ns = kernel wall time / iterations / W, GHz = their ratio\n");
  run<0>("16 exp");
  run<1>("16 fma");
  run<2>("16 exp + 16 fma");
  run<3>("16 exp + 48 fma");
  run<9>("8 exp + 56 fma");
  run<11>("16 ldexp");
  run<12>("16 cvt_pk_bf16");
  run<13>("16 fract");
  run<15>("16 exp_f16");
  run<16>("16 exp_legacy_f32");
  run<17>("16 perm");
  run<18>("16 ds_bpermute");
  run<4>("3 mfma32x32x16");
  run<14>("1 mfma32 + 4 mfma16x16x32");
  run<5>("3 mfma + 16 exp");
  run<6>("3 mfma + 16 exp + 8 perm (round-2 tile)");
  run<19>("3 mfma + 16 exp + 8 cvt_pk (round-3 tile)");
  run<20>("the same, MFMAs back to back, then exps, then packs");
  run<22>("round-3 tile with its REAL dependences (MFMA -> exp -> pack -> MFMA)");
  run<25>("  TWO tiles, real dependences, GROUPED like the compiled key loop (per two tiles)");
  run<23>("  only MFMA -> exp");
  run<24>("  only pack -> MFMA");
  run<21>("two tiles grouped like the real loop (per TWO tiles)");
  run<8>("3 mfma + 64 fma");
  run<7>("3 mfma + 10 exp + 42 fma + 8 perm (hybrid)");
  run<10>("3 mfma + 8 exp + 56 fma + 8 perm (hybrid)");
  return 0;
}
