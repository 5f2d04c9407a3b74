// Micro-benchmark (gfx950): the key loop of the bf16 attention kernel in isolation, 4 waves per SIMD (256 workgroups
// of 16 waves, K / V of one patch-head in LDS), in variants that each remove or re-arrange ONE thing - what keeps the
// real loop at ~216 cycles per 32x32 tile and SIMD when its instruction mix alone runs at ~130 (pipes.hip)?
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/attn_loop tools/ubench/attn_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
typedef __bf16 hw_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float hw_f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const hw_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2_t));
}

constexpr int KV = 1024 * 32;
constexpr int SMEM = 2 * KV + 2048 + 64;

// VAR bits: 1 = no LDS reads in the loop (constant fragments)      2 = exps read registers the MFMAs do not write
//           4 = PV operands do not come from the packs              8 = s_setprio(1) around the MFMAs
//          16 = C operand of QK is the inline constant 0            32 = three key tiles in flight
//          64 = one key tile in flight
template <int VAR>
__device__ __forceinline__ void pv(const f32x16_t& s, unsigned va, f32x16_t& o, const bf16x8_t& cv, f32x16_t& decoy) {
  float pr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (VAR & 2) asm volatile("" : "+v"(decoy[r]));  // opaque: a fresh value as far as the compiler knows, never written
    pr[r] = __builtin_amdgcn_exp2f((VAR & 2) ? decoy[r] : s[r]);
  }
  if (VAR & 2) asm volatile("" ::"v"(s[0]), "v"(s[15]));
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    union { bf16x8_t v; uint32_t u[4]; } pf;
#pragma unroll
    for (int j = 0; j < 4; ++j) pf.u[j] = pack_bf16x2(pr[8 * mf + 2 * j], pr[8 * mf + 2 * j + 1]);
    bf16x8_t vf;
    if (VAR & 1) {
      vf = cv;
      asm volatile("" : "+v"(vf));
    } else {
      const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4_t*>(va + 512 * mf));
      const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4_t*>(va + 512 * mf + 256));
      vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    if (VAR & 4) {
      asm volatile("" ::"v"(pf.u[0]), "v"(pf.u[1]), "v"(pf.u[2]), "v"(pf.u[3]));
      pf.v = cv;
      asm volatile("" : "+v"(pf.v));
    }
    if (VAR & 8) __builtin_amdgcn_s_setprio(1);
    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf.v, o, 0, 0, 0);
    if (VAR & 8) __builtin_amdgcn_s_setprio(0);
  }
}

template <int VAR>
__device__ __forceinline__ f32x16_t qk(const char* kp, bf16x8_t qf, const f32x16_t& c0, const bf16x8_t& cv) {
  bf16x8_t kf = (VAR & 1) ? cv : *reinterpret_cast<const bf16x8_t*>(kp);
  if (VAR & 1) asm volatile("" : "+v"(kf));
  const f32x16_t z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (VAR & 8) __builtin_amdgcn_s_setprio(1);
  const f32x16_t r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf, (VAR & 16) ? z : c0, 0, 0, 0);
  if (VAR & 8) __builtin_amdgcn_s_setprio(0);
  return r;
}

template <int VAR>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float* sink, int qtiles, float seed) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // K, V: small random bf16 values; ones page
  for (int i = tid; i < 2 * KV / 4; i += 1024) {
    unsigned h = (i * 2654435761u) ^ (blockIdx.x * 40503u);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    const unsigned a = 0x3c00u + (h & 0x1ffu) | ((h >> 9) & 1u) << 15, b = 0x3c00u + ((h >> 10) & 0x1ffu) | ((h >> 19) & 1u) << 15;
    reinterpret_cast<unsigned*>(smem)[i] = a | (b << 16);
  }
  for (int w = tid; w < 2048 / 8; w += 1024) *reinterpret_cast<uint2*>(smem + 2 * KV + w * 8) = make_uint2(0x3F80u, 0u);
  __syncthreads();
  const int ql = lane & 31, h = lane >> 5;
  const char* k_lane = smem + ql * 32 + ((h ^ ((ql >> 3) & 1)) << 4);
  const bool v_lane = (lane & 16) == 0;
  const unsigned va0 = v_lane ? lds_base + KV + (4 * h + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8 : lds_base + 2 * KV + (h ? 0 : 128);
  const unsigned vstep = v_lane ? 1024u : 0u;
  bf16x8_t qf, cv;
#pragma unroll
  for (int i = 0; i < 8; ++i) { qf[i] = (short)(0x3c00 + ((lane * 7 + i * 13) & 0xff)); cv[i] = (short)(0x3c80 + ((lane + i) & 0x7f)); }
  const float nm = -seed;
  const f32x16_t negm = {nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm};
  f32x16_t decoy;
#pragma unroll
  for (int i = 0; i < 16; ++i) decoy[i] = -1.f - 0.01f * (float)((lane + i) & 15);
  float total = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int qt = 0; qt < qtiles; ++qt) {
    f32x16_t acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const char* kp = k_lane;
    unsigned va = va0;
    if (VAR & 32) {
      int kt = 0;
      for (; kt + 2 < 32; kt += 3) {
        const f32x16_t sa = qk<VAR>(kp, qf, negm, cv), sb = qk<VAR>(kp + 1024, qf, negm, cv), sc = qk<VAR>(kp + 2048, qf, negm, cv);
        pv<VAR>(sa, va, acc, cv, decoy);
        pv<VAR>(sb, va + 1024, acc, cv, decoy);
        pv<VAR>(sc, va + 2048, acc, cv, decoy);
        kp += 3072; va += 3 * vstep;
      }
      for (; kt < 32; ++kt) { const f32x16_t s = qk<VAR>(kp, qf, negm, cv); pv<VAR>(s, va, acc, cv, decoy); kp += 1024; va += vstep; }
    } else if (VAR & 128) {
      for (int kt = 0; kt < 32; kt += 4) {
        const f32x16_t sa = qk<VAR>(kp, qf, negm, cv), sb = qk<VAR>(kp + 1024, qf, negm, cv);
        const f32x16_t sc = qk<VAR>(kp + 2048, qf, negm, cv), sd = qk<VAR>(kp + 3072, qf, negm, cv);
        pv<VAR>(sa, va, acc, cv, decoy);
        pv<VAR>(sb, va + 1024, acc, cv, decoy);
        pv<VAR>(sc, va + 2048, acc, cv, decoy);
        pv<VAR>(sd, va + 3072, acc, cv, decoy);
        kp += 4096; va += 4 * vstep;
      }
    } else if (VAR & 64) {
      for (int kt = 0; kt < 32; ++kt) { const f32x16_t s = qk<VAR>(kp, qf, negm, cv); pv<VAR>(s, va, acc, cv, decoy); kp += 1024; va += vstep; }
    } else {
      for (int kt = 0; kt < 32; kt += 2) {
        const f32x16_t sa = qk<VAR>(kp, qf, negm, cv), sb = qk<VAR>(kp + 1024, qf, negm, cv);
        pv<VAR>(sa, va, acc, cv, decoy);
        pv<VAR>(sb, va + 1024, acc, cv, decoy);
        kp += 2048; va += 2 * vstep;
      }
    }
    total += acc[0] + acc[8];
    qf[0] = (short)(qf[0] ^ (qt & 1));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (total == 12345.678f) sink[0] = total + decoy[3];
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int VAR>
void run(const char* name) {
  unsigned long long* d; float* sink;
  hipMalloc(&d, 256 * 16 * sizeof(unsigned long long)); hipMalloc(&sink, 64);
  hipFuncSetAttribute((const void*)k<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  const int qtiles = 64;  // 64 query tiles x 32 key tiles per wave (~0.5 ms: the clock settles)
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<VAR>, dim3(256), dim3(1024), SMEM, 0, d, sink, qtiles, 8.0f);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<VAR>, dim3(256), dim3(1024), SMEM, 0, d, sink, qtiles, 8.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> hbuf(256 * 16);
  hipMemcpy(hbuf.data(), d, hbuf.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double mean = 0, mx = 0; for (auto v : hbuf) { mean += (double)v; mx = v > mx ? (double)v : mx; }
  mean /= hbuf.size();
  const double tiles = qtiles * 32.0;
  printf("%-58s mean wave %6.1f, slowest wave %6.1f cyc/tile/SIMD (the launch runs at the slowest), %6.2f ns, %4.2f GHz\n", name,
         mean / tiles / 4, mx / tiles / 4, ms * 1e6 / tiles / 4, mx / (ms * 1e6));
  hipFree(d); hipFree(sink);
}

int main() {
  printf("key loop of attn_bf16_*_kernel, 4 waves per SIMD on every SIMD; cyc/tile/SIMD = mean wave cycles per 32x32 tile / 4\n");
  run<0>("real loop (2 tiles in flight)");
  run<1>("  no LDS reads");
  run<2>("  exps decoupled from the QK MFMA");
  run<4>("  PV operands decoupled from the packs");
  run<6>("  both decoupled");
  run<7>("  both decoupled, no LDS reads");
  run<8>("  s_setprio(1) around MFMAs");
  run<16>("  QK with C = 0 (no shift operand)");
  run<32>("  three tiles in flight");
  run<48>("  three tiles in flight, C = 0");
  run<144>("  four tiles in flight, C = 0");
  run<40>("  three tiles in flight, s_setprio");
  run<33>("  three tiles in flight, no LDS reads");
  run<64>("  one tile in flight");
  return 0;
}
