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
__device__ __forceinline__ void pv(const f32x16_t& s, unsigned va, f32x16_t& o, const bf16x8_t& cv, f32x16_t& decoy,
                                   f32x16_t* o_mf1 = nullptr /* 512: the second PV MFMA of the tile accumulates here */) {
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
    if ((VAR & 512) && mf == 1) *o_mf1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf.v, *o_mf1, 0, 0, 0);
    else o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf.v, o, 0, 0, 0);
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

// THREADS x BPC: block size and blocks per CU (every block has its own K / V image in LDS); DYN: the block's waves claim
// their query tiles from an LDS counter (qtiles = tiles of the whole BLOCK) like the kernel does, instead of qtiles each
template <int VAR, int THREADS = 1024, int BPC = 1, bool DYN = false>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(THREADS / 256 * BPC, THREADS / 256 * BPC)))
void k(unsigned long long* out, float* sink, int qtiles, float seed) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // K, V: small random bf16 values; ones page
  __shared__ unsigned s_next;
  if (tid == 0) s_next = 0u;
  for (int i = tid; i < 2 * KV / 4; i += THREADS) {
    unsigned h = (i * 2654435761u) ^ (blockIdx.x * 40503u);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    const unsigned a = 0x3c00u + (h & 0x1ffu) | ((h >> 9) & 1u) << 15, b = 0x3c00u + ((h >> 10) & 0x1ffu) | ((h >> 19) & 1u) << 15;
    reinterpret_cast<unsigned*>(smem)[i] = a | (b << 16);
  }
  for (int w = tid; w < 2048 / 8; w += THREADS) *reinterpret_cast<uint2*>(smem + 2 * KV + w * 8) = make_uint2(0x3F80u, 0u);
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
  for (int qt = 0;; ++qt) {
    if (DYN) {
      unsigned i = 0;
      if (lane == 0) i = atomicAdd(&s_next, 1u);
      if ((int)__builtin_amdgcn_readfirstlane(i) >= qtiles) break;
    } else if (qt >= qtiles) {
      break;
    }
    f32x16_t acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc2 = acc;
    const char* kp = k_lane;
    unsigned va = va0;
    if (VAR & 1024) {
      // software-pipelined AND interleaved: the QK MFMA of tile t + 1 is issued first, then tile t's exps and packs in two
      // halves with one PV MFMA after each half - in program order no MFMA waits for a VALU result that is not already
      // there, and every MFMA has ~12 VALU instructions behind it to cover its 32 pipe cycles
      // (two tiles per trip with the score registers swapping roles: no register copies)
      auto half = [&](const f32x16_t& sc, int mf, unsigned vaddr) {
        float pr[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) pr[r] = __builtin_amdgcn_exp2f(sc[8 * mf + r]);
        union { bf16x8_t v; uint32_t u[4]; } pf;
#pragma unroll
        for (int j = 0; j < 4; ++j) pf.u[j] = pack_bf16x2(pr[2 * j], pr[2 * j + 1]);
        const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4_t*>(vaddr + 512 * mf));
        const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4_t*>(vaddr + 512 * mf + 256));
        const bf16x8_t vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf.v, acc, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // the two V^T reads
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);  // 8 exps + 4 packs
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // PV
      };
      f32x16_t sA = qk<VAR>(kp, qf, negm, cv), sB;
      for (int kt = 0; kt < 32; kt += 2) {
        sB = qk<VAR>(kp + 1024, qf, negm, cv);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        half(sA, 0, va);
        half(sA, 1, va);
        sA = qk<VAR>(kp + (kt < 30 ? 2048 : 0), qf, negm, cv);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        half(sB, 0, va + vstep);
        half(sB, 1, va + vstep);
        kp += 2048; va += 2 * vstep;
      }
    } else if (VAR & 32) {
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
      // 256: the two tiles in flight accumulate into different registers; 512: the two PV MFMAs of a tile do
      for (int kt = 0; kt < 32; kt += 2) {
        const f32x16_t sa = qk<VAR>(kp, qf, negm, cv), sb = qk<VAR>(kp + 1024, qf, negm, cv);
        pv<VAR>(sa, va, acc, cv, decoy, &acc2);
        pv<VAR>(sb, va + 1024, (VAR & 256) ? acc2 : acc, cv, decoy, &acc2);
        kp += 2048; va += 2 * vstep;
      }
    }
    total += acc[0] + acc[8] + acc2[0] + acc2[8];
    qf[0] = (short)(qf[0] ^ (qt & 1));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (total == 12345.678f) sink[0] = total + decoy[3];
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int VAR, int THREADS = 1024, int BPC = 1, bool DYN = false>
void run(const char* name) {
  constexpr int WPS = THREADS / 256 * BPC;  // waves per SIMD
  unsigned long long* d; float* sink;
  hipMalloc(&d, 256 * BPC * 16 * sizeof(unsigned long long)); hipMalloc(&sink, 64);
  hipMemset(d, 0, 256 * BPC * 16 * sizeof(unsigned long long));
  auto fn = k<VAR, THREADS, BPC, DYN>;
  hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  // static: 64 query tiles x 32 key tiles per wave; dynamic: the same total per block, claimed tile by tile
  const int per_wave = 64 * 4 / WPS;  // equal work per SIMD whatever the wave count
  const int qtiles = DYN ? per_wave * (THREADS / 64) : per_wave;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fn, dim3(256 * BPC), dim3(THREADS), SMEM, 0, d, sink, qtiles, 8.0f);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(fn, dim3(256 * BPC), dim3(THREADS), SMEM, 0, d, sink, qtiles, 8.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> hbuf(256 * BPC * 16);
  hipMemcpy(hbuf.data(), d, hbuf.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double mean = 0, mx = 0; int n = 0;
  for (auto v : hbuf) if (v) { mean += (double)v; mx = v > mx ? (double)v : mx; ++n; }
  mean /= n;
  const double tiles_simd = per_wave * 32.0 * WPS;  // tiles one SIMD processes in the launch
  printf("%-58s %d waves/SIMD: launch %6.2f ns per tile and SIMD = %6.1f cycles at the %4.2f GHz held; mean wave done at %3.0f %% of the launch\n", name,
         WPS, ms * 1e6 / tiles_simd, mx / tiles_simd, mx / (ms * 1e6), 100.0 * mean / mx);
  hipFree(d); hipFree(sink);
}

int main() {
  printf("key loop of attn_bf16_kernel on every SIMD of the chip, equal work per SIMD in every row\n");
  printf("-- more waves per SIMD (blocks per CU x block size), tiles claimed dynamically like in the kernel\n");
  run<0, 1024, 1, true>("2 tiles in flight, 16 waves x 1 block, dynamic");
  run<0, 512, 2, true>("2 tiles in flight, 8 waves x 2 blocks, dynamic");
  run<64, 512, 2, true>("1 tile in flight, 8 waves x 2 blocks, dynamic");
  run<64, 768, 2, true>("1 tile in flight, 12 waves x 2 blocks (80 VGPRs), dynamic");
  run<0, 768, 2, true>("2 tiles in flight, 12 waves x 2 blocks (80 VGPRs), dynamic");
  run<64, 1024, 2, true>("1 tile in flight, 16 waves x 2 blocks (64 VGPRs), dynamic");
  run<32, 512, 2, true>("3 tiles in flight, 8 waves x 2 blocks, dynamic");
  run<1024, 512, 2, true>("software-pipelined + interleaved (1 tile + prefetch), 8 x 2, dynamic");
  run<1024, 768, 2, true>("software-pipelined + interleaved, 12 waves x 2 blocks, dynamic");
  run<256, 512, 2, true>("2 tiles in flight, one PV accumulator per tile, 8 x 2, dynamic");
  run<512, 512, 2, true>("2 tiles in flight, one PV accumulator per MFMA of a tile, 8 x 2, dynamic");
  run<0, 512, 2, true>("2 tiles in flight, 8 waves x 2 blocks, dynamic (again)");
  printf("-- static equal work per wave (round 3's table: the launch ends with the slowest wave)\n");
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
