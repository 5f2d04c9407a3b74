// Micro-benchmark (gfx950): what does the LANE -> ADDRESS map of a gathered 16-byte-per-lane load cost?
// The wide-stage sparse conv (cdsegnet_amd/csrc/conv.hip) loads gathered rows straight into MFMA B fragments: lane
// (j = lane & 15, c = lane >> 4) of v_mfma_f32_16x16x32 holds channels 8c .. 8c+7 of point j, so the four lanes of a QUAD
// (consecutive lanes) read 16 bytes of four DIFFERENT rows.  The alternative reads with lane = 4 * row + chunk (a quad =
// 64 contiguous bytes of ONE row) and moves the data to the MFMA layout with ds_bpermute afterwards.  This program times the
// two maps on the same rows (C = 64: 128-byte rows, two instructions per 16-row set; every instruction = 16 rows x 64 bytes):
//   map M  lane -> (row lane & 15, chunk lane >> 4)      the MFMA operand layout, what conv.hip issues today
//   map Q  lane -> (row lane >> 2, chunk lane & 3)       quad-contiguous
//   rows   "run": the 16 rows of a set are consecutive (neighbours of consecutive output rows mostly are);
//          "scatter": 16 independent rows of the window
// One block per CU (grid 256), W = 8 waves, D instructions in flight per wave, consumed in issue order, repeat.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/gather_map tools/ubench/gather_map.hip ; run: tools/ubench/gather_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// Second kernel (added after the round's last GPU call, not yet run): the conv's own access - raw BUFFER loads with a
// per-lane offset, DEAD of 16 rows of a set absent (index -1 -> out-of-range offset -> zeros, no memory access), and a window
// mix: a set comes from a 1 MB hot window (L2-resident) with probability HOT / 16, else from the whole buffer.
template <int D, int MAP, int DEAD, int HOT>
__global__ __launch_bounds__(512) void kb(const char* src, unsigned long long rows, int iters, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned long long state = (blockIdx.x * 64ull + wave) * 0x9E3779B97F4A7C15ull + 12345ull;
  const int slot = MAP == 0 ? (lane & 15) : (lane >> 2);
  const int chunk = MAP == 0 ? (lane >> 4) : (lane & 3);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(rows * 128), 0x00020000);
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    unsigned off[D];
#pragma unroll
    for (int d = 0; d < D; d += 2) {
      state = state * 6364136223846793005ull + 1442695040888963407ull;
      const unsigned long long span = ((state >> 56) & 15) < (unsigned)HOT ? 8192ull : rows;  // 8192 rows = 1 MB
      unsigned long long r = (state >> 24) & (span - 1);
      r = (r > span - 16 ? r - 16 : r) + slot;
      const bool dead = (((state >> 12) + slot * 7) & 15) < (unsigned)DEAD;  // exactly DEAD of the 16 rows of a set
      off[d] = dead ? 0xFFFFFF80u + chunk * 16 : (unsigned)(r * 128 + chunk * 16);
      off[d + 1] = off[d] + (dead ? 0u : 64u);
      asm volatile("" : "+v"(off[d]), "+v"(off[d + 1]));
    }
    u4 v[D];
#pragma unroll
    for (int d = 0; d < D; ++d) v[d] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[d], 0, 0));
#pragma unroll
    for (int d = 0; d < D; ++d) acc ^= v[d][0] ^ v[d][1] ^ v[d][2] ^ v[d][3];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}


// Third kernel (round 6): the same window mix / dead rows as kb, with (a) map R - lane -> (row lane >> 3, chunk lane & 7): an
// instruction = 8 whole 128-byte rows, one cache line per 8 lanes - as plain global loads (DMA = 0) and (b) the same map as
// LDS-DMA (global_load_lds_dwordx4, DMA = 1: lane-linear LDS destination, D pieces in flight per wave, then vmcnt(0)); dead
// rows read a hot zero line.  Answers: is the wide conv's gather bound by tag look-ups per instruction (then R / DMA are 2 - 8x
// cheaper than M), or by latency / outstanding misses (then nothing moves)?
template <int D, int DMA, int DEAD, int HOT>
__global__ __launch_bounds__(1024) void kr(const char* src, unsigned long long rows, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned long long state = (blockIdx.x * 64ull + wave) * 0x9E3779B97F4A7C15ull + 12345ull;
  const int slot = lane >> 3, chunk = lane & 7;
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * (D * 1024);
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    const char* p[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      state = state * 6364136223846793005ull + 1442695040888963407ull;
      const unsigned long long span = ((state >> 56) & 15) < (unsigned)HOT ? 8192ull : rows;
      unsigned long long r = (state >> 24) & (span - 1);
      r = (r > span - 16 ? r - 16 : r) + slot + 8 * (d & 1);
      const bool dead = (((state >> 12) + (slot + 8 * (d & 1)) * 7) & 15) < (unsigned)DEAD;
      p[d] = dead ? src + chunk * 16 : src + r * 128 + chunk * 16;
      asm volatile("" : "+v"(p[d]));
    }
    if (DMA) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(p[d]), "s"(lds_base + d * 1024) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc ^= *reinterpret_cast<const unsigned*>(smem + wave * (D * 1024) + lane * 4);
    } else {
      u4 v[D];
#pragma unroll
      for (int d = 0; d < D; ++d) v[d] = *reinterpret_cast<const u4*>(p[d]);
#pragma unroll
      for (int d = 0; d < D; ++d) acc ^= v[d][0] ^ v[d][1] ^ v[d][2] ^ v[d][3];
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int D, int DMA, int DEAD, int HOT>
static void runr(const char* src, size_t span, unsigned* sink, int waves) {
  const int iters = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  hipFuncSetAttribute((const void*)kr<D, DMA, DEAD, HOT>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * D * 1024);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kr<D, DMA, DEAD, HOT>), dim3(256), dim3(waves * 64), DMA ? waves * D * 1024 : 0, 0, src,
                       (unsigned long long)(span / 128), iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double instr_per_cu = (double)waves * D * iters;
  printf("map R (8 rows x 128 B) %s, %2d / 16 sets hot, %2d / 16 rows dead, W=%d D=%2d: %6.1f ns per wave-instruction and CU (%5.1f cycles "
         "at 2.1 GHz)\n", DMA ? "LDS-DMA     " : "global loads", HOT, DEAD, waves, D, best * 1e6 / instr_per_cu,
         best * 1e6 / instr_per_cu * 2.1);
}

template <int D, int MAP, int SCATTER, int PERMUTE>
__global__ __launch_bounds__(512) void k(const char* src, unsigned long long rows /* 128-byte rows, a power of two */, int iters,
                                         unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned long long state = (blockIdx.x * 64ull + wave) * 0x9E3779B97F4A7C15ull + 12345ull;
  const int slot = MAP == 0 ? (lane & 15) : (lane >> 2);
  const int chunk = MAP == 0 ? (lane >> 4) : (lane & 3);
  const int paddr = (4 * (lane & 15) + (lane >> 4)) * 4;  // map Q -> MFMA layout: lane (j, c) takes from lane 4 j + c
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    const char* p[D];
#pragma unroll
    for (int d = 0; d < D; d += 2) {
      state = state * 6364136223846793005ull + 1442695040888963407ull;
      unsigned long long r = (state >> 24) & (rows - 1);
      if (SCATTER) {
        r = (r * 2654435761ull + (unsigned long long)slot * 0x9E3779B1ull * ((state >> 40) | 1ull)) & (rows - 1);
      } else {
        r = (r > rows - 16 ? r - 16 : r) + slot;
      }
      p[d] = src + r * 128 + chunk * 16;
      p[d + 1] = p[d] + 64;
      asm volatile("" : "+v"(p[d]), "+v"(p[d + 1]));
    }
    u4 v[D];
#pragma unroll
    for (int d = 0; d < D; ++d) v[d] = *reinterpret_cast<const u4*>(p[d]);  // D loads in flight, consumed in order
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (PERMUTE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[d][e] = __builtin_amdgcn_ds_bpermute(paddr, v[d][e]);
      }
      acc ^= v[d][0] ^ v[d][1] ^ v[d][2] ^ v[d][3];
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;  // (never: keeps the loads and the permutes alive)
}

template <int D, int MAP, int SCATTER, int PERMUTE>
static void run(const char* src, size_t span, const char* where, unsigned* sink) {
  const int iters = 400, waves = 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<D, MAP, SCATTER, PERMUTE>), dim3(256), dim3(waves * 64), 0, 0, src, (unsigned long long)(span / 128), iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double instr_per_cu = (double)waves * D * iters;
  const double bytes = 256.0 * instr_per_cu * 1024.0;
  printf("%-8s map %s%s rows %-7s D=%d: %6.2f TB/s chip, %6.1f ns per wave-instruction and CU (%5.1f cycles at 2.1 GHz)\n", where,
         MAP == 0 ? "M" : "Q", PERMUTE ? "+bpermute" : "         ", SCATTER ? "scatter" : "run", D, bytes / (best * 1e-3) / 1e12,
         best * 1e6 / instr_per_cu, best * 1e6 / instr_per_cu * 2.1);
}

template <int D, int MAP, int DEAD, int HOT>
static void runb(const char* src, size_t span, unsigned* sink) {
  const int iters = 400, waves = 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kb<D, MAP, DEAD, HOT>), dim3(256), dim3(waves * 64), 0, 0, src, (unsigned long long)(span / 128), iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double instr_per_cu = (double)waves * D * iters;
  printf("buffer loads, 64 MB buffer, %2d / 16 row sets from a 1 MB window, %2d / 16 rows dead, map %s D=%d: %6.1f ns per wave-instruction "
         "and CU (%5.1f cycles at 2.1 GHz)\n", HOT, DEAD, MAP == 0 ? "M" : "Q", D, best * 1e6 / instr_per_cu, best * 1e6 / instr_per_cu * 2.1);
}

int main() {
  printf("gathered 16-byte-per-lane loads, 16 rows x 64 bytes per instruction: MFMA-operand lane map (M) vs quad-contiguous (Q)\n");
  const size_t spans[3] = {2ull << 20, 16ull << 20, 64ull << 20};
  const char* names[3] = {"L2 2MB", "16MB", "64MB"};
  unsigned* sink;
  hipMalloc(&sink, 64);
  for (int s = 0; s < 3; ++s) {
    char* buf;
    if (hipMalloc(&buf, spans[s]) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, spans[s]);
    hipDeviceSynchronize();
    run<4, 0, 0, 0>(buf, spans[s], names[s], sink);
    run<4, 1, 0, 0>(buf, spans[s], names[s], sink);
    run<4, 1, 0, 1>(buf, spans[s], names[s], sink);
    run<12, 0, 0, 0>(buf, spans[s], names[s], sink);
    run<12, 1, 0, 0>(buf, spans[s], names[s], sink);
    run<12, 1, 0, 1>(buf, spans[s], names[s], sink);
    run<12, 0, 1, 0>(buf, spans[s], names[s], sink);
    run<12, 1, 1, 0>(buf, spans[s], names[s], sink);
    hipFree(buf);
  }
  {  // the conv's own access: buffer loads, dead lanes, a mostly-hot window
    char* buf;
    const size_t span = 64ull << 20;
    if (hipMalloc(&buf, span) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, span);
    hipDeviceSynchronize();
    runb<12, 0, 0, 16>(buf, span, sink); runb<12, 1, 0, 16>(buf, span, sink);
    runb<12, 0, 6, 16>(buf, span, sink); runb<12, 1, 6, 16>(buf, span, sink);
    runb<12, 0, 0, 14>(buf, span, sink); runb<12, 1, 0, 14>(buf, span, sink);
    runb<12, 0, 6, 14>(buf, span, sink); runb<12, 1, 6, 14>(buf, span, sink);
    runb<12, 0, 6, 0>(buf, span, sink);  runb<12, 1, 6, 0>(buf, span, sink);
    // depth: is the 89 %-hit case latency bound? (4 / 12 / 24 instructions in flight per wave)
    runb<4, 0, 6, 14>(buf, span, sink);  runb<24, 0, 6, 14>(buf, span, sink);
    runb<4, 1, 6, 14>(buf, span, sink);  runb<24, 1, 6, 14>(buf, span, sink);
    // whole-row map, plain and as LDS-DMA
    runr<12, 0, 0, 16>(buf, span, sink, 8); runr<12, 1, 0, 16>(buf, span, sink, 8);
    runr<12, 0, 6, 14>(buf, span, sink, 8); runr<12, 1, 6, 14>(buf, span, sink, 8);
    runr<4, 1, 6, 14>(buf, span, sink, 8);  runr<8, 1, 6, 14>(buf, span, sink, 8);
    runr<12, 1, 6, 14>(buf, span, sink, 4); runr<8, 1, 6, 14>(buf, span, sink, 16);
    runr<12, 0, 6, 0>(buf, span, sink, 8);  runr<12, 1, 6, 0>(buf, span, sink, 8);
    hipFree(buf);
  }
  return 0;
}
