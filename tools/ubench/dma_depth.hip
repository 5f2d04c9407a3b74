// Micro-benchmark (gfx950): how fast does LDS-DMA (global_load_lds_dwordx4) fill a CU's LDS as a function of the bytes
// the CU keeps IN FLIGHT?  Decides whether a deeper stage ring in the LDS-DMA GEMM (cdsegnet_amd/csrc/gemm.hip) pays:
// the 128 x 128 tile keeps 2 blocks x 32 KB in flight per CU and waits ~1.7 us per K step for it.
// One block per CU (grid = 256), W waves per block; every wave loops: issue D DMA instructions (1 KB each: 64 lanes x 16 B,
// per-lane row addresses like the GEMM's gathered A rows: 8 rows x 128 B per instruction), s_waitcnt vmcnt(0), repeat.
// In flight per CU = W x D KB.  The source buffer is `span` bytes (L2-resident: 2 MB ... HBM: 2 GB), rows picked by a
// per-wave LCG so that consecutive instructions do not walk one DRAM page.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/dma_depth tools/ubench/dma_depth.hip ; run: tools/ubench/dma_depth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

template <int D, bool TO_REGS = false>
__global__ __launch_bounds__(1024) void k(const char* src, unsigned long long rows /* 128-byte rows in the buffer */, int iters,
                                          unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * D * 1024;
  unsigned long long state = (blockIdx.x * 64ull + wave) * 0x9E3779B97F4A7C15ull + 12345ull;
  const int lrow = lane >> 3, slot = lane & 7;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const char* p[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      state = state * 6364136223846793005ull + 1442695040888963407ull;
      unsigned long long r = (state >> 24) & (rows - 1);  // (rows is a power of two) 8 neighbouring rows per instruction
      r = (r > rows - 8 ? r - 8 : r) + lrow;
      p[d] = src + r * 128 + slot * 16;
      asm volatile("" : "+v"(p[d]));
    }
    if (TO_REGS) {  // the same bytes through VGPRs (global_load_dwordx4), no LDS write: is the DMA path the limit, or the cache?
      typedef __attribute__((ext_vector_type(4))) unsigned u4;
      u4 v[D];
#pragma unroll
      for (int d = 0; d < D; ++d) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[d]) : "v"(p[d]) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int d = 0; d < D; ++d) asm volatile("" :: "v"(v[d]));
    } else {
#pragma unroll
      for (int d = 0; d < D; ++d) dma16(p[d], lds_base + d * 1024);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int D, bool TO_REGS = false>
static void run(const char* src, size_t span, int waves, const char* where) {
  const int iters = 400;
  unsigned long long* cyc;
  hipMalloc(&cyc, 256 * 16 * 8);
  hipMemset(cyc, 0, 256 * 16 * 8);
  const int lds = waves * D * 1024;
  hipFuncSetAttribute((const void*)k<D, TO_REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<D, TO_REGS>), dim3(256), dim3(waves * 64), lds, 0, src, (unsigned long long)(span / 128), iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * 16);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double c = 0; int n = 0;
  for (auto v : h) if (v) { c += (double)v; ++n; }
  const double bytes = 256.0 * waves * D * 1024.0 * iters;
  printf("%-8s %s W=%2d D=%2d: %3d KB in flight per CU, %6.2f TB/s chip = %6.1f GB/s per CU, round trip %5.2f us (%6.0f cycles)\n", where,
         TO_REGS ? "to VGPRs" : "LDS-DMA ", waves, D, waves * D, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 256 / 1e9, ms * 1e3 / iters, c / n / iters);
  hipFree(cyc);
}

int main() {
  printf("LDS-DMA fill rate vs bytes in flight per CU (one block per CU, W waves x D instructions of 1 KB, wait for all, repeat)\n");
  const size_t spans[4] = {2ull << 20, 16ull << 20, 128ull << 20, 4096ull << 20};
  const char* names[4] = {"L2 2MB", "16MB", "MALL128M", "HBM 4GB"};
  for (int s = 0; s < 4; ++s) {
    char* buf;
    if (hipMalloc(&buf, spans[s]) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, spans[s]);
    hipDeviceSynchronize();
    run<2>(buf, spans[s], 8, names[s]);
    run<4>(buf, spans[s], 8, names[s]);
    run<8>(buf, spans[s], 8, names[s]);
    run<8>(buf, spans[s], 16, names[s]);
    run<2, true>(buf, spans[s], 8, names[s]);
    run<4, true>(buf, spans[s], 8, names[s]);
    run<8, true>(buf, spans[s], 8, names[s]);
    run<8, true>(buf, spans[s], 16, names[s]);
    hipFree(buf);
  }
  return 0;
}
