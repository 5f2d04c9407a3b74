// Stable LSD radix sort of (uint64 key, int32 value) pairs for the serialization step (ref: torch.argsort of the curve
// codes, pointcept/models/utils/structure.py:83, ptv3.py:493) - hand-written for gfx950.
//
// The codes have 30 - 45 significant bits and there are 0.1 - 2.6 M of them: far too few for a sort to be bandwidth
// bound (864 k pairs are 10 MB, ~2.5 us of HBM time per pass), so what a pass costs is its LATENCY structure.  rocPRIM's
// Onesweep resolves the tile offsets of a pass with a decoupled look-back chain through all tiles (38 us per 8-bit pass
// at 864 k keys: ~150 ns per link), its merge-sort path needs 13 dependent passes.  Here a pass is three fully parallel
// launches with no inter-workgroup dependency inside a launch:
//   1. count:   every workgroup histograms the current digit of its 2048-key tile in LDS and writes its 256 counts to a
//               BIN-MAJOR table count[bin][tile] (+ the 256 global bin totals by atomics);
//   2. offsets: one wave per bin turns its table row into exclusive offsets (row prefix + the sum of the lower bins' totals);
//   3. scatter: every workgroup ranks its tile's keys stably per digit - per wave with match masks built from 8 ballots
//               (a key's rank among equal digits = popcount of the matching lanes below it + the wave's running count of
//               that digit in LDS), waves combined by a 4-entry prefix per bin - and writes each pair to
//               offsets[bin][tile] + rank.
// Values: pass 0 takes the key's position when no value array is given (no iota pass).  The passes ping-pong between the
// caller's output and a temporary so that the LAST pass lands in the output.
//
// EXPERIMENT RECORD (round 4, not part of the library): wired behind cdseg_sort_pairs / cdseg_sort_curves it was bit-exact
// (all serialization tests, torch.sort(stable=True) on random keys) but NOT faster than rocPRIM's Onesweep at the plan's
// sizes - 864 k pairs, 34 key bits: 173 us (rocPRIM ~150 - 190 us); 3 x 864 k curve keys in one sort: 444 us (rocPRIM
// ~200 us); per pass (rocprofv3, profiles/r04_sort.txt): scatter 10 / 57 us, count 8.5 / 29.5 us (LDS-atomic conflicts on 256
// bins), offsets 6 / 14 us for 0.86 M / 2.6 M keys - every kernel is a short chain of dependent steps (load, 8 x [9 ballots
// + LDS read-modify-write + shuffle], barrier, table read, barrier, scattered 12-byte writes in runs of ~8 keys), 15
// dependent launches per sort.  What a faster version needs (not built): the next pass's counts taken inside this
// pass's scatter (2 launches per pass), tile-local reordering through LDS so that runs leave coalesced, 11-bit digits.
#include "../../cdsegnet_amd/csrc/common.h"
#include "radix_sort.h"

namespace {

constexpr int RB = 8, BINS = 1 << RB;          // digit width
constexpr int RT = 256, IPT = 8, TILE = RT * IPT;  // threads, keys per thread, keys per workgroup
constexpr int RWAVES = RT / 64;

struct RadixP {
  const uint64_t* kin;
  const int32_t* vin;  // nullptr: value = position
  uint64_t* kout;
  int32_t* vout;
  unsigned* table;   // [BINS][ntiles] counts, then offsets
  unsigned* totals;  // [BINS]
  long n;
  int ntiles, shift, mask;
};

__global__ __launch_bounds__(RT) void radix_count_kernel(RadixP p) {
  __shared__ unsigned cnt[BINS];
  const int tid = threadIdx.x;
  for (int b = tid; b < BINS; b += RT) cnt[b] = 0u;
  __syncthreads();
  const long base = (long)blockIdx.x * TILE;
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const long i = base + j * RT + tid;
    if (i < p.n) atomicAdd(&cnt[(unsigned)(p.kin[i] >> p.shift) & p.mask], 1u);
  }
  __syncthreads();
  for (int b = tid; b < BINS; b += RT) {
    const unsigned c = cnt[b];
    p.table[(size_t)b * p.ntiles + blockIdx.x] = c;
    if (c) atomicAdd(&p.totals[b], c);
  }
}

// one wave per bin: exclusive prefix over the bin's row of tile counts, started at the sum of the lower bins' totals
__global__ __launch_bounds__(RT) void radix_offsets_kernel(RadixP p) {
  const int lane = threadIdx.x & 63;
  const int bin = blockIdx.x * RWAVES + (threadIdx.x >> 6);
  if (bin >= BINS) return;
  unsigned below = 0u;
  for (int b = lane; b < bin; b += 64) below += p.totals[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o, 64);
  unsigned* row = p.table + (size_t)bin * p.ntiles;
  unsigned run = below;
  for (int t0 = 0; t0 < p.ntiles; t0 += 64) {
    const int t = t0 + lane;
    const unsigned c = t < p.ntiles ? row[t] : 0u;
    unsigned inc = c;  // inclusive scan across the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned v = __shfl_up(inc, o, 64);
      if (lane >= o) inc += v;
    }
    if (t < p.ntiles) row[t] = run + inc - c;
    run += __shfl(inc, 63, 64);
  }
}

__global__ __launch_bounds__(RT) void radix_scatter_kernel(RadixP p) {
  __shared__ unsigned wcnt[RWAVES][BINS];  // per wave: keys of each digit seen so far, then the wave's start inside the tile's bin
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int b = tid; b < RWAVES * BINS; b += RT) (&wcnt[0][0])[b] = 0u;
  __syncthreads();
  // wave w owns the tile positions [w * 512, w * 512 + 512): item j = positions j * 64 + lane of that range (tile order =
  // wave, item, lane: the ranks below respect it, so the sort is stable)
  const long base = (long)blockIdx.x * TILE + wave * (IPT * 64);
  uint64_t key[IPT];
  int32_t val[IPT];
  unsigned rank[IPT];
  int dig[IPT];
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const long i = base + j * 64 + lane;
    const bool in = i < p.n;
    key[j] = in ? p.kin[i] : ~0ull;
    val[j] = in ? (p.vin ? p.vin[i] : (int32_t)i) : 0;
    dig[j] = in ? (int)((unsigned)(key[j] >> p.shift) & p.mask) : -1;
  }
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const int d = dig[j];
    unsigned long long m = __ballot(d >= 0);  // lanes with the same digit (keys past the end match nobody)
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const unsigned long long bal = __ballot((d >> b) & 1);
      m &= ((d >> b) & 1) ? bal : ~bal;
    }
    unsigned prev = 0u;
    if (d >= 0) {
      const int leader = __ffsll((long long)m) - 1;
      if (lane == leader) {
        prev = wcnt[wave][d];
        wcnt[wave][d] = prev + (unsigned)__popcll(m);
      }
      prev = __shfl(prev, leader, 64);
    }
    rank[j] = prev + (unsigned)__popcll(m & lt);
  }
  __syncthreads();
  // wave starts inside each bin of the tile (exclusive prefix over the 4 waves), plus the tile's global offset of the bin
  for (int b = tid; b < BINS; b += RT) {
    unsigned run = p.table[(size_t)b * p.ntiles + blockIdx.x];
#pragma unroll
    for (int w = 0; w < RWAVES; ++w) {
      const unsigned c = wcnt[w][b];
      wcnt[w][b] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    if (dig[j] >= 0) {
      const size_t pos = (size_t)wcnt[wave][dig[j]] + rank[j];
      p.kout[pos] = key[j];
      p.vout[pos] = val[j];
    }
  }
}

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

size_t radix_sort_ws_bytes(size_t n) {
  const size_t ntiles = (n + TILE - 1) / TILE;
  return al256(n * 8) + al256(n * 4) + al256((size_t)BINS * (ntiles + 1) * 4) + al256(8 * BINS * 4) + 256;
}

int radix_sort_pairs(const uint64_t* kin, uint64_t* kout, const int32_t* vin, int32_t* vout, size_t n, int end_bit, void* ws,
                     size_t ws_bytes, hipStream_t s) {
  if (n == 0) return CDSEG_OK;
  if (end_bit <= 0 || end_bit > 64) end_bit = 64;
  if (ws_bytes < radix_sort_ws_bytes(n) || n >= (1ull << 31)) return CDSEG_ERR_WORKSPACE;
  const int ntiles = (int)((n + TILE - 1) / TILE);
  char* w = (char*)ws;
  uint64_t* ktmp = (uint64_t*)w;
  int32_t* vtmp = (int32_t*)(w + al256(n * 8));
  unsigned* table = (unsigned*)(w + al256(n * 8) + al256(n * 4));
  unsigned* totals = (unsigned*)((char*)table + al256((size_t)BINS * (ntiles + 1) * 4));
  const int passes = (end_bit + RB - 1) / RB;
  RadixP p;
  p.n = (long)n; p.ntiles = ntiles; p.table = table;
  if (hipMemsetAsync(totals, 0, (size_t)passes * BINS * sizeof(unsigned), s) != hipSuccess) return CDSEG_ERR_LAUNCH;  // one row per pass
  const uint64_t* src_k = kin;
  const int32_t* src_v = vin;
  for (int q = 0; q < passes; ++q) {
    const bool to_out = ((passes - 1 - q) & 1) == 0;
    p.kin = src_k; p.vin = src_v;
    p.kout = to_out ? kout : ktmp;
    p.vout = to_out ? vout : vtmp;
    p.shift = q * RB;
    const int bits = end_bit - q * RB < RB ? end_bit - q * RB : RB;
    p.mask = (1 << bits) - 1;
    p.totals = totals + (size_t)q * BINS;
    hipLaunchKernelGGL(radix_count_kernel, dim3(ntiles), dim3(RT), 0, s, p);
    hipLaunchKernelGGL(radix_offsets_kernel, dim3(BINS / RWAVES), dim3(RT), 0, s, p);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(ntiles), dim3(RT), 0, s, p);
    src_k = p.kout;
    src_v = p.vout;
  }
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}
