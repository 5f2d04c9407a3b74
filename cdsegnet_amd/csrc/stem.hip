// Stem of the PTv3 backbones on gfx950: SubMConv3d(c_in -> C, k = 5, bias = False) + eval BatchNorm + GELU
// (ref: point_transformer_v3m1_base.py:633-663, Embedding; called once per branch, :1781-1784).
//
// As a gathered GEMM the stem needs the 5x5x5 kernel map: 125 int32 per point (432 MB for a batch of eight 120k-point
// scenes), built by 125 parent-cell lookups per point and read again by both branches - 2.7 ms per forward, 10 % of all
// kernel time, for a layer with 6 input channels.  This kernel never materialises that map:
//   * the <= 125 neighbours of a point are ENUMERATED through the next coarser level: the 27 cells around the point's
//     parent (the level-1 3x3x3 map, which stage 1 needs anyway) x their <= 8 children (contiguous in z-order, octants
//     known from an 8-bit occupancy mask per parent): ~9 occupied cells x ~2.5 children instead of 125 probes;
//   * a wave owns 16 points; the 4 lanes of a point split the 27 cells and drop every neighbour into a per-wave
//     [16 points][125 taps] BYTE table in LDS ((parent cell << 3) | rank of the child; the first child row of each of the
//     point's 27 cells sits next to it) - 3.9 KB per wave, so 16 waves fit next to the weight image;
//   * four taps form one MFMA k block (4 x 8 input channels = 32): D^T[32 ch][16 pts] += W_g^T[32 ch][32] X_g^T[32][16 pts],
//     v_mfma_f32_16x16x32: the B operand of lane (point, tap slot) is ONE 16-byte row load (zero if the table says "none"),
//     the A operand one conflict-free 16-byte LDS read of the fragment-ordered weight image; tap groups no point of the
//     tile uses are skipped (ballot);
//   * folded BN + GELU + both output copies (fp32 residual stream, 16-bit shadow) in the epilogue: a lane holds 4
//     consecutive channels of one point (16-byte / 8-byte stores).
// 16-bit operands, fp32 accumulation (the fp32 parity mode keeps the exact-fp32 gathered GEMM).
// History (8 scenes x 120 k points): 125-offset map + gathered GEMM 2.7 ms per forward (round 1) -> map-free walk +
// v_dot2 per (point, neighbour) pair 345 us per launch (round 2: 512 B of LDS weight reads per pair, half of them bank
// conflicts - the LDS bound it, profiles/r02_pmc_stem.txt) -> this form 175 us (round 3: 4 KB of conflict-free weight
// reads per point; 8 -> 16 waves per CU was worth 270 -> 184 us, masking the octants that cannot be in range 184 -> 175;
// of the 175: ~80 inputs + the two output copies, ~25 the walk's loops, ~70 the MFMA phase's four row-load round trips.
// Prefetching the next tile's walk inputs during the MFMA batches did not help: 197 us).
#include <cstdlib>

#include <atomic>

#include "common.h"

namespace {

constexpr int STEM_LPP = 4;  // lanes per point in the tree walk

struct StemP {
  const uint4* x;          // (n, 8) bf16 rows in physical order (16 bytes each)
  const float* scale;      // folded BatchNorm
  const float* shift;
  const int32_t* grid;     // (n, 3)
  const int32_t* cluster;  // (n): parent of every point
  const int32_t* pnbr;     // (27, m) offset-major 3x3x3 map of the parent level
  const int64_t* cinfo;    // (m): (first child << 8) | octant occupancy
  float* out;              // (n, 32) fp32
  bf16_t* out2;            // (n, 32) bf16 or nullptr
  long n, m;
  int depth;               // grid bits of the fine level
};

// per parent: first child row and which octants exist (children of a cell are contiguous and octant-sorted in z-order)
__global__ void child_info_kernel(const int64_t* __restrict__ zc, const int32_t* __restrict__ seg, long m,
                                  int64_t* __restrict__ info) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m) return;
  const int s = seg[p], e = seg[p + 1];
  int occ = 0;
  for (int j = s; j < e; ++j) occ |= 1 << (int)(zc[j] & 7);
  info[p] = ((int64_t)s << 8) | occ;
}

constexpr int SM_WAVES = 16;  // 1 block / CU, 4 waves / SIMD
constexpr int SM_BATCH = 8;   // tap groups whose rows are requested together (4: 192 us, 8: 184, 16: 240)
// per-wave tables: a BYTE per (point, tap) = (parent cell 0..26) << 3 | (rank of the child in its cell), 0xFF = no neighbour,
// and the first child row of each of the point's 27 parent cells.  (A table of int32 row ids - 8.4 KB per wave - allowed 11
// waves per CU next to the 64 KB weight image; a tile is a chain of dependent round trips, so waves are what the kernel
// needs: 8 / 11 / 16 waves = 270 / 218 / 184 us.)
constexpr int SM_TROW = 132;                      // bytes per table row: 128 tap slots + 4 of padding (bank spread)
constexpr int SM_FROW = 28;                       // ints per first-child row: 27 cells + 1
constexpr int SM_WAVE_LDS = 16 * SM_TROW + 16 * SM_FROW * 4;
constexpr int SM_IMG_BYTES = 32 * 2 * 64 * 16;    // [tap group][channel half][lane] x 16 B = 64 KB
constexpr int SM_LDS = SM_IMG_BYTES + SM_WAVES * SM_WAVE_LDS;

// fragment-order image: unit (g, nt, lane) = W[16 nt + (lane & 15)][tap 4 g + (lane >> 4)][0..7] (zeros past tap 124)
__global__ void stem_pack_w_mfma_kernel(const bf16_t* __restrict__ w /* (32, 125 * 8) */, uint4* __restrict__ img) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= 32 * 2 * 64) return;
  const int lane = u & 63, nt = (u >> 6) & 1, g = u >> 7;
  const int c = 16 * nt + (lane & 15), tap = 4 * g + (lane >> 4);
  uint4 v = make_uint4(0, 0, 0, 0);
  if (tap < 125) v = *reinterpret_cast<const uint4*>(w + (long)c * 1000 + tap * 8);
  img[u] = v;
}

__global__ __launch_bounds__(SM_WAVES * 64) void stem5_mfma_kernel(StemP p, const uint4* __restrict__ wimg2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* Wi = reinterpret_cast<uint4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int u = tid; u < SM_IMG_BYTES / 16; u += SM_WAVES * 64) Wi[u] = wimg2[u];
  __syncthreads();
  unsigned char* tab = reinterpret_cast<unsigned char*>(smem + SM_IMG_BYTES + wave * SM_WAVE_LDS);
  int* firsts = reinterpret_cast<int*>(tab + 16 * SM_TROW);
  const int pt = lane >> 2, q = lane & 3;    // tree walk: 4 lanes per point
  const int mi = lane & 15, kq = lane >> 4;  // MFMA: lane = (point, tap slot of the group) / (point, channel quad)
  const long tiles = (p.n + 15) / 16;
  const long tstep = (long)gridDim.x * SM_WAVES;
  constexpr int NCELL = (27 + STEM_LPP - 1) / STEM_LPP;
  for (long tile = (long)blockIdx.x * SM_WAVES + wave; tile < tiles; tile += tstep) {
    // ---- clear the table, then enumerate the tile's neighbours into it
    for (int u = lane; u < 16 * SM_TROW / 16; u += 64) reinterpret_cast<int4*>(tab)[u] = make_int4(-1, -1, -1, -1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      const long i = tile * 16 + pt;
      const bool valid = i < p.n;
      int gx = 0, gy = 0, gz = 0, par = 0;
      if (valid) {
        gx = p.grid[3 * i]; gy = p.grid[3 * i + 1]; gz = p.grid[3 * i + 2];
        par = p.cluster[i];
      }
      int pn[NCELL];
      int64_t info[NCELL];
#pragma unroll
      for (int c = 0; c < NCELL; ++c) {
        const int cell = q + STEM_LPP * c;
        pn[c] = (valid && cell < 27) ? p.pnbr[(long)cell * p.m + par] : -1;
      }
#pragma unroll
      for (int c = 0; c < NCELL; ++c) info[c] = pn[c] >= 0 ? p.cinfo[pn[c]] : 0;
      unsigned char* trow = tab + pt * SM_TROW;
      // octants of a parent cell at distance d whose child can be within +-2 of the point, per axis: only the far child of a
      // cell at d = -1 when the point is the odd child of its parent, only the near one at d = +1 when it is the even child
      const int px = gx & 1, py = gy & 1, pz = gz & 1;
#pragma unroll
      for (int c = 0; c < NCELL; ++c) {
        if (pn[c] < 0) continue;
        const int cell = q + STEM_LPP * c;
        const int dx = cell / 9 - 1, dy = (cell / 3) % 3 - 1, dz = cell % 3 - 1;
        firsts[pt * SM_FROW + cell] = (int)(info[c] >> 8);
        const int occ0 = (int)(info[c] & 255);
        const int mx = (dx < 0 && px) ? 0xF0 : ((dx > 0 && !px) ? 0x0F : 0xFF);
        const int my = (dy < 0 && py) ? 0xCC : ((dy > 0 && !py) ? 0x33 : 0xFF);
        const int mz = (dz < 0 && pz) ? 0xAA : ((dz > 0 && !pz) ? 0x55 : 0xFF);
        int occ = occ0 & mx & my & mz;
        const int bx = 2 * dx - px + 2, by = 2 * dy - py + 2, bz = 2 * dz - pz + 2;  // (neighbour - point + 2) of octant 0
        while (occ) {
          const int oct = __builtin_ctz(occ);
          occ &= occ - 1;
          const int rank = __builtin_popcount(occ0 & ((1 << oct) - 1));
          trow[(bx + (oct >> 2)) * 25 + (by + ((oct >> 1) & 1)) * 5 + (bz + (oct & 1))] = (unsigned char)((cell << 3) | rank);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- 32 tap groups of 4 taps; the rows of SM_BATCH groups are requested together.  A group's 4 table bytes are one
    // dword of the point's row (the same for the point's 4 lanes: broadcast), lane kq takes byte kq.
    f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    const unsigned* mrow = reinterpret_cast<const unsigned*>(tab + mi * SM_TROW);
    const int* frow = firsts + mi * SM_FROW;
#pragma unroll 1
    for (int g0 = 0; g0 < 32; g0 += SM_BATCH) {
      unsigned code[SM_BATCH];
      uint4 xr[SM_BATCH];
#pragma unroll
      for (int b = 0; b < SM_BATCH; ++b) code[b] = (mrow[g0 + b] >> (8 * kq)) & 255u;
#pragma unroll
      for (int b = 0; b < SM_BATCH; ++b) {
        const bool has = code[b] != 255u;
        const int row = has ? frow[has ? code[b] >> 3 : 0u] + (int)(code[b] & 7u) : 0;
        const uint4 v = p.x[row];
        xr[b] = has ? v : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int b = 0; b < SM_BATCH; ++b) {
        if (!__any(code[b] != 255u)) continue;  // no point of the tile has one of these four taps
        const bf16x8_t xb = __builtin_bit_cast(bf16x8_t, xr[b]);
        const uint4* wg = Wi + ((g0 + b) * 2) * 64 + lane;
        acc[0] = mfma_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wg[0]), xb, acc[0]);
        acc[1] = mfma_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wg[64]), xb, acc[1]);
      }
    }
    __builtin_amdgcn_wave_barrier();  // the table is rewritten by the next tile
    // ---- epilogue: lane = (point mi, channels 16 nt + 4 kq .. + 3)
    const long i = tile * 16 + mi;
    if (i < p.n) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int c0 = 16 * nt + 4 * kq;
        const float4 sc = *reinterpret_cast<const float4*>(p.scale + c0), sh = *reinterpret_cast<const float4*>(p.shift + c0);
        float4 v;
        v.x = gelu_erf(acc[nt][0] * sc.x + sh.x);
        v.y = gelu_erf(acc[nt][1] * sc.y + sh.y);
        v.z = gelu_erf(acc[nt][2] * sc.z + sh.z);
        v.w = gelu_erf(acc[nt][3] * sc.w + sh.w);
        *reinterpret_cast<float4*>(p.out + i * 32 + c0) = v;
        if (p.out2) {
          uint2 u;
          u.x = pack_bf16x2(v.x, v.y);
          u.y = pack_bf16x2(v.z, v.w);
          *reinterpret_cast<uint2*>(p.out2 + i * 32 + c0) = u;
        }
      }
    }
  }
}

}  // namespace

extern "C" int cdseg_child_info(const int64_t* zcode_sorted, const int32_t* seg_start, long m, int64_t* info, void* stream) {
  if (m <= 0) return CDSEG_OK;
  hipLaunchKernelGGL(child_info_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, zcode_sorted,
                     seg_start, m, info);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

extern "C" size_t cdseg_stem5_wimg_bytes(void) { return (size_t)SM_IMG_BYTES; }

extern "C" int cdseg_stem5_pack(const void* w, void* wimg, void* stream) {
  if (!w || !wimg) return CDSEG_ERR_ARG;
  hipLaunchKernelGGL(stem_pack_w_mfma_kernel, dim3(32 * 2 * 64 / 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w,
                     reinterpret_cast<uint4*>(wimg));
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

extern "C" int cdseg_stem5(const void* x8, const void* wimg, const float* scale, const float* shift, const int32_t* grid,
                           const int32_t* cluster, const int32_t* parent_nbr3, const int64_t* child_info, long n, long m,
                           int depth, float* out, void* out2, void* stream) {
  if (!x8 || !wimg || !scale || !shift || !grid || !cluster || !parent_nbr3 || !child_info || !out) return CDSEG_ERR_ARG;
  if (n <= 0) return CDSEG_OK;
  if (n >= (1l << 31) || m <= 0) return CDSEG_ERR_UNSUPPORTED;
  static std::atomic<bool> attr_done{false};  // (a concurrent first call sets the attribute twice: harmless)
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)stem5_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SM_LDS) != hipSuccess)
      return CDSEG_ERR_LAUNCH;
    attr_done = true;
  }
  StemP p;
  p.x = (const uint4*)x8; p.scale = scale; p.shift = shift; p.grid = grid;
  p.cluster = cluster; p.pnbr = parent_nbr3; p.cinfo = child_info; p.out = out; p.out2 = (bf16_t*)out2;
  p.n = n; p.m = m; p.depth = depth;
  const long tiles = (n + 15) / 16;
  long blocks = (tiles + SM_WAVES - 1) / SM_WAVES;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(stem5_mfma_kernel, dim3((unsigned)blocks), dim3(SM_WAVES * 64), SM_LDS, (hipStream_t)stream, p,
                     reinterpret_cast<const uint4*>(wimg));
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}
