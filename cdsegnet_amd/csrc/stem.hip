// Stem of the PTv3 backbones on gfx950: SubMConv3d(c_in -> C, k = 5, bias = False) + eval BatchNorm + GELU
// (ref: point_transformer_v3m1_base.py:633-663, Embedding; called once per branch, :1781-1784).
//
// As a gathered GEMM the stem needs the 5x5x5 kernel map: 125 int32 per point (432 MB for a batch of eight 120k-point
// scenes), built by 125 parent-cell lookups per point and read again by both branches - 2.7 ms per forward, 10 % of all
// kernel time, for a layer with 6 input channels.  This kernel never materialises that map:
//   * the <= 125 neighbours of a point are ENUMERATED through the next coarser level: the 27 cells around the point's
//     parent (the level-1 3x3x3 map, which stage 1 needs anyway) x their <= 8 children (contiguous in z-order, octants
//     known from an 8-bit occupancy mask per parent): ~9 occupied cells x ~2.5 children instead of 125 probes;
//   * the 4 lanes that own a point (8 of the 32 output channels each) split the 27 cells, drop their candidates
//     (offset id, row) into per-lane lists in LDS, then every lane walks the point's lists in a fixed order
//     (deterministic fp32 sums): one 16-byte row load per neighbour (8 bf16 channels), the 8 x 8 weights of that offset
//     from the LDS-resident weight image, 32 v_dot2_f32_bf16;
//   * folded BN + GELU + both output copies (fp32 residual stream, bf16 shadow) in the epilogue.
// bf16 operands, fp32 accumulation: the numerics class of the MFMA path it replaces (the fp32 parity mode keeps the
// exact-fp32 gathered GEMM).
#include <cstdlib>

#include "common.h"

namespace {

constexpr int STEM_WAVES = 12;          // 1 block / CU (LDS), 3 waves / SIMD
constexpr int STEM_LPP = 4;             // lanes per point (C = 32: 8 output channels each)
constexpr int STEM_PPW = 64 / STEM_LPP;  // points per wave
constexpr int STEM_RB = 3;              // neighbour rows requested together per list and round (2: 351 us, 3: 345 us at 8 scenes)
constexpr int STEM_CAP = 24;            // list entries per lane and pass (more: another pass, never seen on scans)
constexpr int STEM_WROW = 136;          // u32 per offset row of the weight image: 4 pairs x 32 channels + 8 pad (banks)

struct StemP {
  const uint4* x;          // (n, 8) bf16 rows in physical order (16 bytes each)
  const uint32_t* wimg;    // (125, STEM_WROW) u32: [offset][pair kp][channel] = bf16 pair (W[c][o][2kp], W[c][o][2kp+1])
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

__global__ void stem_pack_w_kernel(const bf16_t* __restrict__ w /* (32, 125 * 8) */, uint32_t* __restrict__ img) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 125 * STEM_WROW) return;
  const int o = t / STEM_WROW, r = t % STEM_WROW;
  uint32_t v = 0;
  if (r < 128) {
    const int kp = r >> 5, c = r & 31;
    const bf16_t lo = w[(long)c * 1000 + o * 8 + 2 * kp], hi = w[(long)c * 1000 + o * 8 + 2 * kp + 1];
    v = (uint32_t)lo | ((uint32_t)hi << 16);
  }
  img[t] = v;
}

template <int RB>
__global__ __launch_bounds__(STEM_WAVES * 64) void stem5_kernel(StemP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* Ws = reinterpret_cast<uint32_t*>(smem);                              // 125 x STEM_WROW
  uint32_t* lists = Ws + 125 * STEM_WROW;                                        // [wave][point][lane q][STEM_CAP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pt = lane >> 2, q = lane & 3;
  for (int u = tid; u < 125 * STEM_WROW / 4; u += STEM_WAVES * 64)
    reinterpret_cast<uint4*>(Ws)[u] = reinterpret_cast<const uint4*>(p.wimg)[u];
  __syncthreads();
  uint32_t* mylist = lists + ((wave * STEM_PPW + pt) * STEM_LPP + q) * STEM_CAP;
  const uint32_t* ptlists = lists + (wave * STEM_PPW + pt) * STEM_LPP * STEM_CAP;
  const int lim = 1 << p.depth;

  const long tiles = (p.n + STEM_PPW - 1) / STEM_PPW;
  for (long tile = (long)blockIdx.x * STEM_WAVES + wave; tile < tiles; tile += (long)gridDim.x * STEM_WAVES) {
    const long i = tile * STEM_PPW + pt;
    const bool valid = i < p.n;
    int gx = 0, gy = 0, gz = 0, par = 0;
    if (valid) {
      gx = p.grid[3 * i]; gy = p.grid[3 * i + 1]; gz = p.grid[3 * i + 2];
      par = p.cluster[i];
    }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int pass = 0;; ++pass) {
      // ---- enumerate: lane q takes the parent cells q, q + 4, ...; candidates [pass * CAP, (pass + 1) * CAP) are listed
      int cnt = 0;
      {
        // the lane's 7 parent cells: all map entries first, then all child_info words (two memory round trips instead
        // of two per cell), then the enumeration from registers
        constexpr int NCELL = (27 + STEM_LPP - 1) / STEM_LPP;
        int pn[NCELL];
        int64_t info[NCELL];
#pragma unroll
        for (int c = 0; c < NCELL; ++c) {
          const int cell = q + STEM_LPP * c;
          pn[c] = (valid && cell < 27) ? p.pnbr[(long)cell * p.m + par] : -1;
        }
#pragma unroll
        for (int c = 0; c < NCELL; ++c) info[c] = pn[c] >= 0 ? p.cinfo[pn[c]] : 0;
#pragma unroll
        for (int c = 0; c < NCELL; ++c) {
          if (pn[c] < 0) continue;
          const int cell = q + STEM_LPP * c;
          const int dx = cell / 9 - 1, dy = (cell / 3) % 3 - 1, dz = cell % 3 - 1;
          const int first = (int)(info[c] >> 8);
          int occ = (int)(info[c] & 255), rank = 0;
          const int bx = (((gx >> 1) + dx) << 1) - gx, by = (((gy >> 1) + dy) << 1) - gy, bz = (((gz >> 1) + dz) << 1) - gz;
          while (occ) {
            const int oct = __builtin_ctz(occ);
            occ &= occ - 1;
            const int ex = bx + (oct >> 2), ey = by + ((oct >> 1) & 1), ez = bz + (oct & 1);  // neighbour - point
            if (ex >= -2 && ex <= 2 && ey >= -2 && ey <= 2 && ez >= -2 && ez <= 2) {
              const int k = cnt - pass * STEM_CAP;
              if (k >= 0 && k < STEM_CAP)
                mylist[k] = ((uint32_t)((ex + 2) * 25 + (ey + 2) * 5 + (ez + 2)) << 24) | (uint32_t)(first + rank);
              ++cnt;
            }
            ++rank;
          }
        }
      }
      // the 4 lanes of a point read each other's lists: same wave, LDS ops of a wave complete in order
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      int c4[STEM_LPP];
      bool more = false;
#pragma unroll
      for (int l = 0; l < STEM_LPP; ++l) {
        const int c = __shfl(cnt, (lane & ~3) | l, 64) - pass * STEM_CAP;
        more |= c > STEM_CAP;
        c4[l] = c < 0 ? 0 : (c > STEM_CAP ? STEM_CAP : c);
      }
      // ---- accumulate: every lane walks the point's 4 lists (fixed summation order: round-major, 3 entries of each
      // list per round).  The <= 12 neighbour rows of a round are requested together, then consumed: one memory round
      // trip per round instead of one per neighbour
      int cmax = 0;
#pragma unroll
      for (int l = 0; l < STEM_LPP; ++l) cmax = c4[l] > cmax ? c4[l] : cmax;
      for (int r0 = 0; r0 < cmax; r0 += RB) {
        uint32_t e[STEM_LPP][RB];
        uint4 xr[STEM_LPP][RB];
#pragma unroll
        for (int l = 0; l < STEM_LPP; ++l)
#pragma unroll
          for (int b = 0; b < RB; ++b) {
            e[l][b] = 0;
            xr[l][b] = make_uint4(0, 0, 0, 0);
            if (r0 + b < c4[l]) {
              e[l][b] = ptlists[l * STEM_CAP + r0 + b];
              xr[l][b] = p.x[e[l][b] & 0xffffffu];
            }
          }
#pragma unroll
        for (int l = 0; l < STEM_LPP; ++l)
#pragma unroll
          for (int b = 0; b < RB; ++b) {
            if (r0 + b >= c4[l]) continue;
            const uint32_t* wr = Ws + (e[l][b] >> 24) * STEM_WROW + q * 8;
            const uint32_t xs[4] = {xr[l][b].x, xr[l][b].y, xr[l][b].z, xr[l][b].w};
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) {
              const uint4 w0 = *reinterpret_cast<const uint4*>(wr + kp * 32);
              const uint4 w1 = *reinterpret_cast<const uint4*>(wr + kp * 32 + 4);
              const uint32_t wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                acc[j] = dot2_bf16(xs[kp], wv[j], acc[j]);
              }
            }
          }
      }
      __builtin_amdgcn_wave_barrier();  // lists are rewritten by the next pass / tile
      if (!__any(more)) break;
    }
    if (valid) {
      const int c0 = q * 8;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = gelu_erf(acc[j] * p.scale[c0 + j] + p.shift[c0 + j]);
      float* o = p.out + i * 32 + c0;
      *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
      if (p.out2) {
        uint4 u;
        u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]);
        u.z = pack_bf16x2(v[4], v[5]); u.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4*>(p.out2 + i * 32 + c0) = u;
      }
    }
  }
}

constexpr int STEM_LDS = (125 * STEM_WROW + STEM_WAVES * STEM_PPW * STEM_LPP * STEM_CAP) * 4;

}  // namespace

extern "C" int cdseg_child_info(const int64_t* zcode_sorted, const int32_t* seg_start, long m, int64_t* info, void* stream) {
  if (m <= 0) return CDSEG_OK;
  hipLaunchKernelGGL(child_info_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, zcode_sorted,
                     seg_start, m, info);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

extern "C" size_t cdseg_stem5_wimg_bytes(void) { return (size_t)125 * STEM_WROW * 4; }

extern "C" int cdseg_stem5_pack(const void* w, void* wimg, void* stream) {
  if (!w || !wimg) return CDSEG_ERR_ARG;
  hipLaunchKernelGGL(stem_pack_w_kernel, dim3((125 * STEM_WROW + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)w, (uint32_t*)wimg);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

extern "C" int cdseg_stem5(const void* x8, const void* wimg, const float* scale, const float* shift, const int32_t* grid,
                           const int32_t* cluster, const int32_t* parent_nbr3, const int64_t* child_info, long n, long m,
                           int depth, float* out, void* out2, void* stream) {
  if (!x8 || !wimg || !scale || !shift || !grid || !cluster || !parent_nbr3 || !child_info || !out) return CDSEG_ERR_ARG;
  if (n <= 0) return CDSEG_OK;
  if (n >= (1l << 24) || m <= 0) return CDSEG_ERR_UNSUPPORTED;  // list entries carry 24-bit row ids
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)stem5_kernel<STEM_RB>, hipFuncAttributeMaxDynamicSharedMemorySize, STEM_LDS) != hipSuccess)
      return CDSEG_ERR_LAUNCH;
    attr_done = true;
  }
  StemP p;
  p.x = (const uint4*)x8; p.wimg = (const uint32_t*)wimg; p.scale = scale; p.shift = shift; p.grid = grid;
  p.cluster = cluster; p.pnbr = parent_nbr3; p.cinfo = child_info; p.out = out; p.out2 = (bf16_t*)out2;
  p.n = n; p.m = m; p.depth = depth;
  const long tiles = (n + STEM_PPW - 1) / STEM_PPW;
  long blocks = (tiles + STEM_WAVES - 1) / STEM_WAVES;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(stem5_kernel<STEM_RB>, dim3((unsigned)blocks), dim3(STEM_WAVES * 64), STEM_LDS, (hipStream_t)stream, p);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}
