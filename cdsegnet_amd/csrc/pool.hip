// Serialized pooling of the wide stages in ONE launch (gfx950, 16-bit trunk):
//
//   out[m] = act(scale * max_{i in children(m)} round16(W x_i + b) + shift)        ref: ptv3.py:506-515, 548-551
//            (SerializedPooling: proj Linear -> torch_scatter.segment_csr(reduce="max") -> BatchNorm (folded) -> GELU)
//
// As two launches (cdseg_gemm, cdseg_segment_max) the projected rows make a round trip through HBM: 960 k x 64 16-bit
// values written and read back (246 MB) to produce 446 k pooled rows - 90 + 84 us on the first pooling of the benchmark
// forward.  Here the projection of 16 children at a time stays on the CU:
//   * the children of a pooled row are contiguous (points are in (batch | z) order, a pooled cell is a code prefix), so
//     a wave takes 16 CONSECUTIVE pooled rows and walks their children 16 at a time;
//   * W is resident in LDS as MFMA A fragments (4 - 16 KB); the product is computed transposed (D^T = W X^T,
//     v_mfma_f32_16x16x32) with the output channels permuted inside the image so that lane (j, q) ends up with the
//     COUT / 4 consecutive channels q * COUT / 4 .. of child j (the register-resident layout of blockrr.hip);
//   * the 16 x COUT projected values are rounded to the 16-bit type (what the two-launch path stores: max and rounding
//     commute, the results are bit-identical to max over the stored rows) and parked in a per-wave LDS strip; lane
//     (p, q) then folds the rows of ITS pooled row m0 + p into a running fp32 maximum - a loop over at most 16 strip
//     rows, no cross-lane traffic, clusters that straddle two strips just continue in the next one;
//   * epilogue per pooled row: folded BatchNorm, GELU (erf), fp32 row + its 16-bit copy, 64 - 128 contiguous bytes per lane.
// HBM traffic = the fine rows once + the pooled rows once.
#include <atomic>

#include "common.h"

namespace {

struct PoolP {
  const bf16_t* x; const uint4* wimg; const float* bias; const int32_t* seg; const float* scale; const float* shift;
  float* out; bf16_t* out2;
  long m;  // pooled rows
  int ldx, ldo, ldo2, act;
};

constexpr int PL_WAVES = 8;

template <int CIN, int COUT>
struct PoolCfg {
  static constexpr int KS = CIN / 32, OT = COUT / 16, QI = CIN / 4, QO = COUT / 4;
  static constexpr int W_BYTES = CIN * COUT * 2;
  static constexpr int ROW = COUT * 2 + 16;             // strip row pitch in bytes (+16: bank spread)
  static constexpr int STRIP = 16 * ROW;                // one wave's strip
  static constexpr int PAR = 3 * COUT * 4;              // bias, scale, shift
  static constexpr int LDS = W_BYTES + PL_WAVES * STRIP + PAR;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(PL_WAVES * 64) void pool_fused_kernel(PoolP p) {
  using K = PoolCfg<CIN, COUT>;
  constexpr int KS = K::KS, OT = K::OT, QI = K::QI, QO = K::QO;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  {
    uint4* d = reinterpret_cast<uint4*>(smem);
    for (int u = tid; u < K::W_BYTES / 16; u += PL_WAVES * 64) d[u] = p.wimg[u];
    float* pr = reinterpret_cast<float*>(smem + K::W_BYTES + PL_WAVES * K::STRIP);
    for (int c = tid; c < COUT; c += PL_WAVES * 64) {
      pr[c] = p.bias ? p.bias[c] : 0.f;
      pr[COUT + c] = p.scale ? p.scale[c] : 1.f;
      pr[2 * COUT + c] = p.shift ? p.shift[c] : 0.f;
    }
  }
  __syncthreads();
  const uint4* W = reinterpret_cast<const uint4*>(smem);  // OT x KS fragments
  char* strip = smem + K::W_BYTES + wave * K::STRIP;
  const float* pr = reinterpret_cast<const float*>(smem + K::W_BYTES + PL_WAVES * K::STRIP);
  // 32 -> 64: the four weight fragments stay in registers for the lifetime of the wave; 64 -> 128 (16 fragments = 64
  // registers next to 32 running maxima) re-reads them from LDS per strip
  constexpr bool KEEPW = OT * KS <= 4;
  uint4 wkeep[KEEPW ? OT : 1][KS];
  if constexpr (KEEPW) {
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
      for (int s = 0; s < KS; ++s) wkeep[t][s] = W[(t * KS + s) * 64 + lane];
  }

  const long chunks = (p.m + 15) / 16;
  for (long chunk = (long)blockIdx.x * PL_WAVES + wave; chunk < chunks; chunk += (long)gridDim.x * PL_WAVES) {
    const long m0 = chunk * 16;
    const long mrow = m0 + j;  // this lane's pooled row (lanes q = 0..3 share it, COUT / 4 channels each)
    const bool valid = mrow < p.m;
    const int s0 = p.seg[valid ? mrow : p.m], s1 = p.seg[valid ? mrow + 1 : p.m];
    const int f0 = __builtin_amdgcn_readfirstlane(s0);
    const long mend = m0 + 16 < p.m ? m0 + 16 : p.m;
    const int f1 = p.seg[mend];  // (wave uniform)
    float mx[QO];
#pragma unroll
    for (int c = 0; c < QO; ++c) mx[c] = -INFINITY;
    // the rows of strip f + 16 are requested while strip f is multiplied and folded (rows past the chunk: duplicates of
    // its last child - never folded, they lie outside every [s0, s1))
    auto fetch = [&](int f, bf16x8_t (&dst)[KS]) {
      int row = f + j;
      row = row < f1 ? row : f1 - 1;
#pragma unroll
      for (int s = 0; s < KS; ++s) dst[s] = *reinterpret_cast<const bf16x8_t*>(p.x + (long)row * p.ldx + 32 * s + 8 * q);
    };
    bf16x8_t xf[KS];
    fetch(f0, xf);
#pragma unroll 1
    for (int f = f0; f < f1; f += 16) {
      bf16x8_t xn[KS];
      fetch(f + 16 < f1 ? f + 16 : f, xn);
      int lo = lane;  // opaque per strip: keeps the weight-fragment reads of the 64 -> 128 form inside the loop
      asm volatile("" : "+v"(lo));
      // projected values of child j, channels q * QO .. + QO - 1, rounded to the 16-bit type -> strip row j
#pragma unroll
      for (int t = 0; t < OT; t += 2) {
        // (bias added behind the product, like the GEMM's epilogue: bit-identical projected values)
        f32x4_t a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        uint4 w0[KS], w1[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          w0[s] = KEEPW ? wkeep[KEEPW ? t : 0][s] : W[(t * KS + s) * 64 + lo];
          w1[s] = KEEPW ? wkeep[KEEPW ? t + 1 : 0][s] : W[((t + 1) * KS + s) * 64 + lo];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          a = mfma_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w0[s]), xf[s], a);
          b = mfma_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w1[s]), xf[s], b);
        }
        a += *reinterpret_cast<const f32x4_t*>(pr + q * QO + 4 * t);
        b += *reinterpret_cast<const f32x4_t*>(pr + q * QO + 4 * t + 4);
        uint4 u;
        u.x = pack_bf16x2(a[0], a[1]); u.y = pack_bf16x2(a[2], a[3]);
        u.z = pack_bf16x2(b[0], b[1]); u.w = pack_bf16x2(b[2], b[3]);
        *reinterpret_cast<uint4*>(strip + j * K::ROW + (q * QO + 4 * t) * 2) = u;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // fold this lane's children that sit in the strip
      const int a0 = (s0 > f ? s0 : f) - f, a1 = (s1 < f + 16 ? s1 : f + 16) - f;
      for (int r = a0; r < a1; ++r) {
        const char* src = strip + r * K::ROW + q * QO * 2;
#pragma unroll
        for (int c8 = 0; c8 < QO / 8; ++c8) {
          const uint4 u = *reinterpret_cast<const uint4*>(src + c8 * 16);
          const uint32_t w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v0, v1;
            unpack_bf16x2(w4[e], v0, v1);
            mx[8 * c8 + 2 * e] = fmaxf(mx[8 * c8 + 2 * e], v0);
            mx[8 * c8 + 2 * e + 1] = fmaxf(mx[8 * c8 + 2 * e + 1], v1);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // the strip is rewritten by the next 16 children
#pragma unroll
      for (int s = 0; s < KS; ++s) xf[s] = xn[s];
    }
    if (valid) {
      float* o = p.out + mrow * p.ldo + q * QO;
#pragma unroll
      for (int c4 = 0; c4 < QO / 4; ++c4) {
        const f32x4_t sc = *reinterpret_cast<const f32x4_t*>(pr + COUT + q * QO + 4 * c4);
        const f32x4_t sh = *reinterpret_cast<const f32x4_t*>(pr + 2 * COUT + q * QO + 4 * c4);
        f32x4_t v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = __builtin_fmaf(mx[4 * c4 + r], sc[r], sh[r]);  // (the expression of segment_max_kernel, bit for bit)
          if (p.act == CDSEG_ACT_GELU) v[r] = gelu_erf(v[r]);
        }
        *reinterpret_cast<f32x4_t*>(o + 4 * c4) = v;
        if (p.out2) {
          uint2 u;
          u.x = pack_bf16x2(v[0], v[1]);
          u.y = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<uint2*>(p.out2 + mrow * p.ldo2 + q * QO + 4 * c4) = u;
        }
      }
    }
  }
}

// fragment image of W (COUT, CIN) row-major: 16-byte unit (ot * KS + s) * 64 + lane, lane = 16 q + i:
//   W[(i >> 2) * (COUT / 4) + 4 ot + (i & 3)][32 s + 8 q + 0..7]      (input channels in natural k-slot order: the fp32 sums
//   of a product are then the GEMM kernel's, bit for bit)
// (MFMA A operand; output row i lands in lane group i >> 2, register i & 3: lane (j, q) holds channels q * COUT / 4 + 4 ot + r)
__global__ void pool_pack_kernel(const bf16_t* __restrict__ w, uint4* __restrict__ img, int cin, int cout) {
  const int KS = cin / 32;
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= (cout / 16) * KS * 64) return;
  const int lane = u & 63, s = (u >> 6) % KS, ot = (u >> 6) / KS;
  const int q = lane >> 4, i = lane & 15;
  const long row = (i >> 2) * (cout / 4) + ot * 4 + (i & 3);
  img[u] = *reinterpret_cast<const uint4*>(w + row * cin + 32 * s + 8 * q);
}

template <int CIN, int COUT>
int launch_pool(const PoolP& p, hipStream_t s) {
  using K = PoolCfg<CIN, COUT>;
  static std::atomic<bool> attr_done{false};  // (a concurrent first call sets the attribute twice: harmless)
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)pool_fused_kernel<CIN, COUT>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS) != hipSuccess)
      return CDSEG_ERR_LAUNCH;
    attr_done = true;
  }
  const long chunks = (p.m + 15) / 16;
  long blocks = (chunks + PL_WAVES - 1) / PL_WAVES;
  const int per_cu = K::LDS > 80 * 1024 ? 1 : (K::LDS > 52 * 1024 ? 2 : 3);
  if (blocks > 256 * per_cu) blocks = 256 * per_cu;
  hipLaunchKernelGGL((pool_fused_kernel<CIN, COUT>), dim3((unsigned)blocks), dim3(PL_WAVES * 64), K::LDS, s, p);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

bool pool_supported(int cin, int cout) { return (cin == 32 && cout == 64) || (cin == 64 && cout == 128); }

}  // namespace

extern "C" size_t cdseg_pool_fused_img_bytes(int cin, int cout) { return pool_supported(cin, cout) ? (size_t)cin * cout * 2 : 0; }

extern "C" int cdseg_pool_fused_pack(const void* w, int cin, int cout, void* wimg, void* stream) {
  if (!w || !wimg || (((uintptr_t)w | (uintptr_t)wimg) & 15)) return CDSEG_ERR_ARG;
  if (!pool_supported(cin, cout)) return CDSEG_ERR_UNSUPPORTED;
  const int units = (cout / 16) * (cin / 32) * 64;
  hipLaunchKernelGGL(pool_pack_kernel, dim3((units + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, (uint4*)wimg, cin,
                     cout);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

extern "C" int cdseg_pool_fused(const void* x, int ldx, const void* wimg, const float* bias, const int32_t* seg_start, long m,
                                const float* scale, const float* shift, int act, float* out, int ldo, void* out2, int ldo2,
                                int cin, int cout, void* stream) {
  if (m <= 0) return CDSEG_OK;
  if (!x || !wimg || !seg_start || !out) return CDSEG_ERR_ARG;
  if (!pool_supported(cin, cout) || (act != CDSEG_ACT_NONE && act != CDSEG_ACT_GELU)) return CDSEG_ERR_UNSUPPORTED;
  if ((ldx & 7) || (ldo & 3) || (out2 && (ldo2 & 3)) || (((uintptr_t)x | (uintptr_t)wimg | (uintptr_t)out) & 15) ||
      (out2 && (((uintptr_t)out2) & 7)) || (bias && (((uintptr_t)bias) & 15)) || (scale && (((uintptr_t)scale) & 15)) ||
      (shift && (((uintptr_t)shift) & 15)) || (!scale != !shift))
    return CDSEG_ERR_ARG;
  PoolP p;
  p.x = (const bf16_t*)x; p.wimg = (const uint4*)wimg; p.bias = bias; p.seg = seg_start; p.scale = scale; p.shift = shift;
  p.out = out; p.out2 = (bf16_t*)out2; p.m = m; p.ldx = ldx; p.ldo = ldo; p.ldo2 = ldo2; p.act = act;
  hipStream_t s = (hipStream_t)stream;
  if (cin == 32) return launch_pool<32, 64>(p, s);
  return launch_pool<64, 128>(p, s);
}
