// Block head and tail of the wide stages (C = 32 / 64, bf16) with the activations kept in REGISTERS (gfx950).
//
//   head (ref: ptv3.py:401-414)   x += LN_cpe(y Wl^T + bl) [+ t bias] ;  h = LN1(x) ;  qkv = h Wqkv^T + bqkv
//   tail (ref: ptv3.py:416-427)   x += proj(o) ;  h = LN2(x) ;  x += fc2(GELU(fc1(h))) ;  xc = bf16(x)
//
// The first fused versions (mlp.hip) tile 64 rows per workgroup and stream every weight matrix from L2 into LDS once
// per tile, with a block barrier around every product: 72 KB of weights per 48 KB of activations at C = 64, ~1 TB/s
// of HBM traffic on kernels whose arithmetic is trivial.  Here
//   * ALL weights of the kernel are resident in LDS for the lifetime of a persistent workgroup (head 32 KB, tail
//     72 KB at C = 64), stored as MFMA A-operand fragments (a fragment read is 1 KB contiguous: conflict free);
//   * a wave owns 32 points and never synchronises with another wave: the products are computed TRANSPOSED
//     (D^T = W X^T, v_mfma_f32_16x16x32_bf16), so an accumulator register holds (4 channels) x (one point per lane)
//     and the accumulators of one product ARE the B operand of the next after a bf16 pack - the k-slot <-> channel
//     assignment is free as long as the weight image uses the same one (tools/sim/blockrr_model.py checks the index
//     algebra on the CPU).  LayerNorm statistics are a per-lane sum + two cross-lane adds; the 4C-wide hidden
//     activation never exists outside registers;
//   * the output channels are permuted inside the weight images so that a lane owns C/4 CONSECUTIVE channels of its
//     point: every load / store of x, qkv, xc is a 16-byte access, 4 lanes cover a whole row.
// The erf-GELU (128 per lane per 32 rows) was half of the tail's VALU work as rcp + exp + Horner; it is the packed
// polynomial form of common.h now (the hidden activation is rounded to 16 bits right after).
#include <cstdlib>

#include <atomic>

#include "common.h"
#include "deep.h"

namespace {

constexpr int RR_WAVES = 8;

// ---- weight images.  16-byte unit ((ot * KS + s) * 64 + lane), lane = 16 q + i: W[out(ot, i)][in(s, q, 0..7)]
enum { OUT_P = 0, OUT_NAT = 1 };   // P: (i >> 2) * (NOUT / 4) + ot * 4 + (i & 3);  NAT: 16 ot + i
enum { IN_P = 0, IN_HID = 1 };     // P: q * (K / 4) + 8 s + e;  HID: 32 s + (e < 4 ? 4 q + e : 16 + 4 q + e - 4)

struct PackP {
  const bf16_t* w;  // (nout, k) row-major
  uint4* img;
  int nout, k, out_mode, in_mode;
};

__global__ void rr_pack_kernel(PackP p) {
  const int KS = p.k / 32;
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= (p.nout / 16) * KS * 64) return;
  const int lane = u & 63, s = (u >> 6) % KS, ot = (u >> 6) / KS;
  const int q = lane >> 4, i = lane & 15;
  const int row = p.out_mode == OUT_P ? (i >> 2) * (p.nout / 4) + ot * 4 + (i & 3) : 16 * ot + i;
  union { uint4 v; bf16_t h[8]; } o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int col = p.in_mode == IN_P ? q * (p.k / 4) + 8 * s + e : 32 * s + (e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4));
    o.h[e] = p.w[(long)row * p.k + col];
  }
  p.img[u] = o.v;
}

template <int C>
struct RRCfg {
  static constexpr int KS = C / 32, CT = C / 16, Q = C / 4;  // k steps, 16-channel tiles, channels per lane
  static constexpr int HEAD_W = 4 * C * C * 2, HEAD_P = 9 * C * 4;   // Wl + Wqkv ; bl lnp_g lnp_b colbias ln1_g ln1_b bqkv(3C)
  static constexpr int TAIL_W = 9 * C * C * 2, TAIL_P = 8 * C * 4;   // Wp + W1 + W2 ; bp ln_g ln_b b2 b1(4C)
};

__device__ __forceinline__ bf16x8_t pack8(const f32x4_t& a, const f32x4_t& b) {
  union { bf16x8_t v; uint32_t u[4]; } r;
  r.u[0] = pack_bf16x2(a[0], a[1]); r.u[1] = pack_bf16x2(a[2], a[3]);
  r.u[2] = pack_bf16x2(b[0], b[1]); r.u[3] = pack_bf16x2(b[2], b[3]);
  return r.v;
}

// LayerNorm of a point's C channels held as v[CT] (4 lanes q = 0..3 share the point: lanes l, l ^ 16, l ^ 32, l ^ 48)
template <int CT>
__device__ __forceinline__ void ln_stats(const f32x4_t (&v)[CT], float inv_c, float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < CT; ++t) s += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  mean = s * inv_c;
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const float a = v[t][0] - mean, b = v[t][1] - mean, c = v[t][2] - mean, d = v[t][3] - mean;
    q += (a * a + b * b) + (c * c + d * d);
  }
  q += __shfl_xor(q, 16, 64);
  q += __shfl_xor(q, 32, 64);
  rstd = 1.0f / sqrtf(q * inv_c + eps);
}

struct HeadRR {
  const bf16_t* y; const uint4* wimg; const float* bl; const float* lnp_g; const float* lnp_b; float* x;
  const float* colbias; const float* ln1_g; const float* ln1_b; const float* bqkv; bf16_t* qkv;
  long n; int ldy, ldx, ldqkv; float eps;
};

template <int C>
__global__ __launch_bounds__(RR_WAVES * 64, 4) void head_rr_kernel(HeadRR p) {
  using K = RRCfg<C>;
  constexpr int KS = K::KS, CT = K::CT, Q = K::Q, OT = 3 * C / 16, QQ = 3 * C / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane0 & 15, q = lane0 >> 4;
  {
    uint4* d = reinterpret_cast<uint4*>(smem);
    for (int u = tid; u < K::HEAD_W / 16; u += RR_WAVES * 64) d[u] = p.wimg[u];
    float* pr = reinterpret_cast<float*>(smem + K::HEAD_W);
    for (int c = tid; c < C; c += RR_WAVES * 64) {
      pr[c] = p.bl[c]; pr[C + c] = p.lnp_g[c]; pr[2 * C + c] = p.lnp_b[c]; pr[3 * C + c] = p.colbias ? p.colbias[c] : 0.f;
      pr[4 * C + c] = p.ln1_g[c]; pr[5 * C + c] = p.ln1_b[c];
    }
    for (int c = tid; c < 3 * C; c += RR_WAVES * 64) pr[6 * C + c] = p.bqkv[c];
  }
  __syncthreads();
  const uint4* Wl = reinterpret_cast<const uint4*>(smem);                        // CT x KS fragments
  const uint4* Wq = reinterpret_cast<const uint4*>(smem + C * C * 2);            // OT x KS fragments
  const float* pr = reinterpret_cast<const float*>(smem + K::HEAD_W);
  const float inv_c = 1.0f / C;

  const long tiles = (p.n + 31) / 32;
  for (long tile = (long)blockIdx.x * RR_WAVES + wave; tile < tiles; tile += (long)gridDim.x * RR_WAVES) {
#pragma unroll 1
    for (int g = 0; g < 2; ++g) {
      const long row = tile * 32 + g * 16 + j;
      const bool ok = row < p.n;
      const long rr = ok ? row : p.n - 1;
      int lane = lane0;  // opaque per iteration: the weight fragments are loop invariant and would otherwise be hoisted
      asm volatile("" : "+v"(lane));  // out of the tile loop into ~130 registers
      // y as the B operand: lane (j, q) holds channels q * C/4 + 8 s .. + 7 of its point
      bf16x8_t yf[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) yf[s] = *reinterpret_cast<const bf16x8_t*>(p.y + rr * p.ldy + q * Q + 8 * s);
      f32x4_t xr[CT];
#pragma unroll
      for (int t = 0; t < CT; ++t) xr[t] = *reinterpret_cast<const f32x4_t*>(p.x + rr * p.ldx + q * Q + 4 * t);
      f32x4_t v[CT];
      {
        uint4 wl[CT][KS];  // all fragments of the product in one LDS round trip (see tail_rr_kernel)
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
          for (int s = 0; s < KS; ++s) wl[t][s] = Wl[(t * KS + s) * 64 + lane];
#pragma unroll
        for (int t = 0; t < CT; ++t) {
          v[t] = *reinterpret_cast<const f32x4_t*>(pr + q * Q + 4 * t);  // + bl (C operand)
#pragma unroll
          for (int s = 0; s < KS; ++s)
            v[t] = mfma_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wl[t][s]), yf[s], v[t]);
        }
      }
      float mean, rstd;
      ln_stats<CT>(v, inv_c, p.eps, mean, rstd);
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const f32x4_t ga = *reinterpret_cast<const f32x4_t*>(pr + C + q * Q + 4 * t);
        const f32x4_t be = *reinterpret_cast<const f32x4_t*>(pr + 2 * C + q * Q + 4 * t);
        const f32x4_t tb = *reinterpret_cast<const f32x4_t*>(pr + 3 * C + q * Q + 4 * t);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[t][r] = ((v[t][r] - mean) * rstd * ga[r] + be[r]) + xr[t][r] + tb[r];
        if (ok) *reinterpret_cast<f32x4_t*>(p.x + row * p.ldx + q * Q + 4 * t) = v[t];
      }
      ln_stats<CT>(v, inv_c, p.eps, mean, rstd);
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const f32x4_t ga = *reinterpret_cast<const f32x4_t*>(pr + 4 * C + q * Q + 4 * t);
        const f32x4_t be = *reinterpret_cast<const f32x4_t*>(pr + 5 * C + q * Q + 4 * t);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[t][r] = (v[t][r] - mean) * rstd * ga[r] + be[r];
      }
      bf16x8_t hf[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) hf[s] = pack8(v[2 * s], v[2 * s + 1]);
      // qkv: two 16-channel tiles at a time -> 8 consecutive channels per lane -> one 16-byte store; the fragments of
      // the next pair are requested before the MFMAs of this one
      uint4 wq[2][2][KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        wq[0][0][s] = Wq[(0 * KS + s) * 64 + lane];
        wq[0][1][s] = Wq[(1 * KS + s) * 64 + lane];
      }
#pragma unroll
      for (int ot = 0; ot < OT; ot += 2) {
        const int cur = (ot >> 1) & 1;
        if (ot + 2 < OT) {
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            wq[cur ^ 1][0][s] = Wq[((ot + 2) * KS + s) * 64 + lane];
            wq[cur ^ 1][1][s] = Wq[((ot + 3) * KS + s) * 64 + lane];
          }
        }
        f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(pr + 6 * C + q * QQ + 4 * ot);
        f32x4_t a1 = *reinterpret_cast<const f32x4_t*>(pr + 6 * C + q * QQ + 4 * ot + 4);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          a0 = mfma_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wq[cur][0][s]), hf[s], a0);
          a1 = mfma_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wq[cur][1][s]), hf[s], a1);
        }
        if (ok) *reinterpret_cast<bf16x8_t*>(p.qkv + row * p.ldqkv + q * QQ + 4 * ot) = pack8(a0, a1);
      }
    }
  }
}

struct TailRR {
  const bf16_t* o; const uint4* wimg; const float* bp; const float* ln_g; const float* ln_b; const float* b1;
  const float* b2; float* x; bf16_t* xc;
  long n; int ldo, ldx, ldxc; float eps;
};

template <int C>
__global__ __launch_bounds__(RR_WAVES * 64, 4) void tail_rr_kernel(TailRR p) {
  using K = RRCfg<C>;
  constexpr int KS = K::KS, CT = K::CT, Q = K::Q, HU = 4 * C / 32;  // hidden k steps of 32
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane0 & 15, q = lane0 >> 4;
  {
    uint4* d = reinterpret_cast<uint4*>(smem);
    for (int u = tid; u < K::TAIL_W / 16; u += RR_WAVES * 64) d[u] = p.wimg[u];
    float* pr = reinterpret_cast<float*>(smem + K::TAIL_W);
    for (int c = tid; c < C; c += RR_WAVES * 64) {
      pr[c] = p.bp[c]; pr[C + c] = p.ln_g[c]; pr[2 * C + c] = p.ln_b[c]; pr[3 * C + c] = p.b2[c];
    }
    for (int c = tid; c < 4 * C; c += RR_WAVES * 64) pr[4 * C + c] = p.b1[c];
  }
  __syncthreads();
  const uint4* Wp = reinterpret_cast<const uint4*>(smem);                    // CT x KS
  const uint4* W1 = reinterpret_cast<const uint4*>(smem + C * C * 2);        // (4C / 16) x KS
  const uint4* W2 = reinterpret_cast<const uint4*>(smem + 5 * C * C * 2);    // CT x HU
  const float* pr = reinterpret_cast<const float*>(smem + K::TAIL_W);
  const float inv_c = 1.0f / C;

  const long tiles = (p.n + 31) / 32;
  for (long tile = (long)blockIdx.x * RR_WAVES + wave; tile < tiles; tile += (long)gridDim.x * RR_WAVES) {
#pragma unroll 1
    for (int g = 0; g < 2; ++g) {
      const long row = tile * 32 + g * 16 + j;
      const bool ok = row < p.n;
      const long rr = ok ? row : p.n - 1;
      int lane = lane0;  // opaque per iteration (see head_rr_kernel)
      asm volatile("" : "+v"(lane));
      bf16x8_t of[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) of[s] = *reinterpret_cast<const bf16x8_t*>(p.o + rr * p.ldo + q * Q + 8 * s);
      // x' = (o Wp^T + bp) + x.  All CT * KS weight fragments are requested before the first MFMA (one LDS round trip
      // for the product instead of one per fragment: a wave's chain of ~100 dependent LDS reads per 16 rows was the
      // kernel's critical path)
      f32x4_t x1[CT];
      {
        uint4 wp[CT][KS];
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
          for (int s = 0; s < KS; ++s) wp[t][s] = Wp[(t * KS + s) * 64 + lane];
#pragma unroll
        for (int t = 0; t < CT; ++t) {
          f32x4_t a = *reinterpret_cast<const f32x4_t*>(pr + q * Q + 4 * t);
#pragma unroll
          for (int s = 0; s < KS; ++s)
            a = mfma_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wp[t][s]), of[s], a);
          const f32x4_t r = *reinterpret_cast<const f32x4_t*>(p.x + rr * p.ldx + q * Q + 4 * t);
          x1[t] = a + r;
        }
      }
      float mean, rstd;
      ln_stats<CT>(x1, inv_c, p.eps, mean, rstd);
      bf16x8_t hf[KS];
      {
        f32x4_t h[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t) {
          const f32x4_t ga = *reinterpret_cast<const f32x4_t*>(pr + C + q * Q + 4 * t);
          const f32x4_t be = *reinterpret_cast<const f32x4_t*>(pr + 2 * C + q * Q + 4 * t);
#pragma unroll
          for (int r = 0; r < 4; ++r) h[t][r] = (x1[t][r] - mean) * rstd * ga[r] + be[r];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) hf[s] = pack8(h[2 * s], h[2 * s + 1]);
      }
      // MLP: hidden k step u = two natural 16-row hidden tiles, bias + GELU + bf16 in registers, straight into fc2
      f32x4_t acc2[CT];
#pragma unroll
      for (int t = 0; t < CT; ++t) acc2[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int u = 0; u < HU; ++u) {
        // the step's 2 KS fc1 fragments and CT fc2 fragments in one batch: the fc2 ones arrive during the GELU
        uint4 w1[2][KS], w2[CT];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          w1[0][s] = W1[((2 * u) * KS + s) * 64 + lane];
          w1[1][s] = W1[((2 * u + 1) * KS + s) * 64 + lane];
        }
#pragma unroll
        for (int t = 0; t < CT; ++t) w2[t] = W2[(t * HU + u) * 64 + lane];
        f32x4_t h0 = *reinterpret_cast<const f32x4_t*>(pr + 4 * C + 32 * u + 4 * q);
        f32x4_t h1 = *reinterpret_cast<const f32x4_t*>(pr + 4 * C + 32 * u + 16 + 4 * q);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          h0 = mfma_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w1[0][s]), hf[s], h0);
          h1 = mfma_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w1[1][s]), hf[s], h1);
        }
        gelu_lp4(h0);  // 16-bit result: the packed polynomial form (common.h)
        gelu_lp4(h1);
        const bf16x8_t Hf = pack8(h0, h1);
#pragma unroll
        for (int t = 0; t < CT; ++t)
          acc2[t] = mfma_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w2[t]), Hf, acc2[t]);
      }
      if (ok) {
#pragma unroll
        for (int t = 0; t < CT; ++t) {
          const f32x4_t b2 = *reinterpret_cast<const f32x4_t*>(pr + 3 * C + q * Q + 4 * t);
          acc2[t] = acc2[t] + (b2 + x1[t]);
          *reinterpret_cast<f32x4_t*>(p.x + row * p.ldx + q * Q + 4 * t) = acc2[t];
        }
        if (p.xc) {
#pragma unroll
          for (int t = 0; t < CT; t += 2)
            *reinterpret_cast<bf16x8_t*>(p.xc + row * p.ldxc + q * Q + 4 * t) = pack8(acc2[t], acc2[t + 1]);
        }
      }
    }
  }
}

int pack_one(const void* w, int nout, int k, int out_mode, int in_mode, void* img, hipStream_t s) {
  PackP p;
  p.w = (const bf16_t*)w; p.img = (uint4*)img; p.nout = nout; p.k = k; p.out_mode = out_mode; p.in_mode = in_mode;
  const int units = (nout / 16) * (k / 32) * 64;
  hipLaunchKernelGGL(rr_pack_kernel, dim3((units + 255) / 256), dim3(256), 0, s, p);
  return hipGetLastError() == hipSuccess ? CDSEG_OK : CDSEG_ERR_LAUNCH;
}

int rr_grid(long n, int lds_bytes) {
  const long tiles = (n + 31) / 32;
  const int per_cu = lds_bytes > 80 * 1024 ? 1 : (lds_bytes > 40 * 1024 ? 2 : 3);
  long blocks = (tiles + RR_WAVES - 1) / RR_WAVES;
  if (blocks > 256 * per_cu) blocks = 256 * per_cu;
  return (int)blocks;
}

}  // namespace

extern "C" size_t cdseg_block_rr_img_bytes(int channels, int which) {
  if (channels != 32 && channels != 64 && !deep_supported(channels)) return 0;
  return (size_t)(which == 0 ? 4 : 9) * channels * channels * 2;
}

// head image: Wl (C, C), Wqkv (3C, C);  tail image: Wp (C, C), W1 (4C, C), W2 (C, 4C).  bf16 row-major inputs.
extern "C" int cdseg_block_rr_pack(int channels, const void* wl, const void* wqkv, void* head_img, const void* wp,
                                   const void* w1, const void* w2, void* tail_img, void* stream) {
  // deep stages (C = 128 / 256): per-wave weight streams in consumption order (csrc/deep.hip)
  if (deep_supported(channels)) return deep_pack(channels, wl, wqkv, head_img, wp, w1, w2, tail_img, (hipStream_t)stream);
  if (channels != 32 && channels != 64) return CDSEG_ERR_UNSUPPORTED;
  const int C = channels;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (head_img) {
    if (!wl || !wqkv) return CDSEG_ERR_ARG;
    if ((rc = pack_one(wl, C, C, OUT_P, IN_P, head_img, s)) != CDSEG_OK) return rc;
    if ((rc = pack_one(wqkv, 3 * C, C, OUT_P, IN_P, (char*)head_img + C * C * 2, s)) != CDSEG_OK) return rc;
  }
  if (tail_img) {
    if (!wp || !w1 || !w2) return CDSEG_ERR_ARG;
    if ((rc = pack_one(wp, C, C, OUT_P, IN_P, tail_img, s)) != CDSEG_OK) return rc;
    if ((rc = pack_one(w1, 4 * C, C, OUT_NAT, IN_P, (char*)tail_img + C * C * 2, s)) != CDSEG_OK) return rc;
    if ((rc = pack_one(w2, C, 4 * C, OUT_P, IN_HID, (char*)tail_img + 5 * C * C * 2, s)) != CDSEG_OK) return rc;
  }
  return CDSEG_OK;
}

extern "C" int cdseg_cpe_head_rr(const void* y, int ldy, const void* head_img, const float* bl, const float* lnp_g,
                                 const float* lnp_b, float* x, int ldx, const float* colbias, const float* ln1_g,
                                 const float* ln1_b, float eps, const float* bqkv, void* qkv, int ldqkv, long n,
                                 int channels, int qkv_flags, void* stream) {
  if (n <= 0) return CDSEG_OK;
  if (qkv_flags & ~CDSEG_ATTN_V_BF16) return CDSEG_ERR_ARG;
  // (the register-resident head of the wide stages - off in the product, tools only - writes v in the build's own type)
  if (qkv_flags && !deep_supported(channels)) return CDSEG_ERR_UNSUPPORTED;
  if (!y || !head_img || !bl || !lnp_g || !lnp_b || !x || !ln1_g || !ln1_b || !bqkv || !qkv) return CDSEG_ERR_ARG;
  if (channels != 32 && channels != 64 && !deep_supported(channels)) return CDSEG_ERR_UNSUPPORTED;
  if ((ldy & 7) || (ldx & 3) || (ldqkv & 7) || (((uintptr_t)y | (uintptr_t)x | (uintptr_t)qkv | (uintptr_t)head_img) & 15))
    return CDSEG_ERR_ARG;
  if (deep_supported(channels))
    return deep_head(y, ldy, head_img, bl, lnp_g, lnp_b, x, ldx, x, ldx, colbias, ln1_g, ln1_b, eps, bqkv, qkv, ldqkv, n,
                     channels, qkv_flags, (hipStream_t)stream);
  HeadRR p;
  p.y = (const bf16_t*)y; p.wimg = (const uint4*)head_img; p.bl = bl; p.lnp_g = lnp_g; p.lnp_b = lnp_b; p.x = x;
  p.colbias = colbias; p.ln1_g = ln1_g; p.ln1_b = ln1_b; p.bqkv = bqkv; p.qkv = (bf16_t*)qkv;
  p.n = n; p.ldy = ldy; p.ldx = ldx; p.ldqkv = ldqkv; p.eps = eps;
  hipStream_t s = (hipStream_t)stream;
  if (channels == 32) {
    constexpr int lds = RRCfg<32>::HEAD_W + RRCfg<32>::HEAD_P;
    hipLaunchKernelGGL(head_rr_kernel<32>, dim3(rr_grid(n, lds)), dim3(RR_WAVES * 64), lds, s, p);
  } else {
    constexpr int lds = RRCfg<64>::HEAD_W + RRCfg<64>::HEAD_P;
    hipLaunchKernelGGL(head_rr_kernel<64>, dim3(rr_grid(n, lds)), dim3(RR_WAVES * 64), lds, s, p);
  }
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

extern "C" int cdseg_attn_tail_rr(const void* o, int ldo, const void* tail_img, const float* bp, const float* ln_g,
                                  const float* ln_b, float eps, const float* b1, const float* b2, float* x, int ldx,
                                  void* xc, int ldxc, long n, int channels, void* stream) {
  if (n <= 0) return CDSEG_OK;
  if (!o || !tail_img || !bp || !ln_g || !ln_b || !b1 || !b2 || !x) return CDSEG_ERR_ARG;
  if (channels != 32 && channels != 64 && !deep_supported(channels)) return CDSEG_ERR_UNSUPPORTED;
  if ((ldo & 7) || (ldx & 3) || (xc && (ldxc & 7)) || (((uintptr_t)o | (uintptr_t)x | (uintptr_t)xc | (uintptr_t)tail_img) & 15))
    return CDSEG_ERR_ARG;
  if (deep_supported(channels))
    return deep_tail(o, ldo, tail_img, bp, ln_g, ln_b, eps, b1, b2, x, ldx, x, ldx, xc, ldxc, n, channels, nullptr, 0,
                     (hipStream_t)stream);
  TailRR p;
  p.o = (const bf16_t*)o; p.wimg = (const uint4*)tail_img; p.bp = bp; p.ln_g = ln_g; p.ln_b = ln_b; p.b1 = b1; p.b2 = b2;
  p.x = x; p.xc = (bf16_t*)xc; p.n = n; p.ldo = ldo; p.ldx = ldx; p.ldxc = ldxc; p.eps = eps;
  hipStream_t s = (hipStream_t)stream;
  if (channels == 32) {
    constexpr int lds = RRCfg<32>::TAIL_W + RRCfg<32>::TAIL_P;
    hipLaunchKernelGGL(tail_rr_kernel<32>, dim3(rr_grid(n, lds)), dim3(RR_WAVES * 64), lds, s, p);
  } else {
    constexpr int lds = RRCfg<64>::TAIL_W + RRCfg<64>::TAIL_P;
    static std::atomic<bool> attr_done{false};  // (a concurrent first call sets the attribute twice: harmless)
    if (!attr_done) {
      if (hipFuncSetAttribute((const void*)tail_rr_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return CDSEG_ERR_LAUNCH;
      attr_done = true;
    }
    hipLaunchKernelGGL(tail_rr_kernel<64>, dim3(rr_grid(n, lds)), dim3(RR_WAVES * 64), lds, s, p);
  }
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

// Deep stages (C = 128 / 256 / 512) with the residual rows read from one buffer and written to another (round 6).  With
// x_out != x the few-row launches of a single scene's deep stages cut a tile's weight stream over several workgroups (head:
// one per q / k / v column block; tail: by hidden chunks, through the fp32 workspace `ws` and a reduce launch) - csrc/deep.hip.
// The native Block executor ping-pongs the residual through its scratch arena with this pair.
extern "C" int cdseg_cpe_head_rr2(const void* y, int ldy, const void* head_img, const float* bl, const float* lnp_g,
                                  const float* lnp_b, const float* x, int ldx, float* x_out, int ldx_out, const float* colbias,
                                  const float* ln1_g, const float* ln1_b, float eps, const float* bqkv, void* qkv, int ldqkv,
                                  long n, int channels, int qkv_flags, void* stream) {
  if (n <= 0) return CDSEG_OK;
  if (qkv_flags & ~CDSEG_ATTN_V_BF16) return CDSEG_ERR_ARG;
  if (!deep_supported(channels)) return CDSEG_ERR_UNSUPPORTED;
  if (!y || !head_img || !bl || !lnp_g || !lnp_b || !x || !x_out || !ln1_g || !ln1_b || !bqkv || !qkv) return CDSEG_ERR_ARG;
  if ((ldy & 7) || (ldx & 3) || (ldx_out & 3) || (ldqkv & 7) ||
      (((uintptr_t)y | (uintptr_t)x | (uintptr_t)x_out | (uintptr_t)qkv | (uintptr_t)head_img) & 15))
    return CDSEG_ERR_ARG;
  return deep_head(y, ldy, head_img, bl, lnp_g, lnp_b, x, ldx, x_out, ldx_out, colbias, ln1_g, ln1_b, eps, bqkv, qkv, ldqkv, n,
                   channels, qkv_flags, (hipStream_t)stream);
}

extern "C" int cdseg_attn_tail_rr2(const void* o, int ldo, const void* tail_img, const float* bp, const float* ln_g,
                                   const float* ln_b, float eps, const float* b1, const float* b2, const float* x_in, int ldx_in,
                                   float* x, int ldx, void* xc, int ldxc, long n, int channels, void* ws, size_t ws_bytes,
                                   void* stream) {
  if (n <= 0) return CDSEG_OK;
  if (!deep_supported(channels)) return CDSEG_ERR_UNSUPPORTED;
  if (!o || !tail_img || !bp || !ln_g || !ln_b || !b1 || !b2 || !x || !x_in) return CDSEG_ERR_ARG;
  if ((ldo & 7) || (ldx & 3) || (ldx_in & 3) || (xc && (ldxc & 7)) ||
      (((uintptr_t)o | (uintptr_t)x | (uintptr_t)x_in | (uintptr_t)xc | (uintptr_t)tail_img) & 15))
    return CDSEG_ERR_ARG;
  return deep_tail(o, ldo, tail_img, bp, ln_g, ln_b, eps, b1, b2, x_in, ldx_in, x, ldx, xc, ldxc, n, channels, ws, ws_bytes,
                   (hipStream_t)stream);
}
