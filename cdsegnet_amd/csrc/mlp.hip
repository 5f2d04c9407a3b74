// Fused PTv3 MLP for the big stages (C = 32 / 64 / 128 at 14k - 120k points per scene):
//     x += fc2(GELU(fc1(h))) ;  xc = T(x)              ref: ptv3.py:299-322 (MLP), :423-427 (Block tail)
// As two GEMM launches the 4C-wide hidden activation makes a round trip through HBM (write + read: 246 MB per
// launch pair for a 4-scene batch at C = 64), which is most of what those launches move.  Here a workgroup owns 64
// rows: the hidden tile (64 x 64) is produced by MFMA from the h rows in LDS, passed through bias + GELU, rounded to
// bf16 (exactly what the unfused path stored) and consumed from LDS as the A operand of the second product; only
// h, the residual and the two outputs touch HBM.  bf16 only (the fp32 parity mode keeps the two-GEMM form).
#include <cstdlib>

#include <atomic>

#include "common.h"

namespace {

struct MlpP {
  const bf16_t* h;   // (n, ldh) LayerNorm output
  const bf16_t* w1;  // (4C, C)
  const float* b1;   // (4C)
  const bf16_t* w2;  // (C, 4C)
  const float* b2;   // (C)
  float* x;          // (n, ldx) fp32 residual stream, updated in place
  bf16_t* xc;        // (n, ldxc) bf16 shadow of x, or nullptr
  long n;
  int ldh, ldx, ldxc;
  // PROJ form (attention tail fused in front): h is the attention output o; x += proj(o); h' = LN2(x) feeds the MLP
  const bf16_t* wp;   // (C, C)
  const float* bp;    // (C)
  const float* ln_g;  // (C)
  const float* ln_b;
  float ln_eps;
};

template <int NCH>
__device__ __forceinline__ int mlp_lds_off(int row, int chunk) {
  constexpr int RB = NCH * 16;
  const int sw = NCH == 16 ? (row & 15) : ((row >> 1) & (NCH - 1));
  return row * RB + ((chunk ^ sw) << 4);
}

// HT: hidden columns per tile
// PROJ: the block's tail after attention in one kernel (ptv3.py:416-427):
//     x += proj(o) ;  h = LN2(x) ;  x += fc2(GELU(fc1(h))) ;  xc = T(x)
// the updated residual rows wait in LDS (fp32) while the MLP runs, h never exists in HBM.
// BM rows per workgroup (4 threads per row; 64, or 128 for the C = 128 MLP: the W1 / W2 slices a workgroup streams from
// L2 - 4 KB per row at BM = 64, more than its HBM bytes - are halved)
template <int C, int HT, bool PROJ, int BM = 64>
__global__ __launch_bounds__(4 * BM) void mlp_fused_kernel(MlpP p) {
  constexpr int NT = 4 * BM;
  constexpr int HID = 4 * C;
  constexpr int NJ = HID / HT;    // hidden tiles
  constexpr int NCA = C / 8;      // 16-byte chunks per h / W1 row (4, 8 or 16)
  constexpr int NCH = HT / 8;     // 16-byte chunks per H / W2-tile row (16 or 8)
  constexpr int TN1 = HT / 32;    // 16-wide hidden column tiles per wave (wave: 32 rows x HT/2 hidden columns)
  constexpr int TN2 = C / 32;     // 16-wide output column tiles per wave (wave: 32 rows x C/2 columns)
  constexpr int A_BYTES = BM * C * 2, W1_BYTES = HT * C * 2, H_BYTES = BM * HT * 2, W2_BYTES = C * HT * 2;
  constexpr int CLD = C + 4;
  // the fp32 C tile of the epilogue aliases W1 + H + W2, or - without PROJ, when h is dead by then - everything from As on
  constexpr bool CS_FROM_A = !PROJ && BM * CLD * 4 > W1_BYTES + H_BYTES + W2_BYTES;
  static_assert(BM * CLD * 4 <= (CS_FROM_A ? A_BYTES : 0) + W1_BYTES + H_BYTES + W2_BYTES, "C tile must fit");
  constexpr int X_BYTES = PROJ ? BM * CLD * 4 : 0;
  static_assert(!PROJ || C * C * 2 <= W1_BYTES + H_BYTES + W2_BYTES, "proj weight must fit the W1 + H + W2 region");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // A_BYTES + W1_BYTES + H_BYTES + W2_BYTES + X_BYTES
  char* As = smem;
  char* W1s = smem + A_BYTES;
  char* Hs = W1s + W1_BYTES;
  char* W2s = Hs + H_BYTES;
  float* Xs = reinterpret_cast<float*>(smem + A_BYTES + W1_BYTES + H_BYTES + W2_BYTES);  // PROJ: x rows after the proj add

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const long m0 = (long)blockIdx.x * BM;

  // h rows -> LDS (once)
  for (int id = tid; id < BM * NCA; id += NT) {
    const int row = id / NCA, ch = id % NCA;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (m0 + row < p.n) v = *reinterpret_cast<const uint4*>(p.h + (m0 + row) * p.ldh + ch * 8);
    *reinterpret_cast<uint4*>(As + mlp_lds_off<NCA>(row, ch)) = v;
  }

  f32x4_t acc2[2][TN2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < TN2; ++t) acc2[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if constexpr (PROJ) {
    // ---- x' = x + o Wp^T + bp  (As holds o; Wp sits in the W1 / H / W2 region, free until the MLP loop)
    for (int id = tid; id < C * NCA; id += NT) {
      const int row = id / NCA, ch = id % NCA;
      *reinterpret_cast<uint4*>(W1s + mlp_lds_off<NCA>(row, ch)) =
          *reinterpret_cast<const uint4*>(p.wp + (long)row * C + ch * 8);
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < NCA / 4; ++kk) {
      bf16x8_t a[2], b[TN2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const bf16x8_t*>(As + mlp_lds_off<NCA>(wm * 32 + i * 16 + fr, 4 * kk + fg));
#pragma unroll
      for (int t = 0; t < TN2; ++t)
        b[t] = *reinterpret_cast<const bf16x8_t*>(W1s + mlp_lds_off<NCA>(wn * (C / 2) + t * 16 + fr, 4 * kk + fg));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < TN2; ++t) acc2[i][t] = mfma_16x16x32_bf16(a[i], b[t], acc2[i][t]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < TN2; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Xs[(wm * 32 + i * 16 + fg * 4 + r) * CLD + wn * (C / 2) + t * 16 + fr] = acc2[i][t][r];
        acc2[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    __syncthreads();  // all waves done with o in As and with Wp; Xs complete
    // rows: + bias + residual, LayerNorm, h -> As (A-operand layout).  4 lanes per row, float4 groups part + 4 i
    {
      constexpr int MAXG = C / 16;
      const int row = tid >> 2, part = tid & 3;
      const long m = m0 + row;
      const bool act = m < p.n;
      float4 v[MAXG];
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < MAXG; ++i) {
        const int cg = part + 4 * i;
        v[i] = *reinterpret_cast<const float4*>(Xs + row * CLD + 4 * cg);
        const float4 b = *reinterpret_cast<const float4*>(p.bp + 4 * cg);
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act) r = *reinterpret_cast<const float4*>(p.x + m * p.ldx + 4 * cg);
        v[i].x += b.x; v[i].y += b.y; v[i].z += b.z; v[i].w += b.w;  // same order as the GEMM epilogue: + bias, + res
        v[i].x += r.x; v[i].y += r.y; v[i].z += r.z; v[i].w += r.w;
        *reinterpret_cast<float4*>(Xs + row * CLD + 4 * cg) = v[i];
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      const float mean = sum * (1.0f / C);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < MAXG; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
      }
      q += __shfl_xor(q, 1, 64);
      q += __shfl_xor(q, 2, 64);
      const float rstd = 1.0f / sqrtf(q * (1.0f / C) + p.ln_eps);
#pragma unroll
      for (int i = 0; i < MAXG; ++i) {
        const int cg = part + 4 * i;
        const float4 ga = *reinterpret_cast<const float4*>(p.ln_g + 4 * cg);
        const float4 be = *reinterpret_cast<const float4*>(p.ln_b + 4 * cg);
        uint2 u;
        u.x = pack_bf16x2((v[i].x - mean) * rstd * ga.x + be.x, (v[i].y - mean) * rstd * ga.y + be.y);
        u.y = pack_bf16x2((v[i].z - mean) * rstd * ga.z + be.z, (v[i].w - mean) * rstd * ga.w + be.w);
        *reinterpret_cast<uint2*>(As + mlp_lds_off<NCA>(row, cg >> 1) + (cg & 1) * 8) = u;
      }
    }
    // (the MLP loop's first barrier orders these As / Xs writes before their readers)
  }

#pragma unroll 1
  for (int j = 0; j < NJ; ++j) {
    // W1 rows [HT j, HT j + HT) and W2 columns [HT j, HT j + HT) -> LDS
    for (int id = tid; id < HT * NCA; id += NT) {
      const int row = id / NCA, ch = id % NCA;
      *reinterpret_cast<uint4*>(W1s + mlp_lds_off<NCA>(row, ch)) =
          *reinterpret_cast<const uint4*>(p.w1 + (long)(HT * j + row) * C + ch * 8);
    }
    for (int id = tid; id < C * NCH; id += NT) {
      const int row = id / NCH, ch = id % NCH;
      *reinterpret_cast<uint4*>(W2s + mlp_lds_off<NCH>(row, ch)) =
          *reinterpret_cast<const uint4*>(p.w2 + (long)row * HID + HT * j + ch * 8);
    }
    __syncthreads();

    // ---- hidden tile = h W1_j^T : wave (wm, wn) owns rows 32 wm .. +32, hidden columns (HT/2) wn .. +HT/2
    f32x4_t acc1[2][TN1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < TN1; ++t) acc1[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < NCA / 4; ++kk) {
      bf16x8_t a[2], b[TN1];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const bf16x8_t*>(As + mlp_lds_off<NCA>(wm * 32 + i * 16 + fr, 4 * kk + fg));
#pragma unroll
      for (int t = 0; t < TN1; ++t)
        b[t] = *reinterpret_cast<const bf16x8_t*>(W1s + mlp_lds_off<NCA>(wn * (HT / 2) + t * 16 + fr, 4 * kk + fg));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < TN1; ++t)
          acc1[i][t] = mfma_16x16x32_bf16(a[i], b[t], acc1[i][t]);
    }
    // bias + GELU + bf16 -> Hs in A-operand layout.  MFMA C layout: col = lane & 15, row = 4 (lane >> 4) + r
#pragma unroll
    for (int t = 0; t < TN1; ++t) {
      const int col = wn * (HT / 2) + t * 16 + fr;
      const float bias = p.b1[HT * j + col];
#pragma unroll
      for (int i = 0; i < 2; ++i)
      {
        f32x4_t hv = acc1[i][t] + bias;
        gelu_lp4(hv);  // 16-bit result: the packed polynomial form (common.h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wm * 32 + i * 16 + fg * 4 + r;
          *reinterpret_cast<bf16_t*>(Hs + mlp_lds_off<NCH>(row, col >> 3) + (col & 7) * 2) = f32_to_bf16(hv[r]);
        }
      }
    }
    __syncthreads();

    // ---- acc2 += H_j W2_j^T : K = HT
#pragma unroll
    for (int kk = 0; kk < NCH / 4; ++kk) {
      bf16x8_t a[2], b[TN2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const bf16x8_t*>(Hs + mlp_lds_off<NCH>(wm * 32 + i * 16 + fr, 4 * kk + fg));
#pragma unroll
      for (int t = 0; t < TN2; ++t)
        b[t] = *reinterpret_cast<const bf16x8_t*>(W2s + mlp_lds_off<NCH>(wn * (C / 2) + t * 16 + fr, 4 * kk + fg));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < TN2; ++t) acc2[i][t] = mfma_16x16x32_bf16(a[i], b[t], acc2[i][t]);
    }
    __syncthreads();  // W1s / W2s / Hs are rewritten by the next hidden tile (or become the C tile)
  }

  // ---- epilogue through an fp32 C tile (aliases W1s + Hs + W2s): + b2 + residual, row-contiguous 16-byte accesses
  float* Cs = reinterpret_cast<float*>(CS_FROM_A ? As : W1s);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < TN2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cs[(wm * 32 + i * 16 + fg * 4 + r) * CLD + wn * (C / 2) + t * 16 + fr] = acc2[i][t][r];
  __syncthreads();
  constexpr int GPR = C / 4;
  for (int item = tid; item < BM * GPR; item += NT) {
    const int row = item / GPR, cg = item % GPR;
    const long m = m0 + row;
    if (m >= p.n) continue;
    float4 v = *reinterpret_cast<const float4*>(Cs + row * CLD + 4 * cg);
    const float4 b = *reinterpret_cast<const float4*>(p.b2 + 4 * cg);
    float* xr = p.x + m * p.ldx + 4 * cg;
    const float4 r = PROJ ? *reinterpret_cast<const float4*>(Xs + row * CLD + 4 * cg)
                          : *reinterpret_cast<const float4*>(xr);
    v.x += b.x + r.x; v.y += b.y + r.y; v.z += b.z + r.z; v.w += b.w + r.w;
    *reinterpret_cast<float4*>(xr) = v;
    if (p.xc) {
      uint2 u;
      u.x = pack_bf16x2(v.x, v.y);
      u.y = pack_bf16x2(v.z, v.w);
      *reinterpret_cast<uint2*>(p.xc + m * p.ldxc + 4 * cg) = u;
    }
  }
}

template <int C, int HT, bool PROJ, int BM>
int launch_mlp(const MlpP& p, hipStream_t s) {
  constexpr int CLD = C + 4;
  constexpr int LDS = BM * C * 2 + HT * C * 2 + BM * HT * 2 + C * HT * 2 + (PROJ ? BM * CLD * 4 : 0);
  static std::atomic<bool> attr_done{false};  // (a concurrent first call sets the attribute twice: harmless)
  if (!attr_done) {
    if (LDS > 64 * 1024 && hipFuncSetAttribute((const void*)mlp_fused_kernel<C, HT, PROJ, BM>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return CDSEG_ERR_LAUNCH;
    attr_done = true;
  }
  const dim3 grid((unsigned)((p.n + BM - 1) / BM));
  hipLaunchKernelGGL((mlp_fused_kernel<C, HT, PROJ, BM>), grid, dim3(4 * BM), LDS, s, p);
  return hipGetLastError() == hipSuccess ? CDSEG_OK : CDSEG_ERR_LAUNCH;
}

}  // namespace

// x (n, ldx) fp32 += fc2(GELU(fc1(h))) with h (n, ldh) bf16; xc (n, ldxc) bf16 copy of the new x, or NULL.
// Supported: bf16, C = 32, 64 or 128 (hidden 4C), row strides multiples of 8 (h) / 4 (x, xc), 16-byte aligned pointers.
extern "C" int cdseg_mlp_fused(const void* h, int ldh, const void* w1, const float* b1, const void* w2, const float* b2,
                               float* x, int ldx, void* xc, int ldxc, long n, int channels, int dtype, void* stream) {
  if (n <= 0) return CDSEG_OK;
  if (!h || !w1 || !b1 || !w2 || !b2 || !x) return CDSEG_ERR_ARG;
  if (dtype != CDSEG_BF16 || (channels != 32 && channels != 64 && channels != 128)) return CDSEG_ERR_UNSUPPORTED;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  if ((ldh & 7) || (ldx & 3) || (xc && (ldxc & 3)) || !al16(h) || !al16(w1) || !al16(w2) || !al16(x) || !al16(b2) ||
      (xc && (((uintptr_t)xc) & 7)))
    return CDSEG_ERR_ARG;
  MlpP p;
  p.h = (const bf16_t*)h; p.w1 = (const bf16_t*)w1; p.b1 = b1; p.w2 = (const bf16_t*)w2; p.b2 = b2;
  p.x = x; p.xc = (bf16_t*)xc; p.n = n; p.ldh = ldh; p.ldx = ldx; p.ldxc = ldxc;
  p.wp = nullptr; p.bp = nullptr; p.ln_g = nullptr; p.ln_b = nullptr; p.ln_eps = 0.f;
  // 64-wide hidden tiles: 20 / 32 / 56 KB of LDS per workgroup (C = 32 / 64 / 128).  128-wide tiles were measured 2-3 %
  // slower end to end at C = 64 (56 KB: two workgroups per CU instead of five) and equal at C = 32.
  hipStream_t s = (hipStream_t)stream;
  static const int bm = cdseg_knob("CDSEG_MLP_BM", 128);
  if (channels == 32) return launch_mlp<32, 64, false, 64>(p, s);
  if (channels == 64) return launch_mlp<64, 64, false, 64>(p, s);
  if (bm == 128 && n >= 128 * 512) return launch_mlp<128, 64, false, 128>(p, s);  // 80 KB: two 8-wave workgroups per CU
  return launch_mlp<128, 64, false, 64>(p, s);
}

// Block tail after attention in one launch (ptv3.py:416-427):
//   x (n, ldx) fp32 += proj(o) ; h = LayerNorm(x; ln_g, ln_b, eps) ; x += fc2(GELU(fc1(h))) ; xc = typed copy of x.
// Supported: bf16, channels 32 or 64; else CDSEG_ERR_UNSUPPORTED.
extern "C" int cdseg_attn_tail_fused(const void* o, int ldo, const void* wp, const float* bp, const float* ln_g,
                                     const float* ln_b, float ln_eps, const void* w1, const float* b1, const void* w2,
                                     const float* b2, float* x, int ldx, void* xc, int ldxc, long n, int channels,
                                     int dtype, void* stream) {
  if (n <= 0) return CDSEG_OK;
  if (!o || !wp || !bp || !ln_g || !ln_b || !w1 || !b1 || !w2 || !b2 || !x) return CDSEG_ERR_ARG;
  if (dtype != CDSEG_BF16 || (channels != 32 && channels != 64)) return CDSEG_ERR_UNSUPPORTED;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  if ((ldo & 7) || (ldx & 3) || (xc && (ldxc & 3)) || !al16(o) || !al16(wp) || !al16(w1) || !al16(w2) || !al16(x) ||
      !al16(bp) || !al16(b2) || !al16(ln_g) || !al16(ln_b) || (xc && (((uintptr_t)xc) & 7)))
    return CDSEG_ERR_ARG;
  MlpP p;
  p.h = (const bf16_t*)o; p.w1 = (const bf16_t*)w1; p.b1 = b1; p.w2 = (const bf16_t*)w2; p.b2 = b2;
  p.x = x; p.xc = (bf16_t*)xc; p.n = n; p.ldh = ldo; p.ldx = ldx; p.ldxc = ldxc;
  p.wp = (const bf16_t*)wp; p.bp = bp; p.ln_g = ln_g; p.ln_b = ln_b; p.ln_eps = ln_eps;
  if (channels == 32) return launch_mlp<32, 64, true, 64>(p, (hipStream_t)stream);
  return launch_mlp<64, 64, true, 64>(p, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Block head after the sparse conv in one launch (ptv3.py:401-414):
//     x += LN_cpe(y Wl^T + bl) [+ t bias] ;  h = LN1(x) ;  qkv = h Wqkv^T + bqkv
// y = conv output (n, C).  h lives only in LDS; x is read and written once.
#ifndef CPE_HEAD_PREFETCH
#define CPE_HEAD_PREFETCH 1
#endif
namespace {

struct HeadP {
  const bf16_t* y;
  const bf16_t* wl;     // (C, C)
  const float* bl;
  const float* lnp_g;   // LayerNorm of the linear's output (before the residual add)
  const float* lnp_b;
  float* x;             // (n, ldx) fp32 residual stream, updated in place
  const float* colbias; // (C) timestep bias or nullptr
  const float* ln1_g;   // LayerNorm of the updated x -> h
  const float* ln1_b;
  const bf16_t* wqkv;   // (3C, C)
  const float* bqkv;    // (3C)
  bf16_t* qkv;          // (n, ldqkv)
  int v_bf16;           // IEEE-half build: write the v third as bfloat16 (what cdseg_attention_ex's CDSEG_ATTN_V_BF16 reads)
  long n;
  int ldy, ldx, ldqkv;
  float eps;
};

// BM rows per workgroup (4 threads per row): 128 rows halve the weight reloads (Wl + 3 x Wqkv tile per workgroup) and the
// barriers per row at the same waves per CU (C = 64: 59 KB LDS, 2 x 8 waves instead of 4 x 4)
template <int C, int BM>
__global__ __launch_bounds__(4 * BM) void cpe_head_fused_kernel(HeadP p) {
  constexpr int NT = 4 * BM;
  constexpr int NCA = C / 8;   // 16-byte chunks per row of y / h / W tiles
  constexpr int TN = C / 32;   // 16-wide column tiles per wave (wave: 32 rows x C/2 columns of a C-wide tile)
  constexpr int CLD = C + 4;
  constexpr int A_BYTES = BM * C * 2, W_BYTES = C * C * 2;
  __shared__ __attribute__((aligned(16))) char smem[A_BYTES + W_BYTES + BM * CLD * 4];
  char* As = smem;
  char* Ws = smem + A_BYTES;
  float* Cs = reinterpret_cast<float*>(smem + A_BYTES + W_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const long m0 = (long)blockIdx.x * BM;

  auto load_w = [&](const bf16_t* w) {  // C rows x C
    for (int id = tid; id < C * NCA; id += NT) {
      const int row = id / NCA, ch = id % NCA;
      *reinterpret_cast<uint4*>(Ws + mlp_lds_off<NCA>(row, ch)) = *reinterpret_cast<const uint4*>(w + (long)row * C + ch * 8);
    }
  };
  auto mma_to_cs = [&]() {  // Cs (BM x C) = As (BM x C) Ws^T
    f32x4_t acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < TN; ++t) acc[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < NCA / 4; ++kk) {
      bf16x8_t a[2], b[TN];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const bf16x8_t*>(As + mlp_lds_off<NCA>(wm * 32 + i * 16 + fr, 4 * kk + fg));
#pragma unroll
      for (int t = 0; t < TN; ++t)
        b[t] = *reinterpret_cast<const bf16x8_t*>(Ws + mlp_lds_off<NCA>(wn * (C / 2) + t * 16 + fr, 4 * kk + fg));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[i][t] = mfma_16x16x32_bf16(a[i], b[t], acc[i][t]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Cs[(wm * 32 + i * 16 + fg * 4 + r) * CLD + wn * (C / 2) + t * 16 + fr] = acc[i][t][r];
  };

  // Round 5: the loads whose latency sat exposed behind a barrier are requested a phase ahead, into registers - every weight
  // tile (8 KB at C = 64 = one 16-byte piece per thread) while the previous product runs, and the residual rows of x at the
  // very start instead of after the first product + LayerNorm (the compiler cannot move a load across __syncthreads)
  constexpr int WPT = (C * NCA + NT - 1) / NT;  // 16-byte weight pieces per thread
  uint4 wreg[WPT];
  auto fetch_w = [&](const bf16_t* w) {
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
      const int id = tid + k * NT;
      if (id < C * NCA) wreg[k] = *reinterpret_cast<const uint4*>(w + (long)(id / NCA) * C + (id % NCA) * 8);
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
      const int id = tid + k * NT;
      if (id < C * NCA) *reinterpret_cast<uint4*>(Ws + mlp_lds_off<NCA>(id / NCA, id % NCA)) = wreg[k];
    }
  };
  constexpr int MAXG0 = C / 16;
  float4 xres[MAXG0];
  if (CPE_HEAD_PREFETCH) {
    fetch_w(p.wl);
    const long mrow = m0 + (tid >> 2);
#pragma unroll
    for (int i = 0; i < MAXG0; ++i)
      xres[i] = mrow < p.n ? *reinterpret_cast<const float4*>(p.x + mrow * p.ldx + 4 * ((tid & 3) + 4 * i))
                           : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int id = tid; id < BM * NCA; id += NT) {
    const int row = id / NCA, ch = id % NCA;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (m0 + row < p.n) v = *reinterpret_cast<const uint4*>(p.y + (m0 + row) * p.ldy + ch * 8);
    *reinterpret_cast<uint4*>(As + mlp_lds_off<NCA>(row, ch)) = v;
  }
  if (CPE_HEAD_PREFETCH) {
    store_w();
    fetch_w(p.wqkv);  // the q tile's weights travel behind the first product and the two LayerNorms
  } else {
    load_w(p.wl);
  }
  __syncthreads();
  mma_to_cs();
  __syncthreads();

  // ---- rows (4 lanes each): + bl, LN_cpe, + x, + t bias -> x ;  LN1 -> h (bf16, A-operand layout in As)
  {
    constexpr int MAXG = C / 16;
    const int row = tid >> 2, part = tid & 3;
    const long m = m0 + row;
    const bool act = m < p.n;
    float4 v[MAXG];
    auto row_stats = [&](float& mean, float& rstd) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < MAXG; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      mean = s * (1.0f / C);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < MAXG; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
      }
      q += __shfl_xor(q, 1, 64);
      q += __shfl_xor(q, 2, 64);
      rstd = 1.0f / sqrtf(q * (1.0f / C) + p.eps);
    };
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
      const int cg = part + 4 * i;
      v[i] = *reinterpret_cast<const float4*>(Cs + row * CLD + 4 * cg);
      const float4 b = *reinterpret_cast<const float4*>(p.bl + 4 * cg);
      v[i].x += b.x; v[i].y += b.y; v[i].z += b.z; v[i].w += b.w;
    }
    float mean, rstd;
    row_stats(mean, rstd);
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
      const int cg = part + 4 * i;
      const float4 ga = *reinterpret_cast<const float4*>(p.lnp_g + 4 * cg);
      const float4 be = *reinterpret_cast<const float4*>(p.lnp_b + 4 * cg);
      v[i].x = (v[i].x - mean) * rstd * ga.x + be.x;
      v[i].y = (v[i].y - mean) * rstd * ga.y + be.y;
      v[i].z = (v[i].z - mean) * rstd * ga.z + be.z;
      v[i].w = (v[i].w - mean) * rstd * ga.w + be.w;
      if (act) {
        float* xr = p.x + m * p.ldx + 4 * cg;
        const float4 r = CPE_HEAD_PREFETCH ? xres[i] : *reinterpret_cast<const float4*>(xr);
        v[i].x += r.x; v[i].y += r.y; v[i].z += r.z; v[i].w += r.w;
        if (p.colbias) {
          const float4 t = *reinterpret_cast<const float4*>(p.colbias + 4 * cg);
          v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
        }
        *reinterpret_cast<float4*>(xr) = v[i];
      } else {
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    row_stats(mean, rstd);
#pragma unroll
    for (int i = 0; i < MAXG; ++i) {
      const int cg = part + 4 * i;
      const float4 ga = *reinterpret_cast<const float4*>(p.ln1_g + 4 * cg);
      const float4 be = *reinterpret_cast<const float4*>(p.ln1_b + 4 * cg);
      uint2 u;
      u.x = pack_bf16x2((v[i].x - mean) * rstd * ga.x + be.x, (v[i].y - mean) * rstd * ga.y + be.y);
      u.y = pack_bf16x2((v[i].z - mean) * rstd * ga.z + be.z, (v[i].w - mean) * rstd * ga.w + be.w);
      *reinterpret_cast<uint2*>(As + mlp_lds_off<NCA>(row, cg >> 1) + (cg & 1) * 8) = u;
    }
  }

  // ---- qkv: three C-wide column tiles (q, k, v)
  constexpr int GPR = C / 4;
#pragma unroll 1
  for (int j = 0; j < 3; ++j) {
    __syncthreads();  // As = h complete / previous tile's Cs and Ws consumed
    if (CPE_HEAD_PREFETCH) {
      store_w();
      if (j < 2) fetch_w(p.wqkv + (long)(j + 1) * C * C);
    } else {
      load_w(p.wqkv + (long)j * C * C);
    }
    __syncthreads();
    mma_to_cs();
    __syncthreads();
    for (int item = tid; item < BM * GPR; item += NT) {
      const int row = item / GPR, cg = item % GPR;
      const long m = m0 + row;
      if (m >= p.n) continue;
      float4 v = *reinterpret_cast<const float4*>(Cs + row * CLD + 4 * cg);
      const float4 b = *reinterpret_cast<const float4*>(p.bqkv + j * C + 4 * cg);
      uint2 u;
      if (LP_IS_F16 && j == 2 && p.v_bf16) {  // the attention's P V product is bfloat16 in both builds (attention.hip)
        u.x = pack_truebf16x2(v.x + b.x, v.y + b.y);
        u.y = pack_truebf16x2(v.z + b.z, v.w + b.w);
      } else {
        u.x = pack_bf16x2(v.x + b.x, v.y + b.y);
        u.y = pack_bf16x2(v.z + b.z, v.w + b.w);
      }
      *reinterpret_cast<uint2*>(p.qkv + m * p.ldqkv + j * C + 4 * cg) = u;
    }
  }
}

}  // namespace

// Supported: bf16, channels 32 or 64; else CDSEG_ERR_UNSUPPORTED.  colbias may be NULL.
extern "C" int cdseg_cpe_head_fused(const void* y, int ldy, const void* wl, const float* bl, const float* lnp_g,
                                    const float* lnp_b, float* x, int ldx, const float* colbias, const float* ln1_g,
                                    const float* ln1_b, float eps, const void* wqkv, const float* bqkv, void* qkv,
                                    int ldqkv, long n, int channels, int dtype, int qkv_flags, void* stream) {
  if (n <= 0) return CDSEG_OK;
  if (!y || !wl || !bl || !lnp_g || !lnp_b || !x || !ln1_g || !ln1_b || !wqkv || !bqkv || !qkv) return CDSEG_ERR_ARG;
  if (qkv_flags & ~CDSEG_ATTN_V_BF16) return CDSEG_ERR_ARG;
  if (dtype != CDSEG_BF16 || (channels != 32 && channels != 64)) return CDSEG_ERR_UNSUPPORTED;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  if ((ldy & 7) || (ldx & 3) || (ldqkv & 3) || !al16(y) || !al16(wl) || !al16(wqkv) || !al16(x) || !al16(bl) ||
      !al16(lnp_g) || !al16(lnp_b) || !al16(ln1_g) || !al16(ln1_b) || !al16(bqkv) || (colbias && !al16(colbias)) ||
      (((uintptr_t)qkv) & 7))
    return CDSEG_ERR_ARG;
  HeadP p;
  p.y = (const bf16_t*)y; p.wl = (const bf16_t*)wl; p.bl = bl; p.lnp_g = lnp_g; p.lnp_b = lnp_b; p.x = x;
  p.colbias = colbias; p.ln1_g = ln1_g; p.ln1_b = ln1_b; p.wqkv = (const bf16_t*)wqkv; p.bqkv = bqkv;
  p.qkv = (bf16_t*)qkv; p.n = n; p.ldy = ldy; p.ldx = ldx; p.ldqkv = ldqkv; p.eps = eps;
  p.v_bf16 = (qkv_flags & CDSEG_ATTN_V_BF16) ? 1 : 0;
  static const int bm = cdseg_knob("CDSEG_HEAD_BM", 128);
  if (bm == 128 && n >= 128 * 512) {  // enough 128-row workgroups to fill the chip twice
    const dim3 grid((unsigned)((n + 127) / 128));
    if (channels == 32) hipLaunchKernelGGL((cpe_head_fused_kernel<32, 128>), grid, dim3(512), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((cpe_head_fused_kernel<64, 128>), grid, dim3(512), 0, (hipStream_t)stream, p);
  } else {
    const dim3 grid((unsigned)((n + 63) / 64));
    if (channels == 32) hipLaunchKernelGGL((cpe_head_fused_kernel<32, 64>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((cpe_head_fused_kernel<64, 64>), grid, dim3(256), 0, (hipStream_t)stream, p);
  }
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}
