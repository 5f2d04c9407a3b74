// Training path, first slices: the backward of a PTv3 Block (attention core, LayerNorm, GELU; weight / bias gradients of
// the Linears and of the submanifold conv), exact fp32.
//
// ref: pointcept/models/default.py:424-493 (training forward), engines/train.py:216-271 (loss.backward()); what
//      autograd differentiates here: point_transformer_v3m1_base.py:246-296 (SerializedAttention core),
//      :399-428 (Block: x += proj(attn); x += fc2(GELU(fc1(LN(x))))).
//
// These are the fp32 kernels of the 1e-3 parity mode's backward, written to pin the arithmetic and the index plumbing
// (slot plan, padding duplicates) against the reference's autograd.  The attention core and the weight gradients run on
// the fp32 matrix pipe (v_mfma_f32_16x16x4_f32); LayerNorm / GELU are row kernels.  The 16-bit recompute-P form of
// attn_bf16_kernel's backward (what an AMP training step would run) is the next step.
#include <mutex>

#include "common.h"

namespace {

constexpr int HD = 16;       // head dim
constexpr int BW_WAVES = 16;  // waves per block of the attention backward kernels (1 block per CU: the LDS holds K + V or Q + dO)
constexpr int BW_MAXL = 1024;

struct AttnBwdP {
  const float* q; const float* k; const float* v; const float* dout;
  const int32_t* q_gidx; const int32_t* kv_gidx; const int32_t* widx; const int32_t* patch_start;
  float* dq; float* dk; float* dv;
  float* stats;  // (slot, head, {m (log2 units of the prescaled scores), 1 / l, D})
  int ldq, ldk, ldv, lddo, lddq, lddk, lddv;
  int num_heads;
  float scale;
};

// The attention backward on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32), exact fp32 like the parity mode's forward
// (attn_f32_kernel).  With s_ij = scale q_i.k_j, P = softmax_j(s), dP_ij = dO_i.v_j, D_i = sum_j P_ij dP_ij,
// dS = P o (dP - D):   dQ = scale dS K,   dK = scale dS^T Q,   dV = P^T dO.
// One block = one (patch, head); two kernels, each in the orientation that needs no transposition:
//   q kernel  (K, V of the patch-head in LDS; a wave owns 16-query tiles): S^T = K Q^T and dP^T = V dO^T come out of the
//             MFMA with a query per lane column and 4 keys per lane - sweep 1 keeps the running (max, sum, sum P dP) of
//             the online softmax and leaves (m, 1 / l, D) per query, sweep 2 recomputes them and feeds dS^T STRAIGHT from
//             the accumulator registers into dQ^T += K^T dS^T (MFMA t takes k-slot kq <-> key 4 kq + t, which is register
//             t of lane group kq: no permute);
//   kv kernel (Q', dO and the statistics in LDS; a wave owns 16-key tiles): S = Q' K^T, dP = dO V^T with a key per lane
//             column, then dV^T += dO^T P and dK^T += Q'^T dS the same way.
// Q is prescaled by scale * log2(e) so that P = exp2(s' - m'); 20 + 16 MFMAs per 16 x 16 (query, key) tile.
// LDS rows are 16 floats; the four 16-byte chunks of a row are XOR-swizzled with (row >> 1) & 3 (conflict-free float4
// operand reads, like attn_f32_kernel's K image).
__device__ __forceinline__ int bw_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 3)) << 4); }
__device__ __forceinline__ float bw_elem(const char* base, int row, int d) {
  return *reinterpret_cast<const float*>(base + bw_off(row, d >> 2) + (d & 3) * 4);
}

__global__ __launch_bounds__(BW_WAVES * 64) void attn_bwd_q_mfma_kernel(AttnBwdP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + BW_MAXL * 64;
  const int patch = blockIdx.x, head = blockIdx.y;
  const int ps = p.patch_start[patch], L = p.patch_start[patch + 1] - ps;
  const int nt = (L + 15) >> 4, Lp = nt << 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int s = tid; s < Lp; s += BW_WAVES * 64) {
    float4 kk[4], vv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) kk[c] = vv[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s < L) {
      const long r = p.kv_gidx[ps + s];
      const float4* kp = reinterpret_cast<const float4*>(p.k + r * p.ldk + head * HD);
      const float4* vp = reinterpret_cast<const float4*>(p.v + r * p.ldv + head * HD);
#pragma unroll
      for (int c = 0; c < 4; ++c) { kk[c] = kp[c]; vv[c] = vp[c]; }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      *reinterpret_cast<float4*>(Ks + bw_off(s, c)) = kk[c];
      *reinterpret_cast<float4*>(Vs + bw_off(s, c)) = vv[c];
    }
  }
  __syncthreads();
  const int ql = lane & 15, g = lane >> 4;
  const float c2 = p.scale * 1.44269504088896340736f;
  for (int qt = wave; qt < nt; qt += BW_WAVES) {
    const int qslot = qt * 16 + ql;
    const bool valid = qslot < L;
    long qrow = 0;
    float qf[4] = {0.f, 0.f, 0.f, 0.f}, gf[4] = {0.f, 0.f, 0.f, 0.f};  // B operands: (query ql, head dims 4 g .. 4 g + 3)
    if (valid) {
      qrow = p.q_gidx[ps + qslot];
      const float4 t = *reinterpret_cast<const float4*>(p.q + qrow * p.ldq + head * HD + 4 * g);
      qf[0] = t.x * c2; qf[1] = t.y * c2; qf[2] = t.z * c2; qf[3] = t.w * c2;
      const int w = p.widx[ps + qslot];  // the slot's output row; padding duplicates have none: dO = 0
      if (w >= 0) {
        const float4 u = *reinterpret_cast<const float4*>(p.dout + (long)w * p.lddo + head * HD + 4 * g);
        gf[0] = u.x; gf[1] = u.y; gf[2] = u.z; gf[3] = u.w;
      }
    }
    auto tile = [&](int kt, f32x4_t& sc, f32x4_t& dp) {  // S'^T and dP^T of key tile kt: rows = keys 4 g + r, column = query ql
      const int key = kt * 16 + ql;
      const float4 kf = *reinterpret_cast<const float4*>(Ks + bw_off(key, g));
      const float4 vf = *reinterpret_cast<const float4*>(Vs + bw_off(key, g));
      const float ka[4] = {kf.x, kf.y, kf.z, kf.w}, va[4] = {vf.x, vf.y, vf.z, vf.w};
      sc = f32x4_t{0.f, 0.f, 0.f, 0.f};
      dp = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        sc = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[t], qf[t], sc, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x4f32(va[t], gf[t], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (kt * 16 + 4 * g + r >= L) sc[r] = -INFINITY;
    };
    // ---- sweep 1: online (m, l, sum e dP) over the lane's keys, then across the four lane groups of the query
    float m = -INFINITY, l = 0.f, dn = 0.f;
    for (int kt = 0; kt < nt; ++kt) {
      f32x4_t sc, dp;
      tile(kt, sc, dp);
      const float mt = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
      const float mn = fmaxf(m, mt);
      if (mn > -INFINITY) {
        const float a = exp2f(m - mn);
        l *= a; dn *= a;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = exp2f(sc[r] - mn);
          l += e;
          dn = fmaf(e, dp[r], dn);
        }
        m = mn;
      }
    }
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64), d2 = __shfl_xor(dn, off, 64);
      const float mn = fmaxf(m, m2);
      if (mn > -INFINITY) {
        const float a = exp2f(m - mn), b = exp2f(m2 - mn);
        l = l * a + l2 * b;
        dn = dn * a + d2 * b;
      }
      m = mn;
    }
    const float il = 1.0f / l, D = dn * il;
    if (valid && g == 0) {
      float* st = p.stats + ((long)(ps + qslot) * p.num_heads + head) * 3;
      st[0] = m; st[1] = il; st[2] = D;
    }
    // ---- sweep 2: dQ^T (rows = head dims 4 g + r, column = query ql) += K^T dS^T
    f32x4_t dq = {0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nt; ++kt) {
      f32x4_t sc, dp;
      tile(kt, sc, dp);
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[r] = exp2f(sc[r] - m) * il * (dp[r] - D);  // (masked keys: exp2(-inf) = 0)
#pragma unroll
      for (int t = 0; t < 4; ++t)  // k-slot g of MFMA t <-> key 4 g + t: A = K[key][head dim ql], B = register t
        dq = __builtin_amdgcn_mfma_f32_16x16x4f32(bw_elem(Ks, kt * 16 + 4 * g + t, ql), ds[t], dq, 0, 0, 0);
    }
    if (valid) {
      float* o = p.dq + qrow * p.lddq + head * HD + 4 * g;  // a point padded into two slots collects both
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(o + r, dq[r] * p.scale);
    }
  }
}

__global__ __launch_bounds__(BW_WAVES * 64) void attn_bwd_kv_mfma_kernel(AttnBwdP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;                                                     // Q' = Q * scale * log2(e)
  char* Gs = smem + BW_MAXL * 64;                                      // dO
  float* sm = reinterpret_cast<float*>(smem + 2 * BW_MAXL * 64);       // m
  float* sil = sm + BW_MAXL;                                           // 1 / l
  float* sD = sil + BW_MAXL;                                           // D
  const int patch = blockIdx.x, head = blockIdx.y;
  const int ps = p.patch_start[patch], L = p.patch_start[patch + 1] - ps;
  const int nt = (L + 15) >> 4, Lp = nt << 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float c2 = p.scale * 1.44269504088896340736f;
  for (int s = tid; s < Lp; s += BW_WAVES * 64) {
    float4 qq[4], gg[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) qq[c] = gg[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    float m = INFINITY, il = 0.f, D = 0.f;  // slots past the end: P = exp2(0 - inf) = 0
    if (s < L) {
      const long r = p.q_gidx[ps + s];
      const float4* qp = reinterpret_cast<const float4*>(p.q + r * p.ldq + head * HD);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 t = qp[c];
        qq[c] = make_float4(t.x * c2, t.y * c2, t.z * c2, t.w * c2);
      }
      const int w = p.widx[ps + s];
      if (w >= 0) {
        const float4* gp = reinterpret_cast<const float4*>(p.dout + (long)w * p.lddo + head * HD);
#pragma unroll
        for (int c = 0; c < 4; ++c) gg[c] = gp[c];
      }
      const float* st = p.stats + ((long)(ps + s) * p.num_heads + head) * 3;
      m = st[0]; il = st[1]; D = st[2];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      *reinterpret_cast<float4*>(Qs + bw_off(s, c)) = qq[c];
      *reinterpret_cast<float4*>(Gs + bw_off(s, c)) = gg[c];
    }
    sm[s] = m; sil[s] = il; sD[s] = D;
  }
  __syncthreads();
  const int kl = lane & 15, g = lane >> 4;
  for (int kt = wave; kt < nt; kt += BW_WAVES) {
    const int kslot = kt * 16 + kl;
    const bool valid = kslot < L;
    long krow = 0;
    float kf[4] = {0.f, 0.f, 0.f, 0.f}, vf[4] = {0.f, 0.f, 0.f, 0.f};  // B operands: (key kl, head dims 4 g .. 4 g + 3)
    if (valid) {
      krow = p.kv_gidx[ps + kslot];
      const float4 a = *reinterpret_cast<const float4*>(p.k + krow * p.ldk + head * HD + 4 * g);
      const float4 b = *reinterpret_cast<const float4*>(p.v + krow * p.ldv + head * HD + 4 * g);
      kf[0] = a.x; kf[1] = a.y; kf[2] = a.z; kf[3] = a.w;
      vf[0] = b.x; vf[1] = b.y; vf[2] = b.z; vf[3] = b.w;
    }
    f32x4_t dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};  // rows = head dims 4 g + r, column = key kl
    for (int qt = 0; qt < nt; ++qt) {
      const int qrow = qt * 16 + kl;  // A operand row = query qt * 16 + (lane & 15)
      const float4 qa4 = *reinterpret_cast<const float4*>(Qs + bw_off(qrow, g));
      const float4 ga4 = *reinterpret_cast<const float4*>(Gs + bw_off(qrow, g));
      const float qa[4] = {qa4.x, qa4.y, qa4.z, qa4.w}, ga[4] = {ga4.x, ga4.y, ga4.z, ga4.w};
      f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};  // rows = queries 4 g + r, column = key kl
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        sc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[t], kf[t], sc, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[t], vf[t], dp, 0, 0, 0);
      }
      const float4 m4 = *reinterpret_cast<const float4*>(sm + qt * 16 + 4 * g);
      const float4 i4 = *reinterpret_cast<const float4*>(sil + qt * 16 + 4 * g);
      const float4 D4 = *reinterpret_cast<const float4*>(sD + qt * 16 + 4 * g);
      const float ms[4] = {m4.x, m4.y, m4.z, m4.w}, is[4] = {i4.x, i4.y, i4.z, i4.w}, Ds[4] = {D4.x, D4.y, D4.z, D4.w};
      float pr[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[r] = exp2f(sc[r] - ms[r]) * is[r];
        ds[r] = pr[r] * (dp[r] - Ds[r]);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {  // k-slot g of MFMA t <-> query 4 g + t: A = dO / Q' [query][head dim kl], B = register t
        const int qq = qt * 16 + 4 * g + t;
        dv = __builtin_amdgcn_mfma_f32_16x16x4f32(bw_elem(Gs, qq, kl), pr[t], dv, 0, 0, 0);
        dk = __builtin_amdgcn_mfma_f32_16x16x4f32(bw_elem(Qs, qq, kl), ds[t], dk, 0, 0, 0);
      }
    }
    if (valid) {
      float* ok = p.dk + krow * p.lddk + head * HD + 4 * g;
      float* ov = p.dv + krow * p.lddv + head * HD + 4 * g;
      const float ln2 = 0.69314718055994530942f;  // Q' carries scale * log2(e): dK = scale dS^T Q = ln 2 dS^T Q'
#pragma unroll
      for (int r = 0; r < 4; ++r) { atomicAdd(ok + r, dk[r] * ln2); atomicAdd(ov + r, dv[r]); }
    }
  }
}

constexpr int BW_Q_LDS = 2 * BW_MAXL * 64;
constexpr int BW_KV_LDS = 2 * BW_MAXL * 64 + 3 * BW_MAXL * 4;

// ---- LayerNorm backward: dx = (1/sigma) (dyg - mean(dyg) - xhat mean(dyg xhat)), dyg = dy * gamma.  One wave per row, 16 rows
// per wave; d gamma / d beta are summed over the block's 64 rows in registers + LDS and leave with ONE atomic per column
// and block (the first version issued one atomic per element onto C addresses: 82 of a 450 ms training step).
constexpr int LNB_ROWS = 16, LNB_MAXC = 512;

__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                            float eps, const float* __restrict__ dy, int lddy, float* dx, int lddx,
                                                            int accumulate, float* dgamma, float* dbeta, long m, int c) {
  __shared__ float red[2][4][LNB_MAXC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row0 = ((long)blockIdx.x * 4 + wave) * LNB_ROWS;
  const bool sums = (dgamma || dbeta) && c <= LNB_MAXC;
  float pg[LNB_MAXC / 64], pb[LNB_MAXC / 64];
#pragma unroll
  for (int t = 0; t < LNB_MAXC / 64; ++t) pg[t] = pb[t] = 0.f;
  for (int r = 0; r < LNB_ROWS; ++r) {
    const long row = row0 + r;
    if (row >= m) break;
    const float* xr = x + row * ldx;
    const float* gr = dy + row * lddy;
    float s = 0.f, ss = 0.f;
    for (int j = lane; j < c; j += 64) { const float t = xr[j]; s += t; }
    s = wave_sum(s);
    const float mean = s / c;
    for (int j = lane; j < c; j += 64) { const float t = xr[j] - mean; ss = fmaf(t, t, ss); }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / c + eps);
    float a = 0.f, b = 0.f;
    for (int j = lane; j < c; j += 64) {
      const float xh = (xr[j] - mean) * rstd, dg = gr[j] * gamma[j];
      a += dg;
      b = fmaf(dg, xh, b);
    }
    a = wave_sum(a) / c;
    b = wave_sum(b) / c;
    if (sums) {
#pragma unroll
      for (int t = 0; t < LNB_MAXC / 64; ++t) {
        const int j = lane + 64 * t;
        if (j < c) {
          const float xh = (xr[j] - mean) * rstd, g = gr[j], dg = g * gamma[j];
          const float v = rstd * (dg - a - xh * b);
          float* o = dx + row * lddx + j;
          *o = accumulate ? *o + v : v;
          pg[t] = fmaf(g, xh, pg[t]);
          pb[t] += g;
        }
      }
    } else {
      for (int j = lane; j < c; j += 64) {
        const float xh = (xr[j] - mean) * rstd, dg = gr[j] * gamma[j];
        const float v = rstd * (dg - a - xh * b);
        float* o = dx + row * lddx + j;
        *o = accumulate ? *o + v : v;
        if (dgamma) atomicAdd(dgamma + j, gr[j] * xh);  // (rows wider than 512: per-element atomics)
        if (dbeta) atomicAdd(dbeta + j, gr[j]);
      }
    }
  }
  if (!sums) return;  // (block-uniform)
#pragma unroll
  for (int t = 0; t < LNB_MAXC / 64; ++t) {
    const int j = lane + 64 * t;
    if (j < c) { red[0][wave][j] = pg[t]; red[1][wave][j] = pb[t]; }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < c; j += 256) {
    const float g = (red[0][0][j] + red[0][1][j]) + (red[0][2][j] + red[0][3][j]);
    const float b = (red[1][0][j] + red[1][1][j]) + (red[1][2][j] + red[1][3][j]);
    if (dgamma) atomicAdd(dgamma + j, g);
    if (dbeta) atomicAdd(dbeta + j, b);
  }
}

// ---- exact (erf) GELU backward on the pre-activation: dx = dy * (Phi(u) + u phi(u))
__global__ void gelu_bwd_kernel(const float* __restrict__ u, const float* __restrict__ dy, float* __restrict__ dx, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float t = u[i];
  const float cdf = 0.5f * (1.0f + erff(t * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * t * t);
  dx[i] = dy[i] * (cdf + t * pdf);
}

// ---- weight gradient of a Linear / of one kernel offset of a submanifold conv:
//        dW[n][k] += sum_m dY[m][n] * X[row(m)][k],   row(m) = xidx ? xidx[m] (-1: no neighbour, skipped) : m
//      (+ db[n] += sum_m dY[m][n]).   ref: what autograd does for nn.Linear (ptv3.py:399-428) and spconv.SubMConv3d
//      (ptv3.py:356-362) weights; dW = dY^T X is a GEMM whose reduction runs over the ROWS (10^5 .. 10^6), so the rows are
//      split over blockIdx.z and the partial tiles are added with fp32 atomics (summation order not fixed: ~1e-7 relative).
//      v_mfma_f32_16x16x4_f32 straight from global memory: A[i][kk] = dY[m0 + kk][n0 + i], B[kk][j] = X[row(m0 + kk)][k0 + j]
//      (lane = i + 16 kk); a block is 4 waves = a 64 (n) x 64 (k) tile of dW.  Not tuned (first slice).
struct WgradP {
  const float* dy; const float* x; const int32_t* xidx; float* dw; float* db;
  long M, rows_per_split, idx_stride;  // idx_stride: elements between the index rows of consecutive offsets (conv form)
  int N, K, lddy, ldx, lddw;
  int splits, dw_off_stride;           // blockIdx.z = offset * splits + split; offset o adds o * dw_off_stride to dw
};

__global__ __launch_bounds__(256) void wgrad_kernel(WgradP p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.x * 64 + wave * 16, k0 = blockIdx.y * 64;
  if (n0 >= p.N) return;
  const int off = blockIdx.z / p.splits;  // kernel offset (conv form; 0 for a Linear)
  const long m_begin = (long)(blockIdx.z - off * p.splits) * p.rows_per_split;
  const long m_end = min(p.M, m_begin + p.rows_per_split);
  const int32_t* xidx = p.xidx ? p.xidx + off * p.idx_stride : nullptr;
  float* dw = p.dw + (long)off * p.dw_off_stride;
  f32x4_t acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  bool kt_ok[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) kt_ok[t] = k0 + 16 * t < p.K;
#pragma unroll 4
  for (long m0 = m_begin; m0 < m_end; m0 += 4) {
    const long m = m0 + kq;
    float a = 0.f, b[4] = {0.f, 0.f, 0.f, 0.f};
    if (m < m_end) {
      a = p.dy[m * p.lddy + n0 + i16];
      const long row = xidx ? (long)xidx[m] : m;
      if (row >= 0) {
        const float* xr = p.x + row * p.ldx + k0 + i16;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (kt_ok[t]) b[t] = xr[16 * t];
      }
    }
    bsum += a;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[t], acc[t], 0, 0, 0);
  }
  // D[i][j]: lane holds j = lane & 15, i = 4 (lane >> 4) + r
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (!kt_ok[t]) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(dw + (long)(n0 + 4 * kq + r) * p.lddw + k0 + 16 * t + i16, acc[t][r]);
  }
  if (p.db && blockIdx.y == 0 && off == 0) {
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (lane < 16) atomicAdd(p.db + n0 + lane, bsum);
  }
}

}  // namespace

extern "C" size_t cdseg_attention_bwd_ws_bytes(long num_slots, int num_heads) {
  return (size_t)num_slots * num_heads * 3 * sizeof(float);
}

extern "C" int cdseg_attention_bwd(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv,
                                   const int32_t* q_gidx, const int32_t* kv_gidx, const int32_t* widx,
                                   const int32_t* patch_start, int num_patches, int num_heads, long num_slots, int max_len,
                                   float scale, const void* dout, int lddo, void* dq, void* dk, void* dv, int lddq, int lddk,
                                   int lddv, int dtype, void* ws, size_t ws_bytes, void* stream) {
  if (num_patches <= 0 || num_heads <= 0) return CDSEG_OK;
  if (dtype != CDSEG_F32) return CDSEG_ERR_UNSUPPORTED;  // first slice: the exact-fp32 mode
  // the patch-head lives in LDS (like the forward): a longer patch would write past the staged K / V / statistics
  if (max_len <= 0 || max_len > CDSEG_MAX_PATCH) return CDSEG_ERR_UNSUPPORTED;
  if (!q || !k || !v || !dout || !dq || !dk || !dv || !patch_start) return CDSEG_ERR_ARG;
  // float4 operand loads: 16-byte aligned rows
  if ((ldq | ldk | ldv | lddo | lddq | lddk | lddv) & 3) return CDSEG_ERR_ARG;
  if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) != 0)
    return CDSEG_ERR_ARG;
  if (!ws || ws_bytes < cdseg_attention_bwd_ws_bytes(num_slots, num_heads)) return CDSEG_ERR_WORKSPACE;
  AttnBwdP p;
  p.q = (const float*)q; p.k = (const float*)k; p.v = (const float*)v; p.dout = (const float*)dout;
  p.q_gidx = q_gidx; p.kv_gidx = kv_gidx; p.widx = widx; p.patch_start = patch_start;
  p.dq = (float*)dq; p.dk = (float*)dk; p.dv = (float*)dv; p.stats = (float*)ws;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
  p.num_heads = num_heads; p.scale = scale;
  hipStream_t s = (hipStream_t)stream;
  static std::once_flag attr_once;  // (one device per process: one process per GPU)
  static bool attr_ok = false;
  std::call_once(attr_once, [] {
    attr_ok = hipFuncSetAttribute((const void*)attn_bwd_q_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BW_Q_LDS) == hipSuccess &&
              hipFuncSetAttribute((const void*)attn_bwd_kv_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BW_KV_LDS) == hipSuccess;
  });
  if (!attr_ok) return CDSEG_ERR_LAUNCH;
  dim3 grid((unsigned)num_patches, (unsigned)num_heads);
  hipLaunchKernelGGL(attn_bwd_q_mfma_kernel, grid, dim3(BW_WAVES * 64), BW_Q_LDS, s, p);
  hipLaunchKernelGGL(attn_bwd_kv_mfma_kernel, grid, dim3(BW_WAVES * 64), BW_KV_LDS, s, p);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

extern "C" int cdseg_layernorm_bwd(const float* x, int ldx, const float* gamma, float eps, const float* dy, int lddy, float* dx,
                                   int lddx, int accumulate, float* dgamma, float* dbeta, long m, int c, void* stream) {
  if (m <= 0) return CDSEG_OK;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)cdiv(m, 4 * LNB_ROWS)), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, eps, dy,
                     lddy, dx, lddx, accumulate, dgamma, dbeta, m, c);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

extern "C" int cdseg_gelu_bwd(const float* u, const float* dy, float* dx, long n, void* stream) {
  if (n <= 0) return CDSEG_OK;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, u, dy, dx, n);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

static int launch_wgrad(WgradP p, int noff, hipStream_t stream) {
  const int tiles = cdiv(p.N, 64) * cdiv(p.K, 64) * noff;
  long splits = (2048 + tiles - 1) / tiles;     // ~8 blocks per CU
  const long max_splits = (p.M + 1023) / 1024;  // at least 1024 rows per block: 4096 atomics per tile
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  p.rows_per_split = ((p.M + splits - 1) / splits + 3) & ~3L;
  splits = (p.M + p.rows_per_split - 1) / p.rows_per_split;
  p.splits = (int)splits;
  hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)cdiv(p.N, 64), (unsigned)cdiv(p.K, 64), (unsigned)(splits * noff)), dim3(256), 0,
                     stream, p);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

extern "C" int cdseg_linear_wgrad(const float* x, int ldx, const int32_t* xidx, const float* dy, int lddy, long m, int k, int n,
                                  float* dw, int lddw, float* db, void* stream) {
  if (m <= 0 || n <= 0 || k <= 0) return CDSEG_OK;
  if ((n & 15) || (k & 15)) return CDSEG_ERR_UNSUPPORTED;
  WgradP p;
  p.dy = dy; p.x = x; p.xidx = xidx; p.dw = dw; p.db = db; p.M = m; p.N = n; p.K = k; p.lddy = lddy; p.ldx = ldx; p.lddw = lddw;
  p.idx_stride = 0; p.dw_off_stride = 0;
  return launch_wgrad(p, 1, (hipStream_t)stream);
}

extern "C" int cdseg_conv_wgrad(const float* x, int ldx, const int32_t* nbr_kmajor, int kvol, const float* dy, int lddy, long m,
                                int cin, int cout, float* dw, float* db, void* stream) {
  if (m <= 0 || kvol <= 0) return CDSEG_OK;
  if ((cin & 15) || (cout & 15)) return CDSEG_ERR_UNSUPPORTED;
  WgradP p;
  p.dy = dy; p.x = x; p.xidx = nbr_kmajor; p.dw = dw; p.db = db; p.M = m; p.N = cout; p.K = cin; p.lddy = lddy; p.ldx = ldx;
  p.lddw = kvol * cin; p.idx_stride = m; p.dw_off_stride = cin;
  return launch_wgrad(p, kvol, (hipStream_t)stream);
}
