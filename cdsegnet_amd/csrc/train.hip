// Training path, first slices: the backward of a PTv3 Block (attention core, LayerNorm, GELU; weight / bias gradients of
// the Linears and of the submanifold conv), exact fp32.
//
// ref: pointcept/models/default.py:424-493 (training forward), engines/train.py:216-271 (loss.backward()); what
//      autograd differentiates here: point_transformer_v3m1_base.py:246-296 (SerializedAttention core),
//      :399-428 (Block: x += proj(attn); x += fc2(GELU(fc1(LN(x))))).
//
// These are the fp32 kernels of the 1e-3 parity mode's backward: straightforward VALU code (one thread per query / per
// key, K / V / Q tiles through LDS), written to pin the arithmetic and the index plumbing (slot plan, padding
// duplicates) against the reference's autograd.  The MFMA recompute-P form of attn_bf16_kernel's backward is the next
// step.
#include "common.h"

namespace {

constexpr int TB = 64;   // slots per block (one wave)
constexpr int HD = 16;   // head dim

struct AttnBwdP {
  const float* q; const float* k; const float* v; const float* dout;
  const int32_t* q_gidx; const int32_t* kv_gidx; const int32_t* widx; const int32_t* patch_start;
  float* dq; float* dk; float* dv;
  float* stats;  // (slot, head, {m, l, D})
  int ldq, ldk, ldv, lddo, lddq, lddk, lddv;
  int num_heads;
  float scale;
};

// block -> (patch, head, tile of 64 slots); grid.x = total tiles (host: prefix over patches), looked up by binary search
__device__ __forceinline__ bool locate(const int32_t* patch_start, int num_patches, int tile, int& patch, int& t0) {
  // tiles are laid out patch by patch: tile index -> patch via the cumulative tile counts recomputed on the fly
  int acc = 0;
  for (int p = 0; p < num_patches; ++p) {
    const int L = patch_start[p + 1] - patch_start[p];
    const int nt = (L + TB - 1) / TB;
    if (tile < acc + nt) { patch = p; t0 = (tile - acc) * TB; return true; }
    acc += nt;
  }
  return false;
}

// ---- pass over the QUERIES of a tile: softmax statistics, D_i = sum_j P_ij (dO_i . V_j), then dQ_i
__global__ __launch_bounds__(TB) void attn_bwd_q_kernel(AttnBwdP p, int num_patches) {
  __shared__ float Ks[TB][HD + 1], Vs[TB][HD + 1];
  int patch, t0;
  if (!locate(p.patch_start, num_patches, blockIdx.x, patch, t0)) return;
  const int head = blockIdx.y;
  const int ps = p.patch_start[patch], L = p.patch_start[patch + 1] - ps;
  const int lane = threadIdx.x;
  const int slot = t0 + lane;
  const bool valid = slot < L;
  float q[HD], g[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) { q[d] = 0.f; g[d] = 0.f; }
  long qrow = -1;
  if (valid) {
    qrow = p.q_gidx[ps + slot];
    const float* qp = p.q + qrow * p.ldq + head * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) q[d] = qp[d] * p.scale;
    const int w = p.widx[ps + slot];  // the slot's output row; padding duplicates have none: dO = 0
    if (w >= 0) {
      const float* gp = p.dout + (long)w * p.lddo + head * HD;
#pragma unroll
      for (int d = 0; d < HD; ++d) g[d] = gp[d];
    }
  }
  auto stage = [&](int k0) {
    __syncthreads();
    const int ks = k0 + lane;
    if (ks < L) {
      const long r = p.kv_gidx[ps + ks];
      const float* kp = p.k + r * p.ldk + head * HD;
      const float* vp = p.v + r * p.ldv + head * HD;
#pragma unroll
      for (int d = 0; d < HD; ++d) { Ks[lane][d] = kp[d]; Vs[lane][d] = vp[d]; }
    }
    __syncthreads();
  };
  // pass 1: row max
  float m = -INFINITY;
  for (int k0 = 0; k0 < L; k0 += TB) {
    stage(k0);
    const int nk = min(TB, L - k0);
    for (int j = 0; j < nk; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) s = fmaf(q[d], Ks[j][d], s);
      m = fmaxf(m, s);
    }
  }
  // pass 2: l = sum exp(s - m), Dn = sum exp(s - m) (dO . v)
  float l = 0.f, dn = 0.f;
  for (int k0 = 0; k0 < L; k0 += TB) {
    stage(k0);
    const int nk = min(TB, L - k0);
    for (int j = 0; j < nk; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) { s = fmaf(q[d], Ks[j][d], s); dp = fmaf(g[d], Vs[j][d], dp); }
      const float e = expf(s - m);
      l += e;
      dn = fmaf(e, dp, dn);
    }
  }
  const float inv_l = 1.0f / l, D = dn * inv_l;
  if (valid) {
    float* st = p.stats + ((long)(ps + slot) * p.num_heads + head) * 3;
    st[0] = m; st[1] = l; st[2] = D;
  }
  // pass 3: dQ_i = scale * sum_j P_ij (dP_ij - D_i) K_j
  float dq[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) dq[d] = 0.f;
  for (int k0 = 0; k0 < L; k0 += TB) {
    stage(k0);
    const int nk = min(TB, L - k0);
    for (int j = 0; j < nk; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) { s = fmaf(q[d], Ks[j][d], s); dp = fmaf(g[d], Vs[j][d], dp); }
      const float ds = expf(s - m) * inv_l * (dp - D);
#pragma unroll
      for (int d = 0; d < HD; ++d) dq[d] = fmaf(ds, Ks[j][d], dq[d]);
    }
  }
  if (valid) {
    float* o = p.dq + qrow * p.lddq + head * HD;  // a point padded into two slots collects both (the gather's backward)
#pragma unroll
    for (int d = 0; d < HD; ++d) atomicAdd(o + d, dq[d] * p.scale);
  }
}

// ---- pass over the KEYS of a tile: dK_j = scale * sum_i dS_ij Q_i, dV_j = sum_i P_ij dO_i
__global__ __launch_bounds__(TB) void attn_bwd_kv_kernel(AttnBwdP p, int num_patches) {
  __shared__ float Qs[TB][HD + 1], Gs[TB][HD + 1], Ss[TB][4];
  int patch, t0;
  if (!locate(p.patch_start, num_patches, blockIdx.x, patch, t0)) return;
  const int head = blockIdx.y;
  const int ps = p.patch_start[patch], L = p.patch_start[patch + 1] - ps;
  const int lane = threadIdx.x;
  const int slot = t0 + lane;
  const bool valid = slot < L;
  float k[HD], v[HD], dk[HD], dv[HD];
  long krow = -1;
#pragma unroll
  for (int d = 0; d < HD; ++d) { k[d] = v[d] = dk[d] = dv[d] = 0.f; }
  if (valid) {
    krow = p.kv_gidx[ps + slot];
    const float* kp = p.k + krow * p.ldk + head * HD;
    const float* vp = p.v + krow * p.ldv + head * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) { k[d] = kp[d]; v[d] = vp[d]; }
  }
  for (int q0 = 0; q0 < L; q0 += TB) {
    __syncthreads();
    const int qs = q0 + lane;
    if (qs < L) {
      const long r = p.q_gidx[ps + qs];
      const float* qp = p.q + r * p.ldq + head * HD;
      const int w = p.widx[ps + qs];
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        Qs[lane][d] = qp[d] * p.scale;
        Gs[lane][d] = w >= 0 ? p.dout[(long)w * p.lddo + head * HD + d] : 0.f;
      }
      const float* st = p.stats + ((long)(ps + qs) * p.num_heads + head) * 3;
      Ss[lane][0] = st[0]; Ss[lane][1] = 1.0f / st[1]; Ss[lane][2] = st[2];
    }
    __syncthreads();
    const int nq = min(TB, L - q0);
    for (int i = 0; i < nq; ++i) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) { s = fmaf(Qs[i][d], k[d], s); dp = fmaf(Gs[i][d], v[d], dp); }
      const float pij = expf(s - Ss[i][0]) * Ss[i][1];
      const float ds = pij * (dp - Ss[i][2]);
#pragma unroll
      for (int d = 0; d < HD; ++d) { dk[d] = fmaf(ds, Qs[i][d], dk[d]); dv[d] = fmaf(pij, Gs[i][d], dv[d]); }
    }
  }
  if (valid) {
    float* ok = p.dk + krow * p.lddk + head * HD;
    float* ov = p.dv + krow * p.lddv + head * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) { atomicAdd(ok + d, dk[d]); atomicAdd(ov + d, dv[d]); }  // Qs carries the scale already
  }
}

// ---- LayerNorm backward: dx = (1/sigma) (dyg - mean(dyg) - xhat mean(dyg xhat)), dyg = dy * gamma; one wave per row
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                            float eps, const float* __restrict__ dy, int lddy, float* dx, int lddx,
                                                            int accumulate, float* dgamma, float* dbeta, long m, int c) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
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
  for (int j = lane; j < c; j += 64) {
    const float xh = (xr[j] - mean) * rstd, dg = gr[j] * gamma[j];
    const float r = rstd * (dg - a - xh * b);
    float* o = dx + row * lddx + j;
    *o = accumulate ? *o + r : r;
    if (dgamma) atomicAdd(dgamma + j, gr[j] * xh);
    if (dbeta) atomicAdd(dbeta + j, gr[j]);
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
                                   const int32_t* patch_start, int num_patches, int num_heads, long num_slots, int num_tiles,
                                   float scale, const void* dout, int lddo, void* dq, void* dk, void* dv, int lddq, int lddk,
                                   int lddv, int dtype, void* ws, size_t ws_bytes, void* stream) {
  if (num_patches <= 0 || num_heads <= 0) return CDSEG_OK;
  if (dtype != CDSEG_F32) return CDSEG_ERR_UNSUPPORTED;  // first slice: the exact-fp32 mode
  if (!ws || ws_bytes < cdseg_attention_bwd_ws_bytes(num_slots, num_heads)) return CDSEG_ERR_WORKSPACE;
  AttnBwdP p;
  p.q = (const float*)q; p.k = (const float*)k; p.v = (const float*)v; p.dout = (const float*)dout;
  p.q_gidx = q_gidx; p.kv_gidx = kv_gidx; p.widx = widx; p.patch_start = patch_start;
  p.dq = (float*)dq; p.dk = (float*)dk; p.dv = (float*)dv; p.stats = (float*)ws;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
  p.num_heads = num_heads; p.scale = scale;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)num_tiles, (unsigned)num_heads);
  hipLaunchKernelGGL(attn_bwd_q_kernel, grid, dim3(TB), 0, s, p, num_patches);
  hipLaunchKernelGGL(attn_bwd_kv_kernel, grid, dim3(TB), 0, s, p, num_patches);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

extern "C" int cdseg_layernorm_bwd(const float* x, int ldx, const float* gamma, float eps, const float* dy, int lddy, float* dx,
                                   int lddx, int accumulate, float* dgamma, float* dbeta, long m, int c, void* stream) {
  if (m <= 0) return CDSEG_OK;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)cdiv(m, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, eps, dy,
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
