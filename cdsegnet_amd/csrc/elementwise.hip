// HBM-bound row kernels of the hot path: LayerNorm (+residual, +t-embedding bias), pooling
// segment max (+folded BN, +GELU), stem convolution (Cin 4..6, VALU), GEMV for the timestep
// MLP, Philox normal draws, casts.  16-byte vector loads/stores per lane, one pass over HBM.
#include "common.h"

namespace {

__device__ __forceinline__ float4 load4(const void* p, int dtype, long idx /* element index, %4==0 */) {
  if (dtype == CDSEG_F32) return *reinterpret_cast<const float4*>((const float*)p + idx);
  const uint2 u = *reinterpret_cast<const uint2*>((const bf16_t*)p + idx);
  float4 r;
  unpack_bf16x2(u.x, r.x, r.y);
  unpack_bf16x2(u.y, r.z, r.w);
  return r;
}

__device__ __forceinline__ void store4(void* p, int dtype, long idx, float4 v) {
  if (dtype == CDSEG_F32) {
    *reinterpret_cast<float4*>((float*)p + idx) = v;
  } else {
    uint2 u;
    u.x = pack_bf16x2(v.x, v.y);
    u.y = pack_bf16x2(v.z, v.w);
    *reinterpret_cast<uint2*>((bf16_t*)p + idx) = u;
  }
}

// ---------------------------------------------------------------- LayerNorm
// TPR lanes cooperate on one row (TPR = power of two <= 64); a wave holds 64/TPR rows.
// Each lane keeps its <= MAXV float4 chunks in registers: one HBM read, two-pass statistics.
template <int MAXV>
__global__ void layernorm_kernel(const void* __restrict__ x, int x_dtype, int ldx, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float eps, const float* __restrict__ res, int ldres,
                                 const float* __restrict__ colbias, void* __restrict__ out, int out_dtype, int ldo,
                                 void* __restrict__ out2, int out2_dtype, int ldo2, long m, int c, int tpr) {
  const int lane = threadIdx.x & 63;
  const int wave = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int rpw = 64 / tpr;
  const long row = (long)wave * rpw + lane / tpr;
  const int sub = lane % tpr;
  const int nchunk = c >> 2;
  const bool active = row < m;
  float4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int ch = sub + i * tpr;
    if (active && ch < nchunk) {
      v[i] = load4(x, x_dtype, row * ldx + 4 * ch);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    } else {
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  for (int o = tpr >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)c;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int ch = sub + i * tpr;
    if (ch < nchunk) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  for (int o = tpr >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.0f / sqrtf(q / (float)c + eps);
  if (!active) return;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int ch = sub + i * tpr;
    if (ch < nchunk) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * ch);
      const float4 b = *reinterpret_cast<const float4*>(beta + 4 * ch);
      float4 y;
      y.x = (v[i].x - mean) * rstd * g.x + b.x;
      y.y = (v[i].y - mean) * rstd * g.y + b.y;
      y.z = (v[i].z - mean) * rstd * g.z + b.z;
      y.w = (v[i].w - mean) * rstd * g.w + b.w;
      if (res) {
        const float4 r = *reinterpret_cast<const float4*>(res + row * ldres + 4 * ch);
        y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
      }
      if (colbias) {
        const float4 t = *reinterpret_cast<const float4*>(colbias + 4 * ch);
        y.x += t.x; y.y += t.y; y.z += t.z; y.w += t.w;
      }
      store4(out, out_dtype, row * ldo + 4 * ch, y);
      if (out2) store4(out2, out2_dtype, row * ldo2 + 4 * ch, y);
    }
  }
}

// ---------------------------------------------------------------- segment max / mean
__global__ void segment_max_kernel(const void* __restrict__ y, int y_dtype, int ldy,
                                   const int32_t* __restrict__ seg_start, long m, int c,
                                   const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                   float* __restrict__ out, int ldo, void* __restrict__ out2, int out2_dtype,
                                   int ldo2) {
  const int nchunk = c >> 2;
  const long total = m * nchunk;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long j = t / nchunk;
    const int ch = (int)(t - j * nchunk);
    const int s = seg_start[j], e = seg_start[j + 1];
    float4 mx = load4(y, y_dtype, (long)s * ldy + 4 * ch);
    for (int i = s + 1; i < e; ++i) {
      const float4 v = load4(y, y_dtype, (long)i * ldy + 4 * ch);
      mx.x = fmaxf(mx.x, v.x); mx.y = fmaxf(mx.y, v.y); mx.z = fmaxf(mx.z, v.z); mx.w = fmaxf(mx.w, v.w);
    }
    if (scale) {
      const float4 sc = *reinterpret_cast<const float4*>(scale + 4 * ch);
      const float4 sh = *reinterpret_cast<const float4*>(shift + 4 * ch);
      // (explicit fused multiply-adds: cdseg_pool_fused evaluates the same expression and must agree bit for bit)
      mx.x = __builtin_fmaf(mx.x, sc.x, sh.x); mx.y = __builtin_fmaf(mx.y, sc.y, sh.y);
      mx.z = __builtin_fmaf(mx.z, sc.z, sh.z); mx.w = __builtin_fmaf(mx.w, sc.w, sh.w);
    }
    if (act == CDSEG_ACT_GELU) {
      mx.x = gelu_erf(mx.x); mx.y = gelu_erf(mx.y); mx.z = gelu_erf(mx.z); mx.w = gelu_erf(mx.w);
    }
    *reinterpret_cast<float4*>(out + j * ldo + 4 * ch) = mx;
    if (out2) store4(out2, out2_dtype, j * ldo2 + 4 * ch, mx);
  }
}

__global__ void segment_mean_kernel(const float* __restrict__ x, int ldx, const int32_t* __restrict__ seg_start,
                                    long m, int c, float* __restrict__ out, int ldo) {
  const long total = m * c;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long j = t / c;
    const int col = (int)(t - j * c);
    const int s = seg_start[j], e = seg_start[j + 1];
    float acc = 0.f;
    for (int i = s; i < e; ++i) acc += x[(long)i * ldx + col];
    out[j * ldo + col] = acc / (float)(e - s);
  }
}

// ---------------------------------------------------------------- GEMV (timestep MLP): one wave per output
__global__ void gemv_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ x,
                            int n, int k, int act, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int row = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (row >= n) return;
  float s = 0.f;
  for (int i = lane; i < k; i += 64) s = fmaf(w[(long)row * k + i], x[i], s);
  s = wave_sum(s);
  if (lane == 0) {
    if (b) s += b[row];
    if (act == CDSEG_ACT_SWISH) s = s / (1.0f + expf(-s));
    else if (act == CDSEG_ACT_GELU) s = gelu_erf(s);
    y[row] = s;
  }
}

// ---------------------------------------------------------------- Philox4x32-10 + Box-Muller
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ void randn_kernel(float* __restrict__ out, long n, uint64_t seed, uint64_t offset) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (4 * t >= n) return;
  uint32_t c[4] = {(uint32_t)t, (uint32_t)(t >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  float z[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)(c[2 * h] >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
    const float u2 = (float)(c[2 * h + 1] >> 8) * (1.0f / 16777216.0f);        // [0, 1)
    const float r = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.28318530717958647692f * u2, &sn, &cs);
    z[2 * h] = r * cs;
    z[2 * h + 1] = r * sn;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (4 * t + i < n) out[4 * t + i] = z[i];
}

__global__ void cast_kernel(const void* __restrict__ src, int sd, void* __restrict__ dst, int dd, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = sd == CDSEG_F32 ? ((const float*)src)[i] : bf16_to_f32(((const bf16_t*)src)[i]);
  if (dd == CDSEG_F32) ((float*)dst)[i] = v;
  else ((bf16_t*)dst)[i] = f32_to_bf16(v);
}

// dst[i, 0:cpad] = cast(src[idx[i], 0:cin]) zero-padded to cpad columns (stem input: 6 -> 8 channels)
__global__ void gather_pad_cast_kernel(const float* __restrict__ src, int ld_src, const int32_t* __restrict__ idx,
                                       long n, int cin, int cpad, void* __restrict__ dst, int dd) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * cpad) return;
  const long r = t / cpad;
  const int c = (int)(t - r * cpad);
  const long sr = idx ? (long)idx[r] : r;
  const float v = c < cin ? src[sr * ld_src + c] : 0.f;
  if (dd == CDSEG_F32) ((float*)dst)[t] = v;
  else ((bf16_t*)dst)[t] = f32_to_bf16(v);
}

// DDIM step on a uniform timestep (ref: default.py:192-214), same operation order as the reference
__global__ void ddim_update_kernel(const float* __restrict__ xt, const float* __restrict__ eps, float s_ab1,
                                   float s_1ab, float s_ab, float s_1ab1, int final_step, float* __restrict__ out,
                                   long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float e = eps[i];
  const float x0 = (xt[i] - s_1ab * e) / s_ab;
  out[i] = final_step ? x0 : (s_ab1 * x0 + s_1ab1 * e);
}

__global__ void axpy_kernel(const float* __restrict__ a, const float* __restrict__ b, float alpha,
                            float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + alpha * b[i];
}

}  // namespace

__global__ __launch_bounds__(256) void split16_kernel(const float* __restrict__ x, int ldx, long rows, int cols, bf16_t* hi,
                                                      bf16_t* lo, int ld16, float lo_scale) {
  const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one group of 4 columns
  const int per_row = cols >> 2;
  if (u >= rows * per_row) return;
  const long m = u / per_row;
  const int c = (int)(u % per_row) * 4;
  const float4 v = *reinterpret_cast<const float4*>(x + m * ldx + c);
  const float f[4] = {v.x, v.y, v.z, v.w};
  uint32_t h[2], l[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    h[e] = pack_bf16x2(f[2 * e], f[2 * e + 1]);  // (saturating in the half build)
    float a, b;
    unpack_bf16x2(h[e], a, b);
    l[e] = pack_bf16x2((f[2 * e] - a) * lo_scale, (f[2 * e + 1] - b) * lo_scale);
  }
  *reinterpret_cast<uint2*>(hi + m * ld16 + c) = make_uint2(h[0], h[1]);
  *reinterpret_cast<uint2*>(lo + m * ld16 + c) = make_uint2(l[0], l[1]);
}

// Diagnostic of the IEEE-half build: how many elements of a 16-bit activation buffer sit exactly at +-65504, the value every
// float -> half conversion of that build clamps to (common.h).  A trained checkpoint whose activations leave half's range
// shows up as a non-zero count instead of silently clamped logits.  (The bfloat16 build never clamps: counts nothing.)
__global__ void count_saturated_kernel(const uint16_t* __restrict__ x, long rows, int cols, int ld,
                                       unsigned long long* __restrict__ counter) {
  unsigned local = 0;
  if (LP_IS_F16) {
    const long total = rows * (long)cols;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
      const long r = t / cols;
      const int c = (int)(t - r * cols);
      local += (x[r * ld + c] & 0x7FFFu) == 0x7BFFu;
    }
  }
  const unsigned long long m = __builtin_amdgcn_ballot_w64(local != 0);
  if (m) {  // rare: one atomic per wave that saw any
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(counter, (unsigned long long)local);
  }
}

extern "C" {

int cdseg_layernorm(const void* x, int x_dtype, int ldx, const float* gamma, const float* beta, float eps,
                    const float* res, int ldres, const float* colbias, void* out, int out_dtype, int ldo,
                    void* out2, int out2_dtype, int ldo2, long m, int c, void* stream) {
  if (m <= 0) return CDSEG_OK;
  if (c <= 0 || (c & 3) || c > 2048 || (ldx & 3) || (ldo & 3) || (res && (ldres & 3)) || (out2 && (ldo2 & 3)))
    return CDSEG_ERR_ARG;
  const int nchunk = c >> 2;
  int tpr = 1;
  while (tpr < nchunk && tpr < 64) tpr <<= 1;
  const int maxv = (nchunk + tpr - 1) / tpr;  // chunks per lane (<= 8)
  const int rpw = 64 / tpr;
  const long waves = (m + rpw - 1) / rpw;
  const int wpb = 4;
  dim3 grid((unsigned)((waves + wpb - 1) / wpb)), block(64 * wpb);
  hipStream_t s = (hipStream_t)stream;
#define LN_LAUNCH(MV)                                                                                             \
  hipLaunchKernelGGL(layernorm_kernel<MV>, grid, block, 0, s, x, x_dtype, ldx, gamma, beta, eps, res, ldres, colbias, \
                     out, out_dtype, ldo, out2, out2_dtype, ldo2, m, c, tpr)
  if (maxv <= 1) LN_LAUNCH(1);
  else if (maxv <= 2) LN_LAUNCH(2);
  else if (maxv <= 4) LN_LAUNCH(4);
  else LN_LAUNCH(8);
#undef LN_LAUNCH
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

int cdseg_segment_max(const void* y, int y_dtype, int ldy, const int32_t* seg_start, long m, int c,
                      const float* scale, const float* shift, int act, float* out, int ldo, void* out2,
                      int out2_dtype, int ldo2, void* stream) {
  if (m <= 0) return CDSEG_OK;
  if (c <= 0 || (c & 3) || (ldy & 3) || (ldo & 3) || (out2 && (ldo2 & 3)) || (scale && !shift)) return CDSEG_ERR_ARG;
  const long total = m * (c >> 2);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(segment_max_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, y_dtype, ldy,
                     seg_start, m, c, scale, shift, act, out, ldo, out2, out2_dtype, ldo2);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

int cdseg_segment_mean(const float* x, int ldx, const int32_t* seg_start, long m, int c, float* out, int ldo,
                       void* stream) {
  if (m <= 0) return CDSEG_OK;
  const long total = m * c;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(segment_mean_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     seg_start, m, c, out, ldo);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

int cdseg_gemv(const float* w, const float* b, const float* x, int n, int k, int act, float* y, void* stream) {
  if (n <= 0) return CDSEG_OK;
  hipLaunchKernelGGL(gemv_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, w, b, x, n, k, act,
                     y);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

int cdseg_randn(float* out, long n, uint64_t seed, uint64_t offset, void* stream) {
  if (n <= 0) return CDSEG_OK;
  const long threads = (n + 3) / 4;
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n,
                     seed, offset);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

int cdseg_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, void* stream) {
  if (n <= 0) return CDSEG_OK;
  hipLaunchKernelGGL(cast_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, src_dtype,
                     dst, dst_dtype, n);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

int cdseg_gather_pad_cast(const float* src, int ld_src, const int32_t* idx, long n, int cin, int cpad, void* dst,
                          int dst_dtype, void* stream) {
  if (n <= 0) return CDSEG_OK;
  if (cin <= 0 || cpad < cin) return CDSEG_ERR_ARG;
  const long total = n * cpad;
  hipLaunchKernelGGL(gather_pad_cast_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     src, ld_src, idx, n, cin, cpad, dst, dst_dtype);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

int cdseg_ddim_update(const float* xt, const float* eps, float sqrt_ab_prev, float sqrt_1m_ab, float sqrt_ab,
                      float sqrt_1m_ab_prev, int final_step, float* out, long n, void* stream) {
  if (n <= 0) return CDSEG_OK;
  hipLaunchKernelGGL(ddim_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xt, eps,
                     sqrt_ab_prev, sqrt_1m_ab, sqrt_ab, sqrt_1m_ab_prev, final_step, out, n);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

int cdseg_count_saturated(const void* x, long rows, int cols, int ld, unsigned long long* counter, void* stream) {
  if (rows <= 0 || cols <= 0) return CDSEG_OK;
  if (!x || !counter || ld < cols) return CDSEG_ERR_ARG;
  const long total = rows * (long)cols;
  const unsigned blocks = (unsigned)((total + 256 * 8 - 1) / (256 * 8));
  hipLaunchKernelGGL(count_saturated_kernel, dim3(blocks < 2048 ? blocks : 2048), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)x, rows, cols, ld, counter);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

// fp32 matrix -> a pair of matrices in the build's 16-bit type: hi = rn(x) (saturating), lo = rn((x - hi) * lo_scale).  With
// lo_scale = 2048 in the IEEE-half build x ~= hi + lo / 2048 to 22 significant bits - the operand form of the "fp32 x3"
// sparse convs, which run the 16-bit gathered GEMM three times into one fp32 output (engine.py, precision fp32x3)
int cdseg_split16(const float* x, int ldx, long rows, int cols, void* hi, void* lo, int ld16, float lo_scale, void* stream) {
  if (rows <= 0 || cols <= 0) return CDSEG_OK;
  if (!x || !hi || !lo || (cols & 3) || (ldx & 3) || (ld16 & 3) || ldx < cols || ld16 < cols) return CDSEG_ERR_ARG;
  if ((((uintptr_t)x) & 15) || (((uintptr_t)hi | (uintptr_t)lo) & 7)) return CDSEG_ERR_ARG;
  const long groups = rows * (long)(cols >> 2);
  hipLaunchKernelGGL(split16_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, rows,
                     cols, (bf16_t*)hi, (bf16_t*)lo, ld16, lo_scale);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

int cdseg_axpy(const float* a, const float* b, float alpha, float* out, long n, void* stream) {
  if (n <= 0) return CDSEG_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, alpha,
                     out, n);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

}  // extern "C"
