// Test-time pipeline around the model, HBM-bound index / scatter kernels:
//   GridSample(mode="test")   ref: pointcept/datasets/transform.py:821-897 (voxel hash, sort, per-voxel counts,
//                             fragment i takes member i % count of every voxel)
//   softmax vote              ref: pointcept/engines/test.py:261-267  (pred[idx_part] += softmax(logits))
//   arg-max                   ref: pointcept/engines/test.py:278
// The sort and the run segmentation reuse cdseg_sort_pairs / cdseg_pool_level (serialize.hip).
#include "common.h"

namespace {

// grid = floor(double(coord) / grid_size)   (the reference divides a float32 array by a float64 scalar)
__global__ void grid_floor_kernel(const float* __restrict__ coord, double grid_size, long n3,
                                  int32_t* __restrict__ grid) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n3) grid[i] = (int32_t)floor((double)coord[i] / grid_size);
}

__global__ void min3_kernel(const int32_t* __restrict__ grid, long n, int32_t* __restrict__ out3) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  int m0 = 0x7fffffff, m1 = 0x7fffffff, m2 = 0x7fffffff;
  for (; i < n; i += stride) {
    m0 = min(m0, grid[3 * i]);
    m1 = min(m1, grid[3 * i + 1]);
    m2 = min(m2, grid[3 * i + 2]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m0 = min(m0, __shfl_xor(m0, o, 64));
    m1 = min(m1, __shfl_xor(m1, o, 64));
    m2 = min(m2, __shfl_xor(m2, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&out3[0], m0);
    atomicMin(&out3[1], m1);
    atomicMin(&out3[2], m2);
  }
}

// grid -= min ; key = x << 42 | y << 21 | z  (one key per voxel; any injective key groups like the reference's hash)
__global__ void voxel_key_kernel(int32_t* __restrict__ grid, const int32_t* __restrict__ min3, long n,
                                 int64_t* __restrict__ key) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t x = grid[3 * i] - min3[0], y = grid[3 * i + 1] - min3[1], z = grid[3 * i + 2] - min3[2];
  grid[3 * i] = (int32_t)x;
  grid[3 * i + 1] = (int32_t)y;
  grid[3 * i + 2] = (int32_t)z;
  key[i] = (x << 42) | (y << 21) | z;
}

__global__ void max_run_kernel(const int32_t* __restrict__ seg_start, long m, int32_t* __restrict__ out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  int mx = 0;
  for (; i < m; i += stride) mx = max(mx, seg_start[i + 1] - seg_start[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, mx);
}

// fragment f: voxel v contributes its member (f % count_v)      ref: transform.py:862-864
__global__ void fragment_select_kernel(const int32_t* __restrict__ idx_sort, const int32_t* __restrict__ seg_start,
                                       long m, int frag, int32_t* __restrict__ idx_part) {
  const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= m) return;
  const int s = seg_start[v], c = seg_start[v + 1] - s;
  idx_part[v] = idx_sort[s + frag % c];
}

// pred[idx[i], :] += softmax(logits[i, :]) ; one wave per row (C up to a few hundred classes)
__global__ void softmax_vote_kernel(const float* __restrict__ logits, int ldl, const int32_t* __restrict__ idx, long m,
                                    int c, float* __restrict__ pred, int ldp) {
  const int lane = threadIdx.x & 63;
  const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= m) return;
  const float* l = logits + row * ldl;
  float mx = -INFINITY;
  for (int j = lane; j < c; j += 64) mx = fmaxf(mx, l[j]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int j = lane; j < c; j += 64) s += expf(l[j] - mx);
  s = wave_sum(s);
  const float inv = 1.0f / s;
  float* p = pred + (long)idx[row] * ldp;
  for (int j = lane; j < c; j += 64) p[j] += expf(l[j] - mx) * inv;
}

// first arg-max of every row (torch.max(1)[1] tie rule: lowest index)
__global__ void argmax_rows_kernel(const float* __restrict__ x, int ldx, long n, int c, int32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= n) return;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int j = lane; j < c; j += 64) {
    const float v = x[row * ldx + j];
    if (v > best) { best = v; bi = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) out[row] = bi;
}

inline dim3 g1(long n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

}  // namespace

extern "C" {

// grid (n,3) int32 = floor(coord / grid_size) - min ; key (n) int64 ; min3_dev (3) int32 scratch/output
int cdseg_voxelize(const float* coord, double grid_size, long n, int32_t* grid, int64_t* key, int32_t* min3_dev,
                   void* stream) {
  if (n <= 0) return CDSEG_OK;
  if (!(grid_size > 0)) return CDSEG_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(grid_floor_kernel, g1(3 * n), dim3(256), 0, s, coord, grid_size, 3 * n, grid);
  if (hipMemsetAsync(min3_dev, 0x7f, 3 * sizeof(int32_t), s) != hipSuccess) return CDSEG_ERR_LAUNCH;
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(min3_kernel, dim3((unsigned)blocks), dim3(256), 0, s, grid, n, min3_dev);
  hipLaunchKernelGGL(voxel_key_kernel, g1(n), dim3(256), 0, s, grid, min3_dev, n, key);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

int cdseg_max_run(const int32_t* seg_start, long m, int32_t* out_dev, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(out_dev, 0, sizeof(int32_t), s) != hipSuccess) return CDSEG_ERR_LAUNCH;
  if (m <= 0) return CDSEG_OK;
  long blocks = (m + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(max_run_kernel, dim3((unsigned)blocks), dim3(256), 0, s, seg_start, m, out_dev);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

int cdseg_fragment_select(const int32_t* idx_sort, const int32_t* seg_start, long m, int frag, int32_t* idx_part,
                          void* stream) {
  if (m <= 0) return CDSEG_OK;
  if (frag < 0) return CDSEG_ERR_ARG;
  hipLaunchKernelGGL(fragment_select_kernel, g1(m), dim3(256), 0, (hipStream_t)stream, idx_sort, seg_start, m, frag,
                     idx_part);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

int cdseg_softmax_vote(const float* logits, int ldl, const int32_t* idx, long m, int c, float* pred, int ldp,
                       void* stream) {
  if (m <= 0 || c <= 0) return CDSEG_OK;
  hipLaunchKernelGGL(softmax_vote_kernel, g1(m * 64), dim3(256), 0, (hipStream_t)stream, logits, ldl, idx, m, c, pred,
                     ldp);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

int cdseg_argmax_rows(const float* x, int ldx, long n, int c, int32_t* out, void* stream) {
  if (n <= 0 || c <= 0) return CDSEG_OK;
  hipLaunchKernelGGL(argmax_rows_kernel, g1(n * 64), dim3(256), 0, (hipStream_t)stream, x, ldx, n, c, out);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

}  // extern "C"
