// Test-time pipeline around the model, HBM-bound index / scatter kernels:
//   GridSample(mode="test")   ref: pointcept/datasets/transform.py:821-897 (voxel hash, sort, per-voxel counts,
//                             fragment i takes member i % count of every voxel)
//   softmax vote              ref: pointcept/engines/test.py:261-267  (pred[idx_part] += softmax(logits))
//   arg-max                   ref: pointcept/engines/test.py:278
// The sort and the run segmentation reuse cdseg_sort_pairs / cdseg_pool_level (serialize.hip).
#include "common.h"

namespace {

// grid = floor(double(coord) / grid_size)   (the reference divides a float32 - or, after a test-time rotation,
// float64 - array by a float64 scalar)
template <typename T>
__global__ void grid_floor_kernel(const T* __restrict__ coord, double grid_size, long n3, int32_t* __restrict__ grid) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n3) grid[i] = (int32_t)floor((double)coord[i] / grid_size);
}

// ---- pre-model transforms of the test pipeline (ref: datasets/transform.py:113-117 NormalizeColor, :142-155 CenterShift,
// :259-294 RandomRotateTargetAngle, :298-309 RandomScale, :313-328 RandomFlip; configs/scannet/CDSegNet.py:253-398)
// per-axis min / max of an (n,3) array -> out6 = [min x y z, max x y z] as doubles (exact for both input types).
// Doubles are compared through an order-preserving integer image so that one atomicMin / atomicMax pair suffices.
__device__ __forceinline__ unsigned long long ord_u64(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double ord_f64(unsigned long long u) {
  return __longlong_as_double((long long)((u >> 63) ? (u & 0x7fffffffffffffffull) : ~u));
}
template <typename T>
__global__ void minmax3_kernel(const T* __restrict__ xyz, long n, unsigned long long* __restrict__ acc6) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (; i < n; i += stride)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double v = (double)xyz[3 * i + a];
      lo[a] = fmin(lo[a], v);
      hi[a] = fmax(hi[a], v);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo[a] = fmin(lo[a], __shfl_xor(lo[a], o, 64));
      hi[a] = fmax(hi[a], __shfl_xor(hi[a], o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&acc6[a], ord_u64(lo[a]));
      atomicMax(&acc6[3 + a], ord_u64(hi[a]));
    }
  }
}
__global__ void minmax3_finish_kernel(const unsigned long long* __restrict__ acc6, double* __restrict__ out6) {
  if (threadIdx.x < 6) out6[threadIdx.x] = ord_f64(acc6[threadIdx.x]);
}
// CenterShift: coord -= [(xmin + xmax) / 2, (ymin + ymax) / 2, apply_z ? zmin : 0], in the array's own precision
template <typename T>
__global__ void center_shift_kernel(const T* __restrict__ in, const double* __restrict__ mm6, int apply_z, long n,
                                    T* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const T sx = ((T)mm6[0] + (T)mm6[3]) / (T)2, sy = ((T)mm6[1] + (T)mm6[4]) / (T)2, sz = apply_z ? (T)mm6[2] : (T)0;
  out[3 * i] = in[3 * i] - sx;
  out[3 * i + 1] = in[3 * i + 1] - sy;
  out[3 * i + 2] = in[3 * i + 2] - sz;
}
// one test-time augmentation as the reference applies it: rotate about the origin (float32 rows times a float64
// matrix: the result IS float64), then scale; or flip x and y in place (stays float32).  R row-major.
struct TtaP {
  double r[9];
  double scale;
  int rotate, flip;
};
template <typename TO>
__global__ void tta_kernel(const float* __restrict__ in, TtaP p, int apply_scale, long n, TO* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
  if (p.rotate) {
    // np.dot(coord, rot_t.T): out_j = sum_k in_k R[j][k], accumulated left to right with fused multiply-adds.
    // Bit-identical to numpy's BLAS dgemm for the reference's test-time angles (multiples of pi / 2: every product is
    // exact or a 6e-17 cross term) - for an arbitrary angle the last bit may differ from the BLAS summation order, and a
    // coordinate that sits within 1 ulp of a voxel border can then land in the neighbouring voxel of GridSample.
    double o[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) o[j] = fma((double)z, p.r[3 * j + 2], fma((double)y, p.r[3 * j + 1], (double)x * p.r[3 * j]));
    if (apply_scale) { o[0] *= p.scale; o[1] *= p.scale; o[2] *= p.scale; }
    out[3 * i] = (TO)o[0]; out[3 * i + 1] = (TO)o[1]; out[3 * i + 2] = (TO)o[2];
  } else {
    out[3 * i] = (TO)(p.flip ? -x : x);
    out[3 * i + 1] = (TO)(p.flip ? -y : y);
    out[3 * i + 2] = (TO)z;
  }
}
// out = in / div + add in float32 (NormalizeColor: color / 127.5 - 1)
__global__ void div_add_kernel(const float* __restrict__ in, float div, float add, long n, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] / div + add;
}
// feat (n, ca + cb) float32 = cat([a.float(), b.float()], 1)  (Collect(feat_keys=...), transform.py:46-49); b may be f64
template <typename TB>
__global__ void collect_feat_kernel(const float* __restrict__ a, int ca, const TB* __restrict__ b, int cb, long n,
                                    float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = ca + cb;
  if (t >= n * c) return;
  const long i = t / c;
  const int j = (int)(t - i * c);
  out[t] = j < ca ? a[i * ca + j] : (float)b[i * cb + (j - ca)];
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

// ---- evaluator: exact 1-nearest-neighbour label transfer + IoU counters
// ref: engines/hooks/evaluator.py:132-140 -> pointops.knn_query(1, ...), a brute-force O(m n) scan per query
// (libs/pointops/src/knn_query/knn_query_cuda_kernel.cu:60-104: strict '<', ascending index => lowest index on ties).
// Here: uniform grid over the reference points (cell keys sorted with the library's radix sort), every query walks
// the cube shells around its own cell and stops as soon as no unvisited cell can hold a closer (or equal, lower-index)
// point; exact, with a brute-force fallback for queries far from every reference point.
struct KnnP {
  const float* ref;          // (n,3)
  const float* qry;          // (m,3)
  const int32_t* ref_off;    // (B) cumulative ends
  const int32_t* qry_off;    // (B)
  const int64_t* key_sorted; // (n) sorted cell keys
  const int32_t* perm;       // (n) sorted position -> reference point
  float ox, oy, oz, cell;    // grid origin (<= every reference coordinate) and cell size
  int nb, gmax;              // batches; cells per axis - 1 (keys are clamped to [0, gmax])
  long n, m;
};

__device__ __forceinline__ int batch_of(const int32_t* off, int nb, long i) {
  int lo = 0, hi = nb - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] > i) hi = mid; else lo = mid + 1;
  }
  return lo;
}

__device__ __forceinline__ int64_t cell_key(int b, int cx, int cy, int cz) {
  return ((int64_t)b << 60) | ((int64_t)cx << 40) | ((int64_t)cy << 20) | (int64_t)cz;
}

__global__ void knn_cell_key_kernel(KnnP p, int64_t* __restrict__ key) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  const int b = batch_of(p.ref_off, p.nb, i);
  const int cx = min(p.gmax, max(0, (int)floorf((p.ref[3 * i] - p.ox) / p.cell)));
  const int cy = min(p.gmax, max(0, (int)floorf((p.ref[3 * i + 1] - p.oy) / p.cell)));
  const int cz = min(p.gmax, max(0, (int)floorf((p.ref[3 * i + 2] - p.oz) / p.cell)));
  key[i] = cell_key(b, cx, cy, cz);
}

__device__ __forceinline__ void knn_consider(const KnnP& p, int j, float qx, float qy, float qz, float& best,
                                             int& best_i) {
  const float dx = qx - p.ref[3 * j], dy = qy - p.ref[3 * j + 1], dz = qz - p.ref[3 * j + 2];
  const float d2 = dx * dx + dy * dy + dz * dz;
  if (d2 < best || (d2 == best && j < best_i)) { best = d2; best_i = j; }
}

__global__ void knn1_kernel(KnnP p, int max_shell, int32_t* __restrict__ idx_out, float* __restrict__ d2_out) {
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= p.m) return;
  const int b = batch_of(p.qry_off, p.nb, q);
  const int rs = b ? p.ref_off[b - 1] : 0, re = p.ref_off[b];
  const float qx = p.qry[3 * q], qy = p.qry[3 * q + 1], qz = p.qry[3 * q + 2];
  float best = 1e10f;  // the reference's initial distance
  int best_i = -1;
  if (re > rs) {
    // the query's own cell (unclamped: a query may lie outside the reference bounding box)
    const float fx = (qx - p.ox) / p.cell, fy = (qy - p.oy) / p.cell, fz = (qz - p.oz) / p.cell;
    const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
    // distance from the query to the nearest face of its own cell: every point outside the searched cube of
    // radius r cells is farther than r * cell + edge
    float edge = fminf(fminf(fx - cx, cx + 1 - fx), fminf(fminf(fy - cy, cy + 1 - fy), fminf(fz - cz, cz + 1 - fz)));
    edge = fmaxf(edge, 0.f) * p.cell * 0.999f;  // slack for the rounding of the divisions above
    bool done = false;
    for (int r = 0; r <= max_shell && !done; ++r) {
      for (int dx = -r; dx <= r; ++dx) {
        const int x = cx + dx;
        if (x < 0 || x > p.gmax) continue;
        for (int dy = -r; dy <= r; ++dy) {
          const int y = cy + dy;
          if (y < 0 || y > p.gmax) continue;
          const bool face = dx == -r || dx == r || dy == -r || dy == r;
          // inside the (dx, dy) column only the two end caps belong to shell r, a face column belongs to it whole
          const int zstep = face ? 1 : (r == 0 ? 1 : 2 * r);
          for (int dz = -r; dz <= r; dz += zstep) {
            const int z = cz + dz;
            if (z < 0 || z > p.gmax) continue;
            const int64_t key = cell_key(b, x, y, z);
            long lo = rs, hi = re;  // keys of batch b occupy [rs, re) of the sorted array (batch is the top field)
            while (lo < hi) {
              const long mid = (lo + hi) >> 1;
              if (p.key_sorted[mid] < key) lo = mid + 1; else hi = mid;
            }
            for (; lo < re && p.key_sorted[lo] == key; ++lo) knn_consider(p, p.perm[lo], qx, qy, qz, best, best_i);
          }
        }
      }
      const float lb = r * p.cell * 0.99999f + edge;
      done = best_i >= 0 && best < lb * lb;  // strict: an equal-distance, lower-index point may still be outside
    }
    if (!done) {  // far from every reference point: the reference's brute-force scan
      best = 1e10f;
      best_i = -1;
      for (int j = rs; j < re; ++j) knn_consider(p, j, qx, qy, qz, best, best_i);
    }
  }
  idx_out[q] = best_i;
  if (d2_out) d2_out[q] = best;
}

// ---- k nearest neighbours, k <= CDSEG_KNN_MAX_K (round 5)
// ref: libs/pointops/functions/query.py:7-24 (KNNQuery: idx (m, nsample) int32, -1 = placeholder; distances = sqrt(dist2)),
//      libs/pointops/src/knn_query/knn_query_cuda_kernel.cu:60-104 - one thread per query, a brute-force scan of the batch
//      element's reference points into a max-heap of nsample entries (strict '<': among equal distances the lower index
//      stays), heap-sorted to ascending distance at the end.
// Same uniform grid and shell walk as knn1_kernel; the candidate list is kept sorted by (distance, index) - insertion into
// at most k entries of private memory - and the walk stops once the k-th best is closer than anything an unvisited cell can
// hold.  Results equal the reference's as sets and in their distances; WITHIN a group of exactly equal distances the
// reference's order is its heap's (not index order), this one's is ascending index.
constexpr int KNN_MAX_K = 64;

__device__ __forceinline__ void knnk_consider(const KnnP& p, int j, float qx, float qy, float qz, int k, int& cnt,
                                              float* bd, int* bi) {
  const float dx = qx - p.ref[3 * j], dy = qy - p.ref[3 * j + 1], dz = qz - p.ref[3 * j + 2];
  const float d2 = dx * dx + dy * dy + dz * dz;
  if (cnt == k && !(d2 < bd[k - 1] || (d2 == bd[k - 1] && j < bi[k - 1]))) return;
  int pos = cnt < k ? cnt : k - 1;  // slot that opens up (the last one is dropped when the list is full)
  while (pos > 0 && (bd[pos - 1] > d2 || (bd[pos - 1] == d2 && bi[pos - 1] > j))) {
    bd[pos] = bd[pos - 1];
    bi[pos] = bi[pos - 1];
    --pos;
  }
  bd[pos] = d2;
  bi[pos] = j;
  if (cnt < k) ++cnt;
}

__global__ void knnk_kernel(KnnP p, int k, int max_shell, int32_t* __restrict__ idx_out, float* __restrict__ d2_out) {
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= p.m) return;
  const int b = batch_of(p.qry_off, p.nb, q);
  const int rs = b ? p.ref_off[b - 1] : 0, re = p.ref_off[b];
  const float qx = p.qry[3 * q], qy = p.qry[3 * q + 1], qz = p.qry[3 * q + 2];
  float bd[KNN_MAX_K];
  int bi[KNN_MAX_K];
  int cnt = 0;
  const int avail = min(k, re - rs);  // a batch element with fewer than k points fills the rest with placeholders
  if (avail > 0) {
    const float fx = (qx - p.ox) / p.cell, fy = (qy - p.oy) / p.cell, fz = (qz - p.oz) / p.cell;
    const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
    float edge = fminf(fminf(fx - cx, cx + 1 - fx), fminf(fminf(fy - cy, cy + 1 - fy), fminf(fz - cz, cz + 1 - fz)));
    edge = fmaxf(edge, 0.f) * p.cell * 0.999f;
    bool done = false;
    for (int r = 0; r <= max_shell && !done; ++r) {
      for (int dx = -r; dx <= r; ++dx) {
        const int x = cx + dx;
        if (x < 0 || x > p.gmax) continue;
        for (int dy = -r; dy <= r; ++dy) {
          const int y = cy + dy;
          if (y < 0 || y > p.gmax) continue;
          const bool face = dx == -r || dx == r || dy == -r || dy == r;
          const int zstep = face ? 1 : (r == 0 ? 1 : 2 * r);
          for (int dz = -r; dz <= r; dz += zstep) {
            const int z = cz + dz;
            if (z < 0 || z > p.gmax) continue;
            const int64_t key = cell_key(b, x, y, z);
            long lo = rs, hi = re;
            while (lo < hi) {
              const long mid = (lo + hi) >> 1;
              if (p.key_sorted[mid] < key) lo = mid + 1; else hi = mid;
            }
            for (; lo < re && p.key_sorted[lo] == key; ++lo) knnk_consider(p, p.perm[lo], qx, qy, qz, avail, cnt, bd, bi);
          }
        }
      }
      const float lb = r * p.cell * 0.99999f + edge;
      done = cnt == avail && bd[avail - 1] < lb * lb;  // strict: an equal-distance, lower-index point may still be outside
    }
    if (!done) {  // sparse neighbourhood: the reference's brute-force scan
      cnt = 0;
      for (int j = rs; j < re; ++j) knnk_consider(p, j, qx, qy, qz, avail, cnt, bd, bi);
    }
  }
  for (int i = 0; i < k; ++i) {
    idx_out[q * k + i] = i < cnt ? bi[i] : -1;
    if (d2_out) d2_out[q * k + i] = i < cnt ? bd[i] : 1e10f;  // the reference's initial heap entries
  }
}

// per-class intersection / prediction / target counts; ignore_index rows are dropped.  ref: utils/misc.py:52-65
// Counters are first accumulated per workgroup in LDS (3 k <= 768 bins: 32-bit LDS atomics), then added to the global
// (3, k) table with ONE 64-bit atomic per non-empty bin and workgroup: with one global atomic per point the 137k points of
// a scene queued on ~40 addresses (1.0 ms for a scene, lanes-1 trace of round 3).  k > 256: the direct form.
constexpr int IOU_MAX_K = 256;
__global__ __launch_bounds__(256) void iou_counts_kernel(const int32_t* __restrict__ pred, const int32_t* __restrict__ pred_idx,
                                                         const int32_t* __restrict__ target, long n, int k, int ignore,
                                                         unsigned long long* __restrict__ out /* (3,k): inter, pred, target */) {
  __shared__ unsigned int bins[3 * IOU_MAX_K];
  const bool local = k <= IOU_MAX_K;
  if (local) {
    for (int b = threadIdx.x; b < 3 * k; b += blockDim.x) bins[b] = 0u;
    __syncthreads();
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int t = target[i];
    if (t == ignore) continue;
    const int pr = pred_idx ? pred[pred_idx[i]] : pred[i];
    if (local) {
      if (pr >= 0 && pr < k) atomicAdd(&bins[k + pr], 1u);
      if (t >= 0 && t < k) {
        atomicAdd(&bins[2 * k + t], 1u);
        if (pr == t) atomicAdd(&bins[t], 1u);
      }
    } else {
      if (pr >= 0 && pr < k) atomicAdd(&out[k + pr], 1ull);
      if (t >= 0 && t < k) {
        atomicAdd(&out[2 * k + t], 1ull);
        if (pr == t) atomicAdd(&out[t], 1ull);
      }
    }
  }
  if (local) {
    __syncthreads();
    for (int b = threadIdx.x; b < 3 * k; b += blockDim.x)
      if (bins[b]) atomicAdd(&out[b], (unsigned long long)bins[b]);
  }
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
  hipLaunchKernelGGL(grid_floor_kernel<float>, g1(3 * n), dim3(256), 0, s, coord, grid_size, 3 * n, grid);
  if (hipMemsetAsync(min3_dev, 0x7f, 3 * sizeof(int32_t), s) != hipSuccess) return CDSEG_ERR_LAUNCH;
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(min3_kernel, dim3((unsigned)blocks), dim3(256), 0, s, grid, n, min3_dev);
  hipLaunchKernelGGL(voxel_key_kernel, g1(n), dim3(256), 0, s, grid, min3_dev, n, key);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

// the same for float64 coordinates (what GridSample sees after a test-time rotation / scale)
int cdseg_voxelize_f64(const double* coord, double grid_size, long n, int32_t* grid, int64_t* key, int32_t* min3_dev,
                       void* stream) {
  if (n <= 0) return CDSEG_OK;
  if (!(grid_size > 0)) return CDSEG_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(grid_floor_kernel<double>, g1(3 * n), dim3(256), 0, s, coord, grid_size, 3 * n, grid);
  if (hipMemsetAsync(min3_dev, 0x7f, 3 * sizeof(int32_t), s) != hipSuccess) return CDSEG_ERR_LAUNCH;
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(min3_kernel, dim3((unsigned)blocks), dim3(256), 0, s, grid, n, min3_dev);
  hipLaunchKernelGGL(voxel_key_kernel, g1(n), dim3(256), 0, s, grid, min3_dev, n, key);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

// CenterShift (transform.py:142-155) on an (n,3) float32 / float64 array; ws: 6 x 8 bytes of scratch + 6 doubles
int cdseg_center_shift(const void* xyz, int is_f64, long n, int apply_z, void* out, void* ws12, void* stream) {
  if (n <= 0) return CDSEG_OK;
  if (!xyz || !out || !ws12) return CDSEG_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  unsigned long long* acc = (unsigned long long*)ws12;
  double* mm = (double*)ws12 + 6;
  if (hipMemsetAsync(acc, 0xff, 3 * 8, s) != hipSuccess || hipMemsetAsync(acc + 3, 0, 3 * 8, s) != hipSuccess)
    return CDSEG_ERR_LAUNCH;
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (is_f64) hipLaunchKernelGGL(minmax3_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, s, (const double*)xyz, n, acc);
  else hipLaunchKernelGGL(minmax3_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)xyz, n, acc);
  hipLaunchKernelGGL(minmax3_finish_kernel, dim3(1), dim3(64), 0, s, acc, mm);
  if (is_f64)
    hipLaunchKernelGGL(center_shift_kernel<double>, g1(n), dim3(256), 0, s, (const double*)xyz, mm, apply_z, n, (double*)out);
  else
    hipLaunchKernelGGL(center_shift_kernel<float>, g1(n), dim3(256), 0, s, (const float*)xyz, mm, apply_z, n, (float*)out);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

// One test-time augmentation of an (n,3) float32 array (coordinates, or normals with apply_scale = 0):
// rot9_host != NULL: out (float64) = (in R^T) [* scale]; else flip != 0: out (float32) = in with x, y negated.
int cdseg_tta_apply(const float* in, long n, const double* rot9_host, double scale, int apply_scale, int flip, void* out,
                    void* stream) {
  if (n <= 0) return CDSEG_OK;
  if (!in || !out) return CDSEG_ERR_ARG;
  TtaP p;
  p.rotate = rot9_host != nullptr;
  p.flip = flip;
  p.scale = scale;
  for (int i = 0; i < 9; ++i) p.r[i] = rot9_host ? rot9_host[i] : 0.0;
  hipStream_t s = (hipStream_t)stream;
  if (p.rotate) hipLaunchKernelGGL(tta_kernel<double>, g1(n), dim3(256), 0, s, in, p, apply_scale, n, (double*)out);
  else hipLaunchKernelGGL(tta_kernel<float>, g1(n), dim3(256), 0, s, in, p, 0, n, (float*)out);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

// out = in / div + add (float32): NormalizeColor is (color, 127.5, -1)
int cdseg_div_add(const float* in, float div, float add, long n, float* out, void* stream) {
  if (n <= 0) return CDSEG_OK;
  hipLaunchKernelGGL(div_add_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, in, div, add, n, out);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

// Collect(feat_keys=(a, b)): feat (n, ca + cb) float32 = cat([a, b.float()], 1)
int cdseg_collect_feat(const float* a, int ca, const void* b, int b_is_f64, int cb, long n, float* out, void* stream) {
  if (n <= 0) return CDSEG_OK;
  hipStream_t s = (hipStream_t)stream;
  if (b_is_f64) hipLaunchKernelGGL(collect_feat_kernel<double>, g1(n * (ca + cb)), dim3(256), 0, s, a, ca, (const double*)b, cb, n, out);
  else hipLaunchKernelGGL(collect_feat_kernel<float>, g1(n * (ca + cb)), dim3(256), 0, s, a, ca, (const float*)b, cb, n, out);
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

// Exact 1-NN of every query among the reference points of the same batch element.
// origin (3 floats, host): <= every reference coordinate; cell: grid cell size (a few voxel sizes).
// ws: >= cdseg_knn1_ws_bytes(n).  idx (m) int32 (-1 for an empty batch element), dist2 (m) float or NULL.
size_t cdseg_knn1_ws_bytes(long n) {
  return cdseg_sort_ws_bytes(n) + 2 * (((size_t)n * 8 + 255) & ~(size_t)255) + (((size_t)n * 4 + 255) & ~(size_t)255);
}

int cdseg_knn1(const float* ref_xyz, const int32_t* ref_offset, long n, const float* qry_xyz, const int32_t* qry_offset,
               long m, int nb, const float* origin, float cell, int32_t* idx, float* dist2, void* ws, size_t ws_bytes,
               void* stream) {
  if (m <= 0) return CDSEG_OK;
  if (nb <= 0 || nb > 8 || !(cell > 0) || !origin) return CDSEG_ERR_ARG;
  if (ws_bytes < cdseg_knn1_ws_bytes(n)) return CDSEG_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  KnnP p;
  p.ref = ref_xyz; p.qry = qry_xyz; p.ref_off = ref_offset; p.qry_off = qry_offset;
  p.ox = origin[0]; p.oy = origin[1]; p.oz = origin[2]; p.cell = cell; p.nb = nb; p.gmax = (1 << 20) - 1;
  p.n = n; p.m = m;
  char* w = (char*)ws;
  const size_t a8 = (((size_t)n * 8) + 255) & ~(size_t)255, a4 = (((size_t)n * 4) + 255) & ~(size_t)255;
  int64_t* key = (int64_t*)w;
  int64_t* key_sorted = (int64_t*)(w + a8);
  int32_t* perm = (int32_t*)(w + 2 * a8);
  if (n > 0) {
    hipLaunchKernelGGL(knn_cell_key_kernel, g1(n), dim3(256), 0, s, p, key);
    const int rc = cdseg_sort_pairs(key, key_sorted, nullptr, perm, n, 63, w + 2 * a8 + a4, ws_bytes - 2 * a8 - a4, stream);
    if (rc != CDSEG_OK) return rc;
  }
  p.key_sorted = key_sorted; p.perm = perm;
  hipLaunchKernelGGL(knn1_kernel, g1(m), dim3(256), 0, s, p, 6, idx, dist2);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

// Exact k nearest reference points (same batch element) of every query, ascending by (squared distance, index):
// idx (m, k) int32 with -1 placeholders, dist2 (m, k) float (SQUARED distances, 1e10 placeholders) or NULL.  k <= 64.
// Workspace and the other arguments as cdseg_knn1.
int cdseg_knn(const float* ref_xyz, const int32_t* ref_offset, long n, const float* qry_xyz, const int32_t* qry_offset,
              long m, int nb, int k, const float* origin, float cell, int32_t* idx, float* dist2, void* ws, size_t ws_bytes,
              void* stream) {
  if (m <= 0) return CDSEG_OK;
  if (k <= 0 || k > KNN_MAX_K) return CDSEG_ERR_UNSUPPORTED;
  if (nb <= 0 || nb > 8 || !(cell > 0) || !origin || !idx) return CDSEG_ERR_ARG;
  if (ws_bytes < cdseg_knn1_ws_bytes(n)) return CDSEG_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  KnnP p;
  p.ref = ref_xyz; p.qry = qry_xyz; p.ref_off = ref_offset; p.qry_off = qry_offset;
  p.ox = origin[0]; p.oy = origin[1]; p.oz = origin[2]; p.cell = cell; p.nb = nb; p.gmax = (1 << 20) - 1;
  p.n = n; p.m = m;
  char* w = (char*)ws;
  const size_t a8 = (((size_t)n * 8) + 255) & ~(size_t)255, a4 = (((size_t)n * 4) + 255) & ~(size_t)255;
  int64_t* key = (int64_t*)w;
  int64_t* key_sorted = (int64_t*)(w + a8);
  int32_t* perm = (int32_t*)(w + 2 * a8);
  if (n > 0) {
    hipLaunchKernelGGL(knn_cell_key_kernel, g1(n), dim3(256), 0, s, p, key);
    const int rc = cdseg_sort_pairs(key, key_sorted, nullptr, perm, n, 63, w + 2 * a8 + a4, ws_bytes - 2 * a8 - a4, stream);
    if (rc != CDSEG_OK) return rc;
  }
  p.key_sorted = key_sorted; p.perm = perm;
  hipLaunchKernelGGL(knnk_kernel, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, s, p, k, 8, idx, dist2);
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

// out (3,k) int64 zeroed here: intersection, prediction and target counts (union = pred + target - intersection).
// pred_idx (n) or NULL: the prediction of row i is pred[pred_idx[i]] (the 1-NN label transfer).
int cdseg_iou_counts(const int32_t* pred, const int32_t* pred_idx, const int32_t* target, long n, int k, int ignore_index,
                     int64_t* out, void* stream) {
  if (k <= 0) return CDSEG_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, (size_t)3 * k * sizeof(int64_t), s) != hipSuccess) return CDSEG_ERR_LAUNCH;
  if (n <= 0) return CDSEG_OK;
  long blocks = (n + 2047) / 2048;  // ~8 points per thread: the per-workgroup flush is 3 k global atomics
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(iou_counts_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pred, pred_idx, target, n, k, ignore_index,
                     (unsigned long long*)out);
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
