// MFMA GEMM with fused epilogues for the CDSegNet hot path (gfx950).
//
//   out = epilogue(A @ W^T)        nn.Linear             (ref: ptv3.py:170-171, 310-313, 463, 597-599, 1562)
//   out = epilogue(sum_o A[nbr[:,o]] @ W[:,o,:]^T)       spconv.SubMConv3d as a gathered-A GEMM
//                                                        (ref call sites: ptv3.py:356-362, 1106-1124)
//
// Tiling: workgroup = 4 waves (2 x 2) computes a 64 x BN output tile, K in steps of 32.
//   bf16: v_mfma_f32_16x16x32_bf16, one MFMA per 16x16 tile per K-step
//   f32 : v_mfma_f32_16x16x4_f32 x 8 per K-step (exact fp32, the parity mode)
// A / W K-tiles are staged through LDS in 16-byte chunks with an XOR swizzle
// (chunk ^= (row >> 1) & (chunks_per_row - 1)) that makes every ds_read_b128 of an MFMA fragment
// bank-conflict free (tools/lds_conflicts.py); global loads for tile k+1 are issued before
// the MFMAs of tile k (register double buffering).  The sparse-conv form gathers A rows through
// the stage's neighbour table and skips (block-uniformly) every kernel offset no row of the tile
// has a neighbour at - on z-ordered points that removes most of the 27 offsets' work.
// Small N (32..2048) and huge M: the op is HBM/L2 bound at the early stages, so the epilogue
// (bias, folded BatchNorm, GELU, residual, un-pooling gather-add, row scatter, second typed copy)
// is fused to keep every activation to one write.
#include "common.h"

namespace {

struct GemmP {
  const void* A;
  const void* W;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* res;
  const float* add_src;
  const int32_t* add_idx;
  const int32_t* nbr;
  const int32_t* out_idx;
  void* out;
  void* out2;
  long M;
  int N, K, kvol;
  int lda, ldo, ldo2, ldres, ldadd;
  int out_dtype, out2_dtype;
  int act, out2_pre_add;
};

template <typename CT> struct Tile;
template <> struct Tile<bf16_t> {
  static constexpr int ROW_BYTES = 64;  // 32 bf16
  static constexpr int NCHUNK = 4;
  static constexpr int CH_PER_PART = 1;  // a loader thread's 8 elements = 1 chunk
};
template <> struct Tile<float> {
  static constexpr int ROW_BYTES = 128;  // 32 f32
  static constexpr int NCHUNK = 8;
  static constexpr int CH_PER_PART = 2;
};

template <typename CT>
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * Tile<CT>::ROW_BYTES + ((chunk ^ ((row >> 1) & (Tile<CT>::NCHUNK - 1))) << 4);
}

__device__ __forceinline__ void store_val(void* p, int dtype, long idx, float v) {
  if (dtype == CDSEG_F32) ((float*)p)[idx] = v;
  else ((bf16_t*)p)[idx] = f32_to_bf16(v);
}

// CT: compute/storage type of A and W.  BN: output-tile width (32, 64, 128).
template <typename CT, int BN, bool GATHER>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP g) {
  constexpr int BM = 64;
  constexpr int TN = BN / 32;  // 16-wide column tiles per wave (wave owns BN/2 columns)
  constexpr int RB = Tile<CT>::ROW_BYTES;
  constexpr int CPP = Tile<CT>::CH_PER_PART;
  constexpr int B_PARTS = BN * 4;                      // 8-element parts in a W tile
  constexpr int B_PER_THREAD = (B_PARTS + 255) / 256;  // 1 or 2 (BN=32: half the threads)

  __shared__ __attribute__((aligned(16))) char smem[BM * RB + BN * RB + 16];
  char* As = smem;
  char* Bs = smem + BM * RB;
  unsigned long long* smask = reinterpret_cast<unsigned long long*>(smem + BM * RB + BN * RB);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const long Ktot = (long)g.kvol * g.K;
  const int nkc = (int)((Ktot + 31) / 32);

  // ---- loader coordinates
  const int a_row = tid >> 2, a_part = tid & 3;
  const long a_m = m0 + a_row;
  const bool a_ok = a_m < g.M;

  if (GATHER) {
    if (tid == 0) smask[0] = 0ull;
    __syncthreads();
    if (a_ok) {
      unsigned long long mine = 0ull;
      for (int o = a_part; o < g.kvol; o += 4)
        if (g.nbr[a_m * g.kvol + o] >= 0) mine |= 1ull << o;
      if (mine) atomicOr(smask, mine);
    }
    __syncthreads();
  }
  unsigned long long mask = ~0ull;  // kernel offsets some row of this tile has a neighbour at
  if (GATHER) mask = smask[0];
  auto chunk_live = [&](int kc) -> bool {
    if (!GATHER) return true;
    const int o_lo = (int)(((long)kc * 32) / g.K);
    long hi = (long)kc * 32 + 31;
    if (hi >= Ktot) hi = Ktot - 1;
    const int o_hi = (int)(hi / g.K);
    for (int o = o_lo; o <= o_hi; ++o)
      if ((mask >> o) & 1ull) return true;
    return false;
  };

  uint4 a_reg[CPP];
  uint4 b_reg[B_PER_THREAD][CPP];

  auto load_tiles = [&](int kc) {
    // A: 8 consecutive K elements of one row
    const long kf = (long)kc * 32 + a_part * 8;
#pragma unroll
    for (int c = 0; c < CPP; ++c) a_reg[c] = make_uint4(0u, 0u, 0u, 0u);
    if (a_ok && kf < Ktot) {
      const CT* src;
      bool ok = true;
      if (GATHER) {
        const int o = (int)(kf / g.K);
        const int cc = (int)(kf - (long)o * g.K);
        const int j = g.nbr[a_m * g.kvol + o];
        ok = j >= 0;
        src = (const CT*)g.A + (long)j * g.lda + cc;
      } else {
        src = (const CT*)g.A + a_m * g.lda + kf;
      }
      if (ok) {
#pragma unroll
        for (int c = 0; c < CPP; ++c) a_reg[c] = *reinterpret_cast<const uint4*>((const char*)src + 16 * c);
      }
    }
    // W: rows n0 .. n0+BN, same K slice
#pragma unroll
    for (int i = 0; i < B_PER_THREAD; ++i) {
      const int part = tid + i * 256;
      const int brow = part >> 2, bp = part & 3;
      const long bk = (long)kc * 32 + bp * 8;
#pragma unroll
      for (int c = 0; c < CPP; ++c) b_reg[i][c] = make_uint4(0u, 0u, 0u, 0u);
      if (part < B_PARTS && (n0 + brow) < g.N && bk < Ktot) {
        const CT* src = (const CT*)g.W + (long)(n0 + brow) * Ktot + bk;
#pragma unroll
        for (int c = 0; c < CPP; ++c) b_reg[i][c] = *reinterpret_cast<const uint4*>((const char*)src + 16 * c);
      }
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int c = 0; c < CPP; ++c)
      *reinterpret_cast<uint4*>(As + lds_off<CT>(a_row, a_part * CPP + c)) = a_reg[c];
#pragma unroll
    for (int i = 0; i < B_PER_THREAD; ++i) {
      const int part = tid + i * 256;
      if (part < B_PARTS) {
        const int brow = part >> 2, bp = part & 3;
#pragma unroll
        for (int c = 0; c < CPP; ++c)
          *reinterpret_cast<uint4*>(Bs + lds_off<CT>(brow, bp * CPP + c)) = b_reg[i][c];
      }
    }
  };

  f32x4_t acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;

  int kc = 0;
  while (kc < nkc && !chunk_live(kc)) ++kc;
  if (kc < nkc) {
    load_tiles(kc);
    store_tiles();
  }
  __syncthreads();
  while (kc < nkc) {
    int kn = kc + 1;
    while (kn < nkc && !chunk_live(kn)) ++kn;
    if (kn < nkc) load_tiles(kn);  // global loads in flight during the MFMAs below

    if constexpr (sizeof(CT) == 2) {
      bf16x8_t a[2], b[TN];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const bf16x8_t*>(As + lds_off<CT>(wm * 32 + i * 16 + fr, fg));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const bf16x8_t*>(Bs + lds_off<CT>(wn * (BN / 2) + j * 16 + fr, fg));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    } else {
      // k-slot g of MFMA step s = 4*half + ss holds k = 16*half + 4*g + ss  (same map for A and W)
      f32x4_t a[2][2], b[TN][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a[i][h] = *reinterpret_cast<const f32x4_t*>(As + lds_off<CT>(wm * 32 + i * 16 + fr, 4 * h + fg));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[j][h] = *reinterpret_cast<const f32x4_t*>(Bs + lds_off<CT>(wn * (BN / 2) + j * 16 + fr, 4 * h + fg));
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ss = 0; ss < 4; ++ss)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][h][ss], b[j][h][ss], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (kn < nkc) store_tiles();
    __syncthreads();
    kc = kn;
  }

  // ---- epilogue.  C layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + r
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long m = m0 + wm * 32 + i * 16 + fg * 4 + r;
      if (m >= g.M) continue;
      long orow = m;
      if (g.out_idx) {
        orow = g.out_idx[m];
        if (orow < 0) continue;
      }
      const long arow = g.add_src ? (long)g.add_idx[m] : 0;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 16 + fr;
        if (n >= g.N) continue;
        float v = acc[i][j][r];
        if (g.bias) v += g.bias[n];
        if (g.scale) v = v * g.scale[n] + g.shift[n];
        if (g.act == CDSEG_ACT_GELU) v = gelu_erf(v);
        else if (g.act == CDSEG_ACT_SWISH) v = v / (1.0f + expf(-v));
        if (g.out2 && g.out2_pre_add) store_val(g.out2, g.out2_dtype, m * g.ldo2 + n, v);
        if (g.res) v += g.res[m * g.ldres + n];
        if (g.add_src) v += g.add_src[arow * g.ldadd + n];
        store_val(g.out, g.out_dtype, orow * g.ldo + n, v);
        if (g.out2 && !g.out2_pre_add) store_val(g.out2, g.out2_dtype, m * g.ldo2 + n, v);
      }
    }
  }
}

template <typename CT, bool GATHER>
int launch(const GemmP& p, hipStream_t s) {
  const unsigned gm = (unsigned)((p.M + 63) / 64);
  if (p.N <= 32) {
    hipLaunchKernelGGL((gemm_kernel<CT, 32, GATHER>), dim3(gm, (unsigned)((p.N + 31) / 32)), dim3(256), 0, s, p);
  } else if (p.N <= 64 || (p.M <= 8192 && p.N <= 256)) {
    hipLaunchKernelGGL((gemm_kernel<CT, 64, GATHER>), dim3(gm, (unsigned)((p.N + 63) / 64)), dim3(256), 0, s, p);
  } else {
    hipLaunchKernelGGL((gemm_kernel<CT, 128, GATHER>), dim3(gm, (unsigned)((p.N + 127) / 128)), dim3(256), 0, s, p);
  }
  return hipGetLastError() == hipSuccess ? CDSEG_OK : CDSEG_ERR_LAUNCH;
}

}  // namespace

extern "C" int cdseg_gemm(const cdseg_gemm_args* a, void* stream) {
  if (!a || !a->A || !a->W || !a->out) return CDSEG_ERR_ARG;
  if (a->M <= 0 || a->N <= 0) return CDSEG_OK;
  if (a->K <= 0 || (a->K & 7) || a->kvol <= 0 || a->kvol > 64) return CDSEG_ERR_ARG;
  if (a->a_dtype != a->compute_dtype) return CDSEG_ERR_UNSUPPORTED;
  if (a->scale && !a->shift) return CDSEG_ERR_ARG;
  if (a->add_src && !a->add_idx) return CDSEG_ERR_ARG;
  if (!a->nbr && a->kvol != 1) return CDSEG_ERR_ARG;
  const int esz = a->compute_dtype == CDSEG_F32 ? 4 : 2;
  if (((long)a->lda * esz) & 15) return CDSEG_ERR_ARG;  // 16-byte row alignment for the vector loads
  GemmP p;
  p.A = a->A; p.W = a->W; p.bias = a->bias; p.scale = a->scale; p.shift = a->shift; p.res = a->res;
  p.add_src = a->add_src; p.add_idx = a->add_idx; p.nbr = a->nbr; p.out_idx = a->out_idx;
  p.out = a->out; p.out2 = a->out2; p.M = a->M; p.N = a->N; p.K = a->K; p.kvol = a->kvol;
  p.lda = a->lda; p.ldo = a->ldo; p.ldo2 = a->ldo2; p.ldres = a->ldres; p.ldadd = a->ldadd;
  p.out_dtype = a->out_dtype; p.out2_dtype = a->out2_dtype; p.act = a->act; p.out2_pre_add = a->out2_pre_add;
  hipStream_t s = (hipStream_t)stream;
  if (a->compute_dtype == CDSEG_BF16) return a->nbr ? launch<bf16_t, true>(p, s) : launch<bf16_t, false>(p, s);
  if (a->compute_dtype == CDSEG_F32) return a->nbr ? launch<float, true>(p, s) : launch<float, false>(p, s);
  return CDSEG_ERR_ARG;
}
