/* cdseg.h - C ABI of libcdseg_hip.so: the MI355X (gfx950) kernels behind CDSegNet's
 * single-step-inference hot path.
 *
 * Drop-in boundary.  The reference (a Python program on PyTorch) reaches its fused
 * arithmetic through third-party Python ops; each entry point below is what a binding
 * for that call site would bind (plain device pointers + sizes, no torch types), and
 * cdsegnet_amd/ops.py is the ctypes binding the build ships (INTEGRATION.md shows the
 * equivalent stubs on the reference side).  "ref:" cites the reference interface
 * replaced, relative to /root/reference/pointcept/.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host;
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *  - every function is asynchronous on `stream`, returns CDSEG_OK (0) or a negative
 *    CDSEG_ERR_* code, and never allocates (callers own all buffers and workspaces);
 *  - dtypes: CDSEG_F32 = float32, CDSEG_BF16 = the build's 16-bit type as raw uint16 bits: bfloat16 in libcdseg_hip.so,
 *    IEEE half in libcdseg_hip_f16.so (the same sources compiled with -DCDSEG_LP_F16; float -> half conversions saturate
 *    at +-65504).  Same entry points, same signatures: a caller picks the library that matches its tensors' dtype;
 *  - index arrays are int32 unless stated; codes are int64 (non-negative).
 *  - attention head dimension is 16 (every stage of every shipped config: C/H = 16).
 */
#ifndef CDSEG_H
#define CDSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDSEG_OK 0
#define CDSEG_ERR_ARG (-1)
#define CDSEG_ERR_LAUNCH (-2)
#define CDSEG_ERR_WORKSPACE (-3)
#define CDSEG_ERR_UNSUPPORTED (-4)

#define CDSEG_F32 0
#define CDSEG_BF16 1
/* compute type only (cdseg_gemm compute_dtype, cdseg_attention_ex dtype, cdseg_block_desc dtype): fp32 tensors in memory, every
 * product as three IEEE-half MFMAs on split operands (x ~= hi + lo' / 2048), fp32 accumulation - csrc/gemm.hip "fp32 x3" */
#define CDSEG_F32X3 2

#define CDSEG_ORDER_Z 0
#define CDSEG_ORDER_Z_TRANS 1
#define CDSEG_ORDER_HILBERT 2
#define CDSEG_ORDER_HILBERT_TRANS 3

#define CDSEG_ACT_NONE 0
#define CDSEG_ACT_GELU 1  /* exact erf GELU (nn.GELU()) */
#define CDSEG_ACT_SWISH 2 /* x * sigmoid(x), ref: models/point_transformer_v3/point_transformer_v3m1_base.py:30-31 */

#define CDSEG_HEAD_DIM 16
#ifndef CDSEG_DEEP512_MIN_ROWS
#define CDSEG_DEEP512_MIN_ROWS 2560 /* C = 512 Blocks with fewer rows keep the separate GEMM launches (csrc/runtime.hip) */
#endif
#define CDSEG_MAX_PATCH 1024

/* library / device identification (host-only helpers) */
int cdseg_abi_version(void);
const char* cdseg_build_info(void);

/* ------------------------------------------------------------------ serialization
 * ref: models/utils/structure.py:47-102 (Point.serialization),
 *      models/utils/serialization/default.py:8-24 (encode), z_order.py:66-101, hilbert.py:91-198 */

/* max over all grid coordinates -> *out_dev (int64); depth = bit_length(max).  ref: structure.py:66 */
int cdseg_grid_max(const void* grid, int elem_bytes, long n3, int64_t* out_dev, void* stream);
/* batch[i] = index of the batch element owning point i.  ref: models/utils/misc.py:18-23 (offset2batch) */
int cdseg_offset2batch(const int64_t* offset, int nb, long n, int32_t* batch, void* stream);
/* code[i] = (batch[i] << 3*depth) | curve_key(order_id, grid[i]).  grid (n,3) int32|int64,
 * batch int32|int64 or NULL.  ref: serialization/default.py:8-24 */
int cdseg_encode(const void* grid, int grid_elem_bytes, const void* batch, int batch_elem_bytes, long n, int depth,
                 int order_id, int64_t* code, void* stream);
/* all four curves, code4 is (4,n) row-major in CDSEG_ORDER_* order */
int cdseg_encode4(const int32_t* grid, const int32_t* batch, long n, int depth, int64_t* code4, void* stream);
/* stable radix sort of (code, value) pairs = torch.argsort(code) when vals_in == NULL (iota).
 * ref: structure.py:83 (torch.argsort), ptv3.py:493.  ws from cdseg_sort_ws_bytes(n). */
size_t cdseg_sort_ws_bytes(long n);
int cdseg_sort_pairs(const int64_t* keys_in, int64_t* keys_out, const int32_t* vals_in, int32_t* vals_out, long n,
                     int end_bit, void* ws, size_t ws_bytes, void* stream);
/* Orders of `count` (<= 4) curves of the same n points with ONE sort: codes (4, n) int64 (row r = curve r), rows[count] =
 * the curve rows wanted (host array), orders (count, n) int32 out (= count argsorts, ref: structure.py:83).  The curve
 * slot rides in the key bits above end_bit (end_bit + 2 <= 64).  ws from cdseg_sort_curves_ws_bytes(n, count). */
size_t cdseg_sort_curves_ws_bytes(long n, int count);
int cdseg_sort_curves(const int64_t* codes, const int* rows, int count, long n, int end_bit, int32_t* orders, void* ws,
                      size_t ws_bytes, void* stream);
/* inv[perm[i]] = i.  ref: structure.py:84-90 (scatter_ of arange) */
int cdseg_invert_perm(const int32_t* perm, long n, int32_t* inv, void* stream);
int cdseg_widen_i32(const int32_t* src, long n, int64_t* dst, void* stream);

/* row gathers / scatters (rows of row_bytes % 4 == 0 bytes); idx < 0 gathers zeros / skips */
int cdseg_gather_rows(const void* src, const int32_t* idx, long n_out, int row_bytes, void* dst, void* stream);
int cdseg_scatter_rows(const void* src, const int32_t* idx, long n_in, int row_bytes, void* dst, void* stream);
int cdseg_gather_i32(const int32_t* src, const int32_t* idx, long n, int32_t* dst, void* stream);

/* engine plan, stage 0: int32 grid + batch in z-sorted ("physical") order */
int cdseg_plan_gather_grid(const void* grid, int grid_elem_bytes, const int32_t* perm, const int64_t* zcode_sorted,
                           long n, int depth, int32_t* grid_out, int32_t* batch_out, void* stream);

/* ------------------------------------------------------------------ pooling structure
 * ref: ptv3.py:477-492 (code >> 3*depth, torch.unique, sort(cluster), idx_ptr, head).
 * On z-sorted points a cluster is a contiguous run: cluster ids are an inclusive scan of the
 * run-start flags, seg_start (count+1 entries) are the run starts, *count_dev the number of runs. */
int cdseg_pool_level(const int64_t* zcode_sorted, long n, int shift_bits, int32_t* cluster, int32_t* seg_start,
                     int32_t* count_dev, void* ws, size_t ws_bytes, void* stream);
/* All pooling levels of a scene in one flag / scan / finish pass (replaces one cdseg_pool_level per level).
 * shifts (host, nlev <= 8) = 3 * cumulative pooling depth; last_idx (nb, device) = last point of each batch element.
 * cluster (nlev, n), seg_start (nlev, n + 1) int32; meta: nlev * (1 + nb) + 1 int32 = per level [count, cluster of
 * last_idx[b] ...], then ONE trailing int = the number of points whose (batch, voxel) code repeats their predecessor's
 * (0 for a valid model input: one point per voxel - the host raises on anything else with the read it does anyway). */
size_t cdseg_pool_levels_ws_bytes(long n, int nlev);
int cdseg_pool_levels(const int64_t* zcode_sorted, long n, const int* shifts, int nlev, const int32_t* last_idx, int nb,
                      int32_t* cluster, int32_t* seg_start, int32_t* meta, void* ws, size_t ws_bytes, void* stream);
/* Link between two pooled levels a (finer, m_a points) and b (m_b) from their links to level 0:
 * cluster_ab (m_a), seg_ab (m_b + 1). */
int cdseg_link_derive(const int32_t* cluster_0a, const int32_t* seg_0a, long ma, const int32_t* cluster_0b,
                      const int32_t* seg_0b, long mb, int32_t* cluster_ab, int32_t* seg_ab, void* stream);
/* Curve orders of ALL pooled levels without sorting: z-order / Hilbert keys are hierarchical, so the coarse order on a
 * curve is the level-0 order with every point replaced by its cluster id and consecutive duplicates dropped
 * (replaces torch.argsort(code >> 3*depth) of SerializedPooling, ref ptv3.py:503-514).
 * clusters[l] (n0): level-0 point -> cluster id at pooled level l; orders[c] (n0): rank -> level-0 point on curve c.
 * out: level l / curve c at int32 offset ncurve * sum_{l'<l} m_l' + c * m_l. */
size_t cdseg_coarse_orders_ws_bytes(long n0, int nlev, int ncurve);
int cdseg_coarse_orders(const int32_t* const* clusters, int nlev, const int32_t* const* orders, int ncurve, long n0,
                        int32_t* out, void* ws, size_t ws_bytes, void* stream);
/* pooled grid / batch / codes from the first fine point of each run.  ref: ptv3.py:489-491, 519-525 */
int cdseg_pool_gather(const int32_t* seg_start, long m, long n_fine, int pooling_depth, const int32_t* grid_f,
                      const int32_t* batch_f, const int64_t* code4_f, int32_t* grid_c, int32_t* batch_c,
                      int64_t* code4_c, void* stream);

/* ------------------------------------------------------------------ sparse-conv kernel map
 * ref: spconv.SubMConv3d indice pairs (third party; call sites ptv3.py:356,647,1106,1118).
 * nbr (n, ksize^3): row index of the occupied voxel at grid + (a-r, b-r, c-r), column
 * a*k*k + b*k + c, or -1 (kmajor != 0: stored transposed, (ksize^3, n)).  Points must be sorted
 * by (batch | z-order) code. */
int cdseg_nbr_table(const int64_t* zcode_sorted, const int32_t* grid, const int32_t* batch, long n, int depth,
                    int ksize, int kmajor, int32_t* nbr, void* stream);

/* The same kernel map derived from the PARENT level's 3x3x3 map (pooling depth 1) instead of searching: a target
 * cell lies in one of the 27 parent cells around the point's parent; empty parent cell => empty target, else the target
 * is one of its <= 8 z-contiguous children.  cluster (n): point -> parent; parent_nbr3 (27, m) offset-major;
 * seg_start (m + 1): children runs of the parents. */
int cdseg_nbr_table_from_parent(const int64_t* zcode_sorted, const int32_t* grid, const int32_t* cluster,
                                const int32_t* parent_nbr3, const int32_t* seg_start, long n, long m, int depth, int ksize,
                                int kmajor, int32_t* nbr, void* stream);
/* ... and with the parents' child_info words (cdseg_child_info: first child row << 8 | octant occupancy) instead of the
 * children runs: target = first + popcount(occupancy below the target's octant). */
int cdseg_nbr_table_from_info(const int32_t* grid, const int32_t* cluster, const int32_t* parent_nbr3,
                              const int64_t* child_info, long n, long m, int depth, int ksize, int kmajor, int32_t* nbr,
                              void* stream);
/* ------------------------------------------------------------------ attention padding plan
 * ref: ptv3.py:188-244 (get_padding_and_inverse) in gather/scatter form: for every padded slot
 * the row to read (gidx) and the row to write (widx, -1 for the borrowed duplicates).
 * order: rank -> row for the curve of this block (NULL = identity); offs/offs_pad (nb+1). */
int cdseg_pad_plan(const int32_t* order, const int32_t* offs, const int32_t* offs_pad, int nb, int patch, long n_pad,
                   int32_t* gidx, int32_t* widx, void* stream);
/* every slot plan of a scene in ONE launch (count <= CDSEG_PAD_BATCH_MAX): plan j = (orders[j] or NULL, offs[j],
 * offs_pad[j], patch[j], n_pad[j]); gidx / widx of plan j start at sum_{i<j} n_pad[i]. Pointer arrays are host arrays. */
#define CDSEG_PAD_BATCH_MAX 48
int cdseg_pad_plan_batch(int count, const int32_t* const* orders, const int32_t* const* offs,
                         const int32_t* const* offs_pad, const int* patch, const long* n_pad, int nb, int32_t* gidx,
                         int32_t* widx, void* stream);

/* ------------------------------------------------------------------ dense / gathered GEMM
 * out = epilogue(A @ W^T): nn.Linear (ref: ptv3.py:170-171, 310-313, 463, 597-599, 1562) and,
 * with nbr != NULL, spconv.SubMConv3d as a gathered-A GEMM over kvol offsets
 * (W is the conv weight (N, kvol, K) flattened = (out, k0, k1, k2, in)).
 * epilogue: v = acc + bias; v = v*scale + shift (folded eval BatchNorm1d, ref: ptv3.py:1440);
 *           v = act(v); [out2 = v if out2_pre_add]; v += res; v += add_src[add_idx[m]];
 *           out[out_idx ? out_idx[m] : m] = v; [out2 = v otherwise]
 * kvol <= 128 (k=5 stem: 125), K % 8 == 0. */
typedef struct cdseg_gemm_args {
  const void* A;          /* (M, lda) a_dtype; with nbr: the gather source (rows indexed by nbr) */
  const void* W;          /* (N, kvol*K) compute dtype, K contiguous */
  const float* bias;      /* (N) or NULL */
  const float* scale;     /* (N) or NULL */
  const float* shift;     /* (N) or NULL (required with scale) */
  const float* res;       /* (M, ldres) or NULL */
  const float* add_src;   /* (*, ldadd) or NULL */
  const int32_t* add_idx; /* (M) */
  const int32_t* nbr;     /* (M, kvol), or (kvol, M) when nbr_kmajor, or NULL */
  const int32_t* out_idx; /* (M) or NULL */
  void* out;              /* (M, ldo) out_dtype */
  void* out2;             /* (M, ldo2) out2_dtype or NULL */
  long M;
  int N, K, kvol;
  int lda, ldo, ldo2, ldres, ldadd;
  int a_dtype, compute_dtype, out_dtype, out2_dtype;
  int act;
  int out2_pre_add;
  void* ws;        /* optional split-K workspace (fp32 partial tiles) or NULL: used when M is small */
  size_t ws_bytes; /*   and K long (deep stages); results are deterministic either way */
  /* fused row LayerNorm (N <= 128; ref: nn.LayerNorm eps 1e-5 at ptv3.py:365,367,381):
   *   v = act(acc + bias);  if ln_pre:  v = LN(v) * ln_pre_g + ln_pre_b      (CPE, ptv3.py:401-404)
   *   v += res + colbias (+ add_src);  out = v;  if ln_post: ln_out = LN(v) * ln_post_g + ln_post_b */
  const float* colbias;   /* (N) or NULL: timestep-embedding bias, ptv3.py:406-411 */
  const float* ln_pre_g;
  const float* ln_pre_b;
  const float* ln_post_g;
  const float* ln_post_b;
  void* ln_out;           /* (M, ldln) ln_out_dtype */
  int ldln, ln_out_dtype;
  float ln_eps;
  int nbr_kmajor;         /* neighbour table is offset-major (kvol, M): coalesced index loads, cheaper to build */
} cdseg_gemm_args;
int cdseg_gemm(const cdseg_gemm_args* args_host, void* stream);

/* ------------------------------------------------------------------ LayerNorm
 * out = [res +] LayerNorm(x) * gamma + beta [+ colbias]; optional second copy out2.
 * ref: nn.LayerNorm(eps 1e-5) at ptv3.py:365, 367, 381 and the CPE residual ptv3.py:401-404,
 * t-embedding bias ptv3.py:406-411. */
int cdseg_layernorm(const void* x, int x_dtype, int ldx, const float* gamma, const float* beta, float eps,
                    const float* res, int ldres, const float* colbias, void* out, int out_dtype, int ldo,
                    void* out2, int out2_dtype, int ldo2, long m, int c, void* stream);

/* ------------------------------------------------------------------ serialized window attention
 * ref: ptv3.py:246-296 (SerializedAttention core; flash_attn_varlen_qkvpacked_func :282-288)
 *      ptv3.py:988-1055 (SerializedCrossAttention core; flash_attn_varlen_kvpacked_func :1038-1047)
 * For patch p (slots patch_start[p] .. patch_start[p+1], length <= 1024) and head h:
 *   O = softmax(scale * Q K^T) V over the full patch (no mask), head dim 16,
 *   Q row of slot s = q[q_gidx[s]*ldq + h*16 ..], K/V rows by kv_gidx, output row widx[s] (skip if -1).
 * The gather by serialized order, the padding duplicates and the scatter by inverse are fused. */
int cdseg_attention(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, const int32_t* q_gidx,
                    const int32_t* kv_gidx, const int32_t* widx, const int32_t* patch_start, int num_patches,
                    int num_heads, int max_len, float scale, void* out, int ldo, int dtype, void* stream);
/* The same with producer-side preprocessing declared in `flags` (what the fused qkv epilogues of this library emit, so the
 * kernel - VALU-issue bound - does not redo per (head, query slice) what a producer does once per row):
 *   CDSEG_ATTN_Q_PRESCALED  q already carries softmax scale * log2(e) (folded into Wq / bq at load time); `scale` is ignored
 *   CDSEG_ATTN_V_BF16       v holds bfloat16 whatever the build's 16-bit type is (the IEEE-half build runs P V in bfloat16:
 *                           P = exp2(s - bound) needs fp32's exponent range); CDSEG_BF16 dtype only
 * q, k, v, out and every row (ld * element size) must be 16-byte aligned (CDSEG_ERR_ARG otherwise). */
#define CDSEG_ATTN_Q_PRESCALED 1
#define CDSEG_ATTN_V_BF16 2
int cdseg_attention_ex(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, const int32_t* q_gidx,
                       const int32_t* kv_gidx, const int32_t* widx, const int32_t* patch_start, int num_patches,
                       int num_heads, int max_len, float scale, void* out, int ldo, int dtype, int flags, void* stream);

/* Diagnostic, host only: the block id -> (patch, head, query slice, slices of that patch-head) table of the launch
 * cdseg_attention_ex issues for this shape (the graded schedule of csrc/attention.hip: the last blocks an XCD runs cover a
 * half / a quarter of a patch-head's queries).  4 ints per block id, -1 x 4 for ids that exit at once.  Returns the number
 * of block ids (may exceed `capacity`; at most `capacity` rows are written), or a negative status. */
long cdseg_attention_schedule(int num_patches, int num_heads, int max_len, int dtype, int32_t* table, long capacity);

/* HIP-event timing of the attention launches on their own stream (bench.py's live roofline measurement):
 * enable(1) starts recording, summary() (after a device sync) returns the summed durations. */
int cdseg_prof_enable(int on);
int cdseg_prof_summary(double* total_ms, long* launches); /* class CDSEG_PROF_ATTENTION */
/* per kernel class: 0 = window attention, 1 = k = 3 sparse convs of the wide stages (cdseg_subm_conv3: HBM / gather bound),
 * 2 = k = 3 sparse convs on the gathered GEMM (cdseg_gemm with a 27-offset map: C >= 128, MFMA / LDS-DMA bound) */
#define CDSEG_PROF_ATTENTION 0
#define CDSEG_PROF_CONV 1
#define CDSEG_PROF_CONV_DEEP 2
int cdseg_prof_summary_class(int cls, double* total_ms, long* launches);

/* ------------------------------------------------------------------ pooling reduce
 * out[j] = act(max_{i in run j} y[i] * scale + shift), runs = seg_start (m+1).
 * ref: ptv3.py:510-515 (torch_scatter.segment_csr(..., "max")) + norm/act :548-551 */
int cdseg_segment_max(const void* y, int y_dtype, int ldy, const int32_t* seg_start, long m, int c,
                      const float* scale, const float* shift, int act, float* out, int ldo, void* out2,
                      int out2_dtype, int ldo2, void* stream);
/* The whole SerializedPooling feature path in one launch (16-bit trunk; ref: ptv3.py:506-515 proj + segment_csr("max"),
 * :548-551 norm + act):   out[j] = act(scale * max_{i in run j} round16(W x_i + bias) + shift),   out2 = 16-bit copy.
 * Bit-identical to cdseg_gemm (16-bit output) followed by cdseg_segment_max; the projected rows never reach HBM.
 * (cin, cout) = (32, 64) or (64, 128), else CDSEG_ERR_UNSUPPORTED (callers use the two-launch form).  wimg: fragment image
 * of W (cout, cin) built once by cdseg_pool_fused_pack (cdseg_pool_fused_img_bytes bytes).  act: NONE or GELU. */
size_t cdseg_pool_fused_img_bytes(int cin, int cout);
int cdseg_pool_fused_pack(const void* w, int cin, int cout, void* wimg, void* stream);
int cdseg_pool_fused(const void* x, int ldx, const void* wimg, const float* bias, const int32_t* seg_start, long m,
                     const float* scale, const float* shift, int act, float* out, int ldo, void* out2, int ldo2,
                     int cin, int cout, void* stream);
/* segment mean of coord (ref: ptv3.py:513-515) */
int cdseg_segment_mean(const float* x, int ldx, const int32_t* seg_start, long m, int c, float* out, int ldo,
                       void* stream);

/* ------------------------------------------------------------------ small dense helpers */
/* y = act(W x + b), W (n,k) fp32: the timestep-embedding MLP, ref: ptv3.py:1772-1778, :406-411 */
int cdseg_gemv(const float* w, const float* b, const float* x, int n, int k, int act, float* y, void* stream);
/* standard normal draws (Philox4x32-10 + Box-Muller), the noise-branch input.  ref: default.py:393 */
int cdseg_randn(float* out, long n, uint64_t seed, uint64_t offset, void* stream);
int cdseg_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, void* stream);
/* dst[i, 0:cpad] = cast(src[idx[i], 0:cin]), zero padded (idx NULL = identity): stem input rows in the
 * engine's point order, padded to a 16-byte multiple so the stem conv runs on the gathered GEMM */
int cdseg_gather_pad_cast(const float* src, int ld_src, const int32_t* idx, long n, int cin, int cpad, void* dst,
                          int dst_dtype, void* stream);
/* DDIM update of the noise-branch input on a uniform timestep t (ref: default.py:192-214, dm_target="noise"):
 * x0 = (xt - sqrt(1-ab_t) eps) / sqrt(ab_t); out = final ? x0 : sqrt(ab_{t-1}) x0 + sqrt(1-ab_{t-1}) eps */
int cdseg_ddim_update(const float* xt, const float* eps, float sqrt_ab_prev, float sqrt_1m_ab, float sqrt_ab,
                      float sqrt_1m_ab_prev, int final_step, float* out, long n, void* stream);
/* Diagnostic of the IEEE-half build: *counter += the number of elements of the 16-bit buffer x (rows x cols, row stride ld
 * elements) that sit exactly at +-65504 - the value every float -> half conversion of that build clamps to.  The bfloat16
 * build never clamps and adds nothing.  cdseg_block_forward runs it over a Block's conv output, q / k, attention output and
 * shadow copy when cdseg_block_io.sat_counter is set. */
int cdseg_count_saturated(const void* x, long rows, int cols, int ld, unsigned long long* counter, void* stream);
/* x (rows, cols) fp32 -> hi = T(x) (saturating), lo = T((x - hi) * lo_scale) in the build's 16-bit type T (cols, ldx, ld16
 * multiples of 4).  The IEEE-half build with lo_scale = 2048 gives x ~= hi + lo / 2048 to 22 significant bits: the operand
 * pairs of precision "fp32x3"'s sparse convs (three 16-bit gathered GEMMs into one fp32 output). */
/* cdseg_subm_conv3 with an fp32 output: yf = (accumulate ? yf : 0) + (conv + bias) * out_scale (C = 32 / 64, 16-bit x and
 * weight image as above).  Three calls on IEEE-half operand pairs give a k = 3 conv to 22 significant bits ("fp32x3"). */
int cdseg_subm_conv3_f32(const void* x, int ldx, const void* wimg, const float* bias, const int32_t* nbr_kmajor, long n,
                         int channels, float* yf, int ldyf, float out_scale, int accumulate, void* stream);
int cdseg_split16(const float* x, int ldx, long rows, int cols, void* hi, void* lo, int ld16, float lo_scale, void* stream);
/* out = a + alpha * b (fp32).  ref: default.py:228-236 (add_gaussian_noise) */
int cdseg_axpy(const float* a, const float* b, float alpha, float* out, long n, void* stream);

/* ------------------------------------------------------------------ test-time pipeline
 * ref: datasets/transform.py:821-897 (GridSample mode="test"), engines/test.py:261-278 (softmax vote, arg-max) */
/* grid = floor(coord / grid_size) - min (int32), key = one int64 per voxel; min3_dev (3 x int32) receives the min */
int cdseg_voxelize(const float* coord, double grid_size, long n, int32_t* grid, int64_t* key, int32_t* min3_dev,
                   void* stream);
int cdseg_voxelize_f64(const double* coord, double grid_size, long n, int32_t* grid, int64_t* key, int32_t* min3_dev,
                       void* stream); /* float64 coordinates: what GridSample sees after a test-time rotation */
/* pre-model transforms of the test pipeline (ref: configs/scannet/CDSegNet.py:253-398, datasets/transform.py):
 * CenterShift :142-155 on an (n,3) float32 / float64 array (ws12: 96 bytes of device scratch);
 * one test-time augmentation :259-328 - rot9_host (row-major R) given: out FLOAT64 (n,3) = (in R^T) [* scale]
 *   (RandomRotateTargetAngle about the origin, then RandomScale), else flip: out float32 = in with x and y negated;
 * NormalizeColor :113-117 as out = in / div + add;  Collect(feat_keys=(a, b)) :46-49 as feat = cat([a, float(b)], 1). */
int cdseg_center_shift(const void* xyz, int is_f64, long n, int apply_z, void* out, void* ws12, void* stream);
int cdseg_tta_apply(const float* in, long n, const double* rot9_host, double scale, int apply_scale, int flip, void* out,
                    void* stream);
int cdseg_div_add(const float* in, float div, float add, long n, float* out, void* stream);
int cdseg_collect_feat(const float* a, int ca, const void* b, int b_is_f64, int cb, long n, float* out, void* stream);
/* largest run length of a seg_start array (m runs) = number of test fragments (count.max()) */
int cdseg_max_run(const int32_t* seg_start, long m, int32_t* out_dev, void* stream);
/* fragment `frag`: idx_part[v] = idx_sort[seg_start[v] + frag % count_v].  ref: transform.py:862-864 */
int cdseg_fragment_select(const int32_t* idx_sort, const int32_t* seg_start, long m, int frag, int32_t* idx_part,
                          void* stream);
/* pred[idx[i], :] += softmax(logits[i, :]).  ref: engines/test.py:261-267 */
int cdseg_softmax_vote(const float* logits, int ldl, const int32_t* idx, long m, int c, float* pred, int ldp,
                       void* stream);
/* out[i] = first arg-max of row i.  ref: engines/test.py:278 */
int cdseg_argmax_rows(const float* x, int ldx, long n, int c, int32_t* out, void* stream);

/* ------------------------------------------------------------------ evaluator
 * ref: engines/hooks/evaluator.py:132-140 (pointops.knn_query(1, ...) label transfer), utils/misc.py:52-65 (IoU counts).
 * Exact 1-NN of every query among the reference points of the same batch element (lowest index on ties, like the
 * reference's brute-force scan, libs/pointops/src/knn_query/knn_query_cuda_kernel.cu:60-104), found through a uniform
 * grid: origin (3 host floats) <= every reference coordinate, cell = grid cell size.  offsets: cumulative ends (nb).
 * idx (m) int32, dist2 (m) float or NULL. */
/* k nearest neighbours (round 5; ref: libs/pointops/functions/query.py:7-24 KNNQuery, src/knn_query/knn_query_cuda_kernel.cu:
 * 60-104 - the neighbourhood query of the reference's other backbones, off the CDSegNet path, SURVEY finding 2): idx (m, k)
 * int32 ascending by (squared distance, index), -1 placeholders when the batch element has fewer than k points; dist2 (m, k)
 * SQUARED distances (the reference's Python wrapper takes the square root), 1e10 placeholders, or NULL.  k <= 64
 * (CDSEG_ERR_UNSUPPORTED beyond); workspace cdseg_knn1_ws_bytes(n); other arguments as cdseg_knn1. */
int cdseg_knn(const float* ref_xyz, const int32_t* ref_offset, long n, const float* qry_xyz, const int32_t* qry_offset,
              long m, int nb, int k, const float* origin, float cell, int32_t* idx, float* dist2, void* ws, size_t ws_bytes,
              void* stream);
size_t cdseg_knn1_ws_bytes(long n);
int cdseg_knn1(const float* ref_xyz, const int32_t* ref_offset, long n, const float* qry_xyz, const int32_t* qry_offset,
               long m, int nb, const float* origin, float cell, int32_t* idx, float* dist2, void* ws, size_t ws_bytes,
               void* stream);
/* out (3,k) int64: per-class intersection, prediction and target counts over rows with target != ignore_index;
 * pred_idx (n) or NULL: row i is predicted pred[pred_idx[i]]. */
int cdseg_iou_counts(const int32_t* pred, const int32_t* pred_idx, const int32_t* target, long n, int k, int ignore_index,
                     int64_t* out, void* stream);

/* ------------------------------------------------------------------ fused MLP (ref: ptv3.py:299-322, :423-427)
 * x (n, ldx) fp32 += fc2(GELU(fc1(h))), h (n, ldh); xc (n, ldxc) = typed copy of the new x or NULL.  The 4C hidden
 * activation stays in LDS.  Supported: dtype bf16, channels 32, 64 or 128 (hidden = 4 * channels); else
 * CDSEG_ERR_UNSUPPORTED (callers fall back to two cdseg_gemm launches). */
int cdseg_mlp_fused(const void* h, int ldh, const void* w1, const float* b1, const void* w2, const float* b2, float* x,
                    int ldx, void* xc, int ldxc, long n, int channels, int dtype, void* stream);

/* Block tail after attention in ONE launch (ptv3.py:416-427): x += proj(o); h = LayerNorm(x); x += fc2(GELU(fc1(h)));
 * xc = typed copy of x.  o (n, ldo); the updated residual rows stay in LDS while the MLP runs, h never exists in HBM.
 * Supported: bf16, channels 32 or 64; else CDSEG_ERR_UNSUPPORTED. */
int cdseg_attn_tail_fused(const void* o, int ldo, const void* wp, const float* bp, const float* ln_g, const float* ln_b,
                          float ln_eps, const void* w1, const float* b1, const void* w2, const float* b2, float* x, int ldx,
                          void* xc, int ldxc, long n, int channels, int dtype, void* stream);

/* Block head after the sparse conv in ONE launch (ptv3.py:401-414): x += LN_cpe(y Wl^T + bl) [+ colbias];
 * h = LN1(x); qkv (n, ldqkv) = h Wqkv^T + bqkv.  y (n, ldy) = conv output; h never exists in HBM.
 * qkv_flags: 0, or CDSEG_ATTN_V_BF16 = write the v third of qkv as bfloat16 whatever the build's 16-bit type is (pass the
 * same flag to cdseg_attention_ex; a no-op in the bfloat16 build).
 * Supported: bf16, channels 32 or 64; else CDSEG_ERR_UNSUPPORTED. */
int cdseg_cpe_head_fused(const void* y, int ldy, const void* wl, const float* bl, const float* lnp_g, const float* lnp_b,
                         float* x, int ldx, const float* colbias, const float* ln1_g, const float* ln1_b, float eps,
                         const void* wqkv, const float* bqkv, void* qkv, int ldqkv, long n, int channels, int dtype,
                         int qkv_flags, void* stream);

/* ------------------------------------------------------------------ 3x3x3 submanifold conv, wide stages
 * ref: spconv.SubMConv3d call sites ptv3.py:356-362 (the CPE conv of every Block).  y (n, ldy) bf16 =
 * bias + sum_o x[nbr[o, i]] W_o^T for C_in = C_out = channels in {32, 64}, bf16: weights stationary in LDS, gathered rows
 * loaded straight into MFMA fragments (csrc/conv.hip).  wimg = fragment-order image of the (C, 27*C) weight, built once
 * per weight tensor by cdseg_subm_conv3_pack into a caller-owned buffer of cdseg_subm_conv3_wimg_bytes(channels) bytes.
 * nbr: OFFSET-MAJOR (27, n) kernel map.  Other shapes / dtypes: CDSEG_ERR_UNSUPPORTED (use cdseg_gemm with nbr). */
size_t cdseg_subm_conv3_wimg_bytes(int channels);
int cdseg_subm_conv3_pack(const void* w, int channels, void* wimg, void* stream);
int cdseg_subm_conv3(const void* x, int ldx, const void* wimg, const float* bias, const int32_t* nbr_kmajor, long n,
                     int channels, void* y, int ldy, void* stream);

/* ------------------------------------------------------------------ stem (k = 5) without a materialised kernel map
 * ref: ptv3.py:633-663 (Embedding: SubMConv3d(c_in -> 32, k = 5, bias = False) + BatchNorm1d(eps 1e-3) + GELU).
 * The <= 125 neighbours of a point are enumerated through the next coarser level (27 parent cells x <= 8 children);
 * bf16 operands, fp32 accumulation (csrc/stem.hip).  x8 (n, 8) bf16 rows in physical order (channels zero-padded),
 * wimg = cdseg_stem5_pack image (MFMA fragment order, cdseg_stem5_wimg_bytes) of the (32, 125 * 8) 16-bit weight, scale / shift = folded BatchNorm, grid (n, 3),
 * cluster (n) parent of every point, parent_nbr3 (27, m) offset-major 3x3x3 map of the parent level,
 * child_info (m) from cdseg_child_info(fine z-codes, children runs).  out (n, 32) fp32, out2 (n, 32) bf16 or NULL. */
int cdseg_child_info(const int64_t* zcode_sorted, const int32_t* seg_start, long m, int64_t* info, void* stream);
size_t cdseg_stem5_wimg_bytes(void);
int cdseg_stem5_pack(const void* w, void* wimg, void* stream);
int cdseg_stem5(const void* x8, const void* wimg, const float* scale, const float* shift, const int32_t* grid,
                const int32_t* cluster, const int32_t* parent_nbr3, const int64_t* child_info, long n, long m, int depth,
                float* out, void* out2, void* stream);

/* ------------------------------------------------------------------ Block head / tail on weight images (16-bit trunk)
 * Same contracts as cdseg_cpe_head_fused / cdseg_attn_tail_fused (ptv3.py:401-427), two machine mappings chosen by the
 * channel count:
 *   C = 32 / 64   (csrc/blockrr.hip) all weights of the kernel resident in LDS as MFMA fragments, a wave owns 32 points,
 *                 the activations never leave registers;
 *   C = 128 / 256 (csrc/deep.hip) the activations of a 128- (or 32-) row tile resident in LDS for the whole chain of products,
 *                 the weights streamed L2 -> registers: a wave owns 32 output channels of every product, its fragments are
 *                 stored in the image in the order it consumes them.
 * Images: cdseg_block_rr_img_bytes(C, 0 = head | 1 = tail) bytes; cdseg_block_rr_pack builds them once per Block from the
 * 16-bit row-major weights (head: cpe linear (C,C), qkv (3C,C); tail: proj (C,C), fc1 (4C,C), fc2 (C,4C)); either image
 * pointer may be NULL.  Other channel counts: CDSEG_ERR_UNSUPPORTED (cdseg_gemm + cdseg_layernorm sequences). */
size_t cdseg_block_rr_img_bytes(int channels, int which);
int cdseg_block_rr_pack(int channels, const void* wl, const void* wqkv, void* head_img, const void* wp, const void* w1,
                        const void* w2, void* tail_img, void* stream);
int cdseg_cpe_head_rr(const void* y, int ldy, const void* head_img, const float* bl, const float* lnp_g, const float* lnp_b,
                      float* x, int ldx, const float* colbias, const float* ln1_g, const float* ln1_b, float eps,
                      const float* bqkv, void* qkv, int ldqkv, long n, int channels, int qkv_flags /* as above; deep stages only */,
                      void* stream);
int cdseg_attn_tail_rr(const void* o, int ldo, const void* tail_img, const float* bp, const float* ln_g, const float* ln_b,
                       float eps, const float* b1, const float* b2, float* x, int ldx, void* xc, int ldxc, long n,
                       int channels, void* stream);
/* Deep stages only (C = 128 / 256 / 512): the same two launches with the residual rows READ from one buffer (x / x_in) and
 * WRITTEN to another (x_out / x; they may alias, which gives the in-place forms above).  With distinct buffers, launches of
 * few rows (a single scene's deep stages: 32-row tiles on a fraction of the CUs) cut a tile's weight stream over several
 * workgroups - the head one per q / k / v column block, the tail by MLP hidden chunks through the fp32 workspace `ws`
 * (>= 4 * n * C * 4 bytes for the four-way split, 16-byte aligned; NULL / smaller: fewer or no splits) and a fixed-order
 * reduce launch.  Results do not depend on scheduling; they differ from the unsplit form by fp32 summation order only.
 * (ref: the Block's cpe linear + norms + qkv, ptv3.py:401-414, and proj + norm2 + MLP, ptv3.py:416-427) */
int cdseg_cpe_head_rr2(const void* y, int ldy, const void* head_img, const float* bl, const float* lnp_g, const float* lnp_b,
                       const float* x, int ldx, float* x_out, int ldx_out, const float* colbias, const float* ln1_g,
                       const float* ln1_b, float eps, const float* bqkv, void* qkv, int ldqkv, long n, int channels,
                       int qkv_flags, void* stream);
int cdseg_attn_tail_rr2(const void* o, int ldo, const void* tail_img, const float* bp, const float* ln_g, const float* ln_b,
                        float eps, const float* b1, const float* b2, const float* x_in, int ldx_in, float* x, int ldx, void* xc,
                        int ldxc, long n, int channels, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ native Block executor
 * One PTv3 Block (ref: ptv3.py:399-428, eval mode) per call: the library issues every launch of the
 * block itself, carving its temporaries from the caller's scratch buffer: sparse-conv CPE, then for the 16-bit trunk with
 * C = 32 / 64 the fused head (cdseg_cpe_head_fused), window attention and the fused tail (cdseg_attn_tail_rr), with
 * C = 128 / 256 and head_img / tail_img given the deep-stage head and tail (cdseg_cpe_head_rr / cdseg_attn_tail_rr);
 * otherwise Linear+LayerNorms, QKV, attention, proj, MLP as separate launches.
 * Replaces ~10 host round trips through the binding by one.  Weights are described once (cdseg_block_desc), the per-scene tensors per call
 * (cdseg_block_io). */
typedef struct cdseg_block_desc {
  int dtype;       /* compute dtype T of weights / operand activations */
  int channels;    /* C */
  int heads;       /* C / 16 */
  int hidden;      /* MLP hidden width */
  float attn_scale;
  float ln_eps;
  const void* cpe_conv_w;  /* (C, 27*C) T */
  const float* cpe_conv_b;
  const void* cpe_lin_w;   /* (C, C) T */
  const float* cpe_lin_b;
  const float* cpe_ln_g;
  const float* cpe_ln_b;
  const float* norm1_g;
  const float* norm1_b;
  const void* qkv_w;       /* (3C, C) T */
  const float* qkv_b;
  const void* proj_w;      /* (C, C) T */
  const float* proj_b;
  const float* norm2_g;
  const float* norm2_b;
  const void* fc1_w;       /* (hidden, C) T */
  const float* fc1_b;
  const void* fc2_w;       /* (C, hidden) T */
  const float* fc2_b;
  const void* cpe_conv_wimg; /* cdseg_subm_conv3_pack image of cpe_conv_w, or NULL: the conv runs on cdseg_gemm */
  const void* head_img;      /* cdseg_block_rr_pack images (C = 32 / 64 / 128 / 256), or NULL: the unfused / tile-fused launches */
  const void* tail_img;
  int attn_flags;            /* CDSEG_ATTN_Q_PRESCALED when qkv_w / qkv_b (and the images built from them) carry
                                attn_scale * log2(e) in their q rows (the engine folds it at load time) */
} cdseg_block_desc;

typedef struct cdseg_block_io {
  long n;                  /* points at this stage */
  float* x;                /* (n, C) fp32 residual stream, updated in place */
  const void* xc_in;       /* (n, C) T: CPE conv input (sparse_conv_feat of the reference) */
  void* xc_out;            /* (n, C) T: shadow copy of the block output (may alias x when T is fp32) */
  const float* tbias;      /* (C) timestep bias or NULL */
  const int32_t* nbr;      /* (27, n) OFFSET-MAJOR kernel map of the stage (cdseg_nbr_table with kmajor = 1) */
  const int32_t* gidx;     /* attention slot plan of the block's curve */
  const int32_t* widx;
  const int32_t* patch_start;
  int num_patches;
  int max_len;
  void* scratch;           /* >= cdseg_block_scratch_bytes(desc, n) */
  size_t scratch_bytes;
  unsigned long long* sat_counter; /* NULL, or (diagnostic, IEEE-half build): += clamped values seen in this Block's 16-bit
                                      buffers that reach memory (cdseg_count_saturated); costs a few extra launches */
} cdseg_block_io;

size_t cdseg_block_scratch_bytes(const cdseg_block_desc* desc, long n);
int cdseg_block_forward(const cdseg_block_desc* desc, const cdseg_block_io* io, void* stream);

/* ------------------------------------------------------------------ native plan builder (round 6)
 * ref: models/utils/structure.py:47-102 (Point.serialization), ptv3.py:188-250 (padding plan), :477-505 (pooling structure).
 * The integer side of one forward - (batch | z) sort, level-0 codes, all pooled levels, links, 3x3x3 kernel maps, curve orders,
 * padding tables, slot plans - in TWO library calls around the forward's one host read (the pooled sizes), written into two
 * caller-owned arenas (int32 / int64 elements; every item starts on a 256-byte boundary).  Same kernels, same results as the
 * per-op entry points above (tests/test_gpu_e2e.py::test_native_plan_equals_per_op_plan); exists because a single scene's
 * plan phase was bound by ~40 binding round trips, not by the device (csrc/plan.hip).
 * Level indices: 0 = the input resolution, 1 .. nlev the pooled levels in ascending cumulative pooling depth. */
#define CDSEG_PLAN_MAX_LEVELS 8
#define CDSEG_PLAN_MAX_LINKS 16
#define CDSEG_PLAN_MAX_PADS 4
typedef struct cdseg_plan_spec {
  int nlev;            /* pooled levels (1 .. 8) */
  int cum[9];          /* cumulative pooling depth per level index, cum[0] = 0, strictly ascending */
  int ncurve;          /* curves other than z in use (0 .. 3) */
  int curve_rows[3];   /* their rows of the (4, n) code array: 1 = z-trans, 2 = hilbert, 3 = hilbert-trans */
  int nslot_curve;     /* curves that get slot plans (1 .. 4) */
  int slot_curve[4];   /* -1 = z (identity order), else an index into curve_rows */
  int nlink;           /* links a -> b between POOLED levels (1 <= a < b); the (0, l) links come out of the pooling pass.
                          Must contain (l, l + 1) for every pooled level l whose parent is one octree step up */
  int link_a[CDSEG_PLAN_MAX_LINKS];
  int link_b[CDSEG_PLAN_MAX_LINKS];
  int npad;            /* padding keys (1 .. 4) */
  int pad_patch[CDSEG_PLAN_MAX_PADS]; /* patch size */
  int pad_flash[CDSEG_PLAN_MAX_PADS]; /* enable_flash: K = patch size; else K = min(smallest batch element, patch size) */
} cdseg_plan_spec;

typedef struct cdseg_plan_begin_io {
  const void* grid;        /* (n, 3) int32 | int64 voxel coordinates, caller's point order */
  int grid_elem_bytes;
  const int64_t* offset;   /* (nb) cumulative batch offsets */
  int nb;
  long n;
  int depth;               /* serialization depth (structure.py:66) */
  int end_bit;             /* significant bits of the (batch | z) code */
  int32_t* i32;            /* arenas: cdseg_plan_begin_layout */
  int64_t* i64;
  void* ws;
  size_t ws_bytes;
  int64_t* gmax_host;      /* PINNED host word <- max(grid), or NULL */
  int32_t* meta_host;      /* PINNED host ints <- cdseg_pool_levels' meta (nlev * (1 + nb) + 1), or NULL */
} cdseg_plan_begin_io;

/* off_out[13]: offsets (elements) of batch, perm0, grid0, bat0, last_idx, cluster (nlev, n), seg_start (nlev, n + 1), meta,
 * orders0 (ncurve, n) in the int32 arena, then gmax, zcode, zcode_sorted, code0 (4, n) in the int64 arena.
 * totals_out[3]: int32 elements, int64 elements, workspace bytes. */
int cdseg_plan_begin_layout(const cdseg_plan_spec* spec, long n, int nb, long* off_out, long* totals_out);
/* phase 0: everything up to the two asynchronous host copies; phase 1: the level-0 orders of the non-z curves (one sort) -
 * independent of the host read, so the caller records its event between the phases and waits for it after phase 1 is queued */
int cdseg_plan_begin(const cdseg_plan_spec* spec, const cdseg_plan_begin_io* io, int phase, void* stream);

typedef struct cdseg_plan_finish_io {
  long n;
  int nb;
  int depth;
  const long* m_host;      /* (nlev) pooled sizes, from meta */
  const int* offs_host;    /* (nlev + 1, nb + 1) batch offsets of every level index, level 0 first */
  const int32_t* grid0;    /* items of the begin arena */
  const int32_t* bat0;
  const int64_t* code0;
  const int32_t* cluster;
  const int32_t* seg;
  const int32_t* orders0;
  int32_t* i32;            /* arenas: cdseg_plan_finish_layout */
  int64_t* i64;
  void* ws;
  size_t ws_bytes;
  int32_t* pads_host;      /* PINNED staging buffer for the padding tables (>= info_out[3] ints), free again once the call's
                              copy has run */
} cdseg_plan_finish_io;

/* off_out (element offsets), in this order: per pooled level l = 1 .. nlev: grid (int32 arena), batch (int32), code4 (int64);
 * per link: cluster, seg_start (int32); per level 0 .. nlev: nbr3 (27, m_l) offset-major (int32); per level 0 .. nlev:
 * child_info (int64) or -1; the coarse-order base (int32: level l / curve c at base + ncurve * sum_{1 <= l' < l} m_l' + c * m_l);
 * per (level 0 .. nlev, pad key): offs, offs_pad, patch_start (int32); slot gidx base, slot widx base (int32: the plan of
 * (level, pad key, slot curve), in that nesting, starts at the sum of the n_pad of the plans before it).
 * info_out: int32 elements, int64 elements, workspace bytes, padding-table ints, padding-table base; then per (level, pad key):
 * K, n_pad, patches, longest patch, sum of squared patch lengths (a double's bits). */
int cdseg_plan_finish_layout(const cdseg_plan_spec* spec, long n, int nb, const long* m_host, const int* offs_host,
                             long* off_out, long* info_out);
int cdseg_plan_finish(const cdseg_plan_spec* spec, const cdseg_plan_finish_io* io, void* stream);

/* ------------------------------------------------------------------ training path, first slice (exact fp32)
 * ref: pointcept/models/default.py:424-493 (training forward), pointcept/engines/train.py:216-271 (loss.backward());
 *      what autograd differentiates: ptv3.py:246-296 (SerializedAttention core), ptv3.py:399-428 (Block tail).
 * cdseg_attention_bwd: gradients of cdseg_attention's inputs.  dout (rows, H*16) is the gradient of its output; dq / dk /
 *   dv are ACCUMULATED into (+=, the caller zeroes them) at the gathered rows - a point that the padding plan put into
 *   two slots collects both (the backward of the reference's `qkv[order]` gather); slots without an output row (widx -1)
 *   receive no output gradient but still act as keys.  num_slots = patch_start[num_patches]; max_len = the longest patch,
 *   <= 1024 (the patch-head lives in LDS: longer -> CDSEG_ERR_UNSUPPORTED); rows 16-byte aligned (else CDSEG_ERR_ARG).
 *   ws: cdseg_attention_bwd_ws_bytes.  dtype: CDSEG_F32 only so far.
 * cdseg_layernorm_bwd: dx (=, or += when accumulate) for y = LayerNorm(x) * gamma + beta; optional dgamma / dbeta (+=).
 * cdseg_gelu_bwd: dx = dy * d/du GELU(u) on the pre-activation u (erf form, torch.nn.GELU()).
 * cdseg_linear_wgrad: dw[n][k] += sum_m dy[m][n] * x[row(m)][k] and (db != NULL) db[n] += sum_m dy[m][n], fp32; row(m) = m,
 *   or xidx[m] with -1 = skip: ONE kernel offset of a submanifold conv (xidx = that offset's row of the offset-major kernel
 *   map, dw = the offset's (Cout, Cin) slice of the (Cout, 27, Cin) weight, lddw = 27 * Cin; ref: spconv.SubMConv3d weight
 *   gradient, ptv3.py:356-362).  n, k multiples of 16.  Accumulates (the caller zeroes dw / db); partial sums are added
 *   with fp32 atomics (summation order not fixed).  The DATA gradient of a Linear / of the conv needs no entry point of
 *   its own: it is cdseg_gemm on the transposed weight / on the mirrored, transposed kernel (W'[ci][o][co] = W[co][26-o][ci])
 *   with the same kernel map. */
size_t cdseg_attention_bwd_ws_bytes(long num_slots, int num_heads);
int cdseg_attention_bwd(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, const int32_t* q_gidx,
                        const int32_t* kv_gidx, const int32_t* widx, const int32_t* patch_start, int num_patches,
                        int num_heads, long num_slots, int max_len, float scale, const void* dout, int lddo, void* dq,
                        void* dk, void* dv, int lddq, int lddk, int lddv, int dtype, void* ws, size_t ws_bytes, void* stream);
int cdseg_layernorm_bwd(const float* x, int ldx, const float* gamma, float eps, const float* dy, int lddy, float* dx, int lddx,
                        int accumulate, float* dgamma, float* dbeta, long m, int c, void* stream);
int cdseg_gelu_bwd(const float* u, const float* dy, float* dx, long n, void* stream);
int cdseg_linear_wgrad(const float* x, int ldx, const int32_t* xidx, const float* dy, int lddy, long m, int k, int n, float* dw,
                       int lddw, float* db, void* stream);
/* all kvol offsets of a submanifold conv in one launch: dw (cout, kvol, cin) += ..., db (cout) += column sums of dy;
 * nbr_kmajor: the (kvol, m) offset-major kernel map */
int cdseg_conv_wgrad(const float* x, int ldx, const int32_t* nbr_kmajor, int kvol, const float* dy, int lddy, long m, int cin,
                     int cout, float* dw, float* db, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CDSEG_H */
