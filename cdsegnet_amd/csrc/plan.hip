// Native plan builder (round 6): the integer side of one forward - serialization, pooled levels, links, kernel maps, curve
// orders, padding tables and slot plans - issued from C++ in TWO library calls around the forward's one host read, into two
// caller-owned arenas whose layout this file defines (include/cdseg.h "native plan builder").
//
// ref: pointcept/models/utils/structure.py:47-102 (Point.serialization: depth, codes, orders),
//      pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py:188-250 (padding), :477-505 (pooling structure)
//
// Why: bs = 1 is the reference's protocol (tools/test_time.py) and there the device idled ~0.5 ms per scene behind ~40 binding
// round trips (10 us each for a 3 us launch) in the plan phase - the Block phase is device-bound, the plan phase was host-bound
// (profiles/r06_bs1_host_timeline.txt).  The kernels are the ones the per-op entry points launch; nothing new runs on the device
// except the two-line last_idx kernel.
#include <cstring>

#include "common.h"

namespace {

constexpr long ALIGN_I32 = 64;  // every arena item starts on a 256-byte boundary (the kernels' 16-byte accesses, cache lines)
constexpr long ALIGN_I64 = 32;

inline long up(long v, long a) { return (v + a - 1) / a * a; }

struct Cursor {
  long i32 = 0, i64 = 0;
  long take32(long count) { const long o = i32; i32 = up(i32 + (count > 0 ? count : 1), ALIGN_I32); return o; }
  long take64(long count) { const long o = i64; i64 = up(i64 + (count > 0 ? count : 1), ALIGN_I64); return o; }
};

__global__ void last_idx_kernel(const int64_t* __restrict__ offset, int nb, int32_t* __restrict__ last) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) last[b] = (int32_t)(offset[b] - 1);
}

bool spec_ok(const cdseg_plan_spec* s) {
  if (!s || s->nlev < 1 || s->nlev > CDSEG_PLAN_MAX_LEVELS) return false;
  if (s->ncurve < 0 || s->ncurve > 3 || s->nslot_curve < 1 || s->nslot_curve > 4) return false;
  if (s->nlink < 0 || s->nlink > CDSEG_PLAN_MAX_LINKS || s->npad < 1 || s->npad > CDSEG_PLAN_MAX_PADS) return false;
  if (s->cum[0] != 0) return false;
  for (int l = 1; l <= s->nlev; ++l)
    if (s->cum[l] <= s->cum[l - 1]) return false;
  for (int c = 0; c < s->ncurve; ++c)
    if (s->curve_rows[c] < 1 || s->curve_rows[c] > 3) return false;
  for (int c = 0; c < s->nslot_curve; ++c)
    if (s->slot_curve[c] < -1 || s->slot_curve[c] >= s->ncurve) return false;
  for (int k = 0; k < s->nlink; ++k)
    if (s->link_a[k] < 1 || s->link_b[k] <= s->link_a[k] || s->link_b[k] > s->nlev) return false;
  for (int p = 0; p < s->npad; ++p)
    if (s->pad_patch[p] <= 0) return false;
  return true;
}

int find_link(const cdseg_plan_spec* s, int a, int b) {
  for (int k = 0; k < s->nlink; ++k)
    if (s->link_a[k] == a && s->link_b[k] == b) return k;
  return -1;
}

// ---- begin arena: item order of cdseg_plan_begin_layout
enum { B_BATCH, B_PERM0, B_GRID0, B_BAT0, B_LAST, B_CLUSTER, B_SEG, B_META, B_ORDERS0, B_GMAX, B_ZC, B_ZS, B_CODE0, B_ITEMS };

void begin_layout(const cdseg_plan_spec* s, long n, int nb, long* off, long* totals) {
  Cursor c;
  off[B_BATCH] = c.take32(n);
  off[B_PERM0] = c.take32(n);
  off[B_GRID0] = c.take32(3 * n);
  off[B_BAT0] = c.take32(n);
  off[B_LAST] = c.take32(nb);
  off[B_CLUSTER] = c.take32((long)s->nlev * n);
  off[B_SEG] = c.take32((long)s->nlev * (n + 1));
  off[B_META] = c.take32((long)s->nlev * (1 + nb) + 1);
  off[B_ORDERS0] = c.take32((long)s->ncurve * n);
  off[B_GMAX] = c.take64(1);
  off[B_ZC] = c.take64(n);
  off[B_ZS] = c.take64(n);
  off[B_CODE0] = c.take64(4 * n);
  totals[0] = c.i32;
  totals[1] = c.i64;
  size_t ws = cdseg_sort_ws_bytes(n);
  const size_t a = cdseg_pool_levels_ws_bytes(n, s->nlev);
  if (a > ws) ws = a;
  if (s->ncurve > 0) {
    const size_t b = cdseg_sort_curves_ws_bytes(n, s->ncurve);
    if (b > ws) ws = b;
  }
  totals[2] = (long)ws;
}

// ---- padding plan of one level / key on the host (ref: ptv3.py:188-250), the arithmetic of engine.Level.pad_host_py
struct PadInfo {
  int K;
  long n_pad;
  long npatch;
  long max_len;
  double sum_l2;
};

// counts offs_pad (nb + 1) and patch_start (npatch + 1) into `dst` when it is not NULL; returns false on a degenerate K
bool pad_host(const int* offs, int nb, int patch, int flash, PadInfo& pi, int32_t* dst_offs_pad, int32_t* dst_patch_start) {
  int K = patch;
  if (!flash) {
    int mn = offs[1] - offs[0];
    for (int b = 1; b < nb; ++b) mn = (offs[b + 1] - offs[b]) < mn ? (offs[b + 1] - offs[b]) : mn;
    K = mn < patch ? mn : patch;
  }
  if (K <= 0) return false;
  long pos = 0, np = 0, mx = 0;
  double s2 = 0.0;
  if (dst_offs_pad) dst_offs_pad[0] = 0;
  for (int b = 0; b < nb; ++b) {
    const long c = offs[b + 1] - offs[b];
    const long pc = c > K ? (c + K - 1) / K * K : c;
    for (long st = pos; st < pos + pc; st += K) {
      if (dst_patch_start) dst_patch_start[np] = (int32_t)st;
      const long len = (st + K < pos + pc ? st + K : pos + pc) - st;
      // (the last patch of a batch element ends where the next one starts: its length is what is left)
      if (len > mx) mx = len;
      s2 += (double)len * (double)len;
      ++np;
    }
    pos += pc;
    if (dst_offs_pad) dst_offs_pad[b + 1] = (int32_t)pos;
  }
  if (dst_patch_start) dst_patch_start[np] = (int32_t)pos;
  pi.K = K;
  pi.n_pad = pos;
  pi.npatch = np;
  pi.max_len = mx;
  pi.sum_l2 = s2;
  return true;
}

// ---- finish arena.  off_out (longs), in this order:
//   for l = 1 .. nlev:              grid_l (i32), batch_l (i32), code4_l (i64)
//   for k < nlink:                  cluster_k (i32), seg_k (i32)
//   for l = 0 .. nlev:              nbr3_l (i32)
//   for l = 0 .. nlev:              child_info_l (i64), or -1 when level l has no parent one octree step up
//   coarse orders base (i32)        level l (>= 1) / curve c at base + ncurve * sum_{1 <= l' < l} m_l' + c * m_l
//   for l = 0 .. nlev, p < npad:    offs (i32), offs_pad (i32), patch_start (i32)  [one contiguous table: pads base = first offs]
//   slot gidx base (i32), slot widx base (i32): plan (l, p, c < nslot_curve) at base + sum of the n_pad of the plans before it
// info_out (longs): [0] i32 elements, [1] i64 elements, [2] workspace bytes, [3] pads table length (ints), [4] pads base (i32
//   offset), then per (l, p): K, n_pad, npatch, max_len, sum_l2 (the double's bits)
struct FinishLayout {
  long grid[9], batch[9], code4[9];
  long lcl[CDSEG_PLAN_MAX_LINKS], lseg[CDSEG_PLAN_MAX_LINKS];
  long nbr[9], info[9];
  long coarse;
  long p_offs[9][CDSEG_PLAN_MAX_PADS], p_offs_pad[9][CDSEG_PLAN_MAX_PADS], p_ps[9][CDSEG_PLAN_MAX_PADS];
  PadInfo pi[9][CDSEG_PLAN_MAX_PADS];
  long pads_base, pads_count;
  long gidx, widx, slots_total;
  long tot32, tot64, ws;
};

bool has_parent(const cdseg_plan_spec* s, int l) { return l < s->nlev && s->cum[l + 1] - s->cum[l] == 1; }

int finish_layout(const cdseg_plan_spec* s, long n, int nb, const long* m_host, const int* offs_host, FinishLayout& L) {
  if (!spec_ok(s) || n <= 0 || nb <= 0 || !m_host || !offs_host) return CDSEG_ERR_ARG;
  long m[9];
  m[0] = n;
  for (int l = 1; l <= s->nlev; ++l) {
    m[l] = m_host[l - 1];
    if (m[l] <= 0 || m[l] > n) return CDSEG_ERR_ARG;
  }
  Cursor c;
  for (int l = 1; l <= s->nlev; ++l) {
    L.grid[l] = c.take32(3 * m[l]);
    L.batch[l] = c.take32(m[l]);
    L.code4[l] = c.take64(4 * m[l]);
  }
  for (int k = 0; k < s->nlink; ++k) {
    L.lcl[k] = c.take32(m[s->link_a[k]]);
    L.lseg[k] = c.take32(m[s->link_b[k]] + 1);
  }
  for (int l = 0; l <= s->nlev; ++l) L.nbr[l] = c.take32(27 * m[l]);
  for (int l = 0; l <= s->nlev; ++l) {
    L.info[l] = -1;
    if (has_parent(s, l)) {
      if (l > 0 && find_link(s, l, l + 1) < 0) return CDSEG_ERR_ARG;  // the link to the parent must be in the spec
      L.info[l] = c.take64(m[l + 1]);
    }
  }
  long msum = 0;
  for (int l = 1; l <= s->nlev; ++l) msum += m[l];
  L.coarse = c.take32((long)s->ncurve * msum);
  // padding tables: one contiguous region (uploaded with one copy); items inside it are NOT aligned individually
  long cnt = 0;
  L.slots_total = 0;
  for (int l = 0; l <= s->nlev; ++l) {
    const int* offs = offs_host + (long)l * (nb + 1);
    if (offs[0] != 0 || offs[nb] != m[l]) return CDSEG_ERR_ARG;
    for (int p = 0; p < s->npad; ++p) {
      if (!pad_host(offs, nb, s->pad_patch[p], s->pad_flash[p], L.pi[l][p], nullptr, nullptr)) return CDSEG_ERR_ARG;
      L.p_offs[l][p] = cnt;
      cnt += nb + 1;
      L.p_offs_pad[l][p] = cnt;
      cnt += nb + 1;
      L.p_ps[l][p] = cnt;
      cnt += L.pi[l][p].npatch + 1;
      L.slots_total += L.pi[l][p].n_pad * s->nslot_curve;
    }
  }
  L.pads_count = cnt;
  L.pads_base = c.take32(cnt);
  for (int l = 0; l <= s->nlev; ++l)
    for (int p = 0; p < s->npad; ++p) {
      L.p_offs[l][p] += L.pads_base;
      L.p_offs_pad[l][p] += L.pads_base;
      L.p_ps[l][p] += L.pads_base;
    }
  L.gidx = c.take32(L.slots_total);
  L.widx = c.take32(L.slots_total);
  L.tot32 = c.i32;
  L.tot64 = c.i64;
  L.ws = s->ncurve > 0 ? (long)cdseg_coarse_orders_ws_bytes(n, s->nlev, s->ncurve) : 0;
  return CDSEG_OK;
}

}  // namespace

extern "C" {

int cdseg_plan_begin_layout(const cdseg_plan_spec* spec, long n, int nb, long* off_out, long* totals_out) {
  if (!spec_ok(spec) || n <= 0 || nb <= 0 || !off_out || !totals_out) return CDSEG_ERR_ARG;
  begin_layout(spec, n, nb, off_out, totals_out);
  return CDSEG_OK;
}

// phase 0: grid maximum, batch ids, z codes, the (batch | z) sort, level-0 grid / batch / codes in sorted order, all pooled
// levels' clusters + run starts + sizes, and the two asynchronous copies to the host (gmax_host, meta_host: pinned).
// phase 1: the level-0 orders of the other curves with one sort (does not depend on the host read: the caller records its
// event between the two phases and waits for it AFTER phase 1 is queued).
int cdseg_plan_begin(const cdseg_plan_spec* spec, const cdseg_plan_begin_io* io, int phase, void* stream) {
  if (!spec_ok(spec) || !io || io->n <= 0 || io->nb <= 0 || !io->grid || !io->offset || !io->i32 || !io->i64 || !io->ws)
    return CDSEG_ERR_ARG;
  if (io->depth < 0 || io->depth > 16 || io->end_bit <= 0 || io->end_bit > 64) return CDSEG_ERR_ARG;
  const long n = io->n;
  const int nb = io->nb;
  long off[B_ITEMS], tot[3];
  begin_layout(spec, n, nb, off, tot);
  if ((long)io->ws_bytes < tot[2]) return CDSEG_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  int32_t* A = io->i32;
  int64_t* Q = io->i64;
  int rc;
  if (phase == 0) {
    if ((rc = cdseg_grid_max(io->grid, io->grid_elem_bytes, 3 * n, Q + off[B_GMAX], stream)) != CDSEG_OK) return rc;
    if (io->gmax_host &&
        hipMemcpyAsync(io->gmax_host, Q + off[B_GMAX], sizeof(int64_t), hipMemcpyDeviceToHost, s) != hipSuccess)
      return CDSEG_ERR_LAUNCH;
    hipLaunchKernelGGL(last_idx_kernel, dim3((nb + 63) / 64), dim3(64), 0, s, io->offset, nb, A + off[B_LAST]);
    if ((rc = cdseg_offset2batch(io->offset, nb, n, A + off[B_BATCH], stream)) != CDSEG_OK) return rc;
    if ((rc = cdseg_encode(io->grid, io->grid_elem_bytes, A + off[B_BATCH], 4, n, io->depth, CDSEG_ORDER_Z, Q + off[B_ZC],
                           stream)) != CDSEG_OK)
      return rc;
    if ((rc = cdseg_sort_pairs(Q + off[B_ZC], Q + off[B_ZS], nullptr, A + off[B_PERM0], n, io->end_bit, io->ws, io->ws_bytes,
                               stream)) != CDSEG_OK)
      return rc;
    if ((rc = cdseg_plan_gather_grid(io->grid, io->grid_elem_bytes, A + off[B_PERM0], Q + off[B_ZS], n, io->depth,
                                     A + off[B_GRID0], A + off[B_BAT0], stream)) != CDSEG_OK)
      return rc;
    if ((rc = cdseg_encode4(A + off[B_GRID0], A + off[B_BAT0], n, io->depth, Q + off[B_CODE0], stream)) != CDSEG_OK) return rc;
    int shifts[CDSEG_PLAN_MAX_LEVELS];
    for (int l = 1; l <= spec->nlev; ++l) shifts[l - 1] = 3 * spec->cum[l];
    if ((rc = cdseg_pool_levels(Q + off[B_ZS], n, shifts, spec->nlev, A + off[B_LAST], nb, A + off[B_CLUSTER], A + off[B_SEG],
                                A + off[B_META], io->ws, io->ws_bytes, stream)) != CDSEG_OK)
      return rc;
    if (io->meta_host && hipMemcpyAsync(io->meta_host, A + off[B_META], ((size_t)spec->nlev * (1 + nb) + 1) * sizeof(int32_t),
                                        hipMemcpyDeviceToHost, s) != hipSuccess)
      return CDSEG_ERR_LAUNCH;
    CDSEG_CHECK_LAUNCH();
    return CDSEG_OK;
  }
  if (phase == 1) {
    if (spec->ncurve == 0) return CDSEG_OK;
    return cdseg_sort_curves(Q + off[B_CODE0], spec->curve_rows, spec->ncurve, n, io->end_bit, A + off[B_ORDERS0], io->ws,
                             io->ws_bytes, stream);
  }
  return CDSEG_ERR_ARG;
}

int cdseg_plan_finish_layout(const cdseg_plan_spec* spec, long n, int nb, const long* m_host, const int* offs_host,
                             long* off_out, long* info_out) {
  if (!off_out || !info_out) return CDSEG_ERR_ARG;
  static thread_local FinishLayout L;
  const int rc = finish_layout(spec, n, nb, m_host, offs_host, L);
  if (rc != CDSEG_OK) return rc;
  long* o = off_out;
  for (int l = 1; l <= spec->nlev; ++l) { *o++ = L.grid[l]; *o++ = L.batch[l]; *o++ = L.code4[l]; }
  for (int k = 0; k < spec->nlink; ++k) { *o++ = L.lcl[k]; *o++ = L.lseg[k]; }
  for (int l = 0; l <= spec->nlev; ++l) *o++ = L.nbr[l];
  for (int l = 0; l <= spec->nlev; ++l) *o++ = L.info[l];
  *o++ = L.coarse;
  for (int l = 0; l <= spec->nlev; ++l)
    for (int p = 0; p < spec->npad; ++p) { *o++ = L.p_offs[l][p]; *o++ = L.p_offs_pad[l][p]; *o++ = L.p_ps[l][p]; }
  *o++ = L.gidx;
  *o++ = L.widx;
  long* q = info_out;
  *q++ = L.tot32; *q++ = L.tot64; *q++ = L.ws; *q++ = L.pads_count; *q++ = L.pads_base;
  for (int l = 0; l <= spec->nlev; ++l)
    for (int p = 0; p < spec->npad; ++p) {
      const PadInfo& pi = L.pi[l][p];
      *q++ = pi.K; *q++ = pi.n_pad; *q++ = pi.npatch; *q++ = pi.max_len;
      long bits;
      static_assert(sizeof(long) == sizeof(double), "the double travels as its bits");
      std::memcpy(&bits, &pi.sum_l2, sizeof(bits));
      *q++ = bits;
    }
  return CDSEG_OK;
}

int cdseg_plan_finish(const cdseg_plan_spec* spec, const cdseg_plan_finish_io* io, void* stream) {
  if (!io || !io->grid0 || !io->bat0 || !io->code0 || !io->cluster || !io->seg || !io->i32 || !io->i64 || !io->pads_host)
    return CDSEG_ERR_ARG;
  if (spec && spec->ncurve > 0 && (!io->orders0 || !io->ws)) return CDSEG_ERR_ARG;
  static thread_local FinishLayout L;
  int rc = finish_layout(spec, io->n, io->nb, io->m_host, io->offs_host, L);
  if (rc != CDSEG_OK) return rc;
  if ((long)io->ws_bytes < L.ws) return CDSEG_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const long n = io->n;
  const int nb = io->nb, nlev = spec->nlev;
  int32_t* A = io->i32;
  int64_t* Q = io->i64;
  long m[9];
  m[0] = n;
  for (int l = 1; l <= nlev; ++l) m[l] = io->m_host[l - 1];
  const int32_t* grid[9];
  const int32_t* batch[9];
  const int64_t* code4[9];
  grid[0] = io->grid0; batch[0] = io->bat0; code4[0] = io->code0;
  // pooled levels: grid / batch / the four codes of every pooled voxel (first child's, shifted)
  for (int l = 1; l <= nlev; ++l) {
    const int32_t* seg0l = io->seg + (long)(l - 1) * (n + 1);
    if ((rc = cdseg_pool_gather(seg0l, m[l], n, spec->cum[l], io->grid0, io->bat0, io->code0, A + L.grid[l], A + L.batch[l],
                                Q + L.code4[l], stream)) != CDSEG_OK)
      return rc;
    grid[l] = A + L.grid[l]; batch[l] = A + L.batch[l]; code4[l] = Q + L.code4[l];
  }
  // links between pooled levels from their links to level 0
  for (int k = 0; k < spec->nlink; ++k) {
    const int a = spec->link_a[k], b = spec->link_b[k];
    if ((rc = cdseg_link_derive(io->cluster + (long)(a - 1) * n, io->seg + (long)(a - 1) * (n + 1), m[a],
                                io->cluster + (long)(b - 1) * n, io->seg + (long)(b - 1) * (n + 1), m[b], A + L.lcl[k],
                                A + L.lseg[k], stream)) != CDSEG_OK)
      return rc;
  }
  // kernel maps, top-down: the coarsest level (and any level without a parent one octree step up) by search, every other from
  // its parent's map through the parents' child_info words
  for (int l = nlev; l >= 0; --l) {
    const int dl = io->depth - spec->cum[l];
    if (has_parent(spec, l)) {
      const int32_t *cl, *sg;
      if (l == 0) { cl = io->cluster; sg = io->seg; }
      else { const int k = find_link(spec, l, l + 1); cl = A + L.lcl[k]; sg = A + L.lseg[k]; }
      if ((rc = cdseg_child_info(code4[l], sg, m[l + 1], Q + L.info[l], stream)) != CDSEG_OK) return rc;
      if ((rc = cdseg_nbr_table_from_info(grid[l], cl, A + L.nbr[l + 1], Q + L.info[l], m[l], m[l + 1], dl, 3, 1, A + L.nbr[l],
                                          stream)) != CDSEG_OK)
        return rc;
    } else {
      if ((rc = cdseg_nbr_table(code4[l], grid[l], batch[l], m[l], dl, 3, 1, A + L.nbr[l], stream)) != CDSEG_OK) return rc;
    }
  }
  // curve orders of the pooled levels, derived from the level-0 orders
  if (spec->ncurve > 0) {
    const int32_t* cls[CDSEG_PLAN_MAX_LEVELS];
    const int32_t* ords[3];
    for (int l = 1; l <= nlev; ++l) cls[l - 1] = io->cluster + (long)(l - 1) * n;
    for (int c = 0; c < spec->ncurve; ++c) ords[c] = io->orders0 + (long)c * n;
    if ((rc = cdseg_coarse_orders(cls, nlev, ords, spec->ncurve, n, A + L.coarse, io->ws, io->ws_bytes, stream)) != CDSEG_OK)
      return rc;
  }
  // padding tables: written into the caller's pinned staging buffer, one copy up
  int32_t* ph = io->pads_host;
  for (int l = 0; l <= nlev; ++l) {
    const int* offs = io->offs_host + (long)l * (nb + 1);
    for (int p = 0; p < spec->npad; ++p) {
      PadInfo pi;
      int32_t* d_offs = ph + (L.p_offs[l][p] - L.pads_base);
      for (int b = 0; b <= nb; ++b) d_offs[b] = offs[b];
      pad_host(offs, nb, spec->pad_patch[p], spec->pad_flash[p], pi, ph + (L.p_offs_pad[l][p] - L.pads_base),
               ph + (L.p_ps[l][p] - L.pads_base));
    }
  }
  if (hipMemcpyAsync(A + L.pads_base, ph, (size_t)L.pads_count * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess)
    return CDSEG_ERR_LAUNCH;
  // slot plans: (level, pad key, curve) in this order, CDSEG_PAD_BATCH_MAX per launch
  {
    const int32_t* orders[CDSEG_PAD_BATCH_MAX];
    const int32_t* offs[CDSEG_PAD_BATCH_MAX];
    const int32_t* offs_pad[CDSEG_PAD_BATCH_MAX];
    int patch[CDSEG_PAD_BATCH_MAX];
    long npad[CDSEG_PAD_BATCH_MAX];
    int cnt = 0;
    long done = 0, chunk = 0, mprev = 0;
    auto flush = [&]() -> int {
      if (cnt == 0) return CDSEG_OK;
      const int r = cdseg_pad_plan_batch(cnt, orders, offs, offs_pad, patch, npad, nb, A + L.gidx + done, A + L.widx + done, stream);
      done += chunk;
      cnt = 0;
      chunk = 0;
      return r;
    };
    for (int l = 0; l <= nlev; ++l) {
      for (int p = 0; p < spec->npad; ++p)
        for (int c = 0; c < spec->nslot_curve; ++c) {
          const int ci = spec->slot_curve[c];
          orders[cnt] = ci < 0 ? nullptr
                               : (l == 0 ? io->orders0 + (long)ci * n : A + L.coarse + (long)spec->ncurve * mprev + (long)ci * m[l]);
          offs[cnt] = A + L.p_offs[l][p];
          offs_pad[cnt] = A + L.p_offs_pad[l][p];
          patch[cnt] = L.pi[l][p].K;
          npad[cnt] = L.pi[l][p].n_pad;
          chunk += L.pi[l][p].n_pad;
          if (++cnt == CDSEG_PAD_BATCH_MAX && (rc = flush()) != CDSEG_OK) return rc;
        }
      if (l >= 1) mprev += m[l];
    }
    if ((rc = flush()) != CDSEG_OK) return rc;
  }
  CDSEG_CHECK_LAUNCH();
  return CDSEG_OK;
}

}  // extern "C"
