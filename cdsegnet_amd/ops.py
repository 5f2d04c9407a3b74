"""Torch-tensor wrappers over the C ABI (include/cdseg.h).

Each function mirrors one reference call site (cited in include/cdseg.h) and is a thin
marshalling layer: PyTorch provides device memory and the stream, every computation runs in
the hand-written HIP kernels of libcdseg_hip.so.  No fallbacks: tensors must live on a GPU.
"""
import ctypes
import os
import threading

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_SWISH, BF16, F32, GemmArgs, check

__all__ = ["ACT_NONE", "ACT_GELU", "ACT_SWISH", "F32", "BF16"]

LP_DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16}  # the 16-bit type of each build of the library


class _Dtypes:
    """torch dtype -> ABI dtype code.  A 16-bit tensor must be of the ACTIVE build's 16-bit type (`_lib.use`): the ABI
    code CDSEG_BF16 means "the build's 16-bit type", handing half bits to the bfloat16 build would compute garbage."""

    def __getitem__(self, dtype):
        if dtype == torch.float32:
            return F32
        if dtype == LP_DTYPES[_lib.active()]:
            return BF16
        raise _lib.CdsegError(f"{dtype} tensors do not belong to the active build of the library ({_lib.active()})")


_DT = _Dtypes()


def is_lp(dtype):
    return dtype in (torch.bfloat16, torch.float16)


_WS = {}


class _LazyLib:
    """Attribute access loads the library on first use (and raises loudly if it is missing)."""

    def __getattr__(self, name):
        return getattr(_lib.load(), name)


_LIB = _LazyLib()


class KernelTimer:
    """HIP-event timing of selected ops on the stream they are launched on (torch's current
    stream).  bench.py installs one over the timed region to get per-kernel durations live."""

    def __init__(self, names=("attention",)):
        self.names = set(names)
        self.events = []

    def begin(self, name):
        if name not in self.names:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return (name, e0)

    def end(self, tok, work=0.0):
        if tok is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.events.append((tok[0], tok[1], e1, work))

    def summary(self):
        """name -> dict(launches, ms, work) ; call after torch.cuda.synchronize()."""
        out = {}
        for name, e0, e1, work in self.events:
            d = out.setdefault(name, dict(launches=0, ms=0.0, work=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["work"] += work
        return out


TIMER = None


def _loaded_libs():
    """Every build of the library this process has loaded (a process normally runs one precision, but the 16-bit type is
    a per-thread choice: profiling switches and summaries must not depend on which thread asks - ADVICE r3)."""
    libs = list(_lib._libs.values())
    return libs if libs else [_lib.load()]


def attention_prof_enable(on=True):
    """Event-time every attention / sparse-conv launch inside the library (works for the native Block executor too); applies
    to every loaded build."""
    for lib in _loaded_libs():
        check(lib.cdseg_prof_enable(1 if on else 0), "prof_enable")


def attention_prof_summary():
    """(total_ms, launches) since attention_prof_enable(True), summed over the loaded builds; call after torch.cuda.synchronize()."""
    return prof_summary(PROF_ATTENTION)


PROF_ATTENTION, PROF_CONV, PROF_CONV_DEEP = 0, 1, 2  # include/cdseg.h


def prof_summary(cls):
    """(total_ms, launches) of one profiled kernel class since attention_prof_enable(True), summed over the loaded builds;
    after a device sync."""
    tot_ms, tot_cnt = 0.0, 0
    for lib in _loaded_libs():
        ms, cnt = ctypes.c_double(0.0), ctypes.c_long(0)
        check(lib.cdseg_prof_summary_class(int(cls), ctypes.byref(ms), ctypes.byref(cnt)), "prof_summary_class")
        tot_ms += ms.value
        tot_cnt += cnt.value
    return tot_ms, tot_cnt


def set_timer(t):
    global TIMER
    TIMER = t


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class _ThreadState(threading.local):
    """Per host thread: the bound stream, the GEMM argument-block cache and the Block I/O struct (inference_many
    issues each lane from its own thread; ctypes structs are filled in place, so they cannot be shared)."""

    def __init__(self):
        self.stream = None      # ctypes.c_void_p of the bound HIP stream
        self.stream_obj = None  # the torch stream object
        self.gemm_cache = {}
        self.f32x3 = False  # fp32 products of this thread's calls run as three half MFMAs on split operands (set_f32x3)
        self.block_io = None


_TLS = _ThreadState()


def bind_stream(stream=None):
    """Cache the HIP stream every op of THIS THREAD launches on (torch.cuda.current_stream() costs ~8 us per query,
    and an inference issues ~500 launches).  The engine binds the current torch stream once per inference call and
    re-binds when it switches to its side stream."""
    if stream is None:
        stream = torch.cuda.current_stream()
    _TLS.stream_obj = stream
    _TLS.stream = ctypes.c_void_p(stream.cuda_stream)
    return _TLS.stream


def unbind_stream():
    _TLS.stream = _TLS.stream_obj = None


def current_stream_id():
    """Identity of the bound stream (None when nothing is bound)."""
    st = _TLS.stream
    return st.value if st is not None else None


def record_event():
    ev = torch.cuda.Event()
    ev.record(_TLS.stream_obj if _TLS.stream_obj is not None else torch.cuda.current_stream())
    return ev


def wait_event(ev):
    (_TLS.stream_obj if _TLS.stream_obj is not None else torch.cuda.current_stream()).wait_event(ev)


def _stream():
    st = _TLS.stream
    return st if st is not None else ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.CdsegError("cdsegnet_amd ops run on the GPU only (HIP kernels); got a CPU tensor")


def dt(t):
    return _DT[t.dtype]


def workspace(nbytes, device):
    """Per-device scratch buffer (grown geometrically, never shrunk)."""
    key = (device.type, device.index, current_stream_id())  # one buffer per stream
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


# ------------------------------------------------------------------ serialization
def grid_max(grid):
    _need_gpu(grid)
    grid = grid.contiguous()
    out = torch.empty(1, dtype=torch.int64, device=grid.device)
    check(_lib.load().cdseg_grid_max(_ptr(grid), grid.element_size(), grid.numel(), _ptr(out), _stream()), "grid_max")
    return out


def offset2batch(offset, n):
    _need_gpu(offset)
    offset = offset.contiguous().to(torch.int64)
    out = torch.empty(n, dtype=torch.int32, device=offset.device)
    check(_lib.load().cdseg_offset2batch(_ptr(offset), offset.numel(), n, _ptr(out), _stream()), "offset2batch")
    return out


def encode(grid, batch, depth, order):
    """serialization/default.py:8-24.  grid (N,3) int32|int64; batch None|int32|int64 -> int64 codes."""
    _need_gpu(grid, batch)
    grid = grid.contiguous()
    n = grid.shape[0]
    code = torch.empty(n, dtype=torch.int64, device=grid.device)
    b = None if batch is None else batch.contiguous()
    oid = _lib.ORDER_IDS[order] if isinstance(order, str) else int(order)
    check(_lib.load().cdseg_encode(_ptr(grid), grid.element_size(), _ptr(b), 0 if b is None else b.element_size(), n,
                                   int(depth), oid, _ptr(code), _stream()), "encode")
    return code


def encode4(grid_i32, batch_i32, depth):
    _need_gpu(grid_i32, batch_i32)
    n = grid_i32.shape[0]
    code = torch.empty((4, n), dtype=torch.int64, device=grid_i32.device)
    check(_lib.load().cdseg_encode4(_ptr(grid_i32), _ptr(batch_i32), n, int(depth), _ptr(code), _stream()), "encode4")
    return code


def sort_pairs(keys, vals=None, end_bit=64):
    """Stable radix sort of non-negative int64 keys with int32 payload (iota if vals is None)."""
    _need_gpu(keys, vals)
    lib = _lib.load()
    keys = keys.contiguous()
    n = keys.numel()
    keys_out = torch.empty_like(keys)
    vals_out = torch.empty(n, dtype=torch.int32, device=keys.device)
    nbytes = lib.cdseg_sort_ws_bytes(n)
    ws = workspace(nbytes, keys.device)
    check(lib.cdseg_sort_pairs(_ptr(keys), _ptr(keys_out), _ptr(vals), _ptr(vals_out), n, int(end_bit), _ptr(ws),
                               ws.numel(), _stream()), "sort_pairs")
    return keys_out, vals_out


def sort_curves(code4, rows, end_bit):
    """Orders (rank -> row, int32) of the curve rows ``rows`` of code4 (4, n) with ONE radix sort (the curve slot rides in
    the key bits above end_bit): returns a (len(rows), n) int32 tensor, row k = argsort of code4[rows[k]]."""
    _need_gpu(code4)
    lib = _lib.load()
    n, count = code4.shape[1], len(rows)
    orders = torch.empty((count, n), dtype=torch.int32, device=code4.device)
    if n == 0 or count == 0:
        return orders
    ws = workspace(lib.cdseg_sort_curves_ws_bytes(n, count), code4.device)
    arr = (ctypes.c_int * count)(*[int(r) for r in rows])
    check(lib.cdseg_sort_curves(_ptr(code4), arr, count, n, int(end_bit), _ptr(orders), _ptr(ws), ws.numel(), _stream()),
          "sort_curves")
    return orders


def invert_perm(perm):
    _need_gpu(perm)
    inv = torch.empty_like(perm)
    check(_lib.load().cdseg_invert_perm(_ptr(perm), perm.numel(), _ptr(inv), _stream()), "invert_perm")
    return inv


def widen(x_i32):
    out = torch.empty(x_i32.shape, dtype=torch.int64, device=x_i32.device)
    check(_lib.load().cdseg_widen_i32(_ptr(x_i32), x_i32.numel(), _ptr(out), _stream()), "widen")
    return out


def gather_rows(src, idx):
    """dst[i] = src[idx[i]] (rows), idx int32, idx<0 -> zero row."""
    _need_gpu(src, idx)
    src = src.contiguous()
    row_bytes = src[0].numel() * src.element_size() if src.dim() > 1 else src.element_size()
    out = torch.empty((idx.numel(),) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    check(_lib.load().cdseg_gather_rows(_ptr(src), _ptr(idx), idx.numel(), row_bytes, _ptr(out), _stream()),
          "gather_rows")
    return out


def scatter_rows(src, idx, out):
    _need_gpu(src, idx, out)
    src = src.contiguous()
    row_bytes = src[0].numel() * src.element_size() if src.dim() > 1 else src.element_size()
    check(_lib.load().cdseg_scatter_rows(_ptr(src), _ptr(idx), idx.numel(), row_bytes, _ptr(out), _stream()),
          "scatter_rows")
    return out


def gather_i32(src, idx):
    out = torch.empty(idx.numel(), dtype=torch.int32, device=src.device)
    check(_lib.load().cdseg_gather_i32(_ptr(src), _ptr(idx), idx.numel(), _ptr(out), _stream()), "gather_i32")
    return out


def plan_gather_grid(grid, perm, zcode_sorted, depth):
    grid = grid.contiguous()
    n = grid.shape[0]
    g = torch.empty((n, 3), dtype=torch.int32, device=grid.device)
    b = torch.empty(n, dtype=torch.int32, device=grid.device)
    check(_lib.load().cdseg_plan_gather_grid(_ptr(grid), grid.element_size(), _ptr(perm), _ptr(zcode_sorted), n,
                                             int(depth), _ptr(g), _ptr(b), _stream()), "plan_gather_grid")
    return g, b


def pool_level(zcode_sorted, shift_bits, count_out=None):
    """Clusters of z-sorted points at one pooling level -> (cluster (n), seg_start (n+1, first count+1 valid),
    count (1,) int32 on device)."""
    lib = _lib.load()
    n = zcode_sorted.numel()
    dev = zcode_sorted.device
    cluster = torch.empty(n, dtype=torch.int32, device=dev)
    seg_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
    count = count_out if count_out is not None else torch.empty(1, dtype=torch.int32, device=dev)
    ws = workspace(lib.cdseg_sort_ws_bytes(n), dev)
    check(lib.cdseg_pool_level(_ptr(zcode_sorted), n, int(shift_bits), _ptr(cluster), _ptr(seg_start), _ptr(count),
                               _ptr(ws), ws.numel(), _stream()), "pool_level")
    return cluster, seg_start, count


def pool_levels(zcode_sorted, shifts, last_idx):
    """All pooling levels at once -> (cluster (L,n), seg_start (L,n+1), meta) int32 on the device; meta is FLAT:
    L x (1+nb) ints [pooled count, cluster of each batch element's last point] then one int = duplicate-voxel count."""
    lib = _lib.load()
    n, L, nb = zcode_sorted.numel(), len(shifts), last_idx.numel()
    dev = zcode_sorted.device
    cluster = torch.empty((L, n), dtype=torch.int32, device=dev)
    seg = torch.empty((L, n + 1), dtype=torch.int32, device=dev)
    meta = torch.empty(L * (1 + nb) + 1, dtype=torch.int32, device=dev)
    ws = workspace(lib.cdseg_pool_levels_ws_bytes(n, L), dev)
    sh = (ctypes.c_int * L)(*[int(v) for v in shifts])
    check(lib.cdseg_pool_levels(_ptr(zcode_sorted), n, sh, L, _ptr(last_idx), nb, _ptr(cluster), _ptr(seg), _ptr(meta),
                                _ptr(ws), ws.numel(), _stream()), "pool_levels")
    return cluster, seg, meta


def link_derive(cl0a, seg0a, ma, cl0b, seg0b, mb):
    dev = cl0a.device
    cluster = torch.empty(int(ma), dtype=torch.int32, device=dev)
    seg = torch.empty(int(mb) + 1, dtype=torch.int32, device=dev)
    check(_lib.load().cdseg_link_derive(_ptr(cl0a), _ptr(seg0a), int(ma), _ptr(cl0b), _ptr(seg0b), int(mb),
                                         _ptr(cluster), _ptr(seg), _stream()), "link_derive")
    return cluster, seg


def coarse_orders(clusters, orders, sizes):
    """Curve orders of all pooled levels from the level-0 orders (no sort).  clusters: per level (n0) int32 cluster
    ids of the level-0 points; orders: per curve (n0) int32 rank -> level-0 point; sizes: per level m_l.
    Returns [[order of level l on curve c for c] for l] as views of one int32 buffer."""
    lib = _lib.load()
    n0 = clusters[0].numel()
    nl, nc = len(clusters), len(orders)
    dev = clusters[0].device
    out = torch.empty(nc * int(sum(sizes)), dtype=torch.int32, device=dev)
    ws = workspace(lib.cdseg_coarse_orders_ws_bytes(n0, nl, nc), dev)
    cp = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in clusters])
    op = (ctypes.c_void_p * nc)(*[t.data_ptr() for t in orders])
    check(lib.cdseg_coarse_orders(cp, nl, op, nc, n0, _ptr(out), _ptr(ws), ws.numel(), _stream()), "coarse_orders")
    parts = out.split([int(m) for m in sizes for _ in range(nc)])
    return [list(parts[i * nc:(i + 1) * nc]) for i in range(nl)]


def pool_gather(seg_start, m, n_fine, pooling_depth, grid_f, batch_f, code4_f):
    dev = grid_f.device
    grid_c = torch.empty((m, 3), dtype=torch.int32, device=dev)
    batch_c = torch.empty(m, dtype=torch.int32, device=dev)
    code_c = torch.empty((4, m), dtype=torch.int64, device=dev)
    check(_lib.load().cdseg_pool_gather(_ptr(seg_start), m, n_fine, int(pooling_depth), _ptr(grid_f), _ptr(batch_f),
                                        _ptr(code4_f), _ptr(grid_c), _ptr(batch_c), _ptr(code_c), _stream()),
          "pool_gather")
    return grid_c, batch_c, code_c


def nbr_table(zcode_sorted, grid_i32, batch_i32, depth, ksize, kmajor=False):
    n = grid_i32.shape[0]
    kv = ksize ** 3
    shape = (kv, n) if kmajor else (n, kv)
    nbr = torch.empty(shape, dtype=torch.int32, device=grid_i32.device)
    check(_lib.load().cdseg_nbr_table(_ptr(zcode_sorted), _ptr(grid_i32), _ptr(batch_i32), n, int(depth), int(ksize),
                                      1 if kmajor else 0, _ptr(nbr), _stream()), "nbr_table")
    return nbr


def nbr_table_from_parent(zcode_sorted, grid_i32, cluster, parent_nbr3, seg_start, m, depth, ksize, kmajor=False):
    """Kernel map of a level from its parent level's (27, m) offset-major 3x3x3 map (pooling depth 1)."""
    n = zcode_sorted.numel()
    kv = ksize ** 3
    nbr = torch.empty((kv, n) if kmajor else (n, kv), dtype=torch.int32, device=zcode_sorted.device)
    check(_lib.load().cdseg_nbr_table_from_parent(_ptr(zcode_sorted), _ptr(grid_i32), _ptr(cluster), _ptr(parent_nbr3),
                                                   _ptr(seg_start), n, int(m), int(depth), int(ksize),
                                                   1 if kmajor else 0, _ptr(nbr), _stream()), "nbr_table_from_parent")
    return nbr


def nbr_table_from_info(grid_i32, cluster, parent_nbr3, child_info_words, m, depth, ksize, kmajor=False):
    """The same from the parents' child_info words (first child row << 8 | octant occupancy) instead of the children runs:
    two dependent reads per (point, offset)."""
    n = grid_i32.shape[0]
    kv = ksize ** 3
    nbr = torch.empty((kv, n) if kmajor else (n, kv), dtype=torch.int32, device=grid_i32.device)
    check(_lib.load().cdseg_nbr_table_from_info(_ptr(grid_i32), _ptr(cluster), _ptr(parent_nbr3), _ptr(child_info_words), n,
                                                 int(m), int(depth), int(ksize), 1 if kmajor else 0, _ptr(nbr), _stream()),
          "nbr_table_from_info")
    return nbr


def pad_plan(order, offs, offs_pad, patch, n_pad):
    dev = offs.device
    gidx = torch.empty(n_pad, dtype=torch.int32, device=dev)
    widx = torch.empty(n_pad, dtype=torch.int32, device=dev)
    check(_lib.load().cdseg_pad_plan(_ptr(order), _ptr(offs), _ptr(offs_pad), offs.numel() - 1, int(patch), n_pad,
                                     _ptr(gidx), _ptr(widx), _stream()), "pad_plan")
    return gidx, widx


# ------------------------------------------------------------------ native plan builder (csrc/plan.hip)
class NativePlanCall:
    """One forward's plan through cdseg_plan_begin / cdseg_plan_finish: owns the two arenas of each call and hands out views.
    `spec` is a filled _lib.PlanSpec (static per model and serialization depth)."""

    def __init__(self, spec, grid, offset_dev, n, nb, depth, end_bit, gmax_pin, meta_pin):
        lib = _lib.load()
        self.lib, self.spec, self.n, self.nb, self.depth = lib, spec, int(n), int(nb), int(depth)
        self.sp = ctypes.byref(spec)
        off = (ctypes.c_long * 13)()
        tot = (ctypes.c_long * 3)()
        check(lib.cdseg_plan_begin_layout(self.sp, self.n, self.nb, off, tot), "plan_begin_layout")
        dev = grid.device
        self.b32 = torch.empty(tot[0], dtype=torch.int32, device=dev)
        self.b64 = torch.empty(tot[1], dtype=torch.int64, device=dev)
        self.ws = workspace(tot[2], dev)
        self.boff = list(off)
        io = self.bio = _lib.PlanBeginIO()
        io.grid, io.grid_elem_bytes, io.offset = grid.data_ptr(), grid.element_size(), offset_dev.data_ptr()
        io.nb, io.n, io.depth, io.end_bit = self.nb, self.n, self.depth, int(end_bit)
        io.i32, io.i64, io.ws, io.ws_bytes = self.b32.data_ptr(), self.b64.data_ptr(), self.ws.data_ptr(), self.ws.numel()
        io.gmax_host = None if gmax_pin is None else gmax_pin.data_ptr()
        io.meta_host = meta_pin.data_ptr()
        self._keep = (grid, offset_dev)

    def begin(self, phase):
        check(self.lib.cdseg_plan_begin(self.sp, ctypes.byref(self.bio), int(phase), _stream()), "plan_begin")

    def begin_views(self):
        """perm0 (n), grid0 (n, 3), bat0 (n), code0 (4, n), cluster (nlev, n), seg_start (nlev, n + 1), orders0 (ncurve, n)."""
        o, n, b32, b64, L, nc = self.boff, self.n, self.b32, self.b64, self.spec.nlev, self.spec.ncurve
        return (b32[o[1]:o[1] + n], b32[o[2]:o[2] + 3 * n].view(n, 3), b32[o[3]:o[3] + n],
                b64[o[12]:o[12] + 4 * n].view(4, n), b32[o[5]:o[5] + L * n].view(L, n),
                b32[o[6]:o[6] + L * (n + 1)].view(L, n + 1), b32[o[8]:o[8] + nc * n].view(nc, n) if nc else None)

    def finish(self, m, offs_rows, pads_pin):
        """m: pooled sizes; offs_rows: (nlev + 1) lists of nb + 1 batch offsets; pads_pin(count) -> pinned int32 tensor with
        at least `count` elements (called once).  Returns (off, info): the layout lists of include/cdseg.h."""
        lib, spec, L, npad = self.lib, self.spec, self.spec.nlev, self.spec.npad
        mh = (ctypes.c_long * L)(*m)
        flat = [v for r in offs_rows for v in r]
        oh = (ctypes.c_int * len(flat))(*flat)
        n_off = 3 * L + 2 * spec.nlink + 2 * (L + 1) + 1 + 3 * (L + 1) * npad + 2
        n_info = 5 + 5 * (L + 1) * npad
        off = (ctypes.c_long * n_off)()
        info = (ctypes.c_long * n_info)()
        check(lib.cdseg_plan_finish_layout(self.sp, self.n, self.nb, mh, oh, off, info), "plan_finish_layout")
        dev = self.b32.device
        self.f32 = torch.empty(info[0], dtype=torch.int32, device=dev)
        self.f64 = torch.empty(max(1, info[1]), dtype=torch.int64, device=dev)
        if info[2] > self.ws.numel():
            self.ws = workspace(info[2], dev)
        pin = pads_pin(info[3])
        o, n, b32, b64 = self.boff, self.n, self.b32, self.b64
        io = _lib.PlanFinishIO()
        io.n, io.nb, io.depth, io.m_host, io.offs_host = self.n, self.nb, self.depth, mh, oh
        io.grid0 = b32.data_ptr() + 4 * o[2]
        io.bat0 = b32.data_ptr() + 4 * o[3]
        io.code0 = b64.data_ptr() + 8 * o[12]
        io.cluster = b32.data_ptr() + 4 * o[5]
        io.seg = b32.data_ptr() + 4 * o[6]
        io.orders0 = b32.data_ptr() + 4 * o[8]
        io.i32, io.i64, io.ws, io.ws_bytes = self.f32.data_ptr(), self.f64.data_ptr(), self.ws.data_ptr(), self.ws.numel()
        io.pads_host = pin.data_ptr()
        check(lib.cdseg_plan_finish(self.sp, ctypes.byref(io), _stream()), "plan_finish")
        return list(off), list(info)


# ------------------------------------------------------------------ float ops
PAD_BATCH_MAX = 48


def pad_plan_batch(items, nb):
    """items: list of (order or None, offs, offs_pad, patch, n_pad) -> list of (gidx, widx), one launch."""
    lib = _lib.load()
    res = []
    for s0 in range(0, len(items), PAD_BATCH_MAX):
        chunk = items[s0:s0 + PAD_BATCH_MAX]
        k = len(chunk)
        total = sum(int(it[4]) for it in chunk)
        dev = chunk[0][1].device
        gidx = torch.empty(total, dtype=torch.int32, device=dev)
        widx = torch.empty(total, dtype=torch.int32, device=dev)
        orders = (ctypes.c_void_p * k)(*[None if it[0] is None else it[0].data_ptr() for it in chunk])
        offs = (ctypes.c_void_p * k)(*[it[1].data_ptr() for it in chunk])
        offs_pad = (ctypes.c_void_p * k)(*[it[2].data_ptr() for it in chunk])
        patch = (ctypes.c_int * k)(*[int(it[3]) for it in chunk])
        npad = (ctypes.c_long * k)(*[int(it[4]) for it in chunk])
        check(lib.cdseg_pad_plan_batch(k, orders, offs, offs_pad, patch, npad, int(nb), _ptr(gidx), _ptr(widx), _stream()),
              "pad_plan_batch")
        sizes = [int(it[4]) for it in chunk]
        res.extend(zip(gidx.split(sizes), widx.split(sizes)))  # (one call per buffer: 2 k Python slices cost ~1.5 us each)
    return res


F32X3 = 2  # include/cdseg.h CDSEG_F32X3: fp32 in memory, split-half arithmetic on the matrix pipe


def set_f32x3(on):
    """This thread's fp32 `gemm` / `attention` calls compute in the split-half "fp32 x3" form (csrc/gemm.hip) from now on;
    returns the previous setting.  The engine of precision "fp32x3" brackets its forwards with it."""
    prev, _TLS.f32x3 = _TLS.f32x3, bool(on)
    return prev


def gemm(A, W, out, *, bias=None, scale=None, shift=None, act=ACT_NONE, res=None, add_src=None, add_idx=None,
         nbr=None, out_idx=None, out2=None, out2_pre_add=False, M=None, kvol=1, colbias=None, ln_pre=None,
         nbr_kmajor=False,
         ln_post=None, ln_out=None, ln_eps=1e-5, cache=True):
    """out = epilogue(A @ W^T) (or the gathered-A sparse-conv form when nbr is given).
    A (M,K) [or the gather source], W (N, kvol*K), both of the compute dtype.
    cache=False: W is a temporary (a transposed / mirrored copy made for this call, as the training backward does): the
    argument block is built on the spot and nothing keeps W alive afterwards."""
    if not A.is_cuda:
        _need_gpu(A, W, out)
    # the weight-side half of the argument block is static per layer: cache it keyed by the weight tensor
    x3 = _TLS.f32x3 and W.dtype == torch.float32
    key = (W.data_ptr(), _dp(bias), _dp(scale), _dp(shift), int(kvol), int(act), x3)
    cache_d = _TLS.gemm_cache
    ent = cache_d.get(key) if cache else None
    if ent is None:
        for t in (bias, scale, shift):
            if t is not None and t.dtype != torch.float32:
                raise _lib.CdsegError("gemm epilogue vectors are float32")
        a = GemmArgs()
        a.W, a.bias, a.scale, a.shift = key[0], key[1], key[2], key[3]
        a.N = W.shape[0]
        a.kvol = int(kvol)
        a.K = W.shape[1] // a.kvol
        a.compute_dtype = F32X3 if x3 else dt(W)
        a.act = int(act)
        ent = (a, ctypes.byref(a), W)  # keep W alive with the cache entry
        if cache:
            if len(cache_d) > 4096:
                cache_d.clear()
            cache_d[key] = ent
    a = ent[0]
    if (res is not None and res.dtype != torch.float32) or (add_src is not None and add_src.dtype != torch.float32):
        raise _lib.CdsegError("gemm residuals are float32")
    a.A = A.data_ptr()
    a.res, a.add_src, a.add_idx, a.nbr, a.out_idx = _dp(res), _dp(add_src), _dp(add_idx), _dp(nbr), _dp(out_idx)
    a.out, a.out2 = out.data_ptr(), _dp(out2)
    m = int(M if M is not None else (nbr.shape[1 if nbr_kmajor else 0] if nbr is not None else A.shape[0]))
    a.nbr_kmajor = 1 if nbr_kmajor else 0
    a.M = m
    a.lda = A.stride(0)
    a.ldo = out.stride(0)
    a.ldo2 = out2.stride(0) if out2 is not None else 0
    a.ldres = res.stride(0) if res is not None else 0
    a.ldadd = add_src.stride(0) if add_src is not None else 0
    a.a_dtype = _DT[A.dtype]
    a.out_dtype = _DT[out.dtype]
    a.out2_dtype = _DT[out2.dtype] if out2 is not None else 0
    a.out2_pre_add = 1 if out2_pre_add else 0
    a.colbias = _dp(colbias)
    if ln_pre is not None:
        a.ln_pre_g, a.ln_pre_b = ln_pre[0].data_ptr(), ln_pre[1].data_ptr()
    else:
        a.ln_pre_g = a.ln_pre_b = None
    if ln_post is not None:
        a.ln_post_g, a.ln_post_b = ln_post[0].data_ptr(), ln_post[1].data_ptr()
        a.ln_out, a.ldln, a.ln_out_dtype = ln_out.data_ptr(), ln_out.stride(0), _DT[ln_out.dtype]
    else:
        a.ln_post_g = a.ln_post_b = a.ln_out = None
    a.ln_eps = float(ln_eps)
    if ((m + 63) >> 6) * ((a.N + 127) >> 7) < 256:  # split-K partials: only when the output has few tiles
        ws = workspace(min(32 * m * a.N * 4, 64 << 20), out.device)
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    elif nbr is not None and a.N >= 256 and ((m + 127) >> 7) * ((a.N + 255) >> 8) < 256:
        # deep sparse convs run 128 x 256 tiles, one block per CU: split-K while that grid is below one round of the chip
        ws = workspace(min(8 * m * a.N * 4, 64 << 20), out.device)
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    elif a.N > 128 and (ln_pre is not None or ln_post is not None):  # rows over several column tiles
        ws = workspace(m * a.N * 4, out.device)
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    else:
        a.ws, a.ws_bytes = None, 0
    tok = TIMER.begin("conv" if nbr is not None else "gemm") if TIMER is not None else None
    check(_lib.load().cdseg_gemm(ent[1], _stream()), "gemm")
    if tok is not None:
        TIMER.end(tok, 2.0 * a.M * a.N * a.K * a.kvol)
    return out


FUSED_MLP_CHANNELS = (32, 64, 128)


def mlp_fused_ok(h, hidden):
    """The fused MLP kernel covers bf16 with C = 32 / 64 / 128 and the standard 4x hidden width."""
    c = h.shape[1]
    return (is_lp(h.dtype) and c in FUSED_MLP_CHANNELS and hidden == 4 * c and
            c <= 128)


def mlp_fused(h, w1, b1, w2, b2, x, xc=None):
    """x += fc2(GELU(fc1(h))), xc = bf16(x); the hidden activation never leaves the CU."""
    _need_gpu(h, x)
    check(_lib.load().cdseg_mlp_fused(_ptr(h), h.stride(0), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(x), x.stride(0),
                                       _ptr(xc), xc.stride(0) if xc is not None else 0, h.shape[0], h.shape[1], _DT[h.dtype],
                                       _stream()), "mlp_fused")
    return x


def cpe_head_fused_ok(y):
    """cpe linear + LN + residual + LN1 + qkv in one launch: bf16, C = 32 / 64."""
    return is_lp(y.dtype) and y.shape[1] in (32, 64)


def cpe_head_fused(y, wl, bl, lnp, x, colbias, ln1, wqkv, bqkv, qkv, eps=1e-5, qkv_flags=0):
    """x += LN_cpe(y Wl^T + bl) [+ colbias]; h = LN1(x); qkv = h Wqkv^T + bqkv.  qkv_flags: ATTN_V_BF16 = v third as bfloat16."""
    _need_gpu(y, x)
    check(_lib.load().cdseg_cpe_head_fused(_ptr(y), y.stride(0), _ptr(wl), _ptr(bl), _ptr(lnp[0]), _ptr(lnp[1]), _ptr(x),
                                            x.stride(0), _ptr(colbias), _ptr(ln1[0]), _ptr(ln1[1]), float(eps), _ptr(wqkv),
                                            _ptr(bqkv), _ptr(qkv), qkv.stride(0), y.shape[0], y.shape[1], _DT[y.dtype],
                                            int(qkv_flags), _stream()), "cpe_head_fused")
    return qkv


def attn_tail_fused_ok(o, hidden):
    """proj + LayerNorm + MLP in one launch: bf16, C = 32 / 64."""
    c = o.shape[1]
    return is_lp(o.dtype) and c in (32, 64) and hidden == 4 * c


def attn_tail_fused(o, wp, bp, ln_g, ln_b, w1, b1, w2, b2, x, xc=None, eps=1e-5):
    """x += proj(o); h = LN(x); x += fc2(GELU(fc1(h))); xc = bf16(x) - the Block's tail after attention."""
    _need_gpu(o, x)
    check(_lib.load().cdseg_attn_tail_fused(_ptr(o), o.stride(0), _ptr(wp), _ptr(bp), _ptr(ln_g), _ptr(ln_b), float(eps),
                                             _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(x), x.stride(0), _ptr(xc),
                                             xc.stride(0) if xc is not None else 0, o.shape[0], o.shape[1], _DT[o.dtype],
                                             _stream()), "attn_tail_fused")
    return x


# include/cdseg.h CDSEG_DEEP512_MIN_ROWS: below, a C = 512 Block keeps the separate GEMM launches.  Experimental builds of the
# library read the CDSEG_DEEP512_MIN_ROWS environment knob (csrc/runtime.hip); the binding path follows the same variable so
# that the two executors never disagree under an A/B run
DEEP512_MIN_ROWS = int(os.environ.get("CDSEG_DEEP512_MIN_ROWS", 2560))
DEEP_CHANNELS = (128, 256, 512)  # deep-stage head / tail kernels (csrc/deep.hip): weights streamed L2 -> registers


def block_rr_ok(channels, dtype):
    """Fused Block head / tail kernels on weight images: C = 32 / 64 (csrc/blockrr.hip: weights resident in LDS,
    activations in registers) and C = 128 / 256 / 512 (csrc/deep.hip: activations resident in LDS, weights streamed), 16-bit."""
    return is_lp(dtype) and (channels in (32, 64) or channels in DEEP_CHANNELS)


def block_rr_head_on(channels=32):
    """Deep stages: on.  The register-resident HEAD of the wide stages is off by default: the 64-row-tile fused head already streams at ~4.3 TB/s and measured
    5-10 % faster (tools/bench_block.py); the register-resident TAIL is 1.4-1.65x faster than its predecessor."""
    return channels in DEEP_CHANNELS


def block_rr_pack(channels, wl, wqkv, wp, w1, w2):
    """Fragment images of a Block's head (cpe linear, qkv) and tail (proj, fc1, fc2) weights -> (head_img, tail_img).
    wl = wqkv = None: the tail image alone (head_img None) - the cross block's tail has the Block's shape, its head not."""
    _need_gpu(wl, wqkv, wp, w1, w2)
    lib = _lib.load()
    dev = wp.device
    head = torch.empty(lib.cdseg_block_rr_img_bytes(channels, 0), dtype=torch.uint8, device=dev) if wl is not None else None
    tail = torch.empty(lib.cdseg_block_rr_img_bytes(channels, 1), dtype=torch.uint8, device=dev)
    check(lib.cdseg_block_rr_pack(int(channels), _ptr(wl), _ptr(wqkv), _ptr(head), _ptr(wp), _ptr(w1), _ptr(w2), _ptr(tail),
                                  _stream()), "block_rr_pack")
    return head, tail


def cpe_head_rr(y, head_img, bl, lnp, x, colbias, ln1, bqkv, qkv, eps=1e-5, qkv_flags=0):
    check(_lib.load().cdseg_cpe_head_rr(_ptr(y), y.stride(0), _ptr(head_img), _ptr(bl), _ptr(lnp[0]), _ptr(lnp[1]), _ptr(x),
                                        x.stride(0), _ptr(colbias), _ptr(ln1[0]), _ptr(ln1[1]), float(eps), _ptr(bqkv),
                                        _ptr(qkv), qkv.stride(0), y.shape[0], y.shape[1], int(qkv_flags), _stream()),
          "cpe_head_rr")
    return qkv


def cpe_head_rr2(y, head_img, bl, lnp, x, x_out, colbias, ln1, bqkv, qkv, eps=1e-5, qkv_flags=0):
    """Deep stages: `cpe_head_rr` with the residual rows read from x and written to x_out (distinct buffers let few-row launches
    split a tile over three workgroups, csrc/deep.hip)."""
    check(_lib.load().cdseg_cpe_head_rr2(_ptr(y), y.stride(0), _ptr(head_img), _ptr(bl), _ptr(lnp[0]), _ptr(lnp[1]), _ptr(x),
                                         x.stride(0), _ptr(x_out), x_out.stride(0), _ptr(colbias), _ptr(ln1[0]), _ptr(ln1[1]),
                                         float(eps), _ptr(bqkv), _ptr(qkv), qkv.stride(0), y.shape[0], y.shape[1],
                                         int(qkv_flags), _stream()), "cpe_head_rr2")
    return qkv


def attn_tail_rr2(o, tail_img, bp, ln_g, ln_b, b1, b2, x_in, x, xc=None, ws=None, eps=1e-5):
    """Deep stages: `attn_tail_rr` with the residual read from x_in, the result written to x; ws: fp32 / byte workspace that
    allows the few-row hidden-chunk split (>= 4 * n * C * 4 bytes for four workgroups per tile)."""
    check(_lib.load().cdseg_attn_tail_rr2(_ptr(o), o.stride(0), _ptr(tail_img), _ptr(bp), _ptr(ln_g), _ptr(ln_b), float(eps),
                                          _ptr(b1), _ptr(b2), _ptr(x_in), x_in.stride(0), _ptr(x), x.stride(0), _ptr(xc),
                                          xc.stride(0) if xc is not None else 0, o.shape[0], o.shape[1], _ptr(ws),
                                          0 if ws is None else ws.numel() * ws.element_size(), _stream()), "attn_tail_rr2")
    return x


def attn_tail_rr(o, tail_img, bp, ln_g, ln_b, b1, b2, x, xc=None, eps=1e-5):
    check(_lib.load().cdseg_attn_tail_rr(_ptr(o), o.stride(0), _ptr(tail_img), _ptr(bp), _ptr(ln_g), _ptr(ln_b), float(eps),
                                         _ptr(b1), _ptr(b2), _ptr(x), x.stride(0), _ptr(xc),
                                         xc.stride(0) if xc is not None else 0, o.shape[0], o.shape[1], _stream()),
          "attn_tail_rr")
    return x


def _dp(t):
    return None if t is None else t.data_ptr()


def make_block_desc(dtype, channels, heads, hidden, attn_scale, ln_eps, tensors, attn_flags=0, x3=False):
    """Describe one Block's weights once (tensors: dict field -> tensor); returns an object to pass to
    block_forward.  The tensors are kept alive by the returned handle.  attn_flags: ATTN_Q_PRESCALED when the q rows of
    qkv_w / qkv_b (and the images packed from them) carry attn_scale * log2(e)."""
    d = _lib.BlockDesc()
    d.dtype, d.channels, d.heads, d.hidden = (F32X3 if (x3 and dtype == torch.float32) else _DT[dtype]), int(channels), int(heads), int(hidden)
    d.attn_scale, d.ln_eps = float(attn_scale), float(ln_eps)
    d.attn_flags = int(attn_flags)
    for k, t in tensors.items():
        setattr(d, k, t.data_ptr())
    return (d, ctypes.byref(d), dict(tensors))


def block_scratch_bytes(desc, n):
    return _lib.load().cdseg_block_scratch_bytes(desc[1], int(n))


def count_saturated(x, counter):
    """Diagnostic of the IEEE-half build: counter (1-element int64 device tensor) += elements of the 16-bit matrix x that
    sit at +-65504, the clamp value of that build's float -> half conversions."""
    _need_gpu(x, counter)
    check(_lib.load().cdseg_count_saturated(_ptr(x), x.shape[0], x.shape[1], x.stride(0), _ptr(counter), _stream()),
          "count_saturated")


def split16(x, lo_scale=2048.0):
    """fp32 (rows, cols) -> (hi, lo) in the active build's 16-bit type: hi = T(x), lo = T((x - hi) * lo_scale)."""
    _need_gpu(x)
    T = LP_DTYPES[_lib.active()]
    hi = torch.empty(x.shape, dtype=T, device=x.device)
    lo = torch.empty(x.shape, dtype=T, device=x.device)
    check(_lib.load().cdseg_split16(_ptr(x), x.stride(0), x.shape[0], x.shape[1], _ptr(hi), _ptr(lo), hi.stride(0),
                                    float(lo_scale), _stream()), "split16")
    return hi, lo


def block_forward(desc, n, x, xc_in, xc_out, tbias, nbr, gidx, widx, patch_start, num_patches, max_len, scratch,
                  sat_counter=None):
    """One PTv3 Block on the native executor (all launches issued by the library, one host call).  sat_counter: 1-element
    int64 device tensor that collects the Block's clamped half values (diagnostic, see count_saturated)."""
    if _TLS.block_io is None:
        io = _lib.BlockIO()
        _TLS.block_io = (io, ctypes.byref(io))
    io, ref = _TLS.block_io
    io.n = int(n)
    io.x, io.xc_in, io.xc_out, io.tbias = x.data_ptr(), xc_in.data_ptr(), xc_out.data_ptr(), _dp(tbias)
    io.nbr, io.gidx, io.widx, io.patch_start = nbr.data_ptr(), gidx.data_ptr(), widx.data_ptr(), patch_start.data_ptr()
    io.num_patches, io.max_len = int(num_patches), int(max_len)
    io.scratch, io.scratch_bytes = scratch.data_ptr(), scratch.numel()
    io.sat_counter = sat_counter.data_ptr() if sat_counter is not None else None
    check(_lib.load().cdseg_block_forward(desc[1], ref, _stream()), "block_forward")


def stem5_ok(cout, dtype):
    """The map-free stem kernel (csrc/stem.hip) covers the shipped stems: 32 output channels, bf16."""
    return cout == 32 and is_lp(dtype)


def child_info(zcode_sorted, seg_start, m):
    """Per parent cell: (first child row << 8) | octant occupancy (children of a cell are contiguous in z-order)."""
    _need_gpu(zcode_sorted, seg_start)
    info = torch.empty(int(m), dtype=torch.int64, device=zcode_sorted.device)
    check(_lib.load().cdseg_child_info(_ptr(zcode_sorted), _ptr(seg_start), int(m), _ptr(info), _stream()), "child_info")
    return info


def stem5_pack(w):
    """(32, 125 * 8) bf16 stem weight -> LDS image of the stem kernel (built once per weight)."""
    _need_gpu(w)
    assert is_lp(w.dtype) and tuple(w.shape) == (32, 1000)
    img = torch.empty(_lib.load().cdseg_stem5_wimg_bytes(), dtype=torch.uint8, device=w.device)
    check(_lib.load().cdseg_stem5_pack(_ptr(w), _ptr(img), _stream()), "stem5_pack")
    return img


def stem5(x8, wimg, scale, shift, grid, cluster, parent_nbr3, cinfo, depth, out, out2=None):
    """out (n, 32) fp32 [, out2 bf16] = GELU(BN(SubMConv3d_k5(x8))) without a 125-offset kernel map (ref: ptv3.py:633-663)."""
    _need_gpu(x8, wimg, grid, cluster, parent_nbr3, cinfo, out)
    n, m = x8.shape[0], cinfo.numel()
    check(_lib.load().cdseg_stem5(_ptr(x8), _ptr(wimg), _ptr(scale), _ptr(shift), _ptr(grid), _ptr(cluster),
                                  _ptr(parent_nbr3), _ptr(cinfo), n, m, int(depth), _ptr(out), _ptr(out2), _stream()), "stem5")
    return out


def subm_conv3_ok(x):
    """The weight-stationary live-list conv (csrc/conv.hip) covers the wide bf16 stages: C = 32 / 64."""
    n = x.shape[0]  # (32-bit buffer offsets inside the kernel: larger inputs take the gathered GEMM)
    return (is_lp(x.dtype) and x.dim() == 2 and x.shape[1] in (32, 64) and x.stride(0) == x.shape[1] and
            n * 27 * 4 < 2 ** 31 and n * x.shape[1] * 2 < 2 ** 31 - 65536)


def subm_conv3_pack(w):
    """(C, 27*C) bf16 weight of a k = 3 submanifold conv -> its MFMA-fragment-order image (built once per weight)."""
    _need_gpu(w)
    c = w.shape[0]
    img = torch.empty(_lib.load().cdseg_subm_conv3_wimg_bytes(c), dtype=torch.uint8, device=w.device)
    check(_lib.load().cdseg_subm_conv3_pack(_ptr(w), c, _ptr(img), _stream()), "subm_conv3_pack")
    return img


def subm_conv3(x, wimg, bias, nbr_kmajor, out):
    """out (n, C) bf16 = bias + sum_o x[nbr[o]] W_o^T   (ref call sites: ptv3.py:356-362)."""
    _need_gpu(x, wimg, nbr_kmajor, out)
    n, c = x.shape
    check(_lib.load().cdseg_subm_conv3(_ptr(x), x.stride(0), _ptr(wimg), _ptr(bias), _ptr(nbr_kmajor), n, c, _ptr(out),
                                       out.stride(0), _stream()), "subm_conv3")
    return out


def subm_conv3_f32(x, wimg, bias, nbr_kmajor, out, out_scale=1.0, accumulate=False):
    """The weight-stationary conv with an fp32 output: out = (accumulate ? out : 0) + (conv + bias) * out_scale."""
    _need_gpu(x, wimg, nbr_kmajor, out)
    n, c = x.shape
    check(_lib.load().cdseg_subm_conv3_f32(_ptr(x), x.stride(0), _ptr(wimg), _ptr(bias), _ptr(nbr_kmajor), n, c, _ptr(out),
                                           out.stride(0), float(out_scale), 1 if accumulate else 0, _stream()),
          "subm_conv3_f32")
    return out


def layernorm(x, gamma, beta, out, *, eps=1e-5, res=None, colbias=None, out2=None):
    m, c = x.shape
    check(_lib.load().cdseg_layernorm(_ptr(x), dt(x), x.stride(0), _ptr(gamma), _ptr(beta), float(eps), _ptr(res),
                                      res.stride(0) if res is not None else 0, _ptr(colbias), _ptr(out), dt(out),
                                      out.stride(0), _ptr(out2), dt(out2) if out2 is not None else 0,
                                      out2.stride(0) if out2 is not None else 0, m, c, _stream()), "layernorm")
    return out


ATTN_Q_PRESCALED, ATTN_V_BF16 = 1, 2  # include/cdseg.h: producer-side preprocessing declared to cdseg_attention_ex


def attention(q, k, v, q_gidx, kv_gidx, widx, patch_start, num_heads, max_len, scale, out, work=0.0, flags=0):
    """q/k/v: 2-D views (rows, H*16) of the projection buffers (any row stride); out (rows, H*16).
    flags: ATTN_Q_PRESCALED (q carries scale * log2 e: `scale` ignored) | ATTN_V_BF16 (v is bfloat16 in the half build too).
    work: algorithmic FLOPs of this launch (4 * 16 * H * sum_p L_p^2), only used by the bench timer."""
    num_patches = patch_start.numel() - 1
    if not (q.dtype == k.dtype == out.dtype) or (v.dtype != q.dtype and not (flags & ATTN_V_BF16)):
        raise _lib.CdsegError("attention: q, k, v, out must share a dtype")
    tok = TIMER.begin("attention") if TIMER is not None else None
    check(_lib.load().cdseg_attention_ex(_ptr(q), _ptr(k), _ptr(v), q.stride(0), k.stride(0), v.stride(0), _ptr(q_gidx),
                                         _ptr(kv_gidx), _ptr(widx), _ptr(patch_start), num_patches, int(num_heads),
                                         int(max_len), float(scale), _ptr(out), out.stride(0),
                                         F32X3 if (_TLS.f32x3 and q.dtype == torch.float32) else dt(q), int(flags), _stream()),
          "attention")
    if tok is not None:
        TIMER.end(tok, work)
    return out


def segment_max(y, seg_start, m, scale, shift, act, out, out2=None):
    c = y.shape[1]
    check(_lib.load().cdseg_segment_max(_ptr(y), dt(y), y.stride(0), _ptr(seg_start), m, c, _ptr(scale), _ptr(shift),
                                        int(act), _ptr(out), out.stride(0), _ptr(out2),
                                        dt(out2) if out2 is not None else 0, out2.stride(0) if out2 is not None else 0,
                                        _stream()), "segment_max")
    return out


def pool_fused_ok(cin, cout, dtype):
    """The one-launch pooling (csrc/pool.hip) covers the wide 16-bit stages: 32 -> 64 and 64 -> 128 channels."""
    return is_lp(dtype) and (int(cin), int(cout)) in ((32, 64), (64, 128))


def pool_fused_pack(w):
    """(cout, cin) 16-bit projection weight -> its MFMA-fragment image (built once per weight)."""
    _need_gpu(w)
    cout, cin = w.shape
    lib = _lib.load()
    img = torch.empty(lib.cdseg_pool_fused_img_bytes(cin, cout), dtype=torch.uint8, device=w.device)
    check(lib.cdseg_pool_fused_pack(_ptr(w), cin, cout, _ptr(img), _stream()), "pool_fused_pack")
    return img


def pool_fused(x, wimg, bias, seg_start, m, scale, shift, act, out, out2=None):
    """out (m, cout) fp32 = act(scale * max over each run of round16(x W^T + bias) + shift), out2 its 16-bit copy."""
    _need_gpu(x, out)
    cin, cout = x.shape[1], out.shape[1]
    check(_lib.load().cdseg_pool_fused(_ptr(x), x.stride(0), _ptr(wimg), _ptr(bias), _ptr(seg_start), int(m), _ptr(scale),
                                       _ptr(shift), int(act), _ptr(out), out.stride(0), _ptr(out2),
                                       out2.stride(0) if out2 is not None else 0, cin, cout, _stream()), "pool_fused")
    return out


def segment_mean(x, seg_start, m):
    out = torch.empty((m, x.shape[1]), dtype=torch.float32, device=x.device)
    check(_lib.load().cdseg_segment_mean(_ptr(x), x.stride(0), _ptr(seg_start), m, x.shape[1], _ptr(out),
                                         out.stride(0), _stream()), "segment_mean")
    return out


def gemv(w, b, x, act=ACT_NONE):
    n, k = w.shape
    y = torch.empty(n, dtype=torch.float32, device=w.device)
    check(_lib.load().cdseg_gemv(_ptr(w), _ptr(b), _ptr(x), n, k, int(act), _ptr(y), _stream()), "gemv")
    return y


def randn(shape, seed, offset, device):
    out = torch.empty(shape, dtype=torch.float32, device=device)
    check(_lib.load().cdseg_randn(_ptr(out), out.numel(), int(seed) & (2 ** 64 - 1), int(offset), _stream()), "randn")
    return out


def cast(src, dtype):
    src = src.contiguous()
    out = torch.empty(src.shape, dtype=dtype, device=src.device)
    check(_lib.load().cdseg_cast(_ptr(src), dt(src), _ptr(out), _DT[dtype], src.numel(), _stream()), "cast")
    return out


def gather_pad_cast(src, idx, cpad, dtype):
    """(n, cin) fp32 rows gathered by idx (None = identity), zero-padded to cpad columns, cast to dtype."""
    src = src.contiguous()
    n = idx.numel() if idx is not None else src.shape[0]
    out = torch.empty((n, cpad), dtype=dtype, device=src.device)
    check(_lib.load().cdseg_gather_pad_cast(_ptr(src), src.stride(0), _ptr(idx), n, src.shape[1], int(cpad), _ptr(out),
                                            _DT[dtype], _stream()), "gather_pad_cast")
    return out


def ddim_update(xt, eps, sqrt_ab_prev, sqrt_1m_ab, sqrt_ab, sqrt_1m_ab_prev, final=False):
    out = torch.empty_like(xt)
    check(_lib.load().cdseg_ddim_update(_ptr(xt), _ptr(eps), float(sqrt_ab_prev), float(sqrt_1m_ab), float(sqrt_ab),
                                 float(sqrt_1m_ab_prev), 1 if final else 0, _ptr(out), xt.numel(), _stream()),
          "ddim_update")
    return out


def axpy(a, b, alpha):
    out = torch.empty_like(a)
    check(_lib.load().cdseg_axpy(_ptr(a), _ptr(b), float(alpha), _ptr(out), a.numel(), _stream()), "axpy")
    return out


# ------------------------------------------------------------------ test-time pipeline ops
def voxelize(coord, grid_size):
    """GridSample's voxel coordinates: (grid int32 (n,3) shifted to start at 0, key int64 (n), min int32 (3))."""
    _need_gpu(coord)
    coord = coord.float().contiguous()
    n = coord.shape[0]
    grid = torch.empty((n, 3), dtype=torch.int32, device=coord.device)
    key = torch.empty(n, dtype=torch.int64, device=coord.device)
    mn = torch.empty(3, dtype=torch.int32, device=coord.device)
    check(_lib.load().cdseg_voxelize(_ptr(coord), float(grid_size), n, _ptr(grid), _ptr(key), _ptr(mn), _stream()), "voxelize")
    return grid, key, mn


def voxelize_any(coord, grid_size):
    """voxelize for float32 or float64 coordinates (GridSample after a test-time rotation sees float64)."""
    if coord.dtype != torch.float64:
        return voxelize(coord, grid_size)
    _need_gpu(coord)
    coord = coord.contiguous()
    n = coord.shape[0]
    grid = torch.empty((n, 3), dtype=torch.int32, device=coord.device)
    key = torch.empty(n, dtype=torch.int64, device=coord.device)
    mn = torch.empty(3, dtype=torch.int32, device=coord.device)
    check(_lib.load().cdseg_voxelize_f64(_ptr(coord), float(grid_size), n, _ptr(grid), _ptr(key), _ptr(mn), _stream()),
          "voxelize_f64")
    return grid, key, mn


def center_shift(coord, apply_z=True):
    """CenterShift (ref: datasets/transform.py:142-155): coord - [(xmin+xmax)/2, (ymin+ymax)/2, zmin or 0], in coord's dtype."""
    _need_gpu(coord)
    assert coord.dtype in (torch.float32, torch.float64) and coord.dim() == 2 and coord.shape[1] == 3
    coord = coord.contiguous()
    out = torch.empty_like(coord)
    ws = torch.empty(12, dtype=torch.float64, device=coord.device)
    check(_lib.load().cdseg_center_shift(_ptr(coord), 1 if coord.dtype == torch.float64 else 0, coord.shape[0],
                                         1 if apply_z else 0, _ptr(out), _ptr(ws), _stream()), "center_shift")
    return out


def tta_apply(xyz, rot=None, scale=None, flip=False):
    """One test-time augmentation of an (n,3) float32 array (ref: transform.py:259-328).  rot (3x3 nested floats):
    float64 result (xyz @ rot^T) [* scale]; else flip: float32 result with x, y negated; neither: xyz itself."""
    _need_gpu(xyz)
    xyz = xyz.float().contiguous()
    n = xyz.shape[0]
    if rot is None and not flip:
        return xyz
    if rot is not None:
        r = (ctypes.c_double * 9)(*[float(v) for row in rot for v in row])
        out = torch.empty((n, 3), dtype=torch.float64, device=xyz.device)
        check(_lib.load().cdseg_tta_apply(_ptr(xyz), n, r, float(scale if scale is not None else 1.0),
                                          1 if scale is not None else 0, 0, _ptr(out), _stream()), "tta_apply")
        return out
    out = torch.empty_like(xyz)
    check(_lib.load().cdseg_tta_apply(_ptr(xyz), n, None, 1.0, 0, 1, _ptr(out), _stream()), "tta_apply")
    return out


def div_add(x, div, add):
    """x / div + add in float32 (NormalizeColor = (color, 127.5, -1), ref: transform.py:113-117)."""
    _need_gpu(x)
    x = x.float().contiguous()
    out = torch.empty_like(x)
    check(_lib.load().cdseg_div_add(_ptr(x), float(div), float(add), x.numel(), _ptr(out), _stream()), "div_add")
    return out


def collect_feat(a, b):
    """Collect(feat_keys=(a, b)): cat([a.float(), b.float()], 1) (ref: transform.py:46-49); b float32 or float64."""
    _need_gpu(a, b)
    a = a.float().contiguous()
    b = b.contiguous()
    n = a.shape[0]
    out = torch.empty((n, a.shape[1] + b.shape[1]), dtype=torch.float32, device=a.device)
    check(_lib.load().cdseg_collect_feat(_ptr(a), a.shape[1], _ptr(b), 1 if b.dtype == torch.float64 else 0, b.shape[1], n,
                                         _ptr(out), _stream()), "collect_feat")
    return out


def max_run(seg_start, m):
    out = torch.empty(1, dtype=torch.int32, device=seg_start.device)
    check(_lib.load().cdseg_max_run(_ptr(seg_start), int(m), _ptr(out), _stream()), "max_run")
    return out


def fragment_select(idx_sort, seg_start, m, frag):
    out = torch.empty(int(m), dtype=torch.int32, device=idx_sort.device)
    check(_lib.load().cdseg_fragment_select(_ptr(idx_sort), _ptr(seg_start), int(m), int(frag), _ptr(out), _stream()),
          "fragment_select")
    return out


def softmax_vote(logits, idx, pred):
    check(_lib.load().cdseg_softmax_vote(_ptr(logits), logits.stride(0), _ptr(idx), logits.shape[0], logits.shape[1], _ptr(pred),
                                  pred.stride(0), _stream()), "softmax_vote")
    return pred


def argmax_rows(x):
    out = torch.empty(x.shape[0], dtype=torch.int32, device=x.device)
    check(_lib.load().cdseg_argmax_rows(_ptr(x), x.stride(0), x.shape[0], x.shape[1], _ptr(out), _stream()), "argmax_rows")
    return out


def knn1(ref_xyz, ref_offset, qry_xyz, qry_offset, origin, cell, want_dist=False):
    """Exact nearest reference point (same batch element) of every query.  offsets: int32 cumulative ends."""
    lib = _lib.load()
    _need_gpu(ref_xyz, qry_xyz)
    ref_xyz, qry_xyz = ref_xyz.float().contiguous(), qry_xyz.float().contiguous()
    n, m = ref_xyz.shape[0], qry_xyz.shape[0]
    idx = torch.empty(m, dtype=torch.int32, device=qry_xyz.device)
    d2 = torch.empty(m, dtype=torch.float32, device=qry_xyz.device) if want_dist else None
    ws = workspace(lib.cdseg_knn1_ws_bytes(n), qry_xyz.device)
    org = (ctypes.c_float * 3)(*[float(v) for v in origin])
    check(lib.cdseg_knn1(_ptr(ref_xyz), _ptr(ref_offset), n, _ptr(qry_xyz), _ptr(qry_offset), m, ref_offset.numel(), org,
                         float(cell), _ptr(idx), _ptr(d2), _ptr(ws), ws.numel(), _stream()), "knn1")
    return (idx, d2) if want_dist else idx


def knn(k, ref_xyz, ref_offset, qry_xyz=None, qry_offset=None, origin=None, cell=None):
    """pointops.knn_query(k, xyz, offset, new_xyz, new_offset) (ref: libs/pointops/functions/query.py:7-24): the k nearest
    reference points (same batch element) of every query -> (idx (m, k) int32 with -1 placeholders, dist (m, k) float32 =
    EUCLIDEAN distances like the reference's wrapper, placeholders sqrt(1e10)).  Ascending by (distance, index).
    origin / cell default to the reference points' minimum and a cell that holds ~2 points on average (one host read)."""
    lib = _lib.load()
    _need_gpu(ref_xyz)
    if qry_xyz is None:
        qry_xyz, qry_offset = ref_xyz, ref_offset
    ref_xyz, qry_xyz = ref_xyz.float().contiguous(), qry_xyz.float().contiguous()
    ref_offset, qry_offset = ref_offset.to(torch.int32).contiguous(), qry_offset.to(torch.int32).contiguous()
    n, m = ref_xyz.shape[0], qry_xyz.shape[0]
    if origin is None or cell is None:
        lo, hi = ref_xyz.min(0).values, ref_xyz.max(0).values
        ext = (hi - lo).clamp_min(1e-6).cpu().tolist()
        origin = lo.cpu().tolist()
        # surface-like clouds: cells of side s hold ~ n s^2 / area points; aim at ~2 per cell
        area = max(ext[0] * ext[1], ext[0] * ext[2], ext[1] * ext[2])
        cell = max((2.0 * area / max(n, 1)) ** 0.5, max(ext) / ((1 << 20) - 2))
    idx = torch.empty((m, int(k)), dtype=torch.int32, device=qry_xyz.device)
    d2 = torch.empty((m, int(k)), dtype=torch.float32, device=qry_xyz.device)
    ws = workspace(lib.cdseg_knn1_ws_bytes(n), qry_xyz.device)
    org = (ctypes.c_float * 3)(*[float(v) for v in origin])
    check(lib.cdseg_knn(_ptr(ref_xyz), _ptr(ref_offset), n, _ptr(qry_xyz), _ptr(qry_offset), m, ref_offset.numel(), int(k), org,
                        float(cell), _ptr(idx), _ptr(d2), _ptr(ws), ws.numel(), _stream()), "knn")
    return idx, torch.sqrt(d2)


def iou_counts(pred, target, num_classes, ignore_index=-1, pred_idx=None):
    """(3, K) int64: intersection, prediction and target counts (rows with target == ignore_index dropped)."""
    out = torch.empty((3, num_classes), dtype=torch.int64, device=pred.device)
    check(_lib.load().cdseg_iou_counts(_ptr(pred), _ptr(pred_idx), _ptr(target), target.numel(), int(num_classes),
                                        int(ignore_index), _ptr(out), _stream()), "iou_counts")
    return out


# ------------------------------------------------------------------ reference-shaped composites
def serialization(grid_coord, batch, orders=("z", "z-trans", "hilbert", "hilbert-trans"), depth=None):
    """Point.serialization before the shuffle (structure.py:47-93): code, order, inverse as (k, N) int64
    in the caller's point order - the drop-in for the reference's encode + argsort + scatter_."""
    _need_gpu(grid_coord, batch)
    if depth is None:
        depth = int(grid_max(grid_coord).item()).bit_length()
    codes, orders_, inverses = [], [], []
    nb = int(batch.max().item()) + 1 if batch.numel() else 1
    end_bit = 3 * depth + max(1, nb.bit_length())
    for o in orders:
        c = encode(grid_coord, batch, depth, o)
        _, perm = sort_pairs(c, None, end_bit=min(64, end_bit))
        codes.append(c)
        orders_.append(widen(perm))
        inverses.append(widen(invert_perm(perm)))
    return torch.stack(codes), torch.stack(orders_), torch.stack(inverses), depth


# ------------------------------------------------------------------ training path, first slice (csrc/train.hip)
def attention_bwd(q, k, v, q_gidx, kv_gidx, widx, patch_start, patch_start_host, num_heads, scale, dout, dq, dk, dv):
    """Gradients of `attention` (fp32): dq / dk / dv (views of zero-initialised buffers, any row stride) += at the
    gathered rows.  patch_start_host: the same patch table as Python ints (tile count of the launch)."""
    _need_gpu(q, dout)
    if q.dtype != torch.float32:
        raise _lib.CdsegError("attention_bwd: exact-fp32 mode only (first slice of the training path)")
    ps = [int(x) for x in patch_start_host]
    num_patches = len(ps) - 1
    max_len = max((ps[i + 1] - ps[i] for i in range(num_patches)), default=0)
    if max_len > 1024:
        raise _lib.CdsegError("attention_bwd: a patch holds at most 1024 slots (the patch-head lives in LDS, like the forward)")
    for t, ld in ((q, q.stride(0)), (k, k.stride(0)), (v, v.stride(0)), (dout, dout.stride(0))):
        if ld % 4 or t.data_ptr() % 16:
            raise _lib.CdsegError("attention_bwd: rows must be 16-byte aligned (float4 operand loads)")
    lib = _lib.load()
    ws = torch.empty(max(1, lib.cdseg_attention_bwd_ws_bytes(ps[-1], int(num_heads))), dtype=torch.uint8, device=q.device)
    check(lib.cdseg_attention_bwd(_ptr(q), _ptr(k), _ptr(v), q.stride(0), k.stride(0), v.stride(0), _ptr(q_gidx),
                                  _ptr(kv_gidx), _ptr(widx), _ptr(patch_start), num_patches, int(num_heads), ps[-1], max(1, max_len),
                                  float(scale), _ptr(dout), dout.stride(0), _ptr(dq), _ptr(dk), _ptr(dv), dq.stride(0),
                                  dk.stride(0), dv.stride(0), dt(q), _ptr(ws), ws.numel(), _stream()), "attention_bwd")


def layernorm_bwd(x, gamma, dy, dx, accumulate=False, eps=1e-5, dgamma=None, dbeta=None):
    """dx (=, or += with accumulate) of y = LayerNorm(x) * gamma + beta; fp32."""
    _need_gpu(x, dy)
    check(_lib.load().cdseg_layernorm_bwd(_ptr(x), x.stride(0), _ptr(gamma), float(eps), _ptr(dy), dy.stride(0), _ptr(dx),
                                          dx.stride(0), int(bool(accumulate)), _ptr(dgamma), _ptr(dbeta), x.shape[0],
                                          x.shape[1], _stream()), "layernorm_bwd")
    return dx


def gelu_bwd(u, dy):
    """dy * GELU'(u) on the pre-activation u (erf form); fp32, contiguous."""
    _need_gpu(u, dy)
    dx = torch.empty_like(u)
    check(_lib.load().cdseg_gelu_bwd(_ptr(u), _ptr(dy), _ptr(dx), u.numel(), _stream()), "gelu_bwd")
    return dx


def linear_wgrad(x, dy, dw, db=None, xidx=None):
    """dw (N, K view, any row stride) += dy^T x[xidx or arange] and db (N) += column sums of dy; fp32 (cdseg_linear_wgrad).
    xidx (M) int32 with -1 = no row: one kernel offset of a submanifold conv."""
    _need_gpu(x, dy, dw)
    assert x.dtype == dy.dtype == dw.dtype == torch.float32 and dy.stride(1) == 1 and x.stride(1) == 1 and dw.stride(1) == 1
    m, n = dy.shape
    k = x.shape[1]
    assert tuple(dw.shape) == (n, k) and (xidx is None or (xidx.dtype == torch.int32 and xidx.numel() == m))
    check(_lib.load().cdseg_linear_wgrad(_ptr(x), x.stride(0), _ptr(xidx), _ptr(dy), dy.stride(0), m, k, n, _ptr(dw),
                                         dw.stride(0), _ptr(db), _stream()), "linear_wgrad")
    return dw


def conv_wgrad(x, nbr_kmajor, dy, dw3, db=None):
    """dw3 (Cout, kvol, Cin) += the weight gradient of a submanifold conv over all kernel offsets (one launch), db += the
    bias gradient; nbr_kmajor (kvol, M) int32 offset-major kernel map.  fp32 (cdseg_conv_wgrad)."""
    _need_gpu(x, dy, dw3)
    kvol, m = nbr_kmajor.shape
    cout, kv, cin = dw3.shape
    assert kv == kvol and dw3.is_contiguous() and x.dtype == dy.dtype == dw3.dtype == torch.float32
    assert nbr_kmajor.dtype == torch.int32 and nbr_kmajor.is_contiguous() and dy.shape == (m, cout) and x.shape[1] == cin
    check(_lib.load().cdseg_conv_wgrad(_ptr(x), x.stride(0), _ptr(nbr_kmajor), kvol, _ptr(dy), dy.stride(0), m, cin, cout,
                                       _ptr(dw3), _ptr(db), _stream()), "conv_wgrad")
    return dw3
