"""Single-step-inference engine: runs DefaultSegmentorV2.inference / PT-v3m1.forward
(ref: pointcept/models/default.py:371-422, point_transformer_v3m1_base.py:1757-1815) on the
HIP kernels of libcdseg_hip.so.

MI355X-first data layout (not the reference's):
  * every stage's points are PHYSICALLY stored in (batch | z-order) sorted order, so sparse-conv
    neighbour gathers, pooling segments (contiguous runs) and un-pooling gathers are
    memory-coherent; the other three curves are int32 rank->row maps used only by the attention
    gather.  All ops are permutation-equivariant, the head scatters logits back to the caller's
    order, so results equal the reference's;
  * the c-branch (stride 4,4) visits the same voxel sets as n-branch stages 0/2/4, so both
    branches share one set of "levels" (codes, orders, neighbour tables, padding plans);
  * residual stream fp32; GEMM / attention operands in the compute dtype T (bf16 or fp32), with a
    T-typed shadow copy `xc` written by the producing epilogue.  `xc` also reproduces the
    reference's stale `sparse_conv_feat` after un-pooling (oracle/model.py: unpooling);
  * the per-point timestep embedding collapses to one vector (t is uniform in SSI): a table row,
    two GEMVs and one GEMV for all t_mlp's give a per-block bias (SURVEY.md finding 6);
  * c-decoder / c-head are dead code in SSI and are skipped (SURVEY.md finding 5).
Host<->device syncs: offset + grid max (depth), pooled point counts (one copy for all levels).
"""
import math
import struct
import os
import threading

import numpy as np
import torch

from . import _lib, ops
from ._lib import CdsegError, DuplicateVoxelsError

CURVES = ("z", "z-trans", "hilbert", "hilbert-trans")


def _pooling_depth(stride):
    return (math.ceil(stride) - 1).bit_length()


class _ShareEvents(threading.local):
    on = False  # set by the engine while a scene uses its side stream (per host thread)


SHARE_EVENTS = _ShareEvents()


def _shared(cache, key, build):
    """Lazily built plan item shared by the two branch streams: the builder's stream records an event, a consumer
    on the other stream waits for it (the item itself lives as long as the plan)."""
    ent = cache.get(key)
    cur = ops.current_stream_id()
    if ent is None:
        val = build()
        ent = cache[key] = (val, cur, ops.record_event() if (cur is not None and SHARE_EVENTS.on) else None)
    elif ent[1] != cur and ent[2] is not None:
        ops.wait_event(ent[2])
    return ent[0]


def _cut(lazy, cache, key):
    """Native plan: the item `key` lives in the plan's arena and its view has not been cut yet - do it now (off the plan
    phase's critical path: the host is ahead of the device once the Blocks are being issued)."""
    lz = lazy.pop(key, None)
    if lz is not None:
        sid, parts = lz
        vals = tuple(A[o:o + c] if shape is None else A[o:o + c].view(shape) for A, o, c, shape in parts)
        cache[key] = (vals[0] if len(vals) == 1 else vals, sid, None)


FUSED_CURVE_SORT = True  # the level-0 orders of all curves in use from ONE sort (False: one sort per curve; tools A/B)

class Level:
    """One voxel resolution of the scene, points in (batch | z) sorted order."""

    def __init__(self, cum, depth, n, grid, batch, code4, offs_host, lazy=None):
        self.cum, self.depth, self.n = cum, depth, n
        # grid (n, 3) int32, batch (n) int32, code4 (4, n) int64 - or, for a pooled level of the native plan, `lazy` = (int32
        # arena, grid offset, batch offset, int64 arena, code offset): the three views are cut when something asks for them (the
        # inference path never does: kernel maps, orders and slot plans of that level already exist)
        self._gbc = (grid, batch, code4) if lazy is None else None
        self._gbc_lazy = lazy
        self.offs_host = offs_host  # (B+1) python ints
        self.parent = None  # (parent Level, (cluster, seg_start)) when the next pooled level is one octree step up
        self._order = {}
        self._nbr = {}
        self._pad = {}
        self._slots = {}
        # native plan (csrc/plan.hip): padding tables / slot plans already on the device inside the plan's arena; their views
        # are cut on first use (a model touches ~20 of the ~110)
        self._pad_lazy = {}
        self._slot_lazy = {}
        self._order_lazy = {}
        self._nbr_lazy = {}

    def _cut_gbc(self):
        A, go, bo, Q, co = self._gbc_lazy
        n = self.n
        self._gbc = (A[go:go + 3 * n].view(n, 3), A[bo:bo + n], Q[co:co + 4 * n].view(4, n))
        self._gbc_lazy = None
        return self._gbc

    @property
    def grid(self):
        return (self._gbc or self._cut_gbc())[0]

    @property
    def batch(self):
        return (self._gbc or self._cut_gbc())[1]

    @property
    def code4(self):
        return (self._gbc or self._cut_gbc())[2]

    def order(self, curve):
        """rank -> physical row for a curve (None = identity for z)."""
        if curve == 0:
            return None
        if self._order_lazy:
            _cut(self._order_lazy, self._order, curve)

        def build():
            nb = len(self.offs_host) - 1
            end_bit = min(64, 3 * self.depth + max(1, nb.bit_length()))
            return ops.sort_pairs(self.code4[curve], None, end_bit=end_bit)[1]
        return _shared(self._order, curve, build)

    def nbr(self, ksize, kmajor=False):
        # offset-major tables: a wave searches one offset of 64 consecutive z-ordered points (same cache lines);
        # an open-addressing hash table was measured and is NOT faster (two dependent random reads per lookup)
        if self._nbr_lazy:
            _cut(self._nbr_lazy, self._nbr, (ksize, kmajor))

        def build():
            if self.parent is not None:  # derive from the parent level's 3x3x3 map: ~3 cached reads per lookup
                par, (cluster, seg) = self.parent
                return ops.nbr_table_from_info(self.grid, cluster, par.nbr(3, True), self.child_info(), par.n, self.depth,
                                               ksize, kmajor)
            return ops.nbr_table(self.code4[0], self.grid, self.batch, self.depth, ksize, kmajor)
        return _shared(self._nbr, (ksize, kmajor), build)

    def child_info(self):
        """Per parent cell of this level's points: first child row + octant occupancy (the stem kernel's traversal)."""
        if self._nbr_lazy:
            _cut(self._nbr_lazy, self._nbr, "child_info")

        def build():
            par, (_, seg) = self.parent
            return ops.child_info(self.code4[0], seg, par.n)
        return _shared(self._nbr, "child_info", build)

    def pad_host_py(self, patch_size, enable_flash):
        """Host side of the padding plan (ref: ptv3.py:188-250) in plain Python ints - a scene has a handful of batch
        elements and numpy's per-call overhead was most of the plan's host time: K, offs, offs_pad, patch_start (lists)."""
        offs = self.offs_host
        counts = [b - a for a, b in zip(offs[:-1], offs[1:])]
        K = int(patch_size) if enable_flash else int(min(min(counts), patch_size))
        offs_pad, patch_start = [0], []
        for c in counts:
            pc = (c + K - 1) // K * K if c > K else c
            patch_start.extend(range(offs_pad[-1], offs_pad[-1] + pc, K))
            offs_pad.append(offs_pad[-1] + pc)
        patch_start.append(offs_pad[-1])
        return K, offs, offs_pad, patch_start

    def pad_host(self, patch_size, enable_flash):
        """The same as int32 arrays."""
        K, offs, offs_pad, patch_start = self.pad_host_py(patch_size, enable_flash)
        return (K, np.asarray(offs, dtype=np.int32), np.asarray(offs_pad, dtype=np.int32),
                np.asarray(patch_start, dtype=np.int32))

    def set_pad(self, key, K, offs_pad, patch_start, offs_dev, offs_pad_dev, patch_start_dev):
        lens = [int(b) - int(a) for a, b in zip(patch_start[:-1], patch_start[1:])]
        self._pad[key] = (K, int(offs_pad[-1]), offs_dev, offs_pad_dev, patch_start_dev,
                          max(lens), float(sum(v * v for v in lens)))

    def pad(self, patch_size, enable_flash):
        """(K, n_pad, offs_dev, offs_pad_dev, patch_start_dev, max_len, sum_L2) - ref: ptv3.py:188-250."""
        key = (patch_size, enable_flash)
        lz = self._pad_lazy.pop(key, None) if self._pad_lazy else None
        if lz is not None:
            K, n_pad, max_len, sum_l2, A, oo, op, ot, nb, npatch = lz
            self._pad[key] = (K, n_pad, A[oo:oo + nb + 1], A[op:op + nb + 1], A[ot:ot + npatch + 1], max_len, sum_l2)
        if key not in self._pad:  # not pre-uploaded by Engine.build_plan: upload now
            K, offs, offs_pad, patch_start = self.pad_host(patch_size, enable_flash)
            dev = self.grid.device
            up = torch.tensor(np.concatenate([offs, offs_pad, patch_start]), device=dev)
            a, b = len(offs), len(offs) + len(offs_pad)
            self.set_pad(key, K, offs_pad, patch_start, up[:a], up[a:b], up[b:])
        return self._pad[key]

    def slots(self, curve, patch_size, enable_flash):
        lz = self._slot_lazy.pop((curve, patch_size, enable_flash), None) if self._slot_lazy else None
        if lz is not None:
            A, go, wo, n_pad, sid = lz
            self._slots[(curve, patch_size, enable_flash)] = ((A[go:go + n_pad], A[wo:wo + n_pad]), sid, None)

        def build():
            K, n_pad, offs, offs_pad = self.pad(patch_size, enable_flash)[:4]
            return ops.pad_plan(self.order(curve), offs, offs_pad, K, n_pad)
        return _shared(self._slots, (curve, patch_size, enable_flash), build)


class Plan:
    def __init__(self):
        self.levels = {}
        self.links = {}
        self._finish = None  # Engine.build_plan(defer_pads=True): the padding / slot plans, not built yet
        self._link_lazy = {}

    def finish_pads(self):
        """Build the deferred padding / slot plans (no-op when build_plan already did).  Called on the stream the plan was
        built on, before any fork: the items carry no events."""
        fin, self._finish = self._finish, None
        if fin is not None:
            fin()

    def link(self, a, b):
        """fine level a -> coarse level b: (cluster (n_a) int32, seg_start (n_b + 1) int32)."""
        if self._link_lazy:
            _cut(self._link_lazy, self.links, (a, b))

        def build():
            la, lb = self.levels[a], self.levels[b]
            if a > 0 and (0, a) in self.links and (0, b) in self.links:  # two gathers instead of a flag/scan pass
                (cl0a, seg0a), (cl0b, seg0b) = self.link(0, a), self.link(0, b)
                return ops.link_derive(cl0a, seg0a, la.n, cl0b, seg0b, lb.n)
            cluster, seg, _ = ops.pool_level(la.code4[0], 3 * (lb.cum - la.cum))
            return (cluster, seg)
        return _shared(self.links, (a, b), build)


class State:
    """A branch's activations at one level."""

    def __init__(self, level, x, xc, curves):
        self.level, self.x, self.xc, self.curves = level, x, xc, curves
        self.parent = None


class _HeadsInFp32:
    """What `Engine._eng` hands out for the seg heads of a 16-bit engine with precision "*+head": the heads' weights in fp32
    and the dtype hand-over (`_fit`) - not a second copy of all 101 M parameters."""
    T = torch.float32

    def __init__(self, eng):
        bb, self.device, self.w = eng.model.backbone, eng.device, {}
        for name in ("n_head", "c_head"):
            mod = getattr(bb, "_" + name, None)
            if isinstance(mod, torch.nn.Linear):
                self.w[name + ".w"] = mod.weight.detach().to(device=eng.device, dtype=torch.float32).contiguous()
                self.w[name + ".b"] = (mod.bias.detach().to(device=eng.device, dtype=torch.float32).contiguous()
                                       if mod.bias is not None else None)

    def prepare(self, device):
        assert device == self.device

    def _fit(self, st, exact=False):
        return Engine._fit(self, st, exact)


class Engine:
    PRECISIONS = ("bf16", "bf16+head", "fp16", "fp16+head", "fp32", "fp32x3")

    def __init__(self, model, precision="bf16", variant=None):
        """precision: the 16-bit trunk type ("bf16*": bfloat16, "fp16*": IEEE half - the library is built once for each,
        csrc/common.h) with "+head" = seg heads in exact fp32, or "fp32" (exact-fp32 MFMA everywhere).
        variant: which build an fp32 engine calls (the fp32 twin of a half engine stays inside the half build)."""
        if precision not in self.PRECISIONS:
            raise ValueError(precision)
        self.model, self.precision = model, precision
        bb = model.backbone
        if getattr(bb, "condition", False) and not (bb.c_num_stages == 3 and bb.n_num_stages == 5):
            # the c / n encoder interleave (which randperm draw a stage consumes) is hard-wired in the reference for exactly
            # this shape (ptv3.py:1785-1794); every shipped config has it - reject anything else here, not mid-forward
            raise ValueError(f"conditional PT-v3m1 needs 3 c-branch and 5 n-branch stages (ptv3.py:1785-1794), got "
                             f"{bb.c_num_stages} / {bb.n_num_stages}")
        # ("fp32x3" lives in the IEEE-half build: its sparse convs hand half pairs to that build's 16-bit gathered GEMM)
        self.variant = variant or ("f16" if precision.startswith("fp16") or precision == "fp32x3" else "bf16")
        # "fp32x3": the fp32 engine (fp32 tensors, the reference's order of operations) with every matrix product computed as
        # three IEEE-half MFMAs on split operands (csrc/gemm.hip, attention.hip): inside north_star's 1e-3 at a multiple of the
        # exact-fp32 MFMA rate
        self.x3 = precision == "fp32x3"
        self.T = torch.float32 if precision in ("fp32", "fp32x3") else ops.LP_DTYPES[self.variant]
        self.device = None
        self.w = None
        self.rng_offset = 0
        self._rng_seed = None  # torch.initial_seed() the device-RNG cursor belongs to
        self.use_native_blocks = True  # one library call per Block instead of ~8 binding calls
        self.fold_attn_scale = True  # 16-bit engines: softmax scale * log2(e) folded into Wq / bq at prepare (tools A/B: False)
        # diagnostic of the IEEE-half trunk: count the 16-bit activations that reach memory at +-65504, the clamp value of the
        # half build's conversions (a checkpoint whose activations leave half's range is then noticed, not silently clamped).
        # Off in the timed path (a few extra launches per Block and one host read per forward); smoke and bench run one
        # forward with it.  Result of the last forward: saturation_count / saturation_checked (elements looked at)
        self.speculate_depth = True  # serialization depth from the previous plan, verified behind the pooled-size read
        self._depth_hint = {}
        self.count_saturation = False
        self.saturation_count = None
        self.saturation_checked = 0
        self._sat = None
        self.exact_attention_core = False  # budget tool only: fp32 attention core inside a 16-bit trunk (binding path)
        self._pad_keys = None
        self.native_plan = True  # build_plan through cdseg_plan_begin / cdseg_plan_finish (False: one binding call per step; tools A/B)
        self._plan_specs = {}
        self._tb_cache = {}  # timestep -> per-Block bias vectors (_t_bias)
        self.eager_kernel_maps = True  # kernel maps of all levels built inside build_plan (False: at first use; tools A/B)
        self._side = {}
        # dominant-branch encoder stage at which the noise-branch encoder is forked onto its side stream (None: serial).  2 (round
        # 6, was 1): the noise branch's throughput-bound 120 k-row Blocks then run next to the dominant branch's deep,
        # latency-bound stages instead of next to its 55 k-row ones: bs = 1 3.87 -> 3.81 ms at 120 k points, 3.11 -> 3.01 ms at
        # 60 k (tools/latency_probe.py, profiles/r06_fork_stage.txt)
        self.fork_stage = int(os.environ["CDSEG_FORK_STAGE"]) if os.environ.get("CDSEG_FORK_STAGE") else 2
        self._work_lock = threading.Lock()
        self._tls = threading.local()  # per host thread: device-RNG cursor
        self.attn_work = 0.0  # algorithmic attention FLOPs issued so far (4 * 16 * H * sum_p L_p^2 per launch)
        # algorithmic HBM bytes of the k = 3 sparse convs issued so far: features in + out, the kernel map as stored
        # (27 x int32 per point), the weights once per launch
        self.conv_bytes = 0.0
        self.conv_deep_bytes = 0.0
        self.attn_bytes = 0.0  # algorithmic attention bytes: q, k, v read once + o written once per launch
        # stages of a bf16 forward that run through the exact-fp32 twin engine instead (keys: n_emb, c_emb, n_enc0..4,
        # c_enc0..2, x, n_dec3..0, c_dec1..0, n_head, c_head).  The error budget of the bf16 mode is measured with it
        # (tools/bf16_budget.py); precision "bf16+head" is hi = {"n_head"}.
        self.hi = frozenset(("n_head", "c_head")) if precision.endswith("+head") else frozenset()
        self._twin = None

    # ------------------------------------------------------------------ weights
    def prepare(self, device):
        if self.w is not None and self.device == device:
            return
        with _lib.use(self.variant):
            self._prepare(device)

    def _prepare(self, device):
        self.device = device
        T = self.T
        w = {}
        self._blocks_to_describe = []

        def f32(t):
            return t.detach().to(device=device, dtype=torch.float32).contiguous()

        def lin(mod, pre):
            w[pre + ".w"] = mod.weight.detach().to(device=device, dtype=T).contiguous()
            w[pre + ".b"] = f32(mod.bias) if mod.bias is not None else None

        # 16-bit engines: the attention kernel is VALU-issue bound, so what can be done once per weight is not done once per
        # (head, query slice): softmax scale * log2(e) is folded into the q rows of the projection (weights and bias, in
        # fp32, before the cast) and the kernel is told so (ops.ATTN_Q_PRESCALED).  The exact-fp32 mode keeps the
        # reference's order of operations (scale applied to the scores).
        self.q_prescaled = T != torch.float32 and self.fold_attn_scale

        def lin_q(mod, pre, scale, rows):
            """Linear whose first `rows` output rows are the attention's q projection."""
            if not self.q_prescaled:
                return lin(mod, pre)
            f = float(scale) * 1.4426950408889634
            wt = mod.weight.detach().to(device=device, dtype=torch.float32).clone()
            wt[:rows] *= f
            w[pre + ".w"] = wt.to(T).contiguous()
            if mod.bias is not None:
                b = f32(mod.bias).clone()
                b[:rows] *= f
                w[pre + ".b"] = b
            else:
                w[pre + ".b"] = None

        def conv(mod, pre):
            w[pre + ".w"] = mod.weight.detach().reshape(mod.weight.shape[0], -1).to(device=device, dtype=T).contiguous()
            w[pre + ".b"] = f32(mod.bias) if mod.bias is not None else None
            if self.x3 and mod.kernel_size == 3 and hasattr(ops, "split16"):
                # fp32x3: the k = 3 convs run the FAST 16-bit gathered GEMM three times into one fp32 output
                # (x_hi W_hi + (x_hi W_lo + x_lo W_hi) / 2048, _conv3): the weight's half pair, made once
                w[pre + ".w_hi"], w[pre + ".w_lo"] = ops.split16(w[pre + ".w"])
                cout = w[pre + ".w"].shape[0]
                if mod.in_channels == cout and cout in (32, 64) and hasattr(ops, "subm_conv3_f32"):
                    # wide stages: the weight-stationary kernel (csrc/conv.hip) with its fp32 accumulate output
                    w[pre + ".wimg_hi"] = ops.subm_conv3_pack(w[pre + ".w_hi"])
                    w[pre + ".wimg_lo"] = ops.subm_conv3_pack(w[pre + ".w_lo"])
                if ("x3.scale", cout) not in w:
                    w[("x3.scale", cout)] = torch.full((cout,), 1.0 / 2048.0, dtype=torch.float32, device=device)
                    w[("x3.zero", cout)] = torch.zeros(cout, dtype=torch.float32, device=device)
            # wide bf16 stages (C = 32 / 64, k = 3): fragment-order image for the weight-stationary conv kernel
            k, cin, cout = mod.kernel_size, mod.in_channels, mod.out_channels
            if (k == 3 and cin == cout and hasattr(ops, "subm_conv3_pack") and
                    ops.subm_conv3_ok(torch.empty((1, cin), dtype=T, device="meta"))):
                w[pre + ".wimg"] = ops.subm_conv3_pack(w[pre + ".w"])

        def ln(mod, pre):
            w[pre + ".g"], w[pre + ".b"] = f32(mod.weight), f32(mod.bias)

        def bn(mod, pre):  # eval BatchNorm1d folded to scale / shift (ref: eps=1e-3, ptv3.py:1440)
            scale = mod.weight.detach().double() / torch.sqrt(mod.running_var.detach().double() + mod.eps)
            shift = mod.bias.detach().double() - mod.running_mean.detach().double() * scale
            w[pre + ".scale"], w[pre + ".shift"] = f32(scale), f32(shift)

        self.block_desc = {}

        def block(mod, pre):
            self._blocks_to_describe.append((mod, pre))
            conv(mod.cpe[0], pre + ".cpe0")
            lin(mod.cpe[1], pre + ".cpe1")
            ln(mod.cpe[2], pre + ".cpe2")
            ln(mod.norm1[0], pre + ".norm1")
            lin_q(mod.attn.qkv, pre + ".qkv", mod.attn.scale, mod.channels)
            lin(mod.attn.proj, pre + ".proj")
            ln(mod.norm2[0], pre + ".norm2")
            lin(mod.mlp[0].fc1, pre + ".fc1")
            lin(mod.mlp[0].fc2, pre + ".fc2")

        def stem(mod, pre):
            # (out, k,k,k, in) -> (out, kvol * in_pad): input channels zero-padded to a 16-byte multiple so
            # the k=5 stem runs on the gathered-A MFMA GEMM like every other sparse conv
            cw = mod.stem.conv.weight.detach()
            cout, cin = cw.shape[0], cw.shape[-1]
            cpad = (cin + 7) // 8 * 8
            wp = torch.zeros(cout, cw.shape[1] ** 3, cpad, dtype=torch.float32)
            wp[:, :, :cin] = cw.reshape(cout, -1, cin).float().cpu()
            w[pre + ".w"] = wp.reshape(cout, -1).to(device=device, dtype=T).contiguous()
            w[pre + ".cpad"] = cpad
            if cw.shape[1] == 5 and cpad == 8 and hasattr(ops, "stem5_pack") and ops.stem5_ok(cout, T):
                w[pre + ".wimg"] = ops.stem5_pack(w[pre + ".w"])  # LDS image of the map-free stem kernel (csrc/stem.hip)
            bn(mod.stem.norm, pre + ".bn")

        def pool(mod, pre):
            lin(mod.proj, pre + ".proj")
            bn(mod.norm[0], pre + ".bn")
            pw = w[pre + ".proj.w"]
            if hasattr(ops, "pool_fused_ok") and ops.pool_fused_ok(pw.shape[1], pw.shape[0], T):
                w[pre + ".proj.wimg"] = ops.pool_fused_pack(pw)  # one-launch pooling of the wide stages (csrc/pool.hip)

        def unpool(mod, pre):
            lin(mod.proj[0], pre + ".proj")
            bn(mod.proj[1], pre + ".proj_bn")
            lin(mod.proj_skip[0], pre + ".skip")
            bn(mod.proj_skip[1], pre + ".skip_bn")
            if mod.skip_connection_mode == "cat":
                f = (2 ** -0.5 if mod.skip_connection_scale else 1.0)
                if mod.skip_connection_scale_i is not None:
                    f *= 0.8 ** (int(mod.skip_connection_scale_i) - 1)
                wc = mod.proj_cat[0].weight.detach().float()
                c = wc.shape[0]
                w[pre + ".cat_a.w"] = (wc[:, :c] * f).to(device=device, dtype=T).contiguous()
                w[pre + ".cat_b.w"] = wc[:, c:].to(device=device, dtype=T).contiguous()
                w[pre + ".cat.b"] = f32(mod.proj_cat[0].bias)

        bb = self.model.backbone
        stem(bb._n_embedding, "n_emb")
        for s in range(bb.n_num_stages):
            enc = getattr(bb._n_enc, f"enc{s}")
            if s > 0:
                pool(enc.down, f"n_enc{s}.down")
            for name, mod in enc._modules.items():
                if name.startswith("block"):
                    block(mod, f"n_enc{s}.{name}")
        for s in range(bb.n_num_stages - 1):
            dec = getattr(bb._n_dec, f"dec{s}")
            unpool(dec.up, f"n_dec{s}.up")
            for name, mod in dec._modules.items():
                if name.startswith("block"):
                    block(mod, f"n_dec{s}.{name}")
        if isinstance(bb._n_head, torch.nn.Linear):
            lin(bb._n_head, "n_head")
        if bb.condition:
            stem(bb._c_embedding, "c_emb")
            tw, tb, toff = [], [], {}
            off = 0
            for s in range(bb.c_num_stages):
                enc = getattr(bb._c_enc, f"enc{s}")
                if s > 0:
                    pool(enc.down, f"c_enc{s}.down")
                for name, mod in enc._modules.items():
                    if name.startswith("block"):
                        block(mod, f"c_enc{s}.{name}")
                        if bb.T_dim != -1:
                            tw.append(f32(mod.t_mlp.weight))
                            tb.append(f32(mod.t_mlp.bias))
                            toff[f"c_enc{s}.{name}"] = (off, off + mod.channels)
                            off += mod.channels
            for s in range(bb.c_num_stages - 1):
                dec = getattr(bb._c_dec, f"dec{s}")
                unpool(dec.up, f"c_dec{s}.up")
                for name, mod in dec._modules.items():
                    if name.startswith("block"):
                        block(mod, f"c_dec{s}.{name}")
                        if bb.T_dim != -1:
                            tw.append(f32(mod.t_mlp.weight))
                            tb.append(f32(mod.t_mlp.bias))
                            toff[f"c_dec{s}.{name}"] = (off, off + mod.channels)
                            off += mod.channels
            if isinstance(bb._c_head, torch.nn.Linear):
                lin(bb._c_head, "c_head")
            if bb.T_dim != -1:
                w["t.fc1.w"], w["t.fc1.b"] = f32(bb.fc_t1.weight), f32(bb.fc_t1.bias)
                w["t.fc2.w"], w["t.fc2.b"] = f32(bb.fc_t2.weight), f32(bb.fc_t2.bias)
                w["t.mlp.w"], w["t.mlp.b"] = torch.cat(tw).contiguous(), torch.cat(tb).contiguous()
                w["t.table"] = f32(self.model.t_emb_table)
                self.t_slices = toff
            cb = bb._tm_dec0.cross_block2
            conv(cb.q_cpe[0], "x.q_cpe0"); lin(cb.q_cpe[1], "x.q_cpe1"); ln(cb.q_cpe[2], "x.q_cpe2")
            conv(cb.kv_cpe[0], "x.kv_cpe0"); lin(cb.kv_cpe[1], "x.kv_cpe1"); ln(cb.kv_cpe[2], "x.kv_cpe2")
            ln(cb.q_norm1[0], "x.q_norm1"); ln(cb.kv_norm1[0], "x.kv_norm1"); ln(cb.q_norm2[0], "x.q_norm2")
            lin_q(cb.attn.q, "x.q", cb.attn.scale, cb.q_channels); lin(cb.attn.kv, "x.kv"); lin(cb.attn.proj, "x.proj")
            lin(cb.mlp[0].fc1, "x.fc1"); lin(cb.mlp[0].fc2, "x.fc2")
            if cb.tm_feat != 1.0:
                c = cb.q_channels
                w["x.feat_scale"] = torch.full((c,), cb.tm_feat, dtype=torch.float32, device=device)
                w["x.feat_zero"] = torch.zeros(c, dtype=torch.float32, device=device)
            elif (hasattr(ops, "block_rr_pack") and ops.block_rr_ok(cb.q_channels, T) and cb.q_channels in ops.DEEP_CHANNELS
                  and w["x.fc1.w"].shape[0] == 4 * cb.q_channels):  # (deep.hip's streamed-weight image only: the LDS-resident
                # C = 32 / 64 pack needs the head weights too)
                # the cross block's tail (x += proj(attn); h = LN(x); x += MLP(h), ptv3.py:1205-1222) has a Block tail's
                # shape: one launch on the streamed-weight kernel (csrc/deep.hip) instead of three GEMMs + their second passes
                _, w["x.tail_img"] = ops.block_rr_pack(cb.q_channels, None, None, w["x.proj.w"], w["x.fc1.w"], w["x.fc2.w"])
        self.w = w
        self._tb_cache = {}
        if T == torch.float16:
            # IEEE half ends at 65504 and torch's cast does not saturate (the kernels' own float -> half conversions do): a weight
            # beyond the range would enter every product as inf.  One reduction over the cast weights, one host read
            flags = [(~torch.isfinite(t)).any() for t in w.values() if torch.is_tensor(t) and t.dtype == torch.float16]
            if flags and bool(torch.stack(flags).any().item()):
                bad = [k for k, t in w.items() if torch.is_tensor(t) and t.dtype == torch.float16 and not bool(torch.isfinite(t).all())]
                raise CdsegError(f"weights outside IEEE half's range (|w| > 65504) in {bad[:4]}: run this checkpoint with "
                                 f"precision='bf16+head' (bfloat16 trunk, fp32's exponent range)")
        # native Block executor: one descriptor per Block (weights never move after prepare)
        # wide bf16 stages: fragment images of the Block's head / tail weights (register-resident kernels, blockrr.hip)
        for mod, pre in self._blocks_to_describe:
            if hasattr(ops, "block_rr_pack") and ops.block_rr_ok(mod.channels, T) and w[pre + ".fc1.w"].shape[0] == 4 * mod.channels:
                himg, w[pre + ".tail_img"] = ops.block_rr_pack(
                    mod.channels, w[pre + ".cpe1.w"], w[pre + ".qkv.w"], w[pre + ".proj.w"], w[pre + ".fc1.w"], w[pre + ".fc2.w"])
                if ops.block_rr_head_on(mod.channels):  # C = 32 / 64: slower than the 64-row-tile fused head; deep: on
                    w[pre + ".head_img"] = himg
        # (fp32x3 issues its Blocks from the binding path: its convs are three launches on operands the descriptor does not carry)
        self.native_blocks = hasattr(ops, "block_forward") and self.use_native_blocks and not self.x3
        if self.native_blocks:
            for mod, pre in self._blocks_to_describe:
                t = dict(cpe_conv_w=w[pre + ".cpe0.w"], cpe_conv_b=w[pre + ".cpe0.b"], cpe_lin_w=w[pre + ".cpe1.w"],
                         cpe_lin_b=w[pre + ".cpe1.b"], cpe_ln_g=w[pre + ".cpe2.g"], cpe_ln_b=w[pre + ".cpe2.b"],
                         norm1_g=w[pre + ".norm1.g"], norm1_b=w[pre + ".norm1.b"], qkv_w=w[pre + ".qkv.w"],
                         qkv_b=w[pre + ".qkv.b"], proj_w=w[pre + ".proj.w"], proj_b=w[pre + ".proj.b"],
                         norm2_g=w[pre + ".norm2.g"], norm2_b=w[pre + ".norm2.b"], fc1_w=w[pre + ".fc1.w"],
                         fc1_b=w[pre + ".fc1.b"], fc2_w=w[pre + ".fc2.w"], fc2_b=w[pre + ".fc2.b"])
                if hasattr(ops, "subm_conv3_pack") and ops.subm_conv3_ok(torch.empty((1, mod.channels), dtype=T, device="meta")):
                    t["cpe_conv_wimg"] = w[pre + ".cpe0.wimg"]
                if (pre + ".tail_img") in w:
                    t["tail_img"] = w[pre + ".tail_img"]
                if (pre + ".head_img") in w:
                    t["head_img"] = w[pre + ".head_img"]
                self.block_desc[pre] = ops.make_block_desc(T, mod.channels, mod.attn.num_heads, w[pre + ".fc1.w"].shape[0],
                                                           mod.attn.scale, 1e-5, t,
                                                           attn_flags=ops.ATTN_Q_PRESCALED if self.q_prescaled else 0,
                                                           x3=self.x3)
        self._scratch = {}
        self._scratch_bytes = {}

    def _eng(self, key):
        """The engine that runs stage `key`: this one, or its exact-fp32 twin for the stages named in self.hi."""
        if key not in self.hi or self.T == torch.float32:
            return self
        heads_only = self.hi <= {"n_head", "c_head"}  # precision "*+head": the heads' weights are all that is needed in fp32
        if self._twin is None or isinstance(self._twin, _HeadsInFp32) != heads_only or self._twin.device != self.device:
            if heads_only:
                self._twin = _HeadsInFp32(self)
            else:
                self._twin = Engine(self.model, "fp32", variant=self.variant)
                self._twin.use_native_blocks = self.use_native_blocks
        self._twin.prepare(self.device)
        return self._twin

    def _fit(self, st, exact=False):
        """Bring a State's shadow copy to this engine's dtype (stages of different precision meet here).  exact: the
        shadow is a plain rounding of the fp32 residual stream (true behind a Block), so an fp32 stage reads the
        stream itself."""
        if st is not None and st.xc.dtype != self.T:
            st.xc = st.x if (exact and self.T == torch.float32) else st.xc.to(self.T)
        return st

    def scratch(self, nbytes):
        key = ops.current_stream_id()  # one scratch arena per stream (the two branches run concurrently)
        buf = self._scratch.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = self._scratch[key] = torch.empty(int(nbytes * 1.1) + 4096, dtype=torch.uint8, device=self.device)
        return buf

    def _on_side_stream(self, fn):
        """Run fn's launches on the engine's side stream, ordered after everything issued so far on the bound
        stream.  Returns (fn's result, completion event).  Tensors allocated inside belong to the side stream's
        allocator pool; they are consumed on the main stream only after the completion event."""
        main = torch.cuda.current_stream()
        side = self._side.get(main.cuda_stream)  # one side stream per calling stream (scenes may run on several)
        if side is None:
            side = self._side[main.cuda_stream] = torch.cuda.Stream(device=self.device)
        fork_ev = ops.record_event()
        try:
            with torch.cuda.stream(side):
                ops.bind_stream(side)
                ops.wait_event(fork_ev)
                out = fn()
                done = ops.record_event()
        finally:
            ops.bind_stream(main)
        return out, done

    # ------------------------------------------------------------------ plan
    def _pin(self, name, numel, ring=1, dtype=torch.int32):
        """Per host thread: pinned int32 staging buffers (`ring` of them, handed out in turn together with the event that
        guards the buffer's previous use).  Returns (tensor, its numpy view, slot dict)."""
        slots = getattr(self._tls, "pins", None)
        if slots is None:
            slots = self._tls.pins = {}
        ent = slots.get(name)
        if ent is None:
            ent = slots[name] = {"i": 0, "bufs": [None] * ring}
        i = ent["i"] = (ent["i"] + 1) % ring
        slot = ent["bufs"][i]
        if slot is None or slot["t"].numel() < numel:
            t = torch.empty(max(256, int(numel * 1.5)), dtype=dtype, pin_memory=True)
            slot = ent["bufs"][i] = {"t": t, "np": t.numpy(), "ev": None}
        elif slot["ev"] is not None:
            slot["ev"].synchronize()  # (a copy out of this buffer issued `ring` plans ago: long done)
        return slot["t"], slot["np"], slot

    def _native_spec(self, depth, n_cum, c_cum, all_cum, end_bit):
        """The static description cdseg_plan_begin / cdseg_plan_finish need (include/cdseg.h cdseg_plan_spec), cached per
        serialization depth; None when this model / depth is outside what the native builder covers (the per-op path runs)."""
        key = (depth, end_bit + 2 <= 64)
        if key in self._plan_specs:
            return self._plan_specs[key]
        bb = self.model.backbone
        ent = None
        if self._pad_keys is None:
            self._pad_keys = self._collect_pad_keys()
        used = sorted({CURVES.index(o) for o in bb.order} - {0})
        curves = sorted({CURVES.index(o) for o in bb.order})
        nlev = len(all_cum) - 1
        strictly = all(b > a for cum in (n_cum, c_cum) for a, b in zip(cum[:-1], cum[1:]))
        li = {c: i for i, c in enumerate(all_cum)}
        links = sorted({(li[a], li[b]) for cum in (n_cum, c_cum) for a, b in zip(cum[:-1], cum[1:]) if li[a] >= 1} |
                       {(i, i + 1) for i in range(1, nlev) if all_cum[i + 1] - all_cum[i] == 1})
        ok = (1 <= nlev <= 8 and strictly and len(used) <= 3 and (not used or (end_bit + 2 <= 64 and FUSED_CURVE_SORT)) and
              len(links) <= 16 and 1 <= len(self._pad_keys) <= 4 and hasattr(_lib.load(), "cdseg_plan_finish"))
        if ok:
            sp = _lib.PlanSpec()
            sp.nlev = nlev
            for i, c in enumerate(all_cum):
                sp.cum[i] = c
            sp.ncurve = len(used)
            for i, c in enumerate(used):
                sp.curve_rows[i] = c
            sp.nslot_curve = len(curves)
            for i, c in enumerate(curves):
                sp.slot_curve[i] = -1 if c == 0 else used.index(c)
            sp.nlink = len(links)
            for i, (a, b) in enumerate(links):
                sp.link_a[i], sp.link_b[i] = a, b
            sp.npad = len(self._pad_keys)
            for i, (ps, fl) in enumerate(self._pad_keys):
                sp.pad_patch[i], sp.pad_flash[i] = int(ps), int(bool(fl))
            ent = (sp, used, curves, links)
        self._plan_specs[key] = ent
        return ent

    def _collect_pad_keys(self):
        bb = self.model.backbone
        return sorted({(int(m_.patch_size), bool(m_.enable_flash)) for m_ in bb.modules()
                       if hasattr(m_, "patch_size") and hasattr(m_, "enable_flash")} |
                      {(int(m_.q_patch_size), bool(m_.enable_flash)) for m_ in bb.modules()
                       if hasattr(m_, "q_patch_size")})

    def _build_plan_native(self, ent, grid, offset_dev, offset_host, n, depth, end_bit, gmax_host, n_cum, c_cum, all_cum,
                           while_device_works):
        """build_plan through the native builder (csrc/plan.hip): two library calls around the one host read, every plan item
        a view into one of two arenas per call.  Same items, bit for bit, as the per-op path below
        (tests/test_gpu_e2e.py::test_native_plan_equals_per_op_plan)."""
        sp, used, curves, links = ent
        nb = len(offset_host) if offset_host is not None else int(offset_dev.numel())
        grid = grid.contiguous()
        offset_dev = offset_dev.contiguous()  # (int64, the caller's cumulative offsets: Engine._setup / TrainGraph hand it over so)
        if offset_dev.dtype != torch.int64:
            offset_dev = offset_dev.to(torch.int64)
        nlev = sp.nlev
        nmeta = nlev * (1 + nb) + 1
        mt, mnp, _ = self._pin("meta", nmeta)
        call = ops.NativePlanCall(sp, grid, offset_dev, n, nb, depth, end_bit, gmax_host, mt)
        opin = None
        if offset_host is None:  # the reference's dict carries no host copy of the offsets: they ride along with the one read
            opin, onp, _ = self._pin("offs", nb, dtype=torch.int64)
            opin[:nb].copy_(offset_dev, non_blocking=True)
        call.begin(0)
        ev = ops.record_event()
        call.begin(1)  # the level-0 curve sort does not need the pooled sizes: queued before the host waits
        if while_device_works is not None:
            while_device_works()
        perm0, grid0, bat0, code0, cl_all, seg_all, orders0 = call.begin_views()
        ev.synchronize()  # the one sync: pooled sizes, duplicate-voxel count, grid maximum
        flat = mnp[:nmeta].tolist()
        if opin is not None:
            offset_host = onp[:nb].tolist()
        if gmax_host is not None:
            true_depth = int(gmax_host.item()).bit_length()
            self._depth_hint[grid.device] = true_depth
            if true_depth != depth:  # the guess was wrong: everything built so far used the wrong code width
                return self.build_plan(grid, offset_dev, offset_host, n, _exact_depth=true_depth)
        if flat[-1]:
            raise DuplicateVoxelsError(
                f"input has {flat[-1]} duplicate voxels (points sharing (batch, grid_coord) with another "
                f"point): the model expects one point per voxel - voxelise first (GridSample)", flat[-1])
        sid = ops.current_stream_id()
        plan = Plan()
        plan.perm0, plan.n_cum, plan.c_cum, plan.depth = perm0, n_cum, c_cum, depth
        plan.native = call  # (keeps the arenas' owner alive with the plan)
        offs0 = [0] + [int(v) for v in offset_host]
        m = [flat[i * (1 + nb)] for i in range(nlev)]
        offs_rows = [offs0] + [[0] + [v + 1 for v in flat[i * (1 + nb) + 1:(i + 1) * (1 + nb)]] for i in range(nlev)]
        slot = {}

        def pads_pin(count):
            pt, _, sl = self._pin("pads", count, ring=4)
            slot["s"] = sl
            return pt

        off, info = call.finish(m, offs_rows, pads_pin)
        slot["s"]["ev"] = ops.record_event()
        f32, f64 = call.f32, call.f64
        it = iter(off)
        sizes = [n] + m
        lv0 = plan.levels[0] = Level(0, depth, n, grid0, bat0, code0, offs0)
        levels = [lv0]
        for i in range(nlev):
            go, bo, co = next(it), next(it), next(it)
            mi, cum = m[i], all_cum[i + 1]
            lv = Level(cum, depth - cum, mi, None, None, None, offs_rows[i + 1], lazy=(f32, go, bo, f64, co))
            plan.levels[cum] = lv
            levels.append(lv)
            plan.links[(0, cum)] = ((cl_all[i], seg_all[i]), sid, None)
        for a, b in links:
            clo, sgo = next(it), next(it)
            plan._link_lazy[(all_cum[a], all_cum[b])] = (sid, ((f32, clo, sizes[a], None), (f32, sgo, sizes[b] + 1, None)))
        for i in range(nlev):
            if all_cum[i + 1] - all_cum[i] == 1:
                levels[i].parent = (levels[i + 1], plan.link(all_cum[i], all_cum[i + 1]))
        for i, lv in enumerate(levels):
            lv._nbr_lazy[(3, True)] = (sid, ((f32, next(it), 27 * sizes[i], (27, sizes[i])),))
        for i, lv in enumerate(levels):
            o = next(it)
            if o >= 0:
                lv._nbr_lazy["child_info"] = (sid, ((f64, o, sizes[i + 1], None),))
        pos = next(it)
        nc = len(used)
        for k, c in enumerate(used):
            lv0._order[c] = (orders0[k], sid, None)
        for i in range(1, nlev + 1):
            for k, c in enumerate(used):
                levels[i]._order_lazy[c] = (sid, ((f32, pos + k * sizes[i], sizes[i], None),))
            pos += nc * sizes[i]
        q = 5
        per = []
        for i, lv in enumerate(levels):
            for key in self._pad_keys:
                oo, op, ot = next(it), next(it), next(it)
                K, n_pad, npatch, max_len, bits = info[q:q + 5]
                q += 5
                lv._pad_lazy[key] = (K, n_pad, max_len, struct.unpack("d", struct.pack("q", bits))[0], f32, oo, op, ot, nb,
                                     npatch)
                per.append((lv, key, n_pad))
        gb, wb = next(it), next(it)
        pos = 0
        for lv, key, n_pad in per:
            for c in curves:
                lv._slot_lazy[(c,) + key] = (f32, gb + pos, wb + pos, n_pad, sid)
                pos += n_pad
        return plan

    def build_plan(self, grid, offset_dev, offset_host, n, _exact_depth=None, while_device_works=None, defer_pads=False):
        """Serialization, pooled levels, kernel-map sources, padding / slot plans of one forward.
        Host reads: ONE in steady state (round 5) - the pooled sizes.  The serialization depth (`int(grid_coord.max())
        .bit_length()`, structure.py:66: the reference's first host sync) is taken from the previous call's plan, the grid
        maximum of THIS call travels to the host behind the pooled-size read, and a mismatch (a scene on a coarser / finer
        grid than the last one) rebuilds the plan with the right depth - results never depend on the guess.

        Round 6 (bs = 1 is the reference's protocol and there the device used to idle ~0.7 ms per scene behind this function's
        host work, profiles/r06_bs1_gaps.txt): nothing in here blocks the host except the one read, and the device has work
        queued across it - the pooled sizes travel through a pinned buffer behind an event, the level-0 curve sort (which does
        not need them) is issued BEFORE the host waits, `while_device_works` (the caller's host-only work: the random draws)
        runs in that shadow too, the kernel maps are built before the padding plans' host arithmetic (plain Python ints), and
        the padding tables go up through a pinned buffer without a blocking copy.  `defer_pads`: the padding / slot plans (host
        arithmetic + one upload + one launch; first needed by the first Block's attention) are left to `plan.finish_pads()`,
        which `backbone` calls once the stem is queued - device work for that stretch of host time."""
        bb = self.model.backbone
        nb = len(offset_host) if offset_host is not None else int(offset_dev.numel())
        on_gpu = grid.is_cuda
        hint = self._depth_hint.get(grid.device) if _exact_depth is None else None
        gmax_host = gmax_dev = None
        if _exact_depth is not None:
            depth = _exact_depth
        elif hint is not None and on_gpu and self.speculate_depth:
            depth = hint
            gmax_host = getattr(self._tls, "gmax_pin", None)  # one pinned word per issuing host thread (build_plan does not
            if gmax_host is None:                               # return before it has read it)
                gmax_host = self._tls.gmax_pin = torch.empty(1, dtype=torch.int64, pin_memory=True)
        else:
            gmax_dev = ops.grid_max(grid)
            depth = int(gmax_dev.item()).bit_length()
            self._depth_hint[grid.device] = depth
        # same guards as the reference (structure.py:69,74)
        assert depth * 3 + nb.bit_length() <= 63, "serialization code does not fit int64"
        assert depth <= 16, "grid extent exceeds 2^16 voxels per axis"
        end_bit = min(64, 3 * depth + max(1, nb.bit_length()))

        def cum_depths(strides):
            cum, d = [0], depth
            for s in strides:
                pd = _pooling_depth(s)
                if pd > d:  # ref: ptv3.py:466-467
                    pd = 0
                d -= pd
                cum.append(cum[-1] + pd)
            return cum

        n_cum = cum_depths(bb.n_stride)
        c_cum = cum_depths(bb.c_stride) if bb.condition else [0]
        all_cum = sorted(set(n_cum + c_cum))
        coarse = [c for c in all_cum if c > 0]
        if on_gpu and self.native_plan and (gmax_host is not None or _exact_depth is not None):
            ent = self._native_spec(depth, n_cum, c_cum, all_cum, end_bit)
            if ent is not None:
                return self._build_plan_native(ent, grid, offset_dev, offset_host, n, depth, end_bit, gmax_host, n_cum, c_cum,
                                               all_cum, while_device_works)
        if offset_host is None:  # per-op path without the caller's hint: one more host read
            offset_host = [int(v) for v in offset_dev.cpu().tolist()]
        if gmax_dev is None:
            gmax_dev = ops.grid_max(grid)
            if gmax_host is not None:
                gmax_host.copy_(gmax_dev, non_blocking=True)  # complete once the pooled-size read below has returned
        # index of every batch element's last point, on the device from the device offsets (a host list would be a blocking
        # pageable copy in the middle of the plan kernels)
        last_idx = (offset_dev.to(torch.int32) - 1) if coarse else None
        batch = ops.offset2batch(offset_dev, n)
        zc = ops.encode(grid, batch, depth, "z")
        zs, perm0 = ops.sort_pairs(zc, None, end_bit=end_bit)
        grid0, bat0 = ops.plan_gather_grid(grid, perm0, zs, depth)
        code0 = ops.encode4(grid0, bat0, depth)

        plan = Plan()
        plan.perm0, plan.n_cum, plan.c_cum, plan.depth = perm0, n_cum, c_cum, depth
        offs0 = [0] + [int(v) for v in offset_host]
        lv0 = plan.levels[0] = Level(0, depth, n, grid0, bat0, code0, offs0)
        used = sorted({CURVES.index(o) for o in bb.order} - {0})

        def sort_level0_curves():  # the level-0 orders of all curves in use with ONE sort (Onesweep's cost is mostly fixed)
            if used and end_bit + 2 <= 64 and FUSED_CURVE_SORT and not all(c in lv0._order for c in used):
                srt = ops.sort_curves(code0, used, end_bit)
                for k, c in enumerate(used):
                    lv0._order[c] = (srt[k], ops.current_stream_id(), None)

        if coarse:
            dev = grid.device
            cl_all, seg_all, meta = ops.pool_levels(zs, [3 * cum for cum in coarse], last_idx)
            tmp = [(cl_all[i], seg_all[i]) for i in range(len(coarse))]
            if on_gpu:
                # the pooled sizes (+ the duplicate-voxel count) come back through a pinned buffer; the host waits for THAT
                # copy only, with the curve sort already queued behind it and its own host-only work done in the meantime
                mt, mnp, _ = self._pin("meta", meta.numel())
                mt[:meta.numel()].copy_(meta, non_blocking=True)
                ev = ops.record_event()
                sort_level0_curves()
                if while_device_works is not None:
                    while_device_works()
                    while_device_works = None
                ev.synchronize()  # the one sync for all pooled sizes
                flat = mnp[:meta.numel()].tolist()
            else:
                flat = meta.cpu().tolist()
            if gmax_host is not None:
                true_depth = int(gmax_host.item()).bit_length()
                self._depth_hint[grid.device] = true_depth
                if true_depth != depth:  # the guess was wrong: everything built so far used the wrong code width
                    return self.build_plan(grid, offset_dev, offset_host, n, _exact_depth=true_depth, defer_pads=defer_pads)
                gmax_host = None
            if flat[-1]:
                # the model's input contract (GridSample upstream, structure.py:39-102 downstream): one point per voxel.
                # The kernel maps and the derived coarse orders assume it - refuse instead of computing something else
                # (the training graph catches this one and folds the surplus points onto their voxel, train_graph.py)
                raise DuplicateVoxelsError(
                    f"input has {flat[-1]} duplicate voxels (points sharing (batch, grid_coord) with another "
                    f"point): the model expects one point per voxel - voxelise first (GridSample)", flat[-1])
            meta_h = [flat[i * (1 + nb):(i + 1) * (1 + nb)] for i in range(len(coarse))]
            host = [r[0] for r in meta_h] + [v for r in meta_h for v in r[1:]]
            for i, cum in enumerate(coarse):
                m = host[i]
                e = host[len(coarse) + i * nb: len(coarse) + (i + 1) * nb]
                g, b, c4 = ops.pool_gather(tmp[i][1], m, n, cum, grid0, bat0, code0)
                plan.levels[cum] = Level(cum, depth - cum, m, g, b, c4, [0] + [v + 1 for v in e])
                plan.links[(0, cum)] = ((tmp[i][0], tmp[i][1]), ops.current_stream_id(), None)
            # kernel maps are derived top-down (coarsest level by search, every finer one from its parent's map)
            cums = sorted(plan.levels)
            for fa, co in zip(cums[:-1], cums[1:]):
                if co - fa == 1:
                    plan.levels[fa].parent = (plan.levels[co], plan.link(fa, co))
            if on_gpu and self.eager_kernel_maps:
                # ... and built NOW (every Block's conv needs its level's map): device work for the stretch of host
                # arithmetic below.  Issued on the plan's stream ahead of any fork, so consumers need no event
                share, SHARE_EVENTS.on = SHARE_EVENTS.on, False
                try:
                    for cum in reversed(cums):
                        plan.levels[cum].nbr(3, True)
                finally:
                    SHARE_EVENTS.on = share
            # curve orders of the pooled levels: derived from the level-0 orders (hierarchical keys), not sorted
            if used:
                sort_level0_curves()
                derived = ops.coarse_orders([t[0] for t in tmp], [lv0.order(c) for c in used], host[:len(coarse)])
                for i, cum in enumerate(coarse):
                    for k, c in enumerate(used):
                        plan.levels[cum]._order[c] = (derived[i][k], ops.current_stream_id(), None)
        if while_device_works is not None:
            while_device_works()
        if gmax_host is not None:  # no pooled level, so no read has happened yet: verify the guessed depth now
            true_depth = int(gmax_dev.item()).bit_length()
            self._depth_hint[grid.device] = true_depth
            if true_depth != depth:
                return self.build_plan(grid, offset_dev, offset_host, n, _exact_depth=true_depth, defer_pads=defer_pads)
        def finish_pads():
            # every padding plan the model will ask for, uploaded with ONE host->device copy
            if self._pad_keys is None:  # static per model: walk the module tree once
                self._pad_keys = self._collect_pad_keys()
            pad_keys = self._pad_keys
            flat_up, meta = [], []
            for cum, lv in plan.levels.items():
                for key in pad_keys:
                    K, offs, offs_pad, patch_start = lv.pad_host_py(*key)
                    meta.append((lv, key, K, offs_pad, patch_start, len(offs), len(offs_pad), len(patch_start)))
                    flat_up += offs
                    flat_up += offs_pad
                    flat_up += patch_start
            if on_gpu:
                pt, pnp, slot = self._pin("pads", len(flat_up), ring=4)
                pnp[:len(flat_up)] = flat_up
                up = torch.empty(len(flat_up), dtype=torch.int32, device=grid.device)
                up.copy_(pt[:len(flat_up)], non_blocking=True)
                slot["ev"] = ops.record_event()
            else:
                up = torch.tensor(flat_up, dtype=torch.int32, device=grid.device)
            pos = 0
            for lv, key, K, offs_pad, patch_start, la, lb, lc in meta:
                lv.set_pad(key, K, offs_pad, patch_start, up[pos:pos + la], up[pos + la:pos + la + lb],
                           up[pos + la + lb:pos + la + lb + lc])
                pos += la + lb + lc
            # ... and every slot plan (level x curve x patch key) with ONE launch
            curves = sorted({CURVES.index(o) for o in bb.order})
            items, where = [], []
            for cum, lv in plan.levels.items():
                for key in pad_keys:
                    K, n_pad, offs, offs_pad = lv.pad(*key)[:4]
                    for c in curves:
                        items.append((lv.order(c), offs, offs_pad, K, n_pad))
                        where.append((lv, (c,) + key))
            cur = ops.current_stream_id()
            for (lv, key), gw in zip(where, ops.pad_plan_batch(items, nb)):
                lv._slots[key] = (gw, cur, None)

        if defer_pads and on_gpu:
            plan._finish = finish_pads
        else:
            finish_pads()
        return plan

    # ------------------------------------------------------------------ layers
    def _buf(self, rows, cols, dtype):
        return torch.empty((rows, cols), dtype=dtype, device=self.device)

    FUSE_LN_MAX_C = 512  # rows up to this width are finished by one GEMM block -> LayerNorm in the epilogue

    def _count_conv(self, n, c, esz):
        if c <= 64 and esz == 2:  # the weight-stationary kernel of the wide stages (conv.hip): HBM / gather bound
            self.conv_bytes += 2.0 * n * c * esz + 27.0 * 4 * n + 27.0 * c * c * esz
        else:  # the gathered GEMM (gemm.hip): MFMA / LDS-DMA bound (its FLOPs: forward_work()["conv_deep"])
            self.conv_deep_bytes += 2.0 * n * c * esz + 27.0 * 4 * n + 27.0 * c * c * esz

    def _conv3(self, xc, pre, lv, y):
        """y = SubMConv3d_k3(xc) (ref: ptv3.py:356-362): the weight-stationary kernel on the wide bf16 stages, the
        gathered-A GEMM elsewhere."""
        w = self.w
        self._count_conv(lv.n, xc.shape[1], xc.element_size())
        if (pre + ".w_hi") in w and y.dtype == torch.float32:
            # fp32x3: x = x_hi + x_lo / 2048, W = W_hi + W_lo / 2048 (IEEE-half pairs, 22 significant bits); the dropped
            # x_lo W_lo term is <= 2^-22 of the product.  Launches 2 and 3 scale their sum by 2^-11 and add the running output
            cout = w[pre + ".w"].shape[0]
            sc, ze = w[("x3.scale", cout)], w[("x3.zero", cout)]
            xh, xl = ops.split16(xc)
            nbr = lv.nbr(3, True)
            if (pre + ".wimg_hi") in w and ops.subm_conv3_ok(xh) and y.stride(0) == y.shape[1]:
                ops.subm_conv3_f32(xh, w[pre + ".wimg_hi"], w[pre + ".b"], nbr, y)
                ops.subm_conv3_f32(xh, w[pre + ".wimg_lo"], None, nbr, y, out_scale=1.0 / 2048.0, accumulate=True)
                ops.subm_conv3_f32(xl, w[pre + ".wimg_hi"], None, nbr, y, out_scale=1.0 / 2048.0, accumulate=True)
                return
            ops.gemm(xh, w[pre + ".w_hi"], y, bias=w[pre + ".b"], nbr=nbr, nbr_kmajor=True, kvol=27)
            ops.gemm(xh, w[pre + ".w_lo"], y, scale=sc, shift=ze, res=y, nbr=nbr, nbr_kmajor=True, kvol=27)
            ops.gemm(xl, w[pre + ".w_hi"], y, scale=sc, shift=ze, res=y, nbr=nbr, nbr_kmajor=True, kvol=27)
            return
        if (pre + ".wimg") in w and ops.subm_conv3_ok(xc):
            ops.subm_conv3(xc, w[pre + ".wimg"], w[pre + ".b"], lv.nbr(3, True), y)
        else:
            ops.gemm(xc, w[pre + ".w"], y, bias=w[pre + ".b"], nbr=lv.nbr(3, True), nbr_kmajor=True, kvol=27)

    def _cpe(self, st, pre, xc, tbias=None, next_norm=None):
        """x += LN(Linear(SubMConv3d(xc)))  [+ t bias]   (ref: ptv3.py:401-411).
        next_norm: weight prefix of the LayerNorm that follows on x; returns its output h (dtype T)."""
        w, lv = self.w, st.level
        c = st.x.shape[1]
        y = self._buf(lv.n, c, self.T)
        self._conv3(xc, pre + "0", lv, y)
        h = self._buf(lv.n, c, self.T) if next_norm else None
        if c <= self.FUSE_LN_MAX_C:
            ops.gemm(y, w[pre + "1.w"], st.x, bias=w[pre + "1.b"], ln_pre=(w[pre + "2.g"], w[pre + "2.b"]), res=st.x,
                     colbias=tbias, ln_post=(w[next_norm + ".g"], w[next_norm + ".b"]) if next_norm else None,
                     ln_out=h)
            return h
        y2 = self._buf(lv.n, c, torch.float32)
        ops.gemm(y, w[pre + "1.w"], y2, bias=w[pre + "1.b"])
        ops.layernorm(y2, w[pre + "2.g"], w[pre + "2.b"], st.x, res=st.x, colbias=tbias)
        if next_norm:
            ops.layernorm(st.x, w[next_norm + ".g"], w[next_norm + ".b"], h)
        return h

    def _mlp(self, st, pre_norm, pre_fc, h=None):
        """x += fc2(GELU(fc1(LN(x)))); h: the LayerNorm output if a previous epilogue already produced it."""
        w = self.w
        n, c = st.x.shape
        if h is None:
            h = self._buf(n, c, self.T)
            ops.layernorm(st.x, w[pre_norm + ".g"], w[pre_norm + ".b"], h)
        hid = w[pre_fc + "1.w"].shape[0]
        if ops.mlp_fused_ok(h, hid):  # big stages (C = 32 / 64): one kernel, the hidden activation stays in LDS
            st.xc = self._buf(n, c, self.T)
            ops.mlp_fused(h, w[pre_fc + "1.w"], w[pre_fc + "1.b"], w[pre_fc + "2.w"], w[pre_fc + "2.b"], st.x, st.xc)
            return
        u = self._buf(n, hid, self.T)
        ops.gemm(h, w[pre_fc + "1.w"], u, bias=w[pre_fc + "1.b"], act=ops.ACT_GELU)
        if self.T == torch.float32:
            ops.gemm(u, w[pre_fc + "2.w"], st.x, bias=w[pre_fc + "2.b"], res=st.x)
            st.xc = st.x
        else:
            st.xc = self._buf(n, c, self.T)
            ops.gemm(u, w[pre_fc + "2.w"], st.x, bias=w[pre_fc + "2.b"], res=st.x, out2=st.xc)

    def run_block(self, st, mod, pre, tbias=None):
        """ref: ptv3.py:399-428."""
        w, lv = self.w, st.level
        n, c = st.x.shape
        if self.native_blocks:
            att = mod.attn
            gidx, widx = lv.slots(st.curves[att.order_index], att.patch_size, att.enable_flash)
            _, _, _, _, patch_start, max_len, sum_l2 = lv.pad(att.patch_size, att.enable_flash)
            self.attn_work += 64.0 * att.num_heads * sum_l2
            self.attn_bytes += 4.0 * n * c * st.xc.element_size()
            self._count_conv(n, c, st.xc.element_size())
            desc = self.block_desc[pre]
            xc_out = st.x if self.T == torch.float32 else self._buf(n, c, self.T)
            sb = self._scratch_bytes.get((pre, n))
            if sb is None:
                if len(self._scratch_bytes) > 4096:
                    self._scratch_bytes.clear()
                sb = self._scratch_bytes[(pre, n)] = ops.block_scratch_bytes(desc, n)
            ops.block_forward(desc, n, st.x, st.xc, xc_out, tbias, lv.nbr(3, True), gidx, widx, patch_start,
                              patch_start.numel() - 1, max_len, self.scratch(sb), sat_counter=self._sat)
            if self._sat is not None:
                self.saturation_checked += n * c * (5 if xc_out is not st.x else 4)
            st.xc = xc_out
            return
        qkv = self._buf(n, 3 * c, self.T)
        aflags = ops.ATTN_Q_PRESCALED if self.q_prescaled else 0  # what the qkv producer has done for the attention kernel
        # deep stages (C = 128 / 256 / 512: csrc/deep.hip), as the native executor runs them (csrc/runtime.hip): the residual rows
        # go head -> tail through a second buffer (`xs`), which lets the few-row launches split a tile over several workgroups
        deep_rr = ((pre + ".head_img") in w and (pre + ".tail_img") in w and not ops.cpe_head_fused_ok(st.xc)
                   and hasattr(ops, "cpe_head_rr2"))
        xs = None
        if deep_rr:
            y = self._buf(n, c, self.T)
            self._conv3(st.xc, pre + ".cpe0", lv, y)
            xs = self._buf(n, c, torch.float32)
            ops.cpe_head_rr2(y, w[pre + ".head_img"], w[pre + ".cpe1.b"], (w[pre + ".cpe2.g"], w[pre + ".cpe2.b"]), st.x, xs,
                             tbias, (w[pre + ".norm1.g"], w[pre + ".norm1.b"]), w[pre + ".qkv.b"], qkv,
                             qkv_flags=ops.ATTN_V_BF16)
            aflags |= ops.ATTN_V_BF16
        elif ops.cpe_head_fused_ok(st.xc):  # big stages: cpe linear + LN + residual + LN1 + qkv in one launch
            y = self._buf(n, c, self.T)
            self._conv3(st.xc, pre + ".cpe0", lv, y)
            if (pre + ".head_img") in w:
                ops.cpe_head_rr(y, w[pre + ".head_img"], w[pre + ".cpe1.b"], (w[pre + ".cpe2.g"], w[pre + ".cpe2.b"]), st.x,
                                tbias, (w[pre + ".norm1.g"], w[pre + ".norm1.b"]), w[pre + ".qkv.b"], qkv)
            else:
                ops.cpe_head_fused(y, w[pre + ".cpe1.w"], w[pre + ".cpe1.b"], (w[pre + ".cpe2.g"], w[pre + ".cpe2.b"]), st.x,
                                   tbias, (w[pre + ".norm1.g"], w[pre + ".norm1.b"]), w[pre + ".qkv.w"], w[pre + ".qkv.b"], qkv,
                                   qkv_flags=ops.ATTN_V_BF16)
                aflags |= ops.ATTN_V_BF16
        else:
            h = self._cpe(st, pre + ".cpe", st.xc, tbias, next_norm=pre + ".norm1")
            ops.gemm(h, w[pre + ".qkv.w"], qkv, bias=w[pre + ".qkv.b"])
        att = mod.attn
        curve = st.curves[att.order_index]
        gidx, widx = lv.slots(curve, att.patch_size, att.enable_flash)
        _, _, _, _, patch_start, max_len, sum_l2 = lv.pad(att.patch_size, att.enable_flash)
        self._add_work(64.0 * att.num_heads * sum_l2, 4.0 * n * c * qkv.element_size())
        o = self._buf(n, c, self.T)
        if self.exact_attention_core and self.T != torch.float32:
            # error-budget tool only (tools/bf16_budget.py): the attention core in fp32 on the 16-bit q k v
            q32 = qkv.float()
            o32 = self._buf(n, c, torch.float32)
            if aflags & ops.ATTN_V_BF16 and self.T == torch.float16:
                q32[:, 2 * c:] = qkv[:, 2 * c:].view(torch.bfloat16).float()
            ops.attention(q32[:, :c], q32[:, c:2 * c], q32[:, 2 * c:], gidx, gidx, widx, patch_start, att.num_heads,
                          max_len, att.scale, o32, work=0.0, flags=aflags & ops.ATTN_Q_PRESCALED)
            o.copy_(o32)
        else:
            ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], gidx, gidx, widx, patch_start, att.num_heads,
                          max_len, att.scale, o, work=64.0 * att.num_heads * sum_l2, flags=aflags)
        hid = w[pre + ".fc1.w"].shape[0]
        if xs is not None:  # deep stages: proj + LN2 + MLP, one launch (+ the reduce launch of a split one)
            st.xc = self._buf(n, c, self.T)
            ops.attn_tail_rr2(o, w[pre + ".tail_img"], w[pre + ".proj.b"], w[pre + ".norm2.g"], w[pre + ".norm2.b"],
                              w[pre + ".fc1.b"], w[pre + ".fc2.b"], xs, st.x, st.xc, ws=self.scratch(4 * n * c * 4 + 256))
            return
        if ops.attn_tail_fused_ok(o, hid):  # big stages: proj + LN2 + MLP in one launch
            st.xc = self._buf(n, c, self.T)
            if (pre + ".tail_img") in w:
                ops.attn_tail_rr(o, w[pre + ".tail_img"], w[pre + ".proj.b"], w[pre + ".norm2.g"], w[pre + ".norm2.b"],
                                 w[pre + ".fc1.b"], w[pre + ".fc2.b"], st.x, st.xc)
            else:
                ops.attn_tail_fused(o, w[pre + ".proj.w"], w[pre + ".proj.b"], w[pre + ".norm2.g"], w[pre + ".norm2.b"],
                                    w[pre + ".fc1.w"], w[pre + ".fc1.b"], w[pre + ".fc2.w"], w[pre + ".fc2.b"], st.x, st.xc)
            return
        if c <= self.FUSE_LN_MAX_C:
            h2 = self._buf(n, c, self.T)
            ops.gemm(o, w[pre + ".proj.w"], st.x, bias=w[pre + ".proj.b"], res=st.x,
                     ln_post=(w[pre + ".norm2.g"], w[pre + ".norm2.b"]), ln_out=h2)
        else:
            h2 = None
            ops.gemm(o, w[pre + ".proj.w"], st.x, bias=w[pre + ".proj.b"], res=st.x)
        self._mlp(st, pre + ".norm2", pre + ".fc", h2)

    def run_embedding(self, plan, feat, perm, pre, curves):
        """ref: ptv3.py:633-663.  feat (N, cin) fp32 in the caller's order (perm None: already physical)."""
        w, lv = self.w, plan.levels[0]
        cout = w[pre + ".w"].shape[0]
        a = ops.gather_pad_cast(feat, perm, w[pre + ".cpad"], self.T)
        x = self._buf(lv.n, cout, torch.float32)
        xc = x if self.T == torch.float32 else self._buf(lv.n, cout, self.T)
        if (pre + ".wimg") in w and lv.parent is not None and lv.n < (1 << 24):
            # bf16 stems: neighbours enumerated through the parent level, no 125-offset kernel map (csrc/stem.hip)
            par, (cluster, seg) = lv.parent
            ops.stem5(a, w[pre + ".wimg"], w[pre + ".bn.scale"], w[pre + ".bn.shift"], lv.grid, cluster, par.nbr(3, True),
                      lv.child_info(), lv.depth, x, None if xc is x else xc)
        else:
            ops.gemm(a, w[pre + ".w"], x, scale=w[pre + ".bn.scale"], shift=w[pre + ".bn.shift"], act=ops.ACT_GELU,
                     nbr=lv.nbr(5, True), nbr_kmajor=True, kvol=125, out2=None if xc is x else xc)
        self._sat_note(xc)
        return State(lv, x, xc, curves)

    def run_pooling(self, plan, st, pre, cum_to, perm):
        """ref: ptv3.py:464-555: Linear -> segment max -> BN -> GELU; orders re-shuffled."""
        w = self.w
        fine, coarse = st.level, plan.levels[cum_to]
        _, seg = plan.link(fine.cum, cum_to)
        cout = w[pre + ".proj.w"].shape[0]
        x = self._buf(coarse.n, cout, torch.float32)
        xc = x if self.T == torch.float32 else self._buf(coarse.n, cout, self.T)
        if (pre + ".proj.wimg") in w and st.xc.stride(0) == st.xc.shape[1] and fine.n <= 5 * coarse.n:
            # wide 16-bit stages: projection + maximum + BatchNorm + GELU in one launch, the projected rows stay on the CU
            # (one octree step, ~2.2 children per pooled row: 118 -> 67 us on the first pooling of a collated forward; the
            # c-branch's two-step pooling, ~8.4 children, folds too many rows per lane and stays on the two launches)
            ops.pool_fused(st.xc, w[pre + ".proj.wimg"], w[pre + ".proj.b"], seg, coarse.n, w[pre + ".bn.scale"],
                           w[pre + ".bn.shift"], ops.ACT_GELU, x, xc)
        else:
            y = self._buf(fine.n, cout, self.T)
            ops.gemm(st.xc, w[pre + ".proj.w"], y, bias=w[pre + ".proj.b"])
            ops.segment_max(y, seg, coarse.n, w[pre + ".bn.scale"], w[pre + ".bn.shift"], ops.ACT_GELU, x,
                            None if xc is x else xc)
        self._sat_note(xc)
        curves = st.curves if perm is None else [st.curves[int(j)] for j in perm]
        out = State(coarse, x, xc, curves)
        out.parent = st
        return out

    def run_unpooling(self, plan, st, pre, mod):
        """ref: ptv3.py:597-630.  xc keeps the PRE-merge, unscaled skip feature: the reference leaves
        sparse_conv_feat stale there and the next CPE conv reads it.
        'add' (n-branch): x = f * skip + child[cluster];  'cat' (c-branch): x = proj_cat([f * skip, child[cluster]])
        with f = 2^-0.5 (skip_connection_scale) * 0.8^(i-1) (skip_connection_scale_i, False -> 1.25)."""
        w = self.w
        parent = st.parent
        fine, coarse = parent.level, st.level
        cluster, _ = plan.link(fine.cum, coarse.cum)
        cout = w[pre + ".proj.w"].shape[0]
        f = (2 ** -0.5 if mod.skip_connection_scale else 1.0)
        if mod.skip_connection_scale_i is not None:
            f *= 0.8 ** (int(mod.skip_connection_scale_i) - 1)
        x = self._buf(fine.n, cout, torch.float32)
        xc = self._buf(fine.n, cout, self.T)
        if mod.skip_connection_mode == "add":
            assert f == 1.0, "rejected by SerializedUnpooling.__init__"
            child = self._buf(coarse.n, cout, torch.float32)
            ops.gemm(st.xc, w[pre + ".proj.w"], child, bias=w[pre + ".proj.b"], scale=w[pre + ".proj_bn.scale"],
                     shift=w[pre + ".proj_bn.shift"], act=ops.ACT_GELU)
            ops.gemm(parent.xc, w[pre + ".skip.w"], x, bias=w[pre + ".skip.b"], scale=w[pre + ".skip_bn.scale"],
                     shift=w[pre + ".skip_bn.shift"], act=ops.ACT_GELU, add_src=child, add_idx=cluster, out2=xc,
                     out2_pre_add=True)
        else:
            # proj_cat([f*par, child[inv]]) = par @ (f*Wa)^T + (child @ Wb^T)[inv] + b
            child = self._buf(coarse.n, cout, self.T)
            ops.gemm(st.xc, w[pre + ".proj.w"], child, bias=w[pre + ".proj.b"], scale=w[pre + ".proj_bn.scale"],
                     shift=w[pre + ".proj_bn.shift"], act=ops.ACT_GELU)
            z = self._buf(coarse.n, cout, torch.float32)
            ops.gemm(child, w[pre + ".cat_b.w"], z)
            ops.gemm(parent.xc, w[pre + ".skip.w"], xc, bias=w[pre + ".skip.b"], scale=w[pre + ".skip_bn.scale"],
                     shift=w[pre + ".skip_bn.shift"], act=ops.ACT_GELU)
            ops.gemm(xc, w[pre + ".cat_a.w"], x, bias=w[pre + ".cat.b"], add_src=z, add_idx=cluster)
        self._sat_note(xc)
        out = State(fine, x, xc, parent.curves)
        out.parent = parent.parent
        return out

    def run_cross_block(self, nst, cst):
        """ref: ptv3.py:1179-1223 + :988-1055 (cross_block2: n <- c, the NN->CN feature injection)."""
        w = self.w
        cb = self.model.backbone._tm_dec0.cross_block2
        att = cb.attn
        lv, clv = nst.level, cst.level
        # the reference pads the kv sequence with the q point's padding plan (ptv3.py:1009): the two bottlenecks must
        # hold the same number of points per batch element.  They are the same voxel set in every shipped config; a
        # scene whose grid is shallower than the pooling depths can end on different levels with equal counts
        if clv is not lv and list(clv.offs_host) != list(lv.offs_host):
            raise CdsegError("cross attention needs the same number of c- and n-branch bottleneck points per batch "
                             "element (ref: ptv3.py:1009 reuses the q padding for kv)")
        n, cq = nst.x.shape
        self._cpe(nst, "x.q_cpe", nst.xc)
        self._cpe(cst, "x.kv_cpe", cst.xc)
        hq = self._buf(n, cq, self.T)
        ops.layernorm(nst.x, w["x.q_norm1.g"], w["x.q_norm1.b"], hq)
        hkv = self._buf(n, cst.x.shape[1], self.T)
        ops.layernorm(cst.x, w["x.kv_norm1.g"], w["x.kv_norm1.b"], hkv)
        # the kv point LEAVES the block holding kv_norm1(c + kv_cpe(c)) in feat and sparse_conv_feat (PointSequential
        # assigns the LayerNorm output in place, ptv3.py:1190-1196, modules.py:68-73): that is what the c-decoder of the
        # multi-step path reads next (oracle/model.py: cross_block, `kvp.feat = hkv`)
        cst.xc = hkv
        if self.T == torch.float32:
            cst.x = hkv
        q = self._buf(n, cq, self.T)
        ops.gemm(hq, w["x.q.w"], q, bias=w["x.q.b"])
        kv = self._buf(n, 2 * cq, self.T)
        ops.gemm(hkv, w["x.kv.w"], kv, bias=w["x.kv.b"])
        K = att.q_patch_size
        q_gidx, widx = lv.slots(nst.curves[att.order_index], K, att.enable_flash)
        kv_gidx, _ = clv.slots(cst.curves[att.order_index], K, att.enable_flash)
        _, _, _, _, patch_start, max_len, sum_l2 = lv.pad(K, att.enable_flash)
        self._add_work(64.0 * att.num_heads * sum_l2, 4.0 * n * cq * q.element_size())
        o = self._buf(n, cq, self.T)
        ops.attention(q, kv[:, :cq], kv[:, cq:], q_gidx, kv_gidx, widx, patch_start, att.num_heads, max_len, att.scale, o,
                      work=64.0 * att.num_heads * sum_l2, flags=ops.ATTN_Q_PRESCALED if self.q_prescaled else 0)
        if "x.tail_img" in w and n >= ops.DEEP512_MIN_ROWS:
            nst.xc = self._buf(n, cq, self.T)
            ops.attn_tail_rr(o, w["x.tail_img"], w["x.proj.b"], w["x.q_norm2.g"], w["x.q_norm2.b"], w["x.fc1.b"], w["x.fc2.b"],
                             nst.x, nst.xc)
            return
        if "x.tail_img" in w and hasattr(ops, "attn_tail_rr2"):
            # few rows (a single scene's bottleneck): the tile's weight stream cut over four workgroups (csrc/deep.hip) - the
            # residual is read from the old rows and the result lands in fresh ones, so no workgroup reads what another writes
            x_new = self._buf(n, cq, torch.float32)
            nst.xc = self._buf(n, cq, self.T)
            ops.attn_tail_rr2(o, w["x.tail_img"], w["x.proj.b"], w["x.q_norm2.g"], w["x.q_norm2.b"], w["x.fc1.b"], w["x.fc2.b"],
                              nst.x, x_new, nst.xc, ws=self.scratch(4 * n * cq * 4 + 256))
            nst.x = x_new
            return
        if cb.tm_feat == 1.0:
            ops.gemm(o, w["x.proj.w"], nst.x, bias=w["x.proj.b"], res=nst.x)
        else:  # q_shortcut + feat_scale * attn
            ops.gemm(o, w["x.proj.w"], nst.x, bias=w["x.proj.b"], scale=w["x.feat_scale"], shift=w["x.feat_zero"],
                     res=nst.x)
        self._mlp(nst, "x.q_norm2", "x.fc")

    # ------------------------------------------------------------------ accounting
    def forward_work(self, plan):
        """Algorithmic FLOPs (2 x multiply-add) of ONE single-step forward on `plan`, by kernel class - SURVEY.md 8(d)'s
        formulas on the plan's real sizes: sparse convs count the OCCUPIED neighbours only, attention 4 * L^2 * 16 per
        patch-head, the dead c-decoder is not counted.  Synchronises (reads the kernel maps' occupancy): for reports."""
        bb, w = self.model.backbone, self.w
        cache = {}

        def occ(lv, k):
            if (lv.cum, k) not in cache:
                cache[(lv.cum, k)] = float((lv.nbr(k, True) >= 0).sum().item())
            return cache[(lv.cum, k)]

        out = dict(conv=0.0, linear=0.0, attention=0.0, stem=0.0, pool_unpool=0.0, head=0.0)
        deep = [0.0]  # the part of "conv" that runs on the gathered GEMM (C >= 128, or any width in the fp32 mode)

        def block(mod, pre, lv):
            c, hid = mod.channels, w[pre + ".fc1.w"].shape[0]
            out["conv"] += 2.0 * occ(lv, 3) * c * c
            if c > 64 or self.T == torch.float32:
                deep[0] += 2.0 * occ(lv, 3) * c * c
            out["linear"] += 2.0 * lv.n * (5.0 * c * c + 2.0 * c * hid)
            out["attention"] += 64.0 * mod.attn.num_heads * lv.pad(mod.attn.patch_size, mod.attn.enable_flash)[6]

        def stages(branch, cum, n_stages):
            for s in range(n_stages):
                lv = plan.levels[cum[s]]
                enc = getattr(getattr(bb, f"_{branch}_enc"), f"enc{s}")
                if s > 0:
                    pw = w[f"{branch}_enc{s}.down.proj.w"]
                    out["pool_unpool"] += 2.0 * plan.levels[cum[s - 1]].n * pw.shape[0] * pw.shape[1]
                for name, mod in enc._modules.items():
                    if name.startswith("block"):
                        block(mod, f"{branch}_enc{s}.{name}", lv)

        lv0 = plan.levels[0]
        stem_in = w["n_emb.w"].shape[1] // 125
        out["stem"] += 2.0 * occ(lv0, 5) * min(stem_in, bb._n_embedding.stem.conv.in_channels) * w["n_emb.w"].shape[0]
        stages("n", plan.n_cum, bb.n_num_stages)
        if bb.condition:
            out["stem"] += 2.0 * occ(lv0, 5) * bb._c_embedding.stem.conv.in_channels * w["c_emb.w"].shape[0]
            stages("c", plan.c_cum, bb.c_num_stages)
            cb = bb._tm_dec0.cross_block2
            lq, lc = plan.levels[plan.n_cum[-1]], plan.levels[plan.c_cum[-1]]
            cq, ck = cb.q_channels, cb.kv_channels
            out["conv"] += 2.0 * occ(lq, 3) * cq * cq + 2.0 * occ(lc, 3) * ck * ck
            deep[0] += 2.0 * occ(lq, 3) * cq * cq + 2.0 * occ(lc, 3) * ck * ck
            out["linear"] += 2.0 * lq.n * (cq * cq * 3 + 2.0 * cq * w["x.fc1.w"].shape[0]) + 2.0 * lc.n * (ck * ck + ck * 2 * cq)
            out["attention"] += 64.0 * cb.attn.num_heads * lq.pad(cb.attn.q_patch_size, cb.attn.enable_flash)[6]
        for s in reversed(range(bb.n_num_stages - 1)):
            lf, lc = plan.levels[plan.n_cum[s]], plan.levels[plan.n_cum[s + 1]]
            pre = f"n_dec{s}.up"
            out["pool_unpool"] += 2.0 * lc.n * w[pre + ".proj.w"].numel() + 2.0 * lf.n * w[pre + ".skip.w"].numel()
            if (pre + ".cat_a.w") in w:
                out["pool_unpool"] += 2.0 * lc.n * w[pre + ".cat_b.w"].numel() + 2.0 * lf.n * w[pre + ".cat_a.w"].numel()
            dec = getattr(bb._n_dec, f"dec{s}")
            for name, mod in dec._modules.items():
                if name.startswith("block"):
                    block(mod, f"n_dec{s}.{name}", lf)
        if "n_head.w" in w:
            out["head"] += 2.0 * lv0.n * w["n_head.w"].numel()
        out["total"] = sum(out.values())
        out["conv_deep"] = deep[0]  # (a part of "conv": not in the total a second time)
        return out

    # ------------------------------------------------------------------ randomness
    def draw(self, n, feat_shape, c_ch, noise_level, n_perms, always_noise=False):
        """The reference's CPU-generator consumption order (SURVEY.md finding 3)."""
        m = self.model
        d = {}
        if noise_level is not None:
            # the reference's add_gaussian_noise (default.py:228-236) calls randn_like on the input tensor: torch's CPU
            # generator when the reference runs on CPU (the golden vectors), the CUDA generator - which does NOT advance
            # the CPU one - when it runs on a GPU (test.py moves the inputs first).  feat_noise_source="device" takes the
            # jitter from the device Philox stream instead, so the CPU-generator order of the later draws (c-noise,
            # randperms) is the one a GPU run of the reference has for the same torch.manual_seed.
            on_cpu_gen = m.noise_source == "torch_cpu" and getattr(m, "feat_noise_source", "torch_cpu") == "torch_cpu"
            d["feat_noise"] = torch.randn(feat_shape) if on_cpu_gen else None
        if m.condition and (always_noise or (m.dm and m.dm_input == "xt")):
            d["noise"] = (torch.normal(0, 1, size=(n, c_ch), dtype=torch.float32)
                          if m.noise_source == "torch_cpu" else None)
        if m.backbone.shuffle_orders:
            no = len(m.backbone.order)
            d["perms"] = [torch.randperm(no).tolist() for _ in range(n_perms)]
        return d

    def predraw(self, input_dict, noise_level=None):
        """The random draws of one single-step inference, taken NOW from torch's CPU generator (and the device-RNG
        stream ids reserved now): lets inference_many issue scenes from several threads while the draws stay in
        scene order."""
        m, bb = self.model, self.model.backbone
        feat = input_dict["feat"]
        per_call = (2 + len(bb.c_stride) + len(bb.n_stride)) if bb.condition else (1 + len(bb.n_stride))
        d = self.draw(feat.shape[0], tuple(feat.shape), m.c_in_channels, noise_level, per_call)
        d["rng_base"] = self.reserve_rng()
        return d

    def _add_work(self, flops, nbytes=0.0):
        with self._work_lock:
            self.attn_work += flops
            self.attn_bytes += nbytes

    RNG_RESERVE = 8  # device-RNG streams one single-step inference may consume

    def reserve_rng(self, k=None):
        """Reserve a block of device-RNG stream ids (call in scene order from ONE thread: keeps the device noise
        of scene i independent of which lane / thread runs it)."""
        if self.model.noise_source == "device":
            # device-noise mode: the stream ids come from torch's CPU generator itself (one draw per call, after the
            # order shuffles), so the logits are a function of the generator state alone: torch.manual_seed(s) followed
            # by the same calls reproduces them, whatever ran before in the process.  62-bit base: the windows of
            # RNG_RESERVE consecutive ids of two scenes practically never overlap (a 31-bit base collided by the
            # birthday bound within ~10^4 scenes).  One draw per inference from the CPU generator; call from ONE host
            # thread (predraw / inference_many do)
            return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        seed = torch.initial_seed()
        if seed != self._rng_seed:  # a new torch.manual_seed restarts the counter (used by feat_noise_source="device")
            self._rng_seed, self.rng_offset = seed, 0
        base = self.rng_offset
        self.rng_offset += self.RNG_RESERVE if k is None else int(k)
        return base

    def _device_randn(self, shape):
        seed = torch.initial_seed()
        out = ops.randn(shape, seed, self._tls.rng, self.device)
        self._tls.rng += 1
        return out

    # ------------------------------------------------------------------ forward
    def inference(self, input_dict, noise_level=None, draws=None):
        """Single-step inference (ref: default.py:371-422)."""
        prev = ops.set_f32x3(self.x3) if hasattr(ops, "set_f32x3") else None
        try:
            with _lib.use(self.variant):
                return self._inference(input_dict, noise_level, draws)
        finally:
            if prev is not None:
                ops.set_f32x3(prev)
            if hasattr(ops, "unbind_stream"):
                ops.unbind_stream()

    def inference_ddim(self, input_dict, step=1, mode="avg", noise_level=None, draws=None):
        """Multi-step inference MSAI (mode="avg") / MSFI ("final") (ref: default.py:278-369)."""
        prev = ops.set_f32x3(self.x3) if hasattr(ops, "set_f32x3") else None
        try:
            with _lib.use(self.variant):
                return self._inference_ddim(input_dict, step, mode, noise_level, draws)
        finally:
            if prev is not None:
                ops.set_f32x3(prev)
            if hasattr(ops, "unbind_stream"):
                ops.unbind_stream()

    # -- shared set-up -------------------------------------------------------------------------------
    def _setup(self, input_dict, noise_level, draws, n_backbone_calls, always_noise=False):
        m, bb = self.model, self.model.backbone
        feat = input_dict["feat"]
        dev = feat.device  # CPU tensors are rejected by every op (cdsegnet_amd.ops): no CPU fallback
        self.prepare(dev)
        if dev.type == "cuda" and hasattr(ops, "bind_stream"):
            ops.bind_stream()
        grid = input_dict["grid_coord"]
        offset = input_dict["offset"]
        n = feat.shape[0]
        # (no hint - the reference's dict -: the native plan builder brings the offsets to the host with the pooled sizes, the
        # per-op path reads them itself; either way no read of its own up here)
        offset_host = [int(v) for v in input_dict["offset_host"]] if "offset_host" in input_dict else None
        cond = bb.condition
        c_ch = m.c_in_channels
        per_call = (2 + len(bb.c_stride) + len(bb.n_stride)) if cond else (1 + len(bb.n_stride))
        box = {"draws": draws}

        def host_draws():
            # the random draws are host-only work (torch's CPU generator, in the reference's consumption order): taken while
            # the device runs the first plan kernels (build_plan calls this once, before it waits for the pooled sizes)
            if box["draws"] is None:
                box["draws"] = self.draw(n, tuple(feat.shape), c_ch, noise_level, per_call * n_backbone_calls, always_noise)
            d = box["draws"]
            self._tls.rng = (d["rng_base"] if "rng_base" in d
                             else self.reserve_rng(max(self.RNG_RESERVE, 2 + n_backbone_calls)))

        plan = self.build_plan(grid, offset.to(torch.int64), offset_host, n, while_device_works=host_draws, defer_pads=True)
        draws = box["draws"]
        feat = feat.float().contiguous()
        if noise_level is not None:  # ref: default.py:373-374 (perturbs feat and rebinds it in input_dict)
            fn = draws.get("feat_noise")
            fn = self._device_randn(tuple(feat.shape)) if fn is None else fn.to(dev, torch.float32)
            feat = ops.axpy(feat, fn.contiguous(), noise_level)
            input_dict["feat"] = feat
        perms = list(draws["perms"]) if bb.shuffle_orders else [None] * (per_call * n_backbone_calls)
        self.last_plan = plan
        return feat, draws, perms, plan, per_call

    def _t_bias(self, t):
        """Per-block timestep bias from the (uniform) embedding row (ref: ptv3.py:1772-1778, :406-411)."""
        w = self.w
        if self.model.backbone.T_dim == -1 or "t.table" not in w:
            return {}
        # a function of the weights and t alone (every point of a forward has the same t; single-step inference always
        # t = T - 1): computed once per (engine, t) - the engine is rebuilt when the weights change - and kept.  The first
        # computation is waited for, so later forwards on other streams read finished values
        ent = self._tb_cache.get(t)
        if ent is None:
            v = ops.gemv(w["t.fc1.w"], w["t.fc1.b"], w["t.table"][t + 1], ops.ACT_SWISH)  # table row 0 is t = -1
            v = ops.gemv(w["t.fc2.w"], w["t.fc2.b"], v, ops.ACT_SWISH)
            tall = ops.gemv(w["t.mlp.w"], w["t.mlp.b"], v, ops.ACT_NONE)
            if tall.is_cuda:
                torch.cuda.current_stream(tall.device).synchronize()
            ent = self._tb_cache[t] = {k: tall[a:b] for k, (a, b) in self.t_slices.items()}
        return ent

    def _sat_begin(self, dev):
        """Start of a forward: arm the saturation diagnostic (IEEE-half trunks only)."""
        self._sat = None
        self.saturation_count, self.saturation_checked = None, 0
        if self.count_saturation and self.T == torch.float16:
            self._sat = torch.zeros(1, dtype=torch.int64, device=dev)

    def _sat_note(self, t):
        """A 16-bit activation outside the Blocks (stem / pooling / un-pooling outputs)."""
        if self._sat is not None and t is not None and t.dtype == torch.float16:
            ops.count_saturated(t, self._sat)
            self.saturation_checked += t.numel()

    def _sat_end(self):
        if self._sat is not None:
            self.saturation_count = int(self._sat.item())  # (diagnostic mode: one host read)
            self._sat = None

    def _inference(self, input_dict, noise_level=None, draws=None):
        m, bb = self.model, self.model.backbone
        feat, draws, perms, plan, _ = self._setup(input_dict, noise_level, draws, 1)
        n, dev = feat.shape[0], feat.device
        c_ch = m.c_in_channels
        c_feat = c_perm = None
        t = 0
        if bb.condition:
            if m.dm and m.dm_input == "xt":  # ref: default.py:392-394
                nz = draws.get("noise")
                if nz is None:
                    c_feat, c_perm = self._device_randn((n, c_ch)), None  # any order is equally random
                else:
                    c_feat, c_perm = nz.to(dev, torch.float32).contiguous(), plan.perm0
                t = m.T - 1
            else:
                c_feat = feat if c_ch == feat.shape[1] else input_dict["coord"].float().contiguous()
                c_perm = plan.perm0
        self._sat_begin(dev)
        logits, _ = self.backbone(plan, feat, c_feat, c_perm, t, perms, want_c=False)
        self._sat_end()
        return logits

    def _inference_ddim(self, input_dict, step, mode, noise_level, draws):
        m, bb = self.model, self.model.backbone
        if not bb.condition:
            return self._inference(input_dict, noise_level, draws)
        if m.dm_target != "noise":
            raise NotImplementedError("dm_target other than 'noise'")
        schedule = np.linspace(-1, m.T - 1, num=step + 1, dtype=int)[::-1]  # ref: default.py:224-226
        feat, draws, perms, plan, per_call = self._setup(input_dict, noise_level, draws, len(schedule), always_noise=True)
        n, dev = feat.shape[0], feat.device
        nz = draws.get("noise")
        # the plan (codes, orders, kernel maps, slot plans) is step-invariant: built once, unlike the reference
        c_xt = self._device_randn((n, m.c_in_channels)) if nz is None else \
            ops.gather_rows(nz.to(dev, torch.float32).contiguous(), plan.perm0)
        ab = m.Alpha_bar
        n_pred = None
        for k, t in enumerate(schedule):
            t = int(t)
            logits, eps = self.backbone(plan, feat, c_xt, None, t, perms[per_call * k:per_call * (k + 1)], want_c=True)
            # DDIM update on the uniform timestep (ref: default.py:192-214); scalars in fp32 like the reference
            s_ab, s_1ab = float(torch.sqrt(ab[t])), float(torch.sqrt(1 - ab[t]))
            if t == 0:
                c_xt = ops.ddim_update(c_xt, eps, 0.0, s_1ab, s_ab, 0.0, final=True)
            else:
                c_xt = ops.ddim_update(c_xt, eps, float(torch.sqrt(ab[t - 1])), s_1ab, s_ab,
                                       float(torch.sqrt(1 - ab[t - 1])), final=False)
            if mode == "avg":
                n_pred = logits if n_pred is None else ops.axpy(n_pred, logits, 1.0)
            else:
                n_pred = logits
            if t <= 0:
                break
        if mode == "avg":
            n_pred = ops.axpy(torch.zeros_like(n_pred), n_pred, 1.0 / len(schedule))
        return n_pred

    # -- backbone ------------------------------------------------------------------------------------
    def backbone(self, plan, feat, c_feat, c_perm, t, perms, want_c):
        """PT-v3m1 forward (ref: ptv3.py:1757-1846).  feat (N,C) in the caller's order; c_feat with c_perm
        (None: already in physical order).  Returns logits in the caller's order and, if want_c, the
        c-head output (noise estimate) in physical order."""
        bb = self.model.backbone
        w = self.w
        cond = bb.condition
        n = feat.shape[0]
        dev = feat.device
        pi = iter(perms)
        base_curves = [CURVES.index(o) for o in bb.order]

        def shuffled(perm):
            return list(base_curves) if perm is None else [base_curves[int(j)] for j in perm]

        n_cum, c_cum = plan.n_cum, plan.c_cum
        if cond:
            c_curves = shuffled(next(pi))
        n_curves = shuffled(next(pi))
        tb = self._t_bias(t) if cond else {}

        def enc_stage(st, branch, s, cum, perm):
            enc = getattr(getattr(bb, f"_{branch}_enc"), f"enc{s}")
            e = self._eng(f"{branch}_enc{s}")
            e._fit(st)
            if s > 0:
                st = e.run_pooling(plan, st, f"{branch}_enc{s}.down", cum[s], perm)
            for name, mod in enc._modules.items():
                if name.startswith("block"):
                    key = f"{branch}_enc{s}.{name}"
                    e.run_block(st, mod, key, tb.get(key))
            return st

        def dec_stage(st, branch, s):
            dec = getattr(getattr(bb, f"_{branch}_dec"), f"dec{s}")
            e = self._eng(f"{branch}_dec{s}")
            e._fit(st)
            e._fit(st.parent)  # the skip feature saved by the encoder
            st = e.run_unpooling(plan, st, f"{branch}_dec{s}.up", dec.up)
            for name, mod in dec._modules.items():
                if name.startswith("block"):
                    key = f"{branch}_dec{s}.{name}"
                    e.run_block(st, mod, key, tb.get(key))
            return st

        # ref: ptv3.py:1781-1794.  The reference interleaves the two encoders (c0 n0 c1 n1 n2 c2 n3 n4); that order
        # only fixes which randperm draw each stage consumes.  The encoders are independent until the cross block, so
        # the noise-branch encoder is issued on a SIDE STREAM, forked when the dominant branch reaches stage
        # `fork_stage`: its throughput-bound 120k-point blocks then fill the CUs that the dominant branch's deep,
        # latency-bound stages (a few hundred to a few thousand points) leave idle.
        if cond:
            # (3 c-branch / 5 n-branch stages: checked in Engine.__init__)
            p_c1, p_n1, p_n2, p_c2, p_n3, p_n4 = (next(pi) for _ in range(6))
            n_perms = [None, p_n1, p_n2, p_n3, p_n4]

            def c_branch():
                st = self._eng("c_emb").run_embedding(plan, c_feat, c_perm, "c_emb", c_curves)
                st = enc_stage(st, "c", 0, c_cum, None)
                st = enc_stage(st, "c", 1, c_cum, p_c1)
                return enc_stage(st, "c", 2, c_cum, p_c2)

            fork = self.fork_stage if (dev.type == "cuda" and self.fork_stage is not None) else None
            SHARE_EVENTS.on = fork is not None
            if fork is None:
                plan.finish_pads()
            cst = c_branch() if fork is None else None
            nst = self._eng("n_emb").run_embedding(plan, feat, plan.perm0, "n_emb", n_curves)
            plan.finish_pads()  # (deferred by _setup: the stem is device work for the plan's last stretch of host time)
            join = None
            for s in range(bb.n_num_stages):
                if fork is not None and s == fork:
                    cst, join = self._on_side_stream(c_branch)
                nst = enc_stage(nst, "n", s, n_cum, n_perms[s])
            if join is not None:
                ops.wait_event(join)
            e = self._eng("x")
            e._fit(nst)
            e._fit(cst)
            e.run_cross_block(nst, cst)
        else:
            nst = self._eng("n_emb").run_embedding(plan, feat, plan.perm0, "n_emb", n_curves)
            plan.finish_pads()
            for s in range(bb.n_num_stages):
                nst = enc_stage(nst, "n", s, n_cum, next(pi) if s > 0 else None)
        self.trace = {"n_bottleneck": nst.x}
        # decoders.  In single-step inference the c-decoder / c-head never reach seg_logits (dead code).
        c_out = None
        if cond and want_c:
            for s in reversed(range(bb.c_num_stages - 1)):
                cst = dec_stage(cst, "c", s)
            e = self._eng("c_head")
            e._fit(cst, exact=True)
            c_out = torch.empty((n, w["c_head.w"].shape[0]), dtype=torch.float32, device=dev)
            ops.gemm(cst.xc, e.w["c_head.w"], c_out, bias=e.w["c_head.b"])
        for s in reversed(range(bb.n_num_stages - 1)):
            nst = dec_stage(nst, "n", s)
        # head, scattered back to the caller's point order (ref: ptv3.py:1813)
        if "n_head.w" in w:
            e = self._eng("n_head")
            e._fit(nst, exact=True)
            logits = torch.empty((n, w["n_head.w"].shape[0]), dtype=torch.float32, device=dev)
            ops.gemm(nst.xc, e.w["n_head.w"], logits, bias=e.w["n_head.b"], out_idx=plan.perm0)
        else:
            logits = torch.empty((n, nst.x.shape[1]), dtype=torch.float32, device=dev)
            ops.scatter_rows(nst.x, plan.perm0, logits)
        return logits, c_out
