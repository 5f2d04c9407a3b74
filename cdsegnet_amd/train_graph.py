"""Training forward of the model under torch autograd, on the HIP kernels (training row of SURVEY 8(f4)).

ref: pointcept/models/default.py:424-493 (DefaultSegmentorV2.forward: per-scene timestep, q_sample of the conditioning
target, both branches and BOTH decoders, the criteria) and pointcept/engines/train.py:216-271 (run_step:
`loss = model(input_dict)["loss"]; loss.backward(); optimizer.step()`).  The reference's training step is torch autograd
over torch ops plus three extension ops with hand-written backward passes (spconv.SubMConv3d, flash_attn,
torch_scatter.segment_csr).  The same boundary here: `DefaultSegmentorV2.forward` returns `dict(loss=...)` whose
`.backward()` fills the `.grad` of the model's own nn.Parameters, so `torch.optim.AdamW` and the reference trainer work
unchanged - and underneath, every product that carries the FLOPs runs on this library's kernels behind
`torch.autograd.Function`s:

    sparse conv (k = 3 CPE convs, k = 5 stems)   forward  cdseg_gemm (gathered form)
                                                 backward cdseg_conv_wgrad (all offsets, one launch) + the data gradient as
                                                          the SAME gathered GEMM on the mirrored, transposed kernel
    Linear (qkv, proj, fc1, fc2, CPE, pool / unpool projections)      cdseg_gemm / cdseg_gemm on W^T / cdseg_linear_wgrad
    LayerNorm                                    cdseg_layernorm / cdseg_layernorm_bwd
    serialized (cross) attention                 cdseg_attention / cdseg_attention_bwd (recompute-P, gathered rows)
    segment max of the pooling                   cdseg_segment_max forward; the arg-max mask is recomputed in the backward

in exact fp32 (the reference trains its trunk in fp32 with an fp16 attention core under AMP; the 16-bit backward is the
next step).  What stays plain torch device ops: train-mode BatchNorm1d (3 per pooling stage, batch
statistics), GELU between them, the swish timestep MLP on B rows, row masks of stochastic depth, q_sample, the two
(C -> classes) heads, and the criteria (cdsegnet_amd.losses).  All integer work - serialization, pooling structure,
kernel maps, padded patch plans - is the inference engine's plan (Engine.build_plan), shared with the inference path.
"""
import warnings

import torch
import torch.nn.functional as F

from . import engine as _engine
from ._lib import DuplicateVoxelsError
from . import ops
from .losses import build_criteria


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _f32(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


# ------------------------------------------------------------------------------------------ autograd functions
class _SubMConv(torch.autograd.Function):
    """y = bias + sum_o W_o x[nbr[o]] (spconv.SubMConv3d; ref call sites ptv3.py:356, 647, 1106, 1118)."""

    @staticmethod
    def forward(ctx, x, w5, b, nbr):
        cout, cin = w5.shape[0], w5.shape[-1]
        kvol = nbr.shape[0]
        cp = (cin + 15) // 16 * 16  # the weight-gradient kernel works on 16-channel groups (stems: 6 -> 16)
        w3 = w5.reshape(cout, kvol, cin)
        if cp != cin:
            x = F.pad(x, (0, cp - cin))
            w3 = F.pad(w3, (0, cp - cin))
        x = _c(x)
        w = _c(w3).reshape(cout, kvol * cp)
        y = _f32((x.shape[0], cout), x)
        # (a padded stem weight is a fresh tensor every step: no cache entry may pin it - ADVICE r4)
        ops.gemm(x, w, y, bias=b, nbr=nbr, kvol=kvol, nbr_kmajor=True, cache=cp == cin)
        ctx.save_for_backward(x, w, nbr)
        ctx.meta = (cin, cp, kvol, b is not None, tuple(w5.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, nbr = ctx.saved_tensors
        cin, cp, kvol, has_b, wshape = ctx.meta
        cout = w.shape[0]
        dy = _c(dy)
        dw3 = torch.zeros((cout, kvol, cp), dtype=torch.float32, device=dy.device)
        db = torch.zeros(cout, dtype=torch.float32, device=dy.device) if has_b else None
        ops.conv_wgrad(x, nbr, dy, dw3, db)
        dw5 = dw3[:, :, :cin].reshape(wshape)
        dx = None
        if ctx.needs_input_grad[0]:
            # submanifold map: nbr[o][i] = j  <=>  nbr[kvol - 1 - o][j] = i, so dx = conv(dy, W') on the SAME map with
            # W'[ci][o][co] = W[co][kvol - 1 - o][ci]
            wt = _c(w.view(cout, kvol, cp).flip(1).permute(2, 1, 0)).view(cp, kvol * cout)
            dxp = _f32((x.shape[0], cp), dy)
            ops.gemm(dy, wt, dxp, nbr=nbr, kvol=kvol, nbr_kmajor=True, cache=False)  # wt: a per-call temporary
            dx = dxp[:, :cin]
        return dx, dw5, db, None


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x, w = _c(x), _c(w)
        y = _f32((x.shape[0], w.shape[0]), x)
        ops.gemm(x, w, y, bias=b)
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _f32(x.shape, dy)
            ops.gemm(dy, _c(w.t()), dx, cache=False)  # the transposed copy is a per-call temporary
        dw = torch.zeros_like(w)
        db = torch.zeros(w.shape[0], dtype=torch.float32, device=dy.device) if ctx.has_b else None
        ops.linear_wgrad(x, dy, dw, db)
        return dx, dw, db


def linear(x, mod):
    """nn.Linear on the library's GEMMs when both widths are multiples of 16 (every Linear of the trunk); the two heads
    (C -> classes / c_in) and the B-row timestep MLP are plain torch."""
    w, b = mod.weight, mod.bias
    if w.shape[0] % 16 or w.shape[1] % 16 or x.shape[0] < 1:
        return F.linear(x, w, b)
    return _Linear.apply(x, w, b)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps):
        x = _c(x)
        y = torch.empty_like(x)
        ops.layernorm(x, g, b, y, eps=eps)
        ctx.save_for_backward(x, g)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(x)
        dg = torch.zeros_like(g)
        db = torch.zeros_like(g)
        ops.layernorm_bwd(x, g, dy, dx, accumulate=False, eps=ctx.eps, dgamma=dg, dbeta=db)
        return dx, dg, db, None


def layernorm(x, mod):
    return _LayerNorm.apply(x, mod.weight, mod.bias, float(mod.eps))


class _Attention(torch.autograd.Function):
    """softmax(q k^T scale) v per padded patch and head on gathered rows (ref: ptv3.py:246-296 / :988-1055).
    q (N, C), kv (N, 2C) (self attention: views of the packed qkv)."""

    @staticmethod
    def forward(ctx, q, k, v, q_gidx, kv_gidx, widx, patch_start, patch_start_host, heads, max_len, scale):
        o = _f32(q.shape, q)
        ops.attention(q, k, v, q_gidx, kv_gidx, widx, patch_start, heads, max_len, scale, o)
        ctx.save_for_backward(q, k, v, q_gidx, kv_gidx, widx, patch_start)
        ctx.meta = (list(patch_start_host), heads, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, q_gidx, kv_gidx, widx, patch_start = ctx.saved_tensors
        psh, heads, scale = ctx.meta
        dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        ops.attention_bwd(q, k, v, q_gidx, kv_gidx, widx, patch_start, psh, heads, scale, _c(do), dq, dk, dv)
        return dq, dk, dv, None, None, None, None, None, None, None, None


class _SegmentMax(torch.autograd.Function):
    """Per-channel maximum over the (contiguous) children of every pooled row (torch_scatter.segment_csr(reduce="max"),
    ptv3.py:510-515).  The backward sends a pooled row's gradient to the child that held the maximum (recomputed)."""

    @staticmethod
    def forward(ctx, y, seg, cluster, m):
        y = _c(y)
        c = y.shape[1]
        one = torch.ones(c, dtype=torch.float32, device=y.device)
        out = _f32((m, c), y)
        ops.segment_max(y, seg, m, one, torch.zeros_like(one), ops.ACT_NONE, out)
        ctx.save_for_backward(y, out, seg[:m + 1], cluster)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, out, seg, cluster = ctx.saved_tensors
        cl = cluster.long()
        hit = y == out[cl]
        # exactly ONE child per (pooled row, channel) receives the gradient - the first that holds the maximum, like
        # segment_csr's arg-max (torch_scatter updates its arg only on a strictly larger value); ties are rare but exist
        # (GELU outputs saturate, duplicate points)
        cs = hit.to(torch.int32).cumsum(0)
        before = (cs - hit.to(torch.int32))[seg[:-1].long()]  # hits in front of each segment, per channel
        first = hit & ((cs - before[cl]) == 1)
        return first.to(dout.dtype) * dout[cl], None, None, None


class _SceneRows(torch.autograd.Function):
    """rows[i] = per_scene[batch[i]] for a level in (batch | z) order: the rows of a scene are contiguous, so the backward is
    one column sum per scene (torch's index backward serialises the N duplicates of a row: 234 of a 450 ms step)."""

    @staticmethod
    def forward(ctx, per_scene, batch, offs):
        ctx.offs = offs
        return per_scene[batch]

    @staticmethod
    def backward(ctx, dy):
        o = ctx.offs
        return torch.stack([dy[o[b]:o[b + 1]].sum(0) for b in range(len(o) - 1)]), None, None


def _swish(x):  # ptv3.py:30-31
    return x * torch.sigmoid(x)


def _bn_gelu(x, bn):
    """nn.BatchNorm1d -> GELU, in the module's own mode: training = batch statistics, running buffers updated like the
    module would; eval (`model.eval(); model(batch)`, e.g. a validation-loss hook) = running statistics, buffers untouched."""
    if not bn.training:
        return F.gelu(F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, bn.momentum, bn.eps))
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return F.gelu(F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, bn.momentum, bn.eps))


def voxel_representatives(grid, offset):
    """Points that share a (batch element, grid_coord) voxel -> (keep, rep, offset_u): `keep` = indices of the first point
    (caller's order) of every voxel, ascending; `rep[i]` = position in `keep` of point i's voxel; `offset_u` = cumulative
    point counts of the kept points per batch element.  torch device ops (training path only)."""
    n, dev = grid.shape[0], grid.device
    ar = torch.arange(n, device=dev)
    batch = torch.bucketize(ar, offset.to(dev), right=True)
    g = grid.long()
    key = ((batch << 17 | g[:, 0]) << 17 | g[:, 1]) << 17 | g[:, 2]  # grid < 2^16 per axis (structure.py:74), batch < 2^12
    _, inverse = torch.unique(key, return_inverse=True)
    first = torch.full((int(inverse.max()) + 1,), n, dtype=torch.long, device=dev).scatter_reduce_(0, inverse, ar, "amin")
    keep = first.sort().values
    pos = torch.empty(n, dtype=torch.long, device=dev)
    pos[keep] = torch.arange(keep.numel(), device=dev)
    rep = pos[first[inverse]]
    offset_u = torch.bincount(batch[keep], minlength=offset.numel()).cumsum(0).to(offset.dtype)
    return keep, rep, offset_u


def feat_is_cuda(input_dict):
    return bool(getattr(input_dict.get("feat"), "is_cuda", False))


class _St:
    """A branch's point set at one level: features in the plan's physical (batch | z) order."""

    def __init__(self, level, x, curves, ref_order, parent=None):
        self.level, self.x, self.curves, self.ref_order, self.parent = level, x, curves, ref_order, parent
        self.conv = None  # stale conv input behind an unpooling (ptv3.py:597-630: sparse_conv_feat is not re-synchronised)


class TrainGraph:
    """Builds the autograd graph of one training forward of `model` (DefaultSegmentorV2 in train mode)."""

    def __init__(self, model):
        self.model = model
        self.eng = _engine.Engine(model, "fp32")  # the plan builder (never prepared: no second copy of the weights)
        self.criteria = build_criteria(model.criteria_cfg, model.loss_type, model.task_num)

    # ---------------------------------------------------------------------------------------- pieces
    def _mask(self, st, name, rate, masks):
        """Stochastic-depth row mask of module `name` (already divided by the keep probability), physical order."""
        if masks is not None:
            q = masks.get(name)
            if not q:
                return None
            m = torch.as_tensor(q.pop(0), dtype=torch.float32, device=st.x.device).reshape(-1, 1)
            # recorded masks are in the REFERENCE's row order of the level: the input order at level 0, the sorted order of
            # the curve that was first in the order list when the level was pooled (ptv3.py:489-493) above it;
            # ref_order: reference row -> physical row
            out = torch.empty_like(m)
            out[st.ref_order.long()] = m
            return out
        if rate <= 0.0 or not self.model.training:  # DropPath is the identity in eval mode (timm; ptv3.py:393)
            return None
        keep = 1.0 - rate
        return torch.empty((st.x.shape[0], 1), dtype=torch.float32, device=st.x.device).bernoulli_(keep) / keep

    def _cpe(self, lv, x, seq):
        y = _SubMConv.apply(x, seq[0].weight, seq[0].bias, lv.nbr(seq[0].kernel_size, True))
        return layernorm(linear(y, seq[1]), seq[2])

    def _mlp(self, h, mlp):
        return linear(F.gelu(linear(h, mlp.fc1)), mlp.fc2)

    def _block(self, st, mod, name, t_scene, masks):
        """ref: ptv3.py:399-428."""
        lv = st.level
        x = st.x
        xconv, st.conv = (x if st.conv is None else st.conv), None
        x = x + self._cpe(lv, xconv, mod.cpe)
        if t_scene is not None and hasattr(mod, "t_mlp"):
            x = x + _SceneRows.apply(F.linear(t_scene, mod.t_mlp.weight, mod.t_mlp.bias), lv.batch.long(), list(lv.offs_host))
        att = mod.attn
        c = x.shape[1]
        qkv = linear(layernorm(x, mod.norm1[0]), att.qkv)
        gidx, widx = lv.slots(st.curves[att.order_index], att.patch_size, att.enable_flash)
        patch_start, max_len = lv.pad(att.patch_size, att.enable_flash)[4:6]
        psh = lv.pad_host(att.patch_size, att.enable_flash)[3].tolist()
        o = _Attention.apply(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], gidx, gidx, widx, patch_start, psh, att.num_heads,
                             max_len, att.scale)
        a = linear(o, att.proj)
        m = self._mask(st, name + ".drop_path.0", mod.drop_prob, masks)
        x = x + (a if m is None else a * m)
        h = self._mlp(layernorm(x, mod.norm2[0]), mod.mlp[0])
        m = self._mask(st, name + ".drop_path.0", mod.drop_prob, masks)
        st.x = x + (h if m is None else h * m)
        return st

    def _embedding(self, plan, feat, emb, curves, inv0):
        """ref: ptv3.py:633-663.  inv0: caller row -> physical row (the reference's level-0 order is the caller's)."""
        lv = plan.levels[0]
        x = feat[plan.perm0.long()]
        y = _SubMConv.apply(x, emb.stem.conv.weight, None, lv.nbr(emb.stem.conv.kernel_size, True))
        return _St(lv, _bn_gelu(y, emb.stem.norm), curves, inv0)

    def _pooling(self, plan, st, down, cum_to, perm):
        """ref: ptv3.py:464-555."""
        fine, coarse = st.level, plan.levels[cum_to]
        cluster, seg = plan.link(fine.cum, cum_to)
        y = _SegmentMax.apply(linear(st.x, down.proj), seg, cluster, coarse.n)
        curves = st.curves if perm is None else [st.curves[int(j)] for j in perm]
        order = coarse.order(st.curves[0])  # the reference numbers the pooled points by unique(code[0]) (ptv3.py:489)
        if order is None:
            order = torch.arange(coarse.n, dtype=torch.int32, device=st.x.device)
        return _St(coarse, _bn_gelu(y, down.norm[0]), curves, order, parent=st)

    def _unpooling(self, plan, st, up):
        """ref: ptv3.py:597-630."""
        parent = st.parent
        fine, coarse = parent.level, st.level
        cluster, _ = plan.link(fine.cum, coarse.cum)
        child = _bn_gelu(linear(st.x, up.proj[0]), up.proj[1])
        par = _bn_gelu(linear(parent.x, up.proj_skip[0]), up.proj_skip[1])
        out = _St(fine, None, parent.curves, parent.ref_order, parent=parent.parent)
        out.conv = par  # what the next Block's CPE conv reads: the skip feature before scaling and merging
        f = 2 ** -0.5 if up.skip_connection_scale else 1.0
        if up.skip_connection_scale_i is not None:
            f *= 0.8 ** (int(up.skip_connection_scale_i) - 1)
        if f != 1.0:
            par = par * f
        gathered = child[cluster.long()]
        if up.skip_connection_mode == "add":
            out.x = par + gathered
        else:
            out.x = linear(torch.cat([par, gathered], dim=-1), up.proj_cat[0])
        return out

    def _cross_block(self, nst, cst, cb, masks):
        """ref: ptv3.py:1179-1223 + :988-1055 (cross_block2: n <- c)."""
        lv, clv = nst.level, cst.level
        if clv is not lv and list(clv.offs_host) != list(lv.offs_host):
            raise _engine.CdsegError("cross attention needs the same number of c- and n-branch bottleneck points per batch element")
        xq = nst.x + self._cpe(lv, nst.x, cb.q_cpe)
        xkv = cst.x + self._cpe(clv, cst.x, cb.kv_cpe)
        hq, hkv = layernorm(xq, cb.q_norm1[0]), layernorm(xkv, cb.kv_norm1[0])
        cst.x = hkv  # the kv point leaves the block holding its normed feature (modules.py:68-73)
        att = cb.attn
        cq = xq.shape[1]
        q = linear(hq, att.q)
        kv = linear(hkv, att.kv)
        K = att.q_patch_size
        q_gidx, widx = lv.slots(nst.curves[att.order_index], K, att.enable_flash)
        kv_gidx, _ = clv.slots(cst.curves[att.order_index], K, att.enable_flash)
        patch_start, max_len = lv.pad(K, att.enable_flash)[4:6]
        psh = lv.pad_host(K, att.enable_flash)[3].tolist()
        o = _Attention.apply(q, kv[:, :cq], kv[:, cq:], q_gidx, kv_gidx, widx, patch_start, psh, att.num_heads, max_len, att.scale)
        a = linear(o, att.proj)
        name = "backbone._tm_dec0.cross_block2.drop_path.0"
        m = self._mask(nst, name, cb.drop_prob, masks)
        x = xq + cb.tm_feat * (a if m is None else a * m)
        h = self._mlp(layernorm(x, cb.q_norm2[0]), cb.mlp[0])
        m = self._mask(nst, name, cb.drop_prob, masks)
        nst.x = x + (h if m is None else h * m)

    # ---------------------------------------------------------------------------------------- the forward
    def forward(self, input_dict, draws=None):
        """input_dict: coord, grid_coord, feat, offset, segment on the model's device.  draws (optional, for replaying a
        recorded step): ts (B, 1), noise (N, c_in), perms (8 x 4), masks {DropPath module name: [row masks]}.
        Returns dict(loss, n_pred, c_pred, c_target).

        The reference trainer calls the model inside `torch.cuda.amp.autocast(enabled=cfg.enable_amp)` and scales the loss
        with a GradScaler (engines/train.py:226-240).  This forward is exact fp32 whatever the context: autocast is switched
        off inside it (torch's own ops here - heads, BatchNorm, GELU - would otherwise hand half tensors to fp32 kernels), so
        the trainer's AMP branch runs unchanged: the scaler multiplies an fp32 loss and finds no overflow."""
        if feat_is_cuda(input_dict):
            with torch.autocast(device_type="cuda", enabled=False):
                return self._forward(input_dict, draws)
        return self._forward(input_dict, draws)

    def _forward(self, input_dict, draws=None):
        model, bb = self.model, self.model.backbone
        feat, coord = input_dict["feat"].float(), input_dict["coord"].float()
        dev = feat.device
        n = feat.shape[0]
        offset = input_dict["offset"]
        offset_host = [int(v) for v in offset.cpu().tolist()]
        B = len(offset_host)
        draws = draws or {}
        masks = draws.get("masks")
        if masks is not None:
            masks = {k: list(v) for k, v in masks.items()}
        n_orders = len(bb.order)
        grid = input_dict["grid_coord"]
        rep = None
        try:
            plan = self.eng.build_plan(grid, offset.to(torch.int64), offset_host, n)
        except DuplicateVoxelsError as e:
            # Mix3D (datasets/utils.py:51-54, mix_prob = 0.8 in every shipped CDSegNet training config) merges two scenes into
            # one batch element; both grids start at 0, so some voxels hold a point of each.  spconv tolerates that (its hash
            # keeps one row per voxel as the neighbour, every row is convolved); the kernel maps here need one point per
            # voxel, so the surplus points are FOLDED onto their voxel: the network runs on the first point (caller's order) of
            # every voxel, and every folded point reads its representative's prediction - it still enters the loss with its own
            # label, and its gradient flows into the shared row.
            keep, rep, offset_u = voxel_representatives(grid, offset)
            if not getattr(self, "_warned_duplicates", False):
                self._warned_duplicates = True
                warnings.warn(f"training batch with {e.count} points in already occupied voxels (Mix3D): folded onto the "
                              f"first point of their voxel (cdsegnet_amd/train_graph.py)")
            feat, coord, grid = feat[keep], coord[keep], grid[keep]
            n, offset_in = feat.shape[0], offset
            offset = offset_u
            offset_host = [int(v) for v in offset.cpu().tolist()]
            if "noise" in draws:
                draws = dict(draws, noise=torch.as_tensor(draws["noise"], dtype=torch.float32)[keep.cpu()])
            plan = self.eng.build_plan(grid, offset.to(torch.int64), offset_host, n)
        lv0 = plan.levels[0]
        inv0 = torch.empty(n, dtype=torch.long, device=dev)
        inv0[plan.perm0.long()] = torch.arange(n, device=dev)
        batch0 = torch.empty(n, dtype=torch.long, device=dev)  # batch index in the CALLER's order
        batch0[plan.perm0.long()] = lv0.batch.long()
        point = {"offset": offset, "loss_mode": "train"}
        base_curves = [_engine.CURVES.index(o) for o in bb.order]

        def shuffled(perm):
            return list(base_curves) if perm is None else [base_curves[int(j)] for j in perm]

        t_scene = None
        if bb.condition:
            x0 = feat if model.c_in_channels == feat.shape[-1] else coord
            c_target = x0
            c_feat = x0
            if model.dm:
                # draws in the reference's order: timesteps, noise, then the backbone's order shuffles (default.py:449-459)
                ts = torch.as_tensor(draws["ts"]) if "ts" in draws else torch.randint(0, model.T, size=(B, 1), dtype=torch.int64)
                ts = ts.to(dev).reshape(B, 1)
                noise = (torch.as_tensor(draws["noise"], dtype=torch.float32) if "noise" in draws
                         else torch.normal(0, 1, size=tuple(x0.shape), dtype=torch.float32)).to(dev)
                if model.T_dim != -1:
                    t_emb = model.t_emb_table.to(dev)[ts[:, 0] + 1]  # rows of calc_t_emb(ts) (comm.py:21-39), one per scene
                    t_scene = _swish(F.linear(_swish(F.linear(t_emb, bb.fc_t1.weight, bb.fc_t1.bias)), bb.fc_t2.weight, bb.fc_t2.bias))
                a = model.Alpha_bar.to(dev)[ts[:, 0]][batch0][:, None]
                c_feat = torch.sqrt(a) * x0 + torch.sqrt(1 - a) * noise  # continuous_q_sample, default.py:216-222
                if model.dm_target == "noise":
                    c_target = noise
                if model.dm_min_snr is not None:
                    raise NotImplementedError("dm_min_snr (SNR loss weights): off in every shipped config")
            point["c_target"] = c_target
        perms = draws.get("perms")
        if perms is None:
            perms = [torch.randperm(n_orders).tolist() if bb.shuffle_orders else None for _ in range(8)]
        pi = iter(perms)
        n_cum, c_cum = plan.n_cum, plan.c_cum

        def enc_stage(st, branch, s, cum, perm, tsc):
            enc = getattr(getattr(bb, f"_{branch}_enc"), f"enc{s}")
            if s > 0:
                st = self._pooling(plan, st, enc.down, cum[s], perm)
            for name, mod in enc._modules.items():
                if name.startswith("block"):
                    self._block(st, mod, f"backbone._{branch}_enc.enc{s}.{name}", tsc, masks)
            return st

        def dec_stage(st, branch, s, tsc):
            dec = getattr(getattr(bb, f"_{branch}_dec"), f"dec{s}")
            st = self._unpooling(plan, st, dec.up)
            for name, mod in dec._modules.items():
                if name.startswith("block"):
                    self._block(st, mod, f"backbone._{branch}_dec.dec{s}.{name}", tsc, masks)
            return st

        if bb.condition:
            # the reference interleaves the encoders c0 n0 c1 n1 n2 c2 n3 n4 (ptv3.py:1781-1794): that fixes which shuffle a
            # stage consumes AND the order in which stochastic-depth masks are drawn; the two encoders are independent
            c_curves = shuffled(next(pi))
            n_curves = shuffled(next(pi))
            p_c1, p_n1, p_n2, p_c2, p_n3, p_n4 = (next(pi) for _ in range(6))
            cst = self._embedding(plan, c_feat, bb._c_embedding, c_curves, inv0)
            nst = self._embedding(plan, feat, bb._n_embedding, n_curves, inv0)
            cst = enc_stage(cst, "c", 0, c_cum, None, t_scene)
            nst = enc_stage(nst, "n", 0, n_cum, None, None)
            cst = enc_stage(cst, "c", 1, c_cum, p_c1, t_scene)
            nst = enc_stage(nst, "n", 1, n_cum, p_n1, None)
            nst = enc_stage(nst, "n", 2, n_cum, p_n2, None)
            cst = enc_stage(cst, "c", 2, c_cum, p_c2, t_scene)
            nst = enc_stage(nst, "n", 3, n_cum, p_n3, None)
            nst = enc_stage(nst, "n", 4, n_cum, p_n4, None)
            self._cross_block(nst, cst, bb._tm_dec0.cross_block2, masks)
            for s in reversed(range(bb.c_num_stages - 1)):
                cst = dec_stage(cst, "c", s, t_scene)
            c_phys = F.linear(cst.x, bb._c_head.weight, bb._c_head.bias) if isinstance(bb._c_head, torch.nn.Linear) else cst.x
            point["c_pred"] = c_phys[inv0]
        else:
            nst = self._embedding(plan, feat, bb._n_embedding, shuffled(next(pi)), inv0)
            for s in range(bb.n_num_stages):
                nst = enc_stage(nst, "n", s, n_cum, next(pi) if s > 0 else None, None)
        for s in reversed(range(bb.n_num_stages - 1)):
            nst = dec_stage(nst, "n", s, None)
        n_phys = F.linear(nst.x, bb._n_head.weight, bb._n_head.bias) if isinstance(bb._n_head, torch.nn.Linear) else nst.x
        point["n_pred"] = n_phys[inv0]
        if rep is not None:  # folded duplicates: back to the caller's N rows (and its offsets, which the criteria sample by)
            point["offset"] = offset_in
            for k in ("n_pred", "c_pred", "c_target"):
                if point.get(k) is not None:
                    point[k] = point[k][rep]
        point["n_target"] = input_dict["segment"]
        loss = self.criteria(point)
        return dict(loss=loss, n_pred=point["n_pred"], c_pred=point.get("c_pred"), c_target=point.get("c_target"))
