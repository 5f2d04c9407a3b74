"""TEST INFRASTRUCTURE: a PyTorch-CPU emulation of the C-ABI ops (same signatures as
cdsegnet_amd.ops), used ONLY to exercise the engine's HOST logic (plan building, curve/slot
bookkeeping, buffer plumbing, quirks) in the GPU-less build container.  It is never imported by
the product; tests monkeypatch ``cdsegnet_amd.engine.ops`` with this module."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import model as OM
from oracle import serialization as S

ACT_NONE, ACT_GELU, ACT_SWISH = 0, 1, 2
F32, BF16 = 0, 1


LP_DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16}


def is_lp(dtype):
    return dtype in (torch.bfloat16, torch.float16)


def grid_max(grid):
    return grid.max().reshape(1).to(torch.int64)


def offset2batch(offset, n):
    return torch.from_numpy(S.offset2batch(offset.numpy())).int()


def encode(grid, batch, depth, order):
    name = order if isinstance(order, str) else S.ORDERS[order]
    return torch.from_numpy(S.encode(grid.numpy(), None if batch is None else batch.numpy(), depth, name))


def encode4(grid, batch, depth):
    return torch.stack([encode(grid, batch, depth, o) for o in S.ORDERS])


def sort_pairs(keys, vals=None, end_bit=64):
    assert int(keys.max()) < (1 << end_bit) if end_bit < 64 else True
    ks, perm = torch.sort(keys, stable=True)
    v = perm.int() if vals is None else vals[perm]
    return ks, v


def sort_curves(code4, rows, end_bit):
    return torch.stack([torch.sort(code4[r], stable=True)[1].int() for r in rows]) if len(rows) else \
        torch.empty((0, code4.shape[1]), dtype=torch.int32)


def invert_perm(perm):
    inv = torch.empty_like(perm)
    inv[perm.long()] = torch.arange(len(perm), dtype=perm.dtype)
    return inv


def widen(x):
    return x.long()


def gather_rows(src, idx):
    out = src[idx.long().clamp(min=0)]
    out[idx < 0] = 0
    return out


def scatter_rows(src, idx, out):
    m = idx >= 0
    out[idx[m].long()] = src[m]
    return out


def gather_i32(src, idx):
    return src[idx.long()]


def plan_gather_grid(grid, perm, zs, depth):
    return grid[perm.long()].int(), (zs >> (3 * depth)).int()


def pool_level(zs, shift, count_out=None):
    sh = zs >> shift
    flag = torch.ones_like(sh, dtype=torch.int32)
    flag[1:] = (sh[1:] != sh[:-1]).int()
    cluster = (torch.cumsum(flag, 0) - 1).int()
    m = int(cluster[-1]) + 1
    seg = torch.full((len(zs) + 1,), -12345, dtype=torch.int32)
    seg[:m] = torch.nonzero(flag).flatten().int()
    seg[m] = len(zs)
    cnt = torch.tensor([m], dtype=torch.int32)
    if count_out is not None:
        count_out.copy_(cnt)
        cnt = count_out
    return cluster, seg, cnt


def pool_levels(zs, shifts, last_idx):
    cl, sg, meta = [], [], []
    for sh in shifts:
        c, s_, cnt = pool_level(zs, sh)
        cl.append(c)
        sg.append(s_)
        meta.append(torch.cat([cnt.int(), c[last_idx.long()].int()]))
    dup = (zs[1:] == zs[:-1]).sum().int().reshape(1)
    return torch.stack(cl), torch.stack(sg), torch.cat(meta + [dup])


def link_derive(cl0a, seg0a, ma, cl0b, seg0b, mb):
    cluster = cl0b[seg0a[:ma].long()].int()
    seg = torch.cat([cl0a[seg0b[:mb].long()].int(), torch.tensor([ma], dtype=torch.int32)])
    return cluster, seg


def coarse_orders(clusters, orders, sizes):
    res = []
    for cl, m in zip(clusters, sizes):
        per = []
        for od in orders:
            v = cl[od.long()]
            keep = torch.ones_like(v, dtype=torch.bool)
            keep[1:] = v[1:] != v[:-1]
            o = v[keep]
            assert o.numel() == m
            per.append(o.int())
        res.append(per)
    return res


def pool_gather(seg, m, n_fine, pd, grid_f, batch_f, code4_f):
    h = seg[:m].long()
    return grid_f[h] >> pd, batch_f[h], code4_f[:, h] >> (3 * pd)


def pad_plan_batch(items, nb):
    return [pad_plan(o, a, b, k, n) for o, a, b, k, n in items]


def nbr_table(zs, grid, batch, depth, ksize, kmajor=False):
    t = torch.from_numpy(OM.subm_neighbors(grid.numpy(), batch.numpy(), ksize)).int()
    return t.t().contiguous() if kmajor else t


def nbr_table_from_parent(zs, grid, cluster, parent_nbr3, seg_start, m, depth, ksize, kmajor=False):
    batch = (zs >> (3 * depth)).int()
    return nbr_table(zs, grid, batch, depth, ksize, kmajor)


def nbr_table_from_info(grid, cluster, parent_nbr3, cinfo, m, depth, ksize, kmajor=False):
    """The kernel's own algebra (csrc/serialize.hip: nbr_from_info_kernel), so that the host-logic tests check the derivation
    end to end: target = first child + popcount(occupancy below the target's octant)."""
    g = grid.numpy().astype(np.int64)
    n, r, kv = len(g), ksize // 2, ksize ** 3
    pn, info, cl = parent_nbr3.numpy().astype(np.int64), cinfo.numpy().astype(np.int64), cluster.numpy().astype(np.int64)
    out = np.full((n, kv), -1, dtype=np.int32)
    pop = np.array([bin(v).count("1") for v in range(256)], dtype=np.int64)
    for o in range(kv):
        a, b, c = o // (ksize * ksize), (o // ksize) % ksize, o % ksize
        if o == kv // 2:
            out[:, o] = np.arange(n)
            continue
        t = g + np.array([a - r, b - r, c - r])
        valid = ((t >= 0) & (t < (1 << depth))).all(1)
        d = (t >> 1) - (g >> 1)
        cell = (d[:, 0] + 1) * 9 + (d[:, 1] + 1) * 3 + (d[:, 2] + 1)
        par = pn[np.clip(cell, 0, 26), cl]
        ok = valid & (par >= 0)
        inf = info[np.where(ok, par, 0)]
        occ, octant = inf & 255, ((t[:, 0] & 1) << 2) | ((t[:, 1] & 1) << 1) | (t[:, 2] & 1)
        hit = ok & (((occ >> octant) & 1) == 1)
        res = (inf >> 8) + pop[occ & ((1 << octant) - 1)]
        out[:, o] = np.where(hit, res, -1)
    tt = torch.from_numpy(out)
    return tt.t().contiguous() if kmajor else tt


def pad_plan(order, offs, offs_pad, patch, n_pad):
    offs, offs_pad = offs.numpy().astype(np.int64), offs_pad.numpy().astype(np.int64)
    gidx = np.empty(n_pad, dtype=np.int32)
    widx = np.empty(n_pad, dtype=np.int32)
    for b in range(len(offs) - 1):
        nb = offs[b + 1] - offs[b]
        for p in range(offs_pad[b], offs_pad[b + 1]):
            local = p - offs_pad[b]
            real = local < nb
            rank = offs[b] + (local if real else local - patch)
            g = rank if order is None else int(order[rank])
            gidx[p] = g
            widx[p] = g if real else -1
    return torch.from_numpy(gidx), torch.from_numpy(widx)


def _act(v, act):
    if act == ACT_GELU:
        return F.gelu(v)
    if act == ACT_SWISH:
        return v * torch.sigmoid(v)
    return v


def gemm(A, W, out, *, bias=None, scale=None, shift=None, act=ACT_NONE, res=None, add_src=None, add_idx=None,
         nbr=None, out_idx=None, out2=None, out2_pre_add=False, M=None, kvol=1, colbias=None, ln_pre=None,
         ln_post=None, ln_out=None, ln_eps=1e-5, nbr_kmajor=False, cache=True):
    assert A.dtype == W.dtype
    Af, Wf = A.float(), W.float()
    if nbr is not None and nbr_kmajor:
        nbr = nbr.t()
    if nbr is None:
        v = Af @ Wf.t()
    else:
        K = W.shape[1] // kvol
        v = torch.zeros(nbr.shape[0], W.shape[0])
        for o in range(kvol):
            j = nbr[:, o].long()
            m = j >= 0
            v[m] += Af[j[m]] @ Wf[:, o * K:(o + 1) * K].t()
    if bias is not None:
        v = v + bias
    if scale is not None:
        v = v * scale + shift
    v = _act(v, act)
    if ln_pre is not None:
        v = F.layer_norm(v, (v.shape[1],), ln_pre[0], ln_pre[1], ln_eps)
    if out2 is not None and out2_pre_add:
        out2.copy_(v.to(out2.dtype))
    if res is not None:
        v = v + res
    if colbias is not None:
        v = v + colbias
    if add_src is not None:
        v = v + add_src[add_idx.long()]
    if out_idx is not None:
        out[out_idx.long()] = v.to(out.dtype)
    else:
        out.copy_(v.to(out.dtype))
    if out2 is not None and not out2_pre_add:
        out2.copy_(v.to(out2.dtype))
    if ln_post is not None:
        ln_out.copy_(F.layer_norm(v, (v.shape[1],), ln_post[0], ln_post[1], ln_eps).to(ln_out.dtype))
    return out


def stem5_ok(cout, dtype):
    return cout == 32 and dtype == torch.bfloat16


def child_info(zs, seg, m):
    zs, seg = zs.long(), seg.long()
    info = torch.zeros(int(m), dtype=torch.int64)
    for p in range(int(m)):
        occ = 0
        for j in range(int(seg[p]), int(seg[p + 1])):
            occ |= 1 << int(zs[j] & 7)
        info[p] = (int(seg[p]) << 8) | occ
    return info


def stem5_pack(w):
    return w


def stem5(x8, wimg, scale, shift, grid, cluster, parent_nbr3, cinfo, depth, out, out2=None):
    """Emulation through the explicit 5x5x5 map (what the device kernel avoids building)."""
    g = grid.long()
    key = {tuple(v): i for i, v in enumerate(g.tolist())}  # one batch element per call in the CPU tests, or disjoint grids
    n = g.shape[0]
    nbr = torch.full((n, 125), -1, dtype=torch.int64)
    # batches: neighbours must share the batch element; recover it from the parent chain (cluster -> same parent map)
    first = (cinfo >> 8)
    par_of = cluster.long()
    pn = parent_nbr3.long()
    for i in range(n):
        for cell in range(27):
            q = int(pn[cell, par_of[i]])
            if q < 0:
                continue
            s, occ = int(first[q]), int(cinfo[q] & 255)
            rank = 0
            for octv in range(8):
                if not (occ >> octv) & 1:
                    continue
                j = s + rank
                rank += 1
                d = g[j] - g[i]
                if int(d.abs().max()) <= 2:
                    nbr[i, int((d[0] + 2) * 25 + (d[1] + 2) * 5 + (d[2] + 2))] = j
    return gemm(x8, wimg, out, scale=scale, shift=shift, act=ACT_GELU, nbr=nbr, kvol=125, out2=out2)


def subm_conv3_ok(x):
    return x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] in (32, 64)


def subm_conv3_pack(w):
    return w  # the emulation convolves with the plain (C, 27*C) weight


def subm_conv3(x, wimg, bias, nbr_kmajor, out):
    return gemm(x, wimg, out, bias=bias, nbr=nbr_kmajor, nbr_kmajor=True, kvol=27)


DEEP512_MIN_ROWS = 2560
DEEP_CHANNELS = (128, 256, 512)


def block_rr_ok(channels, dtype):
    return dtype == torch.bfloat16 and (channels in (32, 64) or channels in DEEP_CHANNELS)


def block_rr_head_on(channels=32):
    return channels in DEEP_CHANNELS


def block_rr_pack(channels, wl, wqkv, wp, w1, w2):
    return ((wl, wqkv) if wl is not None else None), (wp, w1, w2)  # the emulation multiplies by the plain weights


def cpe_head_rr(y, head_img, bl, lnp, x, colbias, ln1, bqkv, qkv, eps=1e-5, qkv_flags=0):
    return cpe_head_fused(y, head_img[0], bl, lnp, x, colbias, ln1, head_img[1], bqkv, qkv, eps, qkv_flags)


def attn_tail_rr(o, tail_img, bp, ln_g, ln_b, b1, b2, x, xc=None, eps=1e-5):
    return attn_tail_fused(o, tail_img[0], bp, ln_g, ln_b, tail_img[1], b1, tail_img[2], b2, x, xc, eps)


def cpe_head_fused_ok(y):
    return y.dtype == torch.bfloat16 and y.shape[1] in (32, 64)


def cpe_head_fused(y, wl, bl, lnp, x, colbias, ln1, wqkv, bqkv, qkv, eps=1e-5, qkv_flags=0):
    # (qkv_flags = ATTN_V_BF16 only matters in the IEEE-half build; the emulation's 16-bit type is bfloat16)
    c = x.shape[1]
    v = F.layer_norm(y.float() @ wl.float().t() + bl, (c,), lnp[0], lnp[1], eps)
    x += v
    if colbias is not None:
        x += colbias
    h = F.layer_norm(x, (c,), ln1[0], ln1[1], eps).to(torch.bfloat16)
    qkv.copy_((h.float() @ wqkv.float().t() + bqkv).to(qkv.dtype))
    return qkv


def attn_tail_fused_ok(o, hidden):
    c = o.shape[1]
    return o.dtype == torch.bfloat16 and c in (32, 64) and hidden == 4 * c


def attn_tail_fused(o, wp, bp, ln_g, ln_b, w1, b1, w2, b2, x, xc=None, eps=1e-5):
    x += o.float() @ wp.float().t() + bp
    h = F.layer_norm(x, (x.shape[1],), ln_g, ln_b, eps).to(torch.bfloat16)
    return mlp_fused(h, w1, b1, w2, b2, x, xc)


def mlp_fused_ok(h, hidden):
    c = h.shape[1]
    return h.dtype == torch.bfloat16 and c in (32, 64, 128) and hidden == 4 * c


def mlp_fused(h, w1, b1, w2, b2, x, xc=None):
    u = F.gelu(h.float() @ w1.float().t() + b1).to(torch.bfloat16)
    x += u.float() @ w2.float().t() + b2
    if xc is not None:
        xc.copy_(x.to(xc.dtype))
    return x


def layernorm(x, gamma, beta, out, *, eps=1e-5, res=None, colbias=None, out2=None):
    v = F.layer_norm(x.float(), (x.shape[1],), gamma, beta, eps)
    if res is not None:
        v = v + res
    if colbias is not None:
        v = v + colbias
    out.copy_(v.to(out.dtype))
    if out2 is not None:
        out2.copy_(v.to(out2.dtype))
    return out


ATTN_Q_PRESCALED, ATTN_V_BF16 = 1, 2


def attention(q, k, v, q_gidx, kv_gidx, widx, patch_start, num_heads, max_len, scale, out, work=0.0, flags=0):
    ps = patch_start.numpy().astype(np.int64)
    assert int(np.diff(ps).max()) == max_len
    qq, kk, vv = q.float()[q_gidx.long()], k.float()[kv_gidx.long()], v.float()[kv_gidx.long()]
    if flags & ATTN_Q_PRESCALED:  # q carries scale * log2(e): softmax over 2^s instead of e^(scale s)
        scale = math.log(2.0)
    o = OM._patch_attention(qq, kk, vv, ps, num_heads, scale)
    m = widx >= 0
    out[widx[m].long()] = o[m].to(out.dtype)
    return out


def segment_max(y, seg_start, m, scale, shift, act, out, out2=None):
    seg = seg_start[:m + 1].long()
    cluster = np.repeat(np.arange(m), np.diff(seg.numpy()))
    v = OM.segment_max(y.float(), cluster, m)
    if scale is not None:
        v = v * scale + shift
    v = _act(v, act)
    out.copy_(v)
    if out2 is not None:
        out2.copy_(v.to(out2.dtype))
    return out


def pool_fused_ok(cin, cout, dtype):
    return is_lp(dtype) and (int(cin), int(cout)) in ((32, 64), (64, 128))


def pool_fused_pack(w):
    return w.clone()  # the emulation keeps the row-major weight as its "image"


def pool_fused(x, wimg, bias, seg_start, m, scale, shift, act, out, out2=None):
    y = torch.empty((x.shape[0], wimg.shape[0]), dtype=x.dtype)
    gemm(x, wimg, y, bias=bias)
    return segment_max(y, seg_start, m, scale, shift, act, out, out2)


def segment_mean(x, seg_start, m):
    seg = seg_start[:m + 1].long()
    cluster = np.repeat(np.arange(m), np.diff(seg.numpy()))
    return OM.segment_mean(x, cluster, m)


def gemv(w, b, x, act=ACT_NONE):
    return _act(w @ x + b, act)


def randn(shape, seed, offset, device):
    g = torch.Generator().manual_seed((seed + offset) & 0x7FFFFFFF)
    return torch.randn(shape, generator=g)


def cast(src, dtype):
    return src.to(dtype)


def gather_pad_cast(src, idx, cpad, dtype):
    rows = src if idx is None else src[idx.long()]
    out = torch.zeros(rows.shape[0], cpad)
    out[:, :src.shape[1]] = rows
    return out.to(dtype)


def ddim_update(xt, eps, sqrt_ab_prev, sqrt_1m_ab, sqrt_ab, sqrt_1m_ab_prev, final=False):
    x0 = (xt - sqrt_1m_ab * eps) / sqrt_ab
    return x0 if final else sqrt_ab_prev * x0 + sqrt_1m_ab_prev * eps


def axpy(a, b, alpha):
    return a + alpha * b


# ---- test-time pipeline ops (cdsegnet_amd/csrc/testtime.hip)
def bind_stream(stream=None):
    pass


def unbind_stream():
    pass


def current_stream_id():
    return None


def record_event():
    return None


def wait_event(ev):
    pass


def voxelize(coord, grid_size):
    g = torch.floor(coord.double() / float(grid_size)).long()
    mn = g.min(0).values
    g = g - mn
    key = (g[:, 0] << 42) | (g[:, 1] << 21) | g[:, 2]
    return g.int(), key, mn.int()


def voxelize_any(coord, grid_size):
    return voxelize(coord, grid_size)


def center_shift(coord, apply_z=True):
    mn, mx = coord.min(0).values, coord.max(0).values
    z = mn[2] if apply_z else torch.zeros((), dtype=coord.dtype)
    return coord - torch.stack([(mn[0] + mx[0]) / 2, (mn[1] + mx[1]) / 2, z])


def tta_apply(xyz, rot=None, scale=None, flip=False):
    xyz = xyz.float()
    if rot is not None:
        out = xyz.double() @ torch.tensor(rot, dtype=torch.float64).t()
        return out * scale if scale is not None else out
    if flip:
        return xyz * torch.tensor([-1.0, -1.0, 1.0])
    return xyz


def div_add(x, div, add):
    return x.float() / div + add


def collect_feat(a, b):
    return torch.cat([a.float(), b.float()], 1)


def max_run(seg_start, m):
    s = seg_start[: m + 1].long()
    return (s[1:] - s[:-1]).max().int().reshape(1)


def fragment_select(idx_sort, seg_start, m, frag):
    s = seg_start[: m + 1].long()
    c = s[1:] - s[:-1]
    return idx_sort[s[:-1] + frag % c].int()


def softmax_vote(logits, idx, pred):
    pred[idx.long()] += torch.softmax(logits.float(), -1)
    return pred


def argmax_rows(x):
    return x.argmax(1).int()


# ---- evaluator ops (cdsegnet_amd/csrc/testtime.hip)
def knn1(ref_xyz, ref_offset, qry_xyz, qry_offset, origin, cell, want_dist=False):
    idx = torch.full((qry_xyz.shape[0],), -1, dtype=torch.int32)
    d2 = torch.full((qry_xyz.shape[0],), 1e10, dtype=torch.float32)
    rs = qs = 0
    for re, qe in zip(ref_offset.tolist(), qry_offset.tolist()):
        if re > rs and qe > qs:
            d = torch.cdist(qry_xyz[qs:qe].double(), ref_xyz[rs:re].double()) ** 2
            v, j = d.min(1)
            idx[qs:qe] = (j + rs).int()
            d2[qs:qe] = v.float()
        rs, qs = re, qe
    return (idx, d2) if want_dist else idx


def iou_counts(pred, target, num_classes, ignore_index=-1, pred_idx=None):
    p = pred.long() if pred_idx is None else pred.long()[pred_idx.long()]
    t = target.long()
    keep = t != ignore_index
    p, t = p[keep], t[keep]
    out = torch.zeros((3, num_classes), dtype=torch.int64)
    out[1] = torch.bincount(p[(p >= 0) & (p < num_classes)], minlength=num_classes)[:num_classes]
    out[2] = torch.bincount(t[(t >= 0) & (t < num_classes)], minlength=num_classes)[:num_classes]
    hit = t[p == t]
    out[0] = torch.bincount(hit[(hit >= 0) & (hit < num_classes)], minlength=num_classes)[:num_classes]
    return out


# ------------------------------------------------------------------ training path (cdsegnet_amd/train.py on the CPU)
def attention_bwd(q, k, v, q_gidx, kv_gidx, widx, patch_start, patch_start_host, num_heads, scale, dout, dq, dk, dv):
    """Autograd through the oracle's patch attention on the library's slot plan; += into dq / dk / dv like the kernel.
    (enable_grad: the callers include torch.autograd.Function.backward, where recording is off.)"""
    ps = np.asarray(patch_start_host, dtype=np.int64)
    with torch.enable_grad():
        qq, kk, vv = (t.detach().clone().float().requires_grad_(True) for t in (q, k, v))
        o = OM._patch_attention(qq[q_gidx.long()], kk[kv_gidx.long()], vv[kv_gidx.long()], ps, num_heads, scale)
        m = widx >= 0
        full = torch.zeros_like(dout)
        full = full.index_put((widx[m].long(),), o[m])
        (full * dout).sum().backward()
    dq += qq.grad
    dk += kk.grad
    dv += vv.grad


def layernorm_bwd(x, gamma, dy, dx, accumulate=False, eps=1e-5, dgamma=None, dbeta=None):
    with torch.enable_grad():
        xr = x.detach().clone().requires_grad_(True)
        g = gamma.detach().clone().requires_grad_(True)
        b = torch.zeros_like(gamma).requires_grad_(True)
        F.layer_norm(xr, (x.shape[1],), g, b, eps).backward(dy)
    if accumulate:
        dx += xr.grad
    else:
        dx.copy_(xr.grad)
    if dgamma is not None:
        dgamma += g.grad
    if dbeta is not None:
        dbeta += b.grad
    return dx


def gelu_bwd(u, dy):
    with torch.enable_grad():
        ur = u.detach().clone().requires_grad_(True)
        F.gelu(ur).backward(dy)
    return ur.grad


def linear_wgrad(x, dy, dw, db=None, xidx=None):
    if xidx is None:
        dw += dy.t() @ x
    else:
        m = xidx >= 0
        dw += dy[m].t() @ x[xidx[m].long()]
    if db is not None:
        db += dy.sum(0)
    return dw


def conv_wgrad(x, nbr_kmajor, dy, dw3, db=None):
    for o in range(nbr_kmajor.shape[0]):
        linear_wgrad(x, dy, dw3[:, o, :], None, xidx=nbr_kmajor[o])
    if db is not None:
        db += dy.sum(0)
    return dw3
