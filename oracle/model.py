"""Oracle: floating-point side of the path (PyTorch-CPU fp32 restatement).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  A functional restatement of
CDSegNet single-step inference over a flat ``state_dict`` (reference key names)
and the backbone kwargs of configs/*/CDSegNet.py.  Follows

* pointcept/models/default.py:371-422              (DefaultSegmentorV2.inference)
* pointcept/utils/comm.py:21-39                    (calc_t_emb)
* pointcept/models/point_transformer_v3/point_transformer_v3m1_base.py
    :246-296  SerializedAttention (materialised K x K softmax, the CPU branch)
    :299-322  MLP
    :399-428  Block
    :464-555  SerializedPooling
    :597-630  SerializedUnpooling
    :633-663  Embedding
    :988-1055 SerializedCrossAttention
    :1179-1223 CrossBlock, :1332-1337 TransferModule
    :1757-1815 PointTransformerV3.forward
* pointcept/models/utils/structure.py:39-140       (Point)
* spconv.SubMConv3d (third party, NOT in /root/reference, version unpinned:
  README.md:89) - restated from its published semantics: submanifold
  cross-correlation, output sites = input sites, weight (out, k0, k1, k2, in),
  kernel axis a pairs with indices[:, 1 + a].  **parity unpinned**.
* torch_scatter.segment_csr (third party): per-segment max / mean.

All point tensors are kept in the caller's point order, exactly like the
reference; randomness (the N(0,1) noise-branch input and the eight
``randperm(4)`` order shuffles) is injected through ``draws``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import serialization as S

DEFAULT_ORDERS = ("z", "z-trans", "hilbert", "hilbert-trans")


# --------------------------------------------------------------------------- RNG
def draw_rng(seed, n_points, c_in, n_perms=8, noise_level_like=None, n_orders=4):
    """Replay the reference's CPU-generator consumption order:
    [randn_like(feat) if noise_level] -> torch.normal(0,1,(N,c_in)) -> randperm(4) x 8
    (default.py:373-374, :393; structure.py:94-98; ptv3.py:501-505)."""
    g = torch.Generator()
    g.manual_seed(seed)
    out = {}
    if noise_level_like is not None:
        out["feat_noise"] = torch.randn(noise_level_like, generator=g)
    out["noise"] = torch.normal(0, 1, size=(n_points, c_in), dtype=torch.float32, generator=g)
    out["perms"] = [torch.randperm(n_orders, generator=g).numpy().copy() for _ in range(n_perms)]
    return out


# ------------------------------------------------------------------ small layers
def linear(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def layernorm(x, sd, p, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


# Training mode of the restatement (oracle/train.py::training_forward sets it; None = eval, everything above the
# training row of SURVEY 8(f4) runs in eval).  TRAIN.masks: {DropPath module name: [mask, ...]} in call order - the
# reference's stochastic-depth draws (timm DropPath on an (N, C) feature drops whole ROWS, ptv3.py:392-394,415,423),
# recorded by make_golden.py; BatchNorm1d normalises with the batch statistics (torch's training-mode semantics).
TRAIN = None


def batchnorm_eval(x, sd, p, eps=1e-3):
    """nn.BatchNorm1d(eps=1e-3) (ptv3.py:1440): running statistics in eval mode, batch statistics under TRAIN."""
    if TRAIN is not None:
        return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.0, eps)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)


def drop_path(x, name):
    """DropPath of module `name` (identity in eval and for modules built with rate 0: they hold no masks)."""
    if TRAIN is None:
        return x
    q = TRAIN.masks.get(name)
    return x * q.pop(0) if q else x


def swish(x):  # ptv3.py:30-31
    return x * torch.sigmoid(x)


def calc_t_emb(ts, t_emb_dim):
    """comm.py:21-39.  ts (N,1) int64 -> (N, t_emb_dim) fp32."""
    half = t_emb_dim // 2
    c = np.log(10000) / (half - 1)
    f = torch.exp(torch.arange(half) * -c)
    e = ts * f
    return torch.cat((torch.sin(e), torch.cos(e)), 1)


# ------------------------------------------------------------------ sparse conv
def voxel_keys(grid, batch):
    g = np.asarray(grid, dtype=np.int64)
    b = np.asarray(batch, dtype=np.int64)
    return (b << 48) | (g[:, 0] << 32) | (g[:, 1] << 16) | g[:, 2]


def subm_neighbors(grid, batch, ksize):
    """(N, ksize^3) int64 table: index of the occupied voxel at grid + (a-r, b-r, c-r), -1 if empty.
    Column a*k*k + b*k + c."""
    g = np.asarray(grid, dtype=np.int64)
    b = np.asarray(batch, dtype=np.int64)
    keys = voxel_keys(g, b)
    srt = np.argsort(keys, kind="stable")
    skeys = keys[srt]
    n = len(keys)
    r = ksize // 2
    out = np.full((n, ksize ** 3), -1, dtype=np.int64)
    col = 0
    for a in range(ksize):
        for bb in range(ksize):
            for c in range(ksize):
                q = g + np.array([a - r, bb - r, c - r], dtype=np.int64)
                ok = (q >= 0).all(1) & (q < 65536).all(1)
                qk = (b << 48) | (q[:, 0] << 32) | (q[:, 1] << 16) | q[:, 2]
                pos = np.searchsorted(skeys, qk)
                pos = np.minimum(pos, n - 1)
                hit = ok & (skeys[pos] == qk)
                out[hit, col] = srt[pos[hit]]
                col += 1
    return out


def subm_conv3d(feat, nbr, weight, bias):
    """out[i] = bias + sum_k W[:, k, :] @ feat[nbr[i, k]] over occupied neighbours."""
    cout = weight.shape[0]
    cin = weight.shape[-1]
    w = weight.reshape(cout, -1, cin)
    out = torch.zeros(feat.shape[0], cout, dtype=feat.dtype)
    nbr_t = torch.from_numpy(nbr)
    for k in range(w.shape[1]):
        j = nbr_t[:, k]
        m = j >= 0
        if m.any():
            out[m] += feat[j[m]] @ w[:, k, :].t()
    if bias is not None:
        out = out + bias
    return out


# ------------------------------------------------------------------------- Point
class OPoint(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def make_point(coord, grid, offset, feat):
    p = OPoint()
    p.coord = coord
    p.grid = np.asarray(grid, dtype=np.int64)
    p.offset = np.asarray(offset, dtype=np.int64)
    p.batch = S.offset2batch(p.offset)
    p.feat = feat
    p.nbr = {}
    return p


def serialize_point(p, orders, perm):
    code, order, inverse, depth = S.serialization(p.grid, p.batch, orders)
    if perm is not None:  # structure.py:94-98
        code, order, inverse = code[perm], order[perm], inverse[perm]
    p.code, p.order, p.inverse, p.depth = code, order, inverse, depth


def neighbors(p, ksize):
    if ksize not in p.nbr:
        p.nbr[ksize] = subm_neighbors(p.grid, p.batch, ksize)
    return p.nbr[ksize]


FLASH_SEMANTICS = True


def plan(p, K):
    """Padding plan, cached on the point like the reference does (ptv3.py:190-199).

    FLASH_SEMANTICS=True  -> the shipped GPU path (enable_flash=True): fixed patch size K,
                             varlen patches via cu_seqlens (ptv3.py:282-288).
    FLASH_SEMANTICS=False -> the reference's CPU branch: K = min(min_b n_b, K) (ptv3.py:247-250).
    The two agree whenever every batch element has >= K points or the batch has one element."""
    if "pad" not in p:
        if not FLASH_SEMANTICS:
            K = min(int(S.offset2bincount(p.offset).min()), K)
        p.pad, p.unpad, p.cu = S.padding_plan(p.offset, K)
    return p.pad, p.unpad, p.cu


# --------------------------------------------------------------------- attention
def _patch_attention(q, k, v, cu, H, scale):
    """q (N',C), k (N',C), v (N',C) already in padded patch order; cu int64 (P+1,).
    softmax(q k^T * scale) v per (patch, head); full L x L, no mask (ptv3.py:264-280)."""
    C = q.shape[1]
    d = C // H
    out = torch.empty_like(q)
    lens = np.diff(cu)
    i = 0
    P = len(lens)
    while i < P:
        L = int(lens[i])
        j = i
        # chunk of equal-length patches, bounded so the score tensor stays < ~256 MB
        cap = max(1, int(64e6 // max(1, H * L * L)))
        while j < P and lens[j] == L and (j - i) < cap:
            j += 1
        s, e = int(cu[i]), int(cu[j])
        np_ = j - i
        qq = q[s:e].reshape(np_, L, H, d).permute(0, 2, 1, 3)
        kk = k[s:e].reshape(np_, L, H, d).permute(0, 2, 1, 3)
        vv = v[s:e].reshape(np_, L, H, d).permute(0, 2, 1, 3)
        attn = (qq * scale) @ kk.transpose(-2, -1)
        attn = torch.softmax(attn, dim=-1)
        out[s:e] = (attn @ vv).transpose(1, 2).reshape(np_ * L, C)
        i = j
    return out


def serialized_attention(p, x, sd, pre, H, oi, K):
    """ptv3.py:246-296."""
    C = x.shape[1]
    scale = (C // H) ** -0.5
    pad, unpad, cu = plan(p, K)
    order = torch.from_numpy(p.order[oi][pad])
    inverse = torch.from_numpy(unpad[p.inverse[oi]])
    qkv = linear(x, sd, pre + ".qkv")[order]
    L3 = qkv.reshape(-1, 3, C)
    feat = _patch_attention(L3[:, 0], L3[:, 1], L3[:, 2], cu, H, scale)
    feat = feat[inverse]
    return linear(feat, sd, pre + ".proj")


def serialized_cross_attention(qp, kvp, xq, xkv, sd, pre, H, oi, K):
    """ptv3.py:988-1055; the kv side reuses the q side's pad (ptv3.py:1009)."""
    C = xq.shape[1]
    scale = (C // H) ** -0.5
    pad, unpad, cu = plan(qp, K)
    assert len(kvp.order[oi]) == len(qp.order[oi]), "cross attention needs len(c)==len(n)"
    q_order = torch.from_numpy(qp.order[oi][pad])
    q_inverse = torch.from_numpy(unpad[qp.inverse[oi]])
    kv_order = torch.from_numpy(kvp.order[oi][pad])
    q = linear(xq, sd, pre + ".q")[q_order]
    kv = linear(xkv, sd, pre + ".kv")[kv_order].reshape(-1, 2, C)
    feat = _patch_attention(q, kv[:, 0], kv[:, 1], cu, H, scale)
    feat = feat[q_inverse].float()
    return linear(feat, sd, pre + ".proj")


# ------------------------------------------------------------------------ blocks
def cpe(p, x, sd, pre):
    w = sd[pre + ".0.weight"]
    y = subm_conv3d(x, neighbors(p, w.shape[1]), w, sd.get(pre + ".0.bias"))
    y = linear(y, sd, pre + ".1")
    return layernorm(y, sd, pre + ".2")


def block(p, sd, pre, H, oi, K, with_t):
    """ptv3.py:399-428 (pre_norm=True, DropPath = identity in eval).

    The CPE conv reads ``point.sparse_conv_feat.features`` (modules.py:63-66), which is
    re-synchronised with ``point.feat`` only at the END of a Block (ptv3.py:427).  After a
    SerializedUnpooling the two differ (see ``unpooling``), so the first block of every
    decoder stage convolves the stale tensor while the residual uses ``feat``."""
    x = p.feat
    x_conv = p.pop("conv_feat", x)
    x = x + cpe(p, x_conv, sd, pre + ".cpe")
    if with_t and "t_emb" in p:
        x = x + linear(p.t_emb, sd, pre + ".t_mlp")
    x = x + drop_path(serialized_attention(p, layernorm(x, sd, pre + ".norm1.0"), sd, pre + ".attn", H, oi, K), pre + ".drop_path.0")
    h = layernorm(x, sd, pre + ".norm2.0")
    h = linear(F.gelu(linear(h, sd, pre + ".mlp.0.fc1")), sd, pre + ".mlp.0.fc2")
    p.feat = x + drop_path(h, pre + ".drop_path.0")
    return p


def segment_max(x, cluster, M):
    idx = torch.from_numpy(cluster)[:, None].expand(-1, x.shape[1])
    out = torch.zeros(M, x.shape[1], dtype=x.dtype)
    return out.scatter_reduce(0, idx, x, "amax", include_self=False)


def segment_mean(x, cluster, M):
    idx = torch.from_numpy(cluster)[:, None].expand(-1, x.shape[1])
    out = torch.zeros(M, x.shape[1], dtype=x.dtype)
    return out.scatter_reduce(0, idx, x, "mean", include_self=False)


def pooling(p, sd, pre, stride, perm, with_t):
    """ptv3.py:464-555."""
    pd = (math.ceil(stride) - 1).bit_length()
    if pd > p.depth:
        pd = 0
    cluster, counts, indices, idx_ptr, head, code, order, inverse = S.pooling_structure(p.code, pd)
    if perm is not None:
        code, order, inverse = code[perm], order[perm], inverse[perm]
    M = len(counts)
    q = OPoint()
    q.feat = segment_max(linear(p.feat, sd, pre + ".proj"), cluster, M)
    q.coord = segment_mean(p.coord, cluster, M)
    q.grid = p.grid[head] >> pd
    q.code, q.order, q.inverse = code, order, inverse
    q.depth = p.depth - pd
    q.batch = p.batch[head]
    q.offset = S.batch2offset(q.batch)
    if with_t and "t_emb" in p:
        q.t_emb = p.t_emb[torch.from_numpy(head)]
    q.pooling_inverse = cluster
    q.pooling_parent = p
    q.nbr = {}
    q.feat = F.gelu(batchnorm_eval(q.feat, sd, pre + ".norm.0"))
    return q


def unpooling(p, sd, pre, mode, scale, scale_i):
    """ptv3.py:597-630 (b = s = 1: FreeU disabled).

    Reference quirk (found by the golden vectors, not in SURVEY.md): ``proj_skip`` runs through
    PointSequential and therefore updates BOTH parent.feat and parent.sparse_conv_feat
    (modules.py:68-73), but the skip scaling and the add / cat+proj_cat that follow
    (ptv3.py:609-626) assign ``parent.feat`` only.  The parent leaves with
    sparse_conv_feat.features = GELU(BN(proj_skip(parent))) - unscaled, un-merged - and the next
    Block's CPE conv consumes exactly that (``conv_feat`` here)."""
    parent = p.pooling_parent
    inv = torch.from_numpy(p.pooling_inverse)
    child = F.gelu(batchnorm_eval(linear(p.feat, sd, pre + ".proj.0"), sd, pre + ".proj.1"))
    par = F.gelu(batchnorm_eval(linear(parent.feat, sd, pre + ".proj_skip.0"), sd, pre + ".proj_skip.1"))
    parent.conv_feat = par
    if scale:  # universal_scalling, ptv3.py:34-35
        par = par * 2 ** (-0.5)
    if scale_i is not None:  # exponentially_scalling, ptv3.py:37-38 (i=False -> 0.8**-1)
        par = par * 0.8 ** (scale_i - 1)
    if mode == "add":
        par = par + child[inv]
    else:
        par = linear(torch.cat([par, child[inv]], dim=-1), sd, pre + ".proj_cat.0")
    parent.feat = par
    return parent


def embedding(p, sd, pre):
    """ptv3.py:633-663: SubMConv3d(k=5, bias=False) -> BN(eps 1e-3) -> GELU."""
    w = sd[pre + ".stem.conv.weight"]
    y = subm_conv3d(p.feat, neighbors(p, w.shape[1]), w, None)
    p.feat = F.gelu(batchnorm_eval(y, sd, pre + ".stem.norm"))
    return p


def cross_block(qp, kvp, sd, pre, H, K):
    """ptv3.py:1179-1223 with tm_feat = 1.0."""
    xq = qp.feat + cpe(qp, qp.feat, sd, pre + ".q_cpe")
    xkv = kvp.feat + cpe(kvp, kvp.feat, sd, pre + ".kv_cpe")
    hq = layernorm(xq, sd, pre + ".q_norm1.0")
    hkv = layernorm(xkv, sd, pre + ".kv_norm1.0")
    kvp.feat = hkv  # the kv point leaves the block holding its normed feature
    a = serialized_cross_attention(qp, kvp, hq, hkv, sd, pre + ".attn", H, 0, K)
    x = xq + 1.0 * drop_path(a, pre + ".drop_path.0")
    h = layernorm(x, sd, pre + ".q_norm2.0")
    h = linear(F.gelu(linear(h, sd, pre + ".mlp.0.fc1")), sd, pre + ".mlp.0.fc2")
    qp.feat = x + drop_path(h, pre + ".drop_path.0")
    return qp


# ------------------------------------------------------------------------ forward
def _stage(p, sd, pre, s, depth, heads, K, stride, perm, with_t, n_orders):
    if s > 0:
        p = pooling(p, sd, f"{pre}.enc{s}.down", stride, perm, with_t)
    for i in range(depth):
        p = block(p, sd, f"{pre}.enc{s}.block{i}", heads, i % n_orders, K, with_t)
    return p


def backbone_forward(cfg, sd, c_in, n_in, perms, run_dead=True, trace=None):
    """PointTransformerV3.forward with condition=True (ptv3.py:1757-1815).

    c_in / n_in: dicts with coord (N,3) f32, grid (N,3) int, offset (B,), feat, and
    c_in['t_emb'] (N, T_dim).  perms: the eight randperm(4) draws in consumption order
    c.serialization, n.serialization, c_enc1.down, n_enc1.down, n_enc2.down,
    c_enc2.down, n_enc3.down, n_enc4.down."""
    B = "backbone"
    orders = cfg.get("order", DEFAULT_ORDERS)
    no = len(orders)
    shuffle = cfg.get("shuffle_orders", True)
    perms = list(perms) if shuffle else [None] * 8
    pi = iter(perms)

    c = make_point(c_in["coord"], c_in["grid"], c_in["offset"], c_in["feat"])
    n = make_point(n_in["coord"], n_in["grid"], n_in["offset"], n_in["feat"])
    serialize_point(c, orders, next(pi))
    serialize_point(n, orders, next(pi))
    T_dim = cfg.get("T_dim", 128)
    if T_dim != -1 and "t_emb" in c_in:
        t = swish(linear(c_in["t_emb"], sd, B + ".fc_t1"))
        c.t_emb = swish(linear(t, sd, B + ".fc_t2"))
    with_t = T_dim != -1

    c = embedding(c, sd, B + "._c_embedding")
    n = embedding(n, sd, B + "._n_embedding")

    cd, cc, ch, cK, cs = (cfg["c_enc_depths"], cfg["c_enc_channels"], cfg["c_enc_num_head"],
                          cfg["c_enc_patch_size"], cfg["c_stride"])
    nd, nc, nh, nK, ns = (cfg["n_enc_depths"], cfg["n_enc_channels"], cfg["n_enc_num_head"],
                          cfg["n_enc_patch_size"], cfg["n_stride"])
    assert len(cd) == 3 and len(nd) == 5, "oracle follows the hard-wired interleave of ptv3.py:1785-1794"

    def cst(p, s, perm):
        return _stage(p, sd, B + "._c_enc", s, cd[s], ch[s], cK[s], cs[s - 1] if s else None, perm, with_t, no)

    def nst(p, s, perm):
        return _stage(p, sd, B + "._n_enc", s, nd[s], nh[s], nK[s], ns[s - 1] if s else None, perm, False, no)

    c = cst(c, 0, None)
    n = nst(n, 0, None)
    c = cst(c, 1, next(pi))
    n = nst(n, 1, next(pi))
    n = nst(n, 2, next(pi))
    c = cst(c, 2, next(pi))
    n = nst(n, 3, next(pi))
    n = nst(n, 4, next(pi))
    if trace is not None:
        trace["n_enc4"] = n.feat.clone()
        trace["c_enc2"] = c.feat.clone()

    # fusion (ptv3.py:1797): only cross_block2(n <- c)
    n = cross_block(n, c, sd, B + "._tm_dec0.cross_block2", nh[-1], nK[-1])
    if trace is not None:
        trace["n_fused"] = n.feat.clone()

    mode = cfg.get("skip_connection_mode", "add")
    c_mode = "add" if mode == "add" else "cat"
    n_mode = "cat" if mode == "cat_all" else "add"
    c_scale = cfg.get("skip_connection_scale", False)
    c_scale_i = False  # ctor default leaks through `is not None` (ptv3.py:610-611, 1672-1683)
    n_scale_i_flag = cfg.get("skip_connection_scale_i", False)
    ndd, ndh, ndK = cfg["n_dec_depths"], cfg["n_dec_num_head"], cfg["n_dec_patch_size"]
    cdd, cdh, cdK = cfg["c_dec_depths"], cfg["c_dec_num_head"], cfg["c_dec_patch_size"]

    def ndec(p, s):
        p = unpooling(p, sd, f"{B}._n_dec.dec{s}.up", n_mode, False, (s + 1) if n_scale_i_flag else None)
        for i in range(ndd[s]):
            p = block(p, sd, f"{B}._n_dec.dec{s}.block{i}", ndh[s], i % no, ndK[s], False)
        if trace is not None:
            trace[f"n_dec{s}"] = p.feat.clone()
        return p

    def cdec(p, s):
        p = unpooling(p, sd, f"{B}._c_dec.dec{s}.up", c_mode, c_scale, c_scale_i)
        for i in range(cdd[s]):
            p = block(p, sd, f"{B}._c_dec.dec{s}.block{i}", cdh[s], i % no, cdK[s], with_t)
        return p

    if run_dead:
        c = cdec(c, 1)
    n = ndec(n, 3)
    n = ndec(n, 2)
    if run_dead:
        c = cdec(c, 0)
    n = ndec(n, 1)
    n = ndec(n, 0)
    c_out = linear(c.feat, sd, B + "._c_head") if run_dead else None
    n_out = linear(n.feat, sd, B + "._n_head")
    return c_out, n_out


def inference(cfg, sd, input_dict, draws, T=1000, noise_level=None, run_dead=False, trace=None,
              flash_semantics=True, dm=True):
    """DefaultSegmentorV2.inference (default.py:371-422) with condition=True, dm_input='xt', eval=False.
    dm=True (CDSegNet / PTv3_CNF configs): the c-branch input is N(0,1) noise at t = T-1;
    dm=False (configs/*/Baseline.py): the c-branch sees the conditioning target itself at t = 0, no draw
    (default.py:391-394).  input_dict: coord, grid_coord, offset, feat (torch/numpy).
    Returns seg_logits (N, num_classes) fp32."""
    feat = torch.as_tensor(input_dict["feat"], dtype=torch.float32)
    coord = torch.as_tensor(input_dict["coord"], dtype=torch.float32)
    grid = np.asarray(input_dict["grid_coord"], dtype=np.int64)
    offset = np.asarray(input_dict["offset"], dtype=np.int64)
    if noise_level is not None:  # default.py:373-374 (perturbs feat)
        feat = feat + noise_level * draws["feat_noise"]
    c_in_ch = cfg.get("c_in_channels", 6)
    target = feat if c_in_ch == feat.shape[-1] else coord  # default.py:386-389
    N = len(target)
    if dm:
        noise = draws["noise"]
        assert tuple(noise.shape) == tuple(target.shape)
        t = T - 1
    else:
        noise, t = target, 0
    ts = t * torch.ones((N, 1), dtype=torch.int64)
    T_dim = cfg.get("T_dim", 128)
    c_in = dict(coord=coord, grid=grid, offset=offset, feat=noise)
    if T_dim != -1:
        c_in["t_emb"] = calc_t_emb(ts, T_dim)
    n_in = dict(coord=coord, grid=grid, offset=offset, feat=feat)
    global FLASH_SEMANTICS
    FLASH_SEMANTICS = flash_semantics
    with torch.no_grad():
        _, n_out = backbone_forward(cfg, sd, c_in, n_in, draws["perms"], run_dead=run_dead, trace=trace)
    return n_out


# ------------------------------------------------------------------ multi-step inference (MSAI / MSFI)
def diffusion_alpha_bar(kind, start, stop, T):
    """default.py:75-189 restricted to the schedules the shipped configs use; returns Alpha_bar fp32 (T,)."""
    if kind == "linear":
        scale = 1000 / T
        beta = torch.linspace(scale * start, scale * stop, T, dtype=torch.float64)
    elif kind == "cosine":
        s = 0.008
        t = torch.linspace(start, stop, T + 1, dtype=torch.float64) / T
        ac = torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** 2
        ac = ac / ac[0]
        beta = torch.clip(1 - ac[1:] / ac[:-1], 0, 0.999)
    else:
        raise NotImplementedError(kind)
    alpha = 1 - beta
    ab = alpha + 0
    for t in range(1, T):
        ab[t] *= ab[t - 1]
    return ab.float()


def ddim_sample(alpha_bar, x_t, t, eps, dm_target="noise"):
    """default.py:192-214 (t uniform over points)."""
    ab = alpha_bar[t]
    if dm_target == "noise":
        x0 = (x_t - torch.sqrt(1 - ab) * eps) / torch.sqrt(ab)
    else:
        x0 = eps
        eps = (x_t - torch.sqrt(ab) * x0) / torch.sqrt(1 - ab)
    if t == 0:
        return x0
    ab1 = alpha_bar[t - 1]
    return torch.sqrt(ab1) * x0 + torch.sqrt(1 - ab1) * eps


def inference_ddim(cfg, model_cfg, sd, input_dict, draws, step=1, mode="avg", noise_level=None, flash_semantics=True):
    """DefaultSegmentorV2.inference_ddim (default.py:278-369), condition=True.  draws: noise (N,c_in) and
    8*(step+1) perms in consumption order.  model_cfg: the DefaultSegmentorV2 kwargs (T, schedule, ...)."""
    global FLASH_SEMANTICS
    FLASH_SEMANTICS = flash_semantics
    T = model_cfg["T"]
    feat = torch.as_tensor(input_dict["feat"], dtype=torch.float32)
    coord = torch.as_tensor(input_dict["coord"], dtype=torch.float32)
    grid = np.asarray(input_dict["grid_coord"], dtype=np.int64)
    offset = np.asarray(input_dict["offset"], dtype=np.int64)
    if noise_level is not None:
        feat = feat + noise_level * draws["feat_noise"]
    N = len(feat)
    ab = diffusion_alpha_bar(model_cfg["noise_schedule"], model_cfg["beta_start"], model_cfg["beta_end"], T)
    c_xt = draws["noise"]
    n_pred = torch.zeros(N, cfg["num_classes"])
    schedule = np.linspace(-1, T - 1, num=step + 1, dtype=int)[::-1]  # default.py:224-226
    perms = list(draws["perms"])
    T_dim = cfg.get("T_dim", 128)
    with torch.no_grad():
        for k, t in enumerate(schedule):
            t = int(t)
            ts = t * torch.ones((N, 1), dtype=torch.int64)
            c_in = dict(coord=coord, grid=grid, offset=offset, feat=c_xt)
            if T_dim != -1:
                c_in["t_emb"] = calc_t_emb(ts, T_dim)
            n_in = dict(coord=coord, grid=grid, offset=offset, feat=feat)
            c_out, n_out = backbone_forward(cfg, sd, c_in, n_in, perms[8 * k:8 * k + 8], run_dead=True)
            c_xt = ddim_sample(ab, c_xt, t, c_out, model_cfg.get("dm_target", "noise")).float()
            if mode == "avg":
                n_pred = n_pred + n_out
            else:
                n_pred = n_out
            if t <= 0:
                break
    return n_pred / len(schedule) if mode == "avg" else n_pred


def inference_ptv3(cfg, sd, input_dict, perms, flash_semantics=True):
    """condition=False: plain PTv3 through DefaultSegmentorV2.inference (default.py:409-412,
    ptv3.py:1818-1845).  perms: 5 randperm draws (serialization + 4 poolings)."""
    global FLASH_SEMANTICS
    FLASH_SEMANTICS = flash_semantics
    B = "backbone"
    orders = cfg.get("order", DEFAULT_ORDERS)
    no = len(orders)
    pi = iter(list(perms) if cfg.get("shuffle_orders", True) else [None] * 5)
    feat = torch.as_tensor(input_dict["feat"], dtype=torch.float32)
    coord = torch.as_tensor(input_dict["coord"], dtype=torch.float32)
    n = make_point(coord, np.asarray(input_dict["grid_coord"], dtype=np.int64),
                   np.asarray(input_dict["offset"], dtype=np.int64), feat)
    with torch.no_grad():
        serialize_point(n, orders, next(pi))
        n = embedding(n, sd, B + "._n_embedding")
        nd, nh, nK, ns = cfg["n_enc_depths"], cfg["n_enc_num_head"], cfg["n_enc_patch_size"], cfg["n_stride"]
        for s in range(len(nd)):
            n = _stage(n, sd, B + "._n_enc", s, nd[s], nh[s], nK[s], ns[s - 1] if s else None,
                       next(pi) if s else None, False, no)
        ndd, ndh, ndK = cfg["n_dec_depths"], cfg["n_dec_num_head"], cfg["n_dec_patch_size"]
        n_mode = "cat" if cfg.get("skip_connection_mode", "add") == "cat_all" else "add"
        for s in reversed(range(len(nd) - 1)):
            n = unpooling(n, sd, f"{B}._n_dec.dec{s}.up", n_mode, False,
                          (s + 1) if cfg.get("skip_connection_scale_i", False) else None)
            for i in range(ndd[s]):
                n = block(n, sd, f"{B}._n_dec.dec{s}.block{i}", ndh[s], i % no, ndK[s], False)
        return linear(n.feat, sd, B + "._n_head")
