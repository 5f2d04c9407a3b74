"""Training path, first slice: forward with saved activations and backward of ONE Block's attention + MLP tail, exact
fp32, on the HIP kernels (csrc/train.hip + the inference GEMM).

ref: pointcept/models/default.py:424-493 (DefaultSegmentorV2.forward: q_sample, backbone, GLS loss) and
     pointcept/engines/train.py:216-271 (run_step: loss.backward(), optimizer step) - what autograd does for
     point_transformer_v3m1_base.py:399-428 (Block) and :246-296 (SerializedAttention).

    x1 = x0 + proj(attn(qkv(LN1(x0))))          x0: the residual stream behind the CPE
    y  = x1 + fc2(GELU(fc1(LN2(x1))))

`block_tail_backward` returns d y / d qkv contracted with an upstream gradient (and the gradients of every tensor on
the way).  Second slice: `block_forward` / `block_backward` - the WHOLE Block (CPE conv -> Linear -> LayerNorm in front
of the tail) with the gradient of its input and of every parameter (weights, biases, LayerNorm affine, the 27-offset
conv kernel).  Scope: one Block in eval mode (DropPath = identity); pooling / unpooling, the loss and the
optimizer are the next steps of SURVEY 8(f4)'s training row; gradient all-reduce: cdsegnet_amd.dist.GradBucketer.
"""
import weakref

import torch

from . import ops


class BlockTape:
    """Activations a Block's backward needs (fp32)."""
    __slots__ = ("x0", "qkv", "o", "x1", "u", "y", "gidx", "widx", "patch_start", "patch_start_host", "num_heads", "scale")


def _lin(x, w, b, out=None, **kw):
    out = torch.empty((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device) if out is None else out
    ops.gemm(x, w, out, bias=b, **kw)
    return out


def block_tail_forward(w, pre, x0, gidx, widx, patch_start, patch_start_host, num_heads, max_len, scale):
    """Forward of the Block tail on fp32 weights `w` (Engine.w of a precision='fp32' engine, prefix `pre`), keeping
    what the backward reads.  gidx / widx / patch_start: the slot plan of the Block's curve (Level.slots / Level.pad)."""
    t = BlockTape()
    n, c = x0.shape
    ops.bind_stream()
    try:
        h1 = torch.empty_like(x0)
        ops.layernorm(x0, w[pre + ".norm1.g"], w[pre + ".norm1.b"], h1)
        t.qkv = _lin(h1, w[pre + ".qkv.w"], w[pre + ".qkv.b"])
        t.o = torch.empty_like(x0)
        ops.attention(t.qkv[:, :c], t.qkv[:, c:2 * c], t.qkv[:, 2 * c:], gidx, gidx, widx, patch_start, num_heads, max_len,
                      scale, t.o)
        t.x1 = x0.clone()
        ops.gemm(t.o, w[pre + ".proj.w"], t.x1, bias=w[pre + ".proj.b"], res=t.x1)
        h2 = torch.empty_like(x0)
        ops.layernorm(t.x1, w[pre + ".norm2.g"], w[pre + ".norm2.b"], h2)
        t.u = _lin(h2, w[pre + ".fc1.w"], w[pre + ".fc1.b"])  # pre-activation (the inference path fuses the GELU)
        g = _lin(h2, w[pre + ".fc1.w"], w[pre + ".fc1.b"], act=ops.ACT_GELU)  # (recomputed: the GEMM epilogue fuses the GELU)
        t.y = t.x1.clone()
        ops.gemm(g, w[pre + ".fc2.w"], t.y, bias=w[pre + ".fc2.b"], res=t.y)
    finally:
        ops.unbind_stream()
    t.x0, t.gidx, t.widx, t.patch_start, t.patch_start_host = x0, gidx, widx, patch_start, list(patch_start_host)
    t.num_heads, t.scale = num_heads, scale
    return t


_DERIVED = {}  # id(weight tensor) -> {kind: (version, data_ptr, derived tensor)}; dropped when the weight is collected


def _derived(src, kind):
    """Transposed ("T": dX = dY W is the inference GEMM on W^T) or mirrored-transposed ("convT": the conv's data gradient,
    `_conv_bwd_weight`) copy of a weight, built ONCE per weight and reused by every backward call until the weight
    changes (an optimizer step bumps `Tensor._version`).  Rebuilding them per call handed `ops.gemm` a fresh tensor as W
    every time: its argument cache is keyed by W's address and keeps W alive, so every Block backward pinned up to six
    weight-sized temporaries."""
    ent = _DERIVED.get(id(src))
    if ent is None:
        ent = _DERIVED[id(src)] = {}
        weakref.finalize(src, _DERIVED.pop, id(src), None)
    hit = ent.get(kind)
    if hit is not None and hit[0] == src._version and hit[1] == src.data_ptr():
        return hit[2]
    if kind == "T":
        t = src.t().contiguous()
    else:
        t = _conv_bwd_weight(src, src.shape[0], src.shape[1] // 27)
    ent[kind] = (src._version, src.data_ptr(), t)
    return t


def _wgrad(w, grads, key, x, dyy):
    dw = torch.zeros_like(w[key + ".w"])
    db = torch.zeros_like(w[key + ".b"]) if w.get(key + ".b") is not None else None
    ops.linear_wgrad(x, dyy, dw, db)
    grads[key + ".w"] = dw
    if db is not None:
        grads[key + ".b"] = db


def _ln_bwd(w, grads, key, x, dyy, dx, accumulate):
    if grads is None:
        return ops.layernorm_bwd(x, w[key + ".g"], dyy, dx, accumulate=accumulate)
    c = x.shape[1]
    dg = torch.zeros(c, dtype=torch.float32, device=x.device)
    db = torch.zeros(c, dtype=torch.float32, device=x.device)
    ops.layernorm_bwd(x, w[key + ".g"], dyy, dx, accumulate=accumulate, dgamma=dg, dbeta=db)
    grads[key + ".g"], grads[key + ".b"] = dg, db
    return dx


def _tail_backward(w, pre, t, dy, grads):
    """Shared by both slices (caller holds the stream binding).  grads: None, or a dict that receives the gradients of
    the tail's parameters (Linears: dW = dY^T X by cdseg_linear_wgrad, LayerNorms: d gamma / d beta); activations the
    inference kernels fuse away (LN outputs, GELU output) are recomputed."""
    n, c = t.x0.shape
    wt = lambda k: _derived(w[k], "T")  # noqa: E731 - dX = dY W is the inference GEMM on the transposed weight
    # y = x1 + fc2(GELU(u)):  d g = dy W2 ; d u = d g * GELU'(u) ; d h2 = d u W1
    if grads is not None:
        h2 = torch.empty_like(t.x1)
        ops.layernorm(t.x1, w[pre + ".norm2.g"], w[pre + ".norm2.b"], h2)
        _wgrad(w, grads, pre + ".fc2", _lin(h2, w[pre + ".fc1.w"], w[pre + ".fc1.b"], act=ops.ACT_GELU), dy)
    dg = _lin(dy, wt(pre + ".fc2.w"), None)
    du = ops.gelu_bwd(t.u, dg)
    if grads is not None:
        _wgrad(w, grads, pre + ".fc1", h2, du)
    dh2 = _lin(du, wt(pre + ".fc1.w"), None)
    # x1 feeds the residual and LN2:  d x1 = dy + LN2'(x1)^T d h2
    dx1 = dy.clone()
    _ln_bwd(w, grads, pre + ".norm2", t.x1, dh2, dx1, True)
    # x1 = x0 + proj(o):  d o = d x1 Wp
    if grads is not None:
        _wgrad(w, grads, pre + ".proj", t.o, dx1)
    do = _lin(dx1, wt(pre + ".proj.w"), None)
    # attention core: gradients land at the gathered qkv rows
    dqkv = torch.zeros((n, 3 * c), dtype=torch.float32, device=dy.device)
    ops.attention_bwd(t.qkv[:, :c], t.qkv[:, c:2 * c], t.qkv[:, 2 * c:], t.gidx, t.gidx, t.widx, t.patch_start,
                      t.patch_start_host, t.num_heads, t.scale, do, dqkv[:, :c], dqkv[:, c:2 * c], dqkv[:, 2 * c:])
    dx0 = None
    if grads is not None:  # x0 feeds the residual and LN1 -> qkv
        h1 = torch.empty_like(t.x0)
        ops.layernorm(t.x0, w[pre + ".norm1.g"], w[pre + ".norm1.b"], h1)
        _wgrad(w, grads, pre + ".qkv", h1, dqkv)
        dh1 = _lin(dqkv, wt(pre + ".qkv.w"), None)
        dx0 = dx1.clone()
        _ln_bwd(w, grads, pre + ".norm1", t.x0, dh1, dx0, True)
    return dict(d_qkv=dqkv, d_o=do, d_x1=dx1, d_u=du, d_x0=dx0)


def block_tail_backward(w, pre, t, dy, param_grads=False):
    """Backward of `block_tail_forward`: dy (N, C) is the gradient of y.  Returns dict(d_qkv, d_o, d_x1, d_u) and, with
    param_grads, also d_x0 and grads = {parameter name: gradient} of the tail's twelve parameter tensors."""
    ops.bind_stream()
    try:
        grads = {} if param_grads else None
        out = _tail_backward(w, pre, t, dy, grads)
    finally:
        ops.unbind_stream()
    if param_grads:
        out["grads"] = grads
    return out


# ------------------------------------------------------------------------------------------ whole Block
def _conv_bwd_weight(w_conv, cout, cin):
    """Kernel of the conv's DATA gradient: dx = conv(dy, W') on the SAME kernel map with W'[ci][o][co] = W[co][26-o][ci]
    (submanifold map: nbr[o][i] = j  <=>  nbr[26-o][j] = i)."""
    return w_conv.view(cout, 27, cin).flip(1).permute(2, 1, 0).contiguous().view(cin, 27 * cout)


def block_forward(w, pre, x_in, nbr_kmajor, gidx, widx, patch_start, patch_start_host, num_heads, max_len, scale, x_conv=None):
    """x0 = x_in + LN(Linear(conv3x3x3(x_conv or x_in))) (ref: ptv3.py:399-406, modules.py:63-66), then the tail.
    nbr_kmajor: (27, N) int32 offset-major kernel map (-1 = no neighbour).  Returns the tape."""
    n, c = x_in.shape
    xc = x_in if x_conv is None else x_conv
    ops.bind_stream()
    try:
        yc = torch.empty_like(x_in)
        ops.gemm(xc, w[pre + ".cpe0.w"], yc, bias=w[pre + ".cpe0.b"], nbr=nbr_kmajor, kvol=27, nbr_kmajor=True)
        z = _lin(yc, w[pre + ".cpe1.w"], w[pre + ".cpe1.b"])
        x0 = x_in.clone()
        ops.layernorm(z, w[pre + ".cpe2.g"], w[pre + ".cpe2.b"], x0, res=x_in)
    finally:
        ops.unbind_stream()
    t = block_tail_forward(w, pre, x0, gidx, widx, patch_start, patch_start_host, num_heads, max_len, scale)
    t_full = {"tail": t, "x_in": x_in, "x_conv": xc, "yc": yc, "z": z, "nbr": nbr_kmajor}
    return t_full


def block_backward(w, pre, tape, dy):
    """Backward of `block_forward`.  Returns (d_x_in, d_x_conv or None, grads) with grads[name] for every parameter of the
    Block under the engine's names (pre + '.cpe0.w' (Cout, 27 * Cin), '.cpe1.w', '.cpe2.g', '.norm1.g', '.qkv.w', ...)."""
    t = tape["tail"]
    n, c = t.x0.shape
    dev = dy.device
    f32 = dict(dtype=torch.float32, device=dev)
    grads = {}
    ops.bind_stream()
    try:
        wt = lambda k: _derived(w[k], "T")  # noqa: E731
        dx0 = _tail_backward(w, pre, t, dy, grads)["d_x0"]
        # ---- CPE: x0 = x_in + LN(z), z = Linear(yc), yc = conv(x_conv)
        dz = torch.empty_like(dx0)
        _ln_bwd(w, grads, pre + ".cpe2", tape["z"], dx0, dz, False)
        _wgrad(w, grads, pre + ".cpe1", tape["yc"], dz)
        dyc = _lin(dz, wt(pre + ".cpe1.w"), None)
        wc = w[pre + ".cpe0.w"]
        cout, cin = wc.shape[0], wc.shape[1] // 27
        dwc = torch.zeros_like(wc)
        dbc = torch.zeros_like(w[pre + ".cpe0.b"]) if w.get(pre + ".cpe0.b") is not None else None
        dw3 = dwc.view(cout, 27, cin)
        nbr = tape["nbr"]
        ops.conv_wgrad(tape["x_conv"], nbr, dyc, dw3, dbc)  # 27 gathered dY^T X in one launch
        grads[pre + ".cpe0.w"] = dwc
        if dbc is not None:
            grads[pre + ".cpe0.b"] = dbc
        dxc = torch.empty((n, cin), **f32)
        ops.gemm(dyc, _derived(wc, "convT"), dxc, nbr=nbr, kvol=27, nbr_kmajor=True)
    finally:
        ops.unbind_stream()
    if tape["x_conv"] is tape["x_in"]:
        return dx0 + dxc, None, grads
    return dx0, dxc, grads
