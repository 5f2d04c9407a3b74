"""Training path, first slice: forward with saved activations and backward of ONE Block's attention + MLP tail, exact
fp32, on the HIP kernels (csrc/train.hip + the inference GEMM).

ref: pointcept/models/default.py:424-493 (DefaultSegmentorV2.forward: q_sample, backbone, GLS loss) and
     pointcept/engines/train.py:216-271 (run_step: loss.backward(), optimizer step) - what autograd does for
     point_transformer_v3m1_base.py:399-428 (Block) and :246-296 (SerializedAttention).

    x1 = x0 + proj(attn(qkv(LN1(x0))))          x0: the residual stream behind the CPE
    y  = x1 + fc2(GELU(fc1(LN2(x1))))

`block_tail_backward` returns d y / d qkv contracted with an upstream gradient (and the gradients of every tensor on
the way).  Scope of the slice: data gradients of the Block tail; weight gradients, the CPE conv, pooling,
the loss and the optimizer are the next steps of SURVEY 8(f4)'s training row.
"""
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


def block_tail_backward(w, pre, t, dy):
    """Backward of `block_tail_forward`: dy (N, C) is the gradient of y.  Returns dict(d_qkv, d_o, d_x1, d_u)."""
    n, c = t.x0.shape
    dev = dy.device
    ops.bind_stream()
    try:
        # y = x1 + fc2(GELU(u)):  d g = dy W2 ; d u = d g * GELU'(u) ; d h2 = d u W1
        wt = lambda k: w[k].t().contiguous()  # noqa: E731 - dX = dY W is the inference GEMM on the transposed weight
        dg = _lin(dy, wt(pre + ".fc2.w"), None)
        du = ops.gelu_bwd(t.u, dg)
        dh2 = _lin(du, wt(pre + ".fc1.w"), None)
        # x1 feeds the residual and LN2:  d x1 = dy + LN2'(x1)^T d h2
        dx1 = dy.clone()
        ops.layernorm_bwd(t.x1, w[pre + ".norm2.g"], dh2, dx1, accumulate=True)
        # x1 = x0 + proj(o):  d o = d x1 Wp
        do = _lin(dx1, wt(pre + ".proj.w"), None)
        # attention core: gradients land at the gathered qkv rows
        dqkv = torch.zeros((n, 3 * c), dtype=torch.float32, device=dev)
        ops.attention_bwd(t.qkv[:, :c], t.qkv[:, c:2 * c], t.qkv[:, 2 * c:], t.gidx, t.gidx, t.widx, t.patch_start,
                          t.patch_start_host, t.num_heads, t.scale, do, dqkv[:, :c], dqkv[:, c:2 * c], dqkv[:, 2 * c:])
    finally:
        ops.unbind_stream()
    return dict(d_qkv=dqkv, d_o=do, d_x1=dx1, d_u=du)
