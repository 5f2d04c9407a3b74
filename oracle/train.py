"""TEST INFRASTRUCTURE ONLY - CPU checker of the training path's first slice (the product never imports oracle/).

Restates the tail of a PTv3 Block on PyTorch-CPU fp32 with the same functions as oracle/model.py and lets torch's
autograd differentiate it - what the reference's training step does for point_transformer_v3m1_base.py:399-428 / :246-296
(pointcept/engines/train.py:216-271: loss.backward()).  Pinned by tests/golden/train_block_tail.npz, captured from the
reference's own Block under autograd (oracle/make_golden.py train)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import model as OM


def block_tail(sd, pre, x0, order, inverse, cu, H):
    """x0 (N, C): residual stream behind the CPE.  order (N') / inverse (N) / cu (P + 1): the padded patch plan of the
    Block's curve (ptv3.py:188-244, 255-262).  Returns (y, qkv) with qkv kept in the autograd graph."""
    C = x0.shape[1]
    scale = (C // H) ** -0.5
    qkv = OM.linear(OM.layernorm(x0, sd, pre + ".norm1.0"), sd, pre + ".attn.qkv")
    if qkv.requires_grad:
        qkv.retain_grad()
    L3 = qkv[torch.as_tensor(order, dtype=torch.long)].reshape(-1, 3, C)
    feat = OM._patch_attention(L3[:, 0], L3[:, 1], L3[:, 2], np.asarray(cu), H, scale)
    x1 = x0 + OM.linear(feat[torch.as_tensor(inverse, dtype=torch.long)], sd, pre + ".attn.proj")
    h = OM.layernorm(x1, sd, pre + ".norm2.0")
    y = x1 + OM.linear(F.gelu(OM.linear(h, sd, pre + ".mlp.0.fc1")), sd, pre + ".mlp.0.fc2")
    return y, qkv


def block_tail_qkv_grad(sd, pre, x0, order, inverse, cu, H, dy):
    """(y, d<y, dy>/d qkv) by autograd on the restatement above."""
    x0 = torch.as_tensor(x0, dtype=torch.float32).clone().requires_grad_(True)
    sd = {k: torch.as_tensor(v) for k, v in sd.items()}
    y, qkv = block_tail(sd, pre, x0, order, inverse, cu, H)
    (y * torch.as_tensor(dy, dtype=torch.float32)).sum().backward()
    return y.detach(), qkv.grad.detach()


def block_full_grads(sd, pre, x_in, nbr, order, inverse, cu, H, dy):
    """The WHOLE Block (ptv3.py:399-428 in eval mode: CPE conv -> Linear -> LayerNorm, attention, MLP) under autograd:
    returns (y, d x_in, {parameter key: gradient}) for <y, dy>.  nbr: (N, 27) kernel map (oracle.model.subm_neighbors).
    The sparse conv is the oracle's restatement (spconv is not importable on CPU: parity of its arithmetic is unpinned,
    DESIGN.md 2) - everything behind it is pinned by the reference fixture of the tail."""
    x_in = torch.as_tensor(x_in, dtype=torch.float32).clone().requires_grad_(True)
    sd = {k: torch.as_tensor(v).clone().requires_grad_(True) for k, v in sd.items()}
    w = sd[pre + ".cpe.0.weight"]
    yc = OM.subm_conv3d(x_in, np.asarray(nbr), w, sd.get(pre + ".cpe.0.bias"))
    x0 = x_in + OM.layernorm(OM.linear(yc, sd, pre + ".cpe.1"), sd, pre + ".cpe.2")
    y, _ = block_tail(sd, pre, x0, order, inverse, cu, H)
    (y * torch.as_tensor(dy, dtype=torch.float32)).sum().backward()
    return y.detach(), x_in.grad.detach(), {k: v.grad.detach() for k, v in sd.items() if v.grad is not None}
