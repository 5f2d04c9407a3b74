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


# ------------------------------------------------------------------------------------------ training forward + loss
class _TrainCtx:
    def __init__(self, masks):
        self.masks = {k: [torch.as_tensor(m) for m in v] for k, v in masks.items()}


def lovasz_softmax(logits, labels, ignore=-1):
    """Multi-class Lovasz-Softmax over the classes present (restates pointcept/models/losses/lovasz.py:22-34, 118-165,
    244-262 with mode='multiclass', per_image=False, class_seen=None)."""
    prob = logits.softmax(dim=1)
    valid = labels != ignore
    prob, lab = prob[valid], labels[valid]
    if prob.numel() == 0:
        return prob.sum() * 0.0
    terms = []
    for c in torch.unique(lab):
        fg = (lab == c).to(prob.dtype)
        err = (fg - prob[:, c]).abs()
        err_s, order = torch.sort(err, 0, descending=True)
        fg_s = fg[order]
        total = fg_s.sum()
        jac = 1.0 - (total - fg_s.cumsum(0)) / (total + (1.0 - fg_s).cumsum(0))
        grad = torch.cat([jac[:1], jac[1:] - jac[:-1]])
        terms.append(torch.dot(err_s, grad))
    return torch.stack(terms).mean()


def gls_loss(c_pred, c_target, n_pred, n_target, ignore=-1):
    """The shipped criteria (configs/scannet/CDSegNet.py:117-123): MSE on the noise branch over the labelled points
    (losses/misc.py:24-93, batch_sample_point = -1), cross entropy + Lovasz on the logits, combined by
    losses/builder.py:36-52 with loss_type='GLS', task_num=2: sqrt(MSE * (CE + Lovasz)).  Returns (loss, parts)."""
    valid = n_target != ignore
    mse = ((c_pred[valid] - c_target[valid]) ** 2).mean()
    ce = F.cross_entropy(n_pred[valid], n_target[valid])
    lov = lovasz_softmax(n_pred, n_target, ignore)
    return torch.pow(mse * (ce + lov), 0.5), (mse, ce, lov)


def training_forward(cfg, sd, input_dict, draws, T=1000, alpha_bar=None):
    """DefaultSegmentorV2.forward in train mode (default.py:424-493, condition=True, dm=True, dm_input='xt',
    dm_target='noise'): per-scene timestep, q_sample of the conditioning target, BOTH decoders, batch-statistics
    BatchNorm, the recorded DropPath masks; then the GLS loss.  draws: ts (B, 1) int64, noise (N, c_in), perms (8 x 4),
    masks {DropPath module name: [(N_level, 1) mask, ...]}.  sd: parameters as tensors (requires_grad allowed).
    Returns dict(loss, mse, ce, lovasz, n_pred, c_pred, c_target)."""
    feat = torch.as_tensor(input_dict["feat"], dtype=torch.float32)
    coord = torch.as_tensor(input_dict["coord"], dtype=torch.float32)
    grid = np.asarray(input_dict["grid_coord"], dtype=np.int64)
    offset = np.asarray(input_dict["offset"], dtype=np.int64)
    seg = torch.as_tensor(input_dict["segment"], dtype=torch.int64)
    bcfg = cfg["backbone"]
    c_in_ch = cfg.get("c_in_channels", 6)
    x0 = feat if c_in_ch == feat.shape[-1] else coord
    batch = torch.as_tensor(np.repeat(np.arange(len(offset)), np.diff(np.concatenate([[0], offset]))))
    ts = torch.as_tensor(draws["ts"], dtype=torch.int64)[batch]  # (N, 1)
    noise = torch.as_tensor(draws["noise"], dtype=torch.float32)
    ab = torch.as_tensor(alpha_bar, dtype=torch.float32) if alpha_bar is not None else OM.diffusion_alpha_bar(
        cfg["noise_schedule"], cfg["beta_start"], cfg["beta_end"], T)
    a = ab[ts]  # (N, 1)
    c_xt = torch.sqrt(a) * x0 + torch.sqrt(1 - a) * noise  # default.py:216-222
    c_in = dict(coord=coord, grid=grid, offset=offset, feat=c_xt)
    T_dim = bcfg.get("T_dim", 128)
    if T_dim != -1:
        c_in["t_emb"] = OM.calc_t_emb(ts, T_dim)
    n_in = dict(coord=coord, grid=grid, offset=offset, feat=feat)
    OM.FLASH_SEMANTICS = False
    OM.TRAIN = _TrainCtx(draws["masks"])
    try:
        c_out, n_out = OM.backbone_forward(bcfg, sd, c_in, n_in, draws["perms"], run_dead=True)
    finally:
        OM.TRAIN = None
    loss, (mse, ce, lov) = gls_loss(c_out, noise, n_out, seg)
    return dict(loss=loss, mse=mse, ce=ce, lovasz=lov, n_pred=n_out, c_pred=c_out, c_target=noise)


def adamw_first_step(p, g, lr, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.AdamW's FIRST step (decoupled weight decay, bias-corrected moments): what engines/train.py:216-271 applies
    with the shipped optimizer (configs/scannet/CDSegNet.py:143: lr 2e-3, weight decay 0.05; parameter group "block": 2e-4)."""
    p = p * (1.0 - lr * weight_decay)
    m_hat = ((1 - betas[0]) * g) / (1 - betas[0])
    v_hat = ((1 - betas[1]) * g * g) / (1 - betas[1])
    return p - lr * m_hat / (v_hat.sqrt() + eps)
