"""GPU: training path, first slice (SURVEY 8(f4) tail / VERDICT r2 item 9) - forward with saved activations and
backward of one Block's attention + MLP tail in exact fp32 on the HIP kernels, against the REFERENCE's autograd
(tests/golden/train_block_tail.npz, oracle/make_golden.py train) and against the CPU oracle on further shapes.
Tolerance: 1e-3 (north_star's fp32 bound), measured ~1e-6."""
import numpy as np
import pytest
import torch

from cdsegnet_amd import ops, train
from oracle import train as OT
from tests.helpers import load_fixture

pytestmark = pytest.mark.gpu


def _weights(sd, pre, dev):
    def f(k):
        return torch.as_tensor(np.asarray(sd[pre + k]), dtype=torch.float32).to(dev).contiguous()
    return {"B.norm1.g": f(".norm1.0.weight"), "B.norm1.b": f(".norm1.0.bias"), "B.qkv.w": f(".attn.qkv.weight"),
            "B.qkv.b": f(".attn.qkv.bias"), "B.proj.w": f(".attn.proj.weight"), "B.proj.b": f(".attn.proj.bias"),
            "B.norm2.g": f(".norm2.0.weight"), "B.norm2.b": f(".norm2.0.bias"), "B.fc1.w": f(".mlp.0.fc1.weight"),
            "B.fc1.b": f(".mlp.0.fc1.bias"), "B.fc2.w": f(".mlp.0.fc2.weight"), "B.fc2.b": f(".mlp.0.fc2.bias")}


def _slot_plan(order, inverse, dev):
    """The library's slot plan from the reference's (order, inverse): slot s reads row order[s]; it writes row r iff it
    is the slot the inverse map points at (the padding duplicates have no output row)."""
    gidx = torch.as_tensor(order, dtype=torch.int32).to(dev)
    w = np.full(len(order), -1, dtype=np.int32)
    w[np.asarray(inverse)] = np.arange(len(inverse), dtype=np.int32)
    return gidx, torch.as_tensor(w).to(dev)


def _run(sd, pre, x0, order, inverse, cu, H, dy):
    dev = torch.device("cuda")
    w = _weights(sd, pre, dev)
    gidx, widx = _slot_plan(order, inverse, dev)
    ps = torch.as_tensor(np.asarray(cu), dtype=torch.int32).to(dev)
    C = x0.shape[1]
    t = train.block_tail_forward(w, "B", torch.as_tensor(x0, dtype=torch.float32).to(dev).contiguous(), gidx, widx, ps,
                                 [int(v) for v in cu], H, int(np.diff(cu).max()), (C // H) ** -0.5)
    g = train.block_tail_backward(w, "B", t, torch.as_tensor(dy, dtype=torch.float32).to(dev).contiguous())
    torch.cuda.synchronize()
    return t.y.cpu().numpy(), g["d_qkv"].cpu().numpy()


def test_block_tail_backward_matches_the_reference_autograd():
    fx = load_fixture("train_block_tail.npz")
    pre = str(fx["prefix"])
    sd = {k[3:]: fx[k] for k in fx.files if k.startswith("sd.")}
    y, dqkv = _run(sd, pre, fx["x0"], fx["order"], fx["inverse"], fx["cu"], int(fx["num_heads"]), fx["dy"])
    ey = float(np.abs(y - fx["y"]).max())
    eg = float(np.abs(dqkv - fx["d_qkv"]).max())
    print(f"[measure] train slice vs reference autograd: forward max_abs_err={ey:.3e}, d_qkv max_abs_err={eg:.3e} "
          f"(|d_qkv| mean {float(np.abs(fx['d_qkv']).mean()):.3e}, max {float(np.abs(fx['d_qkv']).max()):.3e})")
    assert ey < 1e-3 and eg < 1e-3
    assert eg < 1e-4 * max(1.0, float(np.abs(fx["d_qkv"]).max()))  # in fact at fp32 round-off


@pytest.mark.parametrize("lens,H", [([700], 2), ([1024, 1024, 300], 4), ([64, 1, 130], 8)])
def test_block_tail_backward_vs_oracle(lens, H):
    """Ragged patches (incl. a 1-slot patch), more heads, slots that repeat rows (padding duplicates)."""
    rng = np.random.default_rng(sum(lens) + H)
    C = 16 * H
    n = sum(lens) - 37 if sum(lens) > 200 else sum(lens)  # fewer rows than slots: the last rows are padded in twice
    order = np.concatenate([rng.permutation(n), rng.integers(0, n, sum(lens) - n)]).astype(np.int64)
    inverse = np.zeros(n, dtype=np.int64)
    for s in range(n):
        inverse[order[s]] = s  # primary slot of every row (a permutation on the first n slots)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pre = "blk"
    sd = {}
    for k, shape in ((".norm1.0.weight", (C,)), (".norm1.0.bias", (C,)), (".attn.qkv.weight", (3 * C, C)), (".attn.qkv.bias", (3 * C,)),
                     (".attn.proj.weight", (C, C)), (".attn.proj.bias", (C,)), (".norm2.0.weight", (C,)), (".norm2.0.bias", (C,)),
                     (".mlp.0.fc1.weight", (4 * C, C)), (".mlp.0.fc1.bias", (4 * C,)), (".mlp.0.fc2.weight", (C, 4 * C)),
                     (".mlp.0.fc2.bias", (C,))):
        sd[pre + k] = (rng.standard_normal(shape) * (0.3 if len(shape) == 2 else 0.1) + (1.0 if k.endswith("0.weight") and len(shape) == 1 else 0.0)).astype(np.float32)
    x0 = rng.standard_normal((n, C)).astype(np.float32)
    dy = rng.standard_normal((n, C)).astype(np.float32)
    ry, rg = OT.block_tail_qkv_grad(sd, pre, x0, order, inverse, cu, H, dy)
    y, dqkv = _run(sd, pre, x0, order, inverse, cu, H, dy)
    ey, eg = float(np.abs(y - ry.numpy()).max()), float(np.abs(dqkv - rg.numpy()).max())
    print(f"[measure] train slice vs oracle lens={lens} H={H}: forward {ey:.3e}, d_qkv {eg:.3e} (|d_qkv| max {float(rg.abs().max()):.3e})")
    assert ey < 1e-3 and eg < 1e-3 * max(1.0, float(rg.abs().max()))


def test_layernorm_and_gelu_backward_kernels():
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(777, 96, generator=g)
    gamma, beta = torch.randn(96, generator=g), torch.randn(96, generator=g)
    dy = torch.randn(777, 96, generator=g)
    xr = x.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (96,), gr, br, 1e-5).backward(dy)
    dx = torch.zeros(777, 96, device=dev)
    dg, db = torch.zeros(96, device=dev), torch.zeros(96, device=dev)
    ops.bind_stream()
    try:
        ops.layernorm_bwd(x.to(dev), gamma.to(dev), dy.to(dev), dx, dgamma=dg, dbeta=db)
        u = torch.randn(5000, generator=g) * 2
        du = ops.gelu_bwd(u.to(dev), torch.ones(5000, device=dev))
    finally:
        ops.unbind_stream()
    assert float((dx.cpu() - xr.grad).abs().max()) < 1e-4
    assert float((dg.cpu() - gr.grad).abs().max()) < 1e-3 and float((db.cpu() - br.grad).abs().max()) < 1e-3
    ur = u.clone().requires_grad_(True)
    torch.nn.functional.gelu(ur).sum().backward()
    assert float((du.cpu() - ur.grad).abs().max()) < 1e-5
