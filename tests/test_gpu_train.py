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


def test_block_tail_parameter_gradients_match_the_reference_autograd():
    """Second slice: weight / bias / LayerNorm-affine gradients of the tail's twelve parameter tensors against the
    gradients the REFERENCE's autograd left on its own parameters (fixture keys g.*)."""
    fx = load_fixture("train_block_tail.npz")
    pre = str(fx["prefix"])
    sd = {k[3:]: fx[k] for k in fx.files if k.startswith("sd.")}
    dev = torch.device("cuda")
    w = _weights(sd, pre, dev)
    gidx, widx = _slot_plan(fx["order"], fx["inverse"], dev)
    cu = fx["cu"]
    ps = torch.as_tensor(np.asarray(cu), dtype=torch.int32).to(dev)
    H = int(fx["num_heads"])
    C = fx["x0"].shape[1]
    t = train.block_tail_forward(w, "B", torch.as_tensor(fx["x0"], dtype=torch.float32).to(dev).contiguous(), gidx, widx, ps,
                                 [int(v) for v in cu], H, int(np.diff(cu).max()), (C // H) ** -0.5)
    out = train.block_tail_backward(w, "B", t, torch.as_tensor(fx["dy"], dtype=torch.float32).to(dev).contiguous(), param_grads=True)
    torch.cuda.synchronize()
    names = {"B.norm1.g": ".norm1.0.weight", "B.norm1.b": ".norm1.0.bias", "B.qkv.w": ".attn.qkv.weight", "B.qkv.b": ".attn.qkv.bias",
             "B.proj.w": ".attn.proj.weight", "B.proj.b": ".attn.proj.bias", "B.norm2.g": ".norm2.0.weight", "B.norm2.b": ".norm2.0.bias",
             "B.fc1.w": ".mlp.0.fc1.weight", "B.fc1.b": ".mlp.0.fc1.bias", "B.fc2.w": ".mlp.0.fc2.weight", "B.fc2.b": ".mlp.0.fc2.bias"}
    assert set(out["grads"]) == set(names)
    worst = 0.0
    for mine, ref in names.items():
        r = fx["g." + pre + ref]
        e = float(np.abs(out["grads"][mine].cpu().numpy() - r).max()) / max(1.0, float(np.abs(r).max()))
        worst = max(worst, e)
        assert e < 1e-3, (mine, e)
    print(f"[measure] tail parameter gradients vs reference autograd: worst rel err {worst:.3e} over 12 tensors")


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


# ------------------------------------------------------------------------------------------ second slice: the whole Block
def test_linear_wgrad_kernel_plain_and_gathered():
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(11)
    M, K, N = 5003, 48, 96
    x, dy = torch.randn(M, K, generator=g), torch.randn(M, N, generator=g)
    idx = torch.randint(-1, M, (M,), generator=g).to(torch.int32)
    dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    wide = torch.zeros(N, 3 * K, device=dev)  # a strided (N, K) slice of a wider matrix, like one conv offset
    ops.bind_stream()
    try:
        ops.linear_wgrad(x.to(dev), dy.to(dev), dw, db)
        ops.linear_wgrad(x.to(dev), dy.to(dev), wide[:, K:2 * K], None, xidx=idx.to(dev))
    finally:
        ops.unbind_stream()
    ref = dy.double().t() @ x.double()
    assert float((dw.cpu().double() - ref).abs().max()) < 1e-3 * float(ref.abs().max())
    assert float((db.cpu().double() - dy.double().sum(0)).abs().max()) < 1e-3 * float(dy.double().sum(0).abs().max())
    m = idx >= 0
    refg = dy[m].double().t() @ x[idx[m].long()].double()
    assert float((wide[:, K:2 * K].cpu().double() - refg).abs().max()) < 1e-3 * float(refg.abs().max())
    assert float(wide[:, :K].abs().max()) == 0.0 and float(wide[:, 2 * K:].abs().max()) == 0.0


@pytest.mark.parametrize("npts,H", [(900, 2), (2300, 4)])
def test_whole_block_backward_vs_oracle(npts, H):
    """CPE conv + Linear + LayerNorm + attention + MLP of one Block: gradient of the input and of every parameter against
    torch autograd on the oracle restatement (real kernel map of a synthetic scene, real padded patch plan)."""
    from cdsegnet_amd import synth
    from oracle import model as OM
    from oracle import serialization as S
    rng = np.random.default_rng(npts + H)
    sc = synth.room_scene(7, npts)
    grid = np.asarray(sc["grid_coord"], dtype=np.int64)
    n = len(grid)
    C = 16 * H
    nbr = OM.subm_neighbors(grid, np.zeros(n, dtype=np.int64), 3)  # (n, 27)
    K = 1024
    offset = np.array([n])
    pad, unpad, cu = S.padding_plan(offset, K)
    perm = rng.permutation(n)  # any serialization order
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    order, inverse = perm[pad], unpad[inv]
    pre = "blk"
    sd = {}
    for k, shape in ((".cpe.0.weight", (C, 3, 3, 3, C)), (".cpe.0.bias", (C,)), (".cpe.1.weight", (C, C)), (".cpe.1.bias", (C,)),
                     (".cpe.2.weight", (C,)), (".cpe.2.bias", (C,)),
                     (".norm1.0.weight", (C,)), (".norm1.0.bias", (C,)), (".attn.qkv.weight", (3 * C, C)), (".attn.qkv.bias", (3 * C,)),
                     (".attn.proj.weight", (C, C)), (".attn.proj.bias", (C,)), (".norm2.0.weight", (C,)), (".norm2.0.bias", (C,)),
                     (".mlp.0.fc1.weight", (4 * C, C)), (".mlp.0.fc1.bias", (4 * C,)), (".mlp.0.fc2.weight", (C, 4 * C)),
                     (".mlp.0.fc2.bias", (C,))):
        scale = {1: 0.1, 2: 0.3, 5: 0.3 / 27 ** 0.5}[len(shape)]
        sd[pre + k] = (rng.standard_normal(shape) * scale + (1.0 if k.endswith(".weight") and len(shape) == 1 else 0.0)).astype(np.float32)
    x_in = rng.standard_normal((n, C)).astype(np.float32)
    dy = rng.standard_normal((n, C)).astype(np.float32)
    ry, rdx, rg = OT.block_full_grads(sd, pre, x_in, nbr, order, inverse, cu, H, dy)

    dev = torch.device("cuda")
    w = _weights(sd, pre, dev)
    f = lambda k: torch.as_tensor(sd[pre + k], dtype=torch.float32).to(dev).contiguous()  # noqa: E731
    w.update({"B.cpe0.w": f(".cpe.0.weight").reshape(C, -1).contiguous(), "B.cpe0.b": f(".cpe.0.bias"), "B.cpe1.w": f(".cpe.1.weight"),
              "B.cpe1.b": f(".cpe.1.bias"), "B.cpe2.g": f(".cpe.2.weight"), "B.cpe2.b": f(".cpe.2.bias")})
    gidx, widx = _slot_plan(order, inverse, dev)
    ps = torch.as_tensor(np.asarray(cu), dtype=torch.int32).to(dev)
    nbr_k = torch.as_tensor(nbr.T.astype(np.int32)).contiguous().to(dev)
    tape = train.block_forward(w, "B", torch.as_tensor(x_in).to(dev), nbr_k, gidx, widx, ps, [int(v) for v in cu], H,
                               int(np.diff(cu).max()), (C // H) ** -0.5)
    dx, dxc, grads = train.block_backward(w, "B", tape, torch.as_tensor(dy).to(dev))
    torch.cuda.synchronize()
    assert dxc is None
    ey = float(np.abs(tape["tail"].y.cpu().numpy() - ry.numpy()).max())
    ex = float(np.abs(dx.cpu().numpy() - rdx.numpy()).max()) / max(1.0, float(rdx.abs().max()))
    names = {"B.cpe0.w": ".cpe.0.weight", "B.cpe0.b": ".cpe.0.bias", "B.cpe1.w": ".cpe.1.weight", "B.cpe1.b": ".cpe.1.bias",
             "B.cpe2.g": ".cpe.2.weight", "B.cpe2.b": ".cpe.2.bias", "B.norm1.g": ".norm1.0.weight", "B.norm1.b": ".norm1.0.bias",
             "B.qkv.w": ".attn.qkv.weight", "B.qkv.b": ".attn.qkv.bias", "B.proj.w": ".attn.proj.weight", "B.proj.b": ".attn.proj.bias",
             "B.norm2.g": ".norm2.0.weight", "B.norm2.b": ".norm2.0.bias", "B.fc1.w": ".mlp.0.fc1.weight", "B.fc1.b": ".mlp.0.fc1.bias",
             "B.fc2.w": ".mlp.0.fc2.weight", "B.fc2.b": ".mlp.0.fc2.bias"}
    assert set(grads) == set(names)
    worst = 0.0
    for mine, ref in names.items():
        r = rg[pre + ref].reshape(grads[mine].shape)
        e = float((grads[mine].cpu() - r).abs().max()) / max(1.0, float(r.abs().max()))
        worst = max(worst, e)
        assert e < 1e-3, (mine, e)
    print(f"[measure] whole Block backward vs oracle autograd n={n} H={H}: forward {ey:.3e}, d_x_in rel {ex:.3e}, "
          f"worst parameter gradient rel {worst:.3e} (18 tensors incl. the 27-offset conv kernel)")
    assert ey < 1e-3 and ex < 1e-3


def _mini_training_model(fx, dev):
    from cdsegnet_amd import configs
    from cdsegnet_amd.param_init import fill_state_dict
    from cdsegnet_amd.registry import build_model
    import cdsegnet_amd.models  # noqa: F401
    cfg = configs.mini_config()
    cfg["backbone"]["enable_flash"] = False  # the fixture was captured on the reference's non-flash (CPU) attention path
    cfg["criteria"] = [dict(type="MSELoss", loss_weight=1.0, ignore_index=-1, batch_sample_point=-1),
                       dict(type="CrossEntropyLoss", loss_weight=1.0, ignore_index=-1),
                       dict(type="LovaszLoss", mode="multiclass", loss_weight=1.0, ignore_index=-1)]
    model = build_model(cfg)
    sd = fill_state_dict(model.state_dict(), seed=int(fx["sd_seed"]))
    model.load_state_dict(sd)
    return model.to(dev).train(), sd


def test_whole_training_step_matches_the_reference_train_step():
    """The model's training forward + loss.backward() + AdamW step on the device (cdsegnet_amd/train_graph.py: sparse convs,
    Linears, LayerNorms, (cross) attention and the pooling maximum on the HIP kernels behind torch.autograd.Functions, fp32)
    against the REFERENCE's own training step on two scenes (tests/golden/train_step_mini.npz, oracle/make_golden.py
    trainstep: default.py:424-493 + engines/train.py:216-271 with every random draw recorded): loss, both predictions, the
    norm of EVERY one of the 508 parameter gradients, eight gradients in full, and the first AdamW step (two learning-rate
    groups, configs/scannet/CDSegNet.py:143-147).  Bound 1e-3 relative (north_star's fp32 tolerance)."""
    fx = load_fixture("train_step_mini.npz")
    dev = torch.device("cuda")
    model, sd = _mini_training_model(fx, dev)
    masks = {str(k): [fx[f"mask.{i}.{j}"] for j in range(int(fx["mask_counts"][i]))] for i, k in enumerate(fx["mask_names"])}
    draws = dict(ts=fx["ts"], noise=fx["noise"], perms=[list(p) for p in fx["perms"]], masks=masks)
    inp = {k: torch.as_tensor(fx[k]).to(dev) for k in ("coord", "grid_coord", "feat", "offset", "segment")}
    named = dict(model.named_parameters())
    blk = [p for k, p in named.items() if "block" in k]
    rest = [p for k, p in named.items() if "block" not in k]
    opt = torch.optim.AdamW([dict(params=rest, lr=0.002), dict(params=blk, lr=0.0002)], lr=0.002, weight_decay=0.05)
    opt.zero_grad()
    out = model(inp, draws=draws)
    e_loss = abs(float(out["loss"].detach()) - float(fx["loss"]))
    e_n = float((out["n_pred"].detach().cpu() - torch.as_tensor(fx["n_pred"])).abs().max())
    e_c = float((out["c_pred"].detach().cpu() - torch.as_tensor(fx["c_pred"])).abs().max())
    out["loss"].backward()
    torch.cuda.synchronize()
    names = [str(n) for n in fx["grad_names"]]
    assert all(named[k].grad is not None for k in names) and len(names) == 508
    gn = np.array([float(named[k].grad.norm()) for k in names])
    ref = fx["grad_norms"]
    rel = np.abs(gn - ref) / (ref + 1e-3 * ref.max())
    worst_full = 0.0
    checked = 0
    for k in fx.files:
        if k.startswith("g."):
            r = fx[k]
            worst_full = max(worst_full, float((named[k[2:]].grad.cpu() - torch.as_tensor(r)).abs().max()) / float(np.abs(r).max()))
            checked += 1
    print(f"[measure] whole training step vs reference: loss err {e_loss:.3e}, n_pred err {e_n:.3e}, c_pred err {e_c:.3e}, "
          f"worst gradient-norm rel err over 508 parameters {rel.max():.3e}, worst of {checked} full gradients rel {worst_full:.3e}")
    assert e_loss < 1e-4 and e_n < 1e-3 and e_c < 1e-3
    assert rel.max() < 1e-3 and checked == 8 and worst_full < 1e-3
    opt.step()
    torch.cuda.synchronize()
    worst = 0.0
    nchk = 0
    for i, k in enumerate(names):
        if ref[i] < 1e-4 * ref.max():
            continue  # (a gradient that is rounding noise makes g / (|g| + eps) noise: see tests/test_oracle.py)
        dn = float((named[k].detach().cpu() - sd[k].float()).norm())
        worst = max(worst, abs(dn - float(fx["step_norms"][i])) / max(float(fx["step_norms"][i]), 1e-12))
        if "p1." + k in fx.files:
            assert float((named[k].detach().cpu() - torch.as_tensor(fx["p1." + k])).abs().max()) < 5e-6
            nchk += 1
    print(f"[measure] first AdamW step vs reference: worst step-norm rel err {worst:.3e} ({nchk} parameters compared in full)")
    assert worst < 2e-2 and nchk == 2


def test_training_forward_default_draws_runs_and_descends():
    """Without injected draws (timesteps, noise, order shuffles from torch's CPU generator like the reference, stochastic-depth
    masks on the device): three optimizer steps on one batch lower the loss, every parameter receives a finite gradient."""
    fx = load_fixture("train_step_mini.npz")
    dev = torch.device("cuda")
    model, _ = _mini_training_model(fx, dev)
    inp = {k: torch.as_tensor(fx[k]).to(dev) for k in ("coord", "grid_coord", "feat", "offset", "segment")}
    opt = torch.optim.AdamW(model.parameters(), lr=0.002, weight_decay=0.05)
    torch.manual_seed(7)
    losses = []
    for _ in range(4):
        opt.zero_grad()
        torch.manual_seed(7)  # the same draws every step: the loss of THIS batch under THESE draws must go down
        loss = model(inp)["loss"]
        loss.backward()
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters())
        losses.append(float(loss.detach()))
        opt.step()
    print(f"[measure] training loop on one batch, 4 steps: loss {losses}")
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


def test_reference_run_step_sequence_with_amp_and_grad_scaler():
    """The reference trainer's run_step (engines/train.py:216-271) with cfg.enable_amp = True, verbatim: the model called
    inside torch.cuda.amp.autocast, scaler.scale(loss).backward(), gradient clipping, scaler.step(optimizer),
    scaler.update(), scheduler.step().  The forward stays exact fp32 inside the autocast context (cdsegnet_amd/train_graph.py),
    so the step must equal the plain fp32 step on the same draws: same loss, same updated parameters."""
    fx = load_fixture("train_step_mini.npz")
    dev = torch.device("cuda")
    masks = lambda: {str(k): [fx[f"mask.{i}.{j}"] for j in range(int(fx["mask_counts"][i]))] for i, k in enumerate(fx["mask_names"])}  # noqa: E731
    inp = {k: torch.as_tensor(fx[k]).to(dev) for k in ("coord", "grid_coord", "feat", "offset", "segment")}
    results = []
    for amp in (False, True):
        model, _ = _mini_training_model(fx, dev)
        opt = torch.optim.AdamW(model.parameters(), lr=0.002, weight_decay=0.05)
        sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=0.002, total_steps=10)
        scaler = torch.cuda.amp.GradScaler() if amp else None
        draws = dict(ts=fx["ts"], noise=fx["noise"], perms=[list(p) for p in fx["perms"]], masks=masks())
        with torch.cuda.amp.autocast(enabled=amp):
            loss = model(inp, draws=draws)["loss"]
        assert loss.dtype == torch.float32
        opt.zero_grad()
        if amp:
            scaler.scale(loss).backward()
            scaler.step(opt)
            scale = scaler.get_scale()
            scaler.update()
            if scale <= scaler.get_scale():
                sched.step()
        else:
            loss.backward()
            opt.step()
            sched.step()
        torch.cuda.synchronize()
        results.append((float(loss.detach()), {k: p.detach().clone() for k, p in model.named_parameters()}))
    (l0, p0), (l1, p1) = results
    worst = max(float((p0[k] - p1[k]).abs().max()) for k in p0)
    print(f"[measure] run_step with AMP context + GradScaler vs plain fp32 step: loss {l0:.6f} / {l1:.6f}, worst parameter difference {worst:.3e}")
    assert abs(l0 - l1) < 1e-6 and abs(l0 - float(fx["loss"])) < 1e-4
    assert worst < 1e-5  # (the scaler's 65536x on the gradients and back is exact in fp32 up to rounding of the scaled values)
