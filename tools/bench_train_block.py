"""Measurement of the training path's second slice: forward-with-tape + whole-Block backward (input gradient + all 18
parameter gradients, exact fp32) on the HIP kernels, at a real stage shape, next to torch autograd on the CPU oracle.
usage: python tools/bench_train_block.py [n_points=56000] [heads=4] [cpu=1]
Prints ms per forward / backward, the algorithmic FLOPs (backward = 2x forward for the Linears and the conv, 2.5x for
the attention core: recomputed scores + four products) and the CPU time of the same gradients."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import ops, synth, train

n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 56000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 4
do_cpu = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
C = 16 * H
dev = torch.device("cuda")
sc = synth.room_scene(3, n_req)
grid = torch.as_tensor(sc["grid_coord"]).to(dev).int().contiguous()
n = grid.shape[0]
batch = torch.zeros(n, dtype=torch.int32, device=dev)
depth = int(ops.grid_max(grid.long()).item()).bit_length()
zs, perm0 = ops.sort_pairs(ops.encode(grid.long(), batch.long(), depth, "z"))
gz = ops.gather_rows(grid, perm0)
nbr = ops.nbr_table(zs, gz, batch, depth, 3, True)  # (27, n) offset-major, rows in z order
code4 = ops.encode4(gz, batch, depth)
_, order = ops.sort_pairs(code4[2].contiguous())
K = 1024
npad = (n + K - 1) // K * K
offs = torch.tensor([0, n], dtype=torch.int32, device=dev)
offs_pad = torch.tensor([0, npad], dtype=torch.int32, device=dev)
gidx, widx = ops.pad_plan(order, offs, offs_pad, K, npad)
ps_host = list(range(0, npad + 1, K))
ps = torch.tensor(ps_host, dtype=torch.int32, device=dev)
g = torch.Generator().manual_seed(0)
def rnd(*sh, s=1.0): return (torch.randn(*sh, generator=g) * s).to(dev)
w = {"B.cpe0.w": rnd(C, 27 * C, s=0.3 / (27 * C) ** 0.5 * C ** 0.5), "B.cpe0.b": rnd(C, s=0.1), "B.cpe1.w": rnd(C, C, s=C ** -0.5), "B.cpe1.b": rnd(C, s=0.1),
     "B.cpe2.g": 1 + rnd(C, s=0.1), "B.cpe2.b": rnd(C, s=0.1), "B.norm1.g": 1 + rnd(C, s=0.1), "B.norm1.b": rnd(C, s=0.1),
     "B.qkv.w": rnd(3 * C, C, s=C ** -0.5), "B.qkv.b": rnd(3 * C, s=0.1), "B.proj.w": rnd(C, C, s=C ** -0.5), "B.proj.b": rnd(C, s=0.1),
     "B.norm2.g": 1 + rnd(C, s=0.1), "B.norm2.b": rnd(C, s=0.1), "B.fc1.w": rnd(4 * C, C, s=C ** -0.5), "B.fc1.b": rnd(4 * C, s=0.1),
     "B.fc2.w": rnd(C, 4 * C, s=(4 * C) ** -0.5), "B.fc2.b": rnd(C, s=0.1)}
x = rnd(n, C)
dy = rnd(n, C)
def fwd(): return train.block_forward(w, "B", x, nbr, gidx, widx, ps, ps_host, H, K, 0.25)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); t.append(1e3 * (time.perf_counter() - t0))
    return sorted(t)[len(t) // 2], r
ms_f, tape = timeit(fwd)
ms_b, (dx, _, grads) = timeit(lambda: train.block_backward(w, "B", tape, dy))
occ = float((nbr >= 0).float().mean()) * 27
lens = np.diff(ps_host).astype(np.float64); lens[-1] = n - ps_host[-2] if n % K else K
f_lin = 2.0 * n * C * C * (1 + 3 + 1 + 4 + 4)
f_conv = 2.0 * n * occ * C * C
f_attn = 4.0 * 16 * H * float((lens ** 2).sum())
fwd_flop = f_lin + f_conv + f_attn
bwd_flop = 2 * (f_lin + f_conv) + 2.5 * f_attn
print(f"whole Block, n={n} C={C} H={H} (fp32 exact mode): forward with tape {ms_f:.2f} ms ({fwd_flop / ms_f / 1e9:.1f} TFLOP/s), "
      f"backward (d input + 18 parameter gradients) {ms_b:.2f} ms ({bwd_flop / ms_b / 1e9:.1f} TFLOP/s algorithmic); "
      f"|d x| mean {float(dx.abs().mean()):.3e}, gradients {len(grads)} tensors")
if do_cpu:
    from oracle import train as OT
    sd = {"b.cpe.0.weight": w["B.cpe0.w"].view(C, 3, 3, 3, C), "b.cpe.0.bias": w["B.cpe0.b"], "b.cpe.1.weight": w["B.cpe1.w"], "b.cpe.1.bias": w["B.cpe1.b"],
          "b.cpe.2.weight": w["B.cpe2.g"], "b.cpe.2.bias": w["B.cpe2.b"], "b.norm1.0.weight": w["B.norm1.g"], "b.norm1.0.bias": w["B.norm1.b"],
          "b.attn.qkv.weight": w["B.qkv.w"], "b.attn.qkv.bias": w["B.qkv.b"], "b.attn.proj.weight": w["B.proj.w"], "b.attn.proj.bias": w["B.proj.b"],
          "b.norm2.0.weight": w["B.norm2.g"], "b.norm2.0.bias": w["B.norm2.b"], "b.mlp.0.fc1.weight": w["B.fc1.w"], "b.mlp.0.fc1.bias": w["B.fc1.b"],
          "b.mlp.0.fc2.weight": w["B.fc2.w"], "b.mlp.0.fc2.bias": w["B.fc2.b"]}
    sd = {k: v.cpu().numpy() for k, v in sd.items()}
    order_np = gidx.cpu().numpy().astype(np.int64)
    wnp = widx.cpu().numpy()
    inverse = np.empty(n, dtype=np.int64)
    slots = np.nonzero(wnp >= 0)[0]
    inverse[wnp[slots]] = slots
    torch.set_num_threads(16)
    t0 = time.perf_counter()
    ry, rdx, rg = OT.block_full_grads(sd, "b", x.cpu().numpy(), nbr.cpu().numpy().T.astype(np.int64), order_np, inverse, np.array(ps_host), H, dy.cpu().numpy())
    cpu_s = time.perf_counter() - t0
    e = float((dx.cpu() - rdx).abs().max()) / max(1.0, float(rdx.abs().max()))
    print(f"CPU oracle (torch autograd, 16 threads), same Block forward + backward: {cpu_s:.2f} s = {1e3 * cpu_s / (ms_f + ms_b):.0f}x the HIP time; "
          f"d_x_in rel err {e:.2e}")
# ---- where the backward's time goes (each piece alone, synchronised)
c = C
t = tape["tail"]
do = rnd(n, c)
def attn_b():
    dqkv = torch.zeros((n, 3 * c), dtype=torch.float32, device=dev)
    ops.bind_stream()
    ops.attention_bwd(t.qkv[:, :c], t.qkv[:, c:2 * c], t.qkv[:, 2 * c:], t.gidx, t.gidx, t.widx, t.patch_start, t.patch_start_host,
                      t.num_heads, t.scale, do, dqkv[:, :c], dqkv[:, c:2 * c], dqkv[:, 2 * c:])
    ops.unbind_stream()
def conv_w():
    dwc = torch.zeros(c, 27 * c, device=dev)
    ops.bind_stream()
    if hasattr(ops, "conv_wgrad"):
        ops.conv_wgrad(x, nbr, dy, dwc.view(c, 27, c))
    else:
        for o in range(27):
            ops.linear_wgrad(x, dy, dwc.view(c, 27, c)[:, o, :], None, xidx=nbr[o])
    ops.unbind_stream()
def lin_w():
    ops.bind_stream()
    ops.linear_wgrad(x, t.u, torch.zeros(4 * c, c, device=dev), torch.zeros(4 * c, device=dev))
    ops.unbind_stream()
def conv_d():
    ops.bind_stream()
    ops.gemm(dy, train._conv_bwd_weight(w["B.cpe0.w"], c, c), torch.empty(n, c, device=dev), nbr=nbr, kvol=27, nbr_kmajor=True)
    ops.unbind_stream()
for name, fn in (("attention backward", attn_b), ("conv weight gradient (27 offsets)", conv_w), ("one Linear weight gradient (4C x C)", lin_w),
                 ("conv data gradient (gathered GEMM, mirrored kernel)", conv_d)):
    ms, _ = timeit(fn)
    print(f"  {name}: {ms:.2f} ms")
