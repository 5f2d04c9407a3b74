"""Micro-benchmark of the serialized-attention kernel alone on a stage-shaped problem.
usage: python tools/bench_attention.py [n_points] [heads] [dtype] [iters]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import ops, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dtype = torch.bfloat16 if (len(sys.argv) <= 3 or sys.argv[3] == "bf16") else torch.float32
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
C = 16 * H
dev = torch.device("cuda")
# realistic gather pattern: physical z order, attention along the hilbert curve
sc = synth.room_scene(0, n)
grid = torch.as_tensor(sc["grid_coord"]).to(dev)
batch = torch.zeros(n, dtype=torch.int64, device=dev)
depth = int(ops.grid_max(grid).item()).bit_length()
zs, perm0 = ops.sort_pairs(ops.encode(grid, batch, depth, "z"))
g0, b0 = ops.plan_gather_grid(grid, perm0, zs, depth)
code4 = ops.encode4(g0, b0, depth)
_, order = ops.sort_pairs(code4[2].contiguous())
K = 1024
npad = (n + K - 1) // K * K
offs = torch.tensor([0, n], dtype=torch.int32, device=dev)
offs_pad = torch.tensor([0, npad], dtype=torch.int32, device=dev)
gidx, widx = ops.pad_plan(order, offs, offs_pad, K, npad)
ps = torch.arange(0, npad + 1, K, dtype=torch.int32, device=dev)
qkv = torch.randn(n, 3 * C, device=dev).to(dtype)
out = torch.empty(n, C, dtype=dtype, device=dev)
P = ps.numel() - 1
flops = 64.0 * H * P * K * K
def run():
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], gidx, gidx, widx, ps, H, K, 0.25, out)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / iters
print(f"attention n={n} H={H} {dtype}: {us:.1f} us/launch, {flops / us / 1e6:.1f} TFLOP/s algorithmic, "
      f"{P * H} patch-heads, q+k+v+o bytes {4 * npad * C * qkv.element_size() / 1e6:.1f} MB")
