"""Micro-benchmark of the serialized-attention kernel alone on a stage-shaped problem.
usage: python tools/bench_attention.py [n_points] [heads] [dtype] [iters] [curve] [scenes] [flags]
  flags: cdseg_attention_ex flags (1: q pre-scaled, 2: v bfloat16 - what the engine's fused qkv producers declare)
  CDSEG_AB_LIB=path/to/other/libcdseg_hip.so  benchmark another build of the library (A/B runs; tools only)
  curve: 0 z, 1 z-trans, 2 hilbert (default), 3 hilbert-trans - the curve the patches are cut from
Prints min / median over `rounds` timed groups (the chip re-clocks between launches: quote the median)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import _lib
if os.environ.get("CDSEG_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CDSEG_AB_LIB"])
from cdsegnet_amd import ops, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dtype = torch.bfloat16 if (len(sys.argv) <= 3 or sys.argv[3] == "bf16") else torch.float32
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
curve = int(sys.argv[5]) if len(sys.argv) > 5 else 2
C = 16 * H
dev = torch.device("cuda")
# realistic gather pattern: physical z order, attention along another curve
scenes = int(sys.argv[6]) if len(sys.argv) > 6 else 1  # > 1: that many collated scenes of n / scenes points
flags = int(sys.argv[7]) if len(sys.argv) > 7 else 0
grids, batches = [], []
for i in range(scenes):
    sc = synth.room_scene(i, n // scenes)
    grids.append(torch.as_tensor(sc["grid_coord"]))
    batches.append(torch.full((len(sc["grid_coord"]),), i, dtype=torch.int64))
grid = torch.cat(grids).to(dev)
batch = torch.cat(batches).to(dev)
n = grid.shape[0]
counts = torch.bincount(batch.cpu(), minlength=scenes).numpy()
depth = int(ops.grid_max(grid).item()).bit_length()
zs, perm0 = ops.sort_pairs(ops.encode(grid, batch, depth, "z"))
g0, b0 = ops.plan_gather_grid(grid, perm0, zs, depth)
code4 = ops.encode4(g0, b0, depth)
_, order = ops.sort_pairs(code4[curve].contiguous())
K = 1024
pads = [(c + K - 1) // K * K if c > K else c for c in counts]
offs = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device=dev)
offs_pad = torch.tensor(np.concatenate([[0], np.cumsum(pads)]), dtype=torch.int32, device=dev)
npad = int(sum(pads))
gidx, widx = ops.pad_plan(order, offs, offs_pad, K, npad)
starts = []
for s, p in zip(np.concatenate([[0], np.cumsum(pads)])[:-1], pads):
    starts += list(range(int(s), int(s + p), K))
ps = torch.tensor(starts + [npad], dtype=torch.int32, device=dev)
qkv = torch.randn(n, 3 * C, device=dev).to(dtype)
out = torch.empty(n, C, dtype=dtype, device=dev)
P = ps.numel() - 1
lens = (ps[1:] - ps[:-1]).double()
flops = 64.0 * H * float((lens * lens).sum())
def run():
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], gidx, gidx, widx, ps, H, K, 0.25, out, flags=flags)
for _ in range(5): run()
torch.cuda.synchronize()
rounds = 7
us = []
for _ in range(rounds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    us.append(1e3 * e0.elapsed_time(e1) / iters)
us.sort()
med, mn = us[len(us) // 2], us[0]
tag = os.path.basename(os.environ.get("CDSEG_AB_LIB", "libcdseg_hip.so"))
print(f"attention[{tag}] n={n} H={H} curve={curve} {dtype}: median {med:.1f} us/launch (min {mn:.1f}), "
      f"{flops / med / 1e6:.1f} TFLOP/s algorithmic = {flops / med / 1e6 / 2500:.3f} of the bf16 MFMA peak, "
      f"{P * H} patch-heads, q+k+v+o bytes {4 * npad * C * qkv.element_size() / 1e6:.1f} MB, checksum {float(out.float().abs().sum()):.6e}")
